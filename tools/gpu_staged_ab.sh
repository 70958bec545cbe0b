#!/bin/bash
# round 6: the staged (four-stage) 128x128 GEMM kernel against what it replaces, per shape at a lone caller's row counts
# (diagnostic library: TT_GEMM_STAGED=0 = the two-stage 128x128 kernel / the 256x256 split-plane kernel), then the new parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_x3_gpu.py tests/test_f16_gpu.py -x -q -m gpu 2>&1 | tail -6
for M in 1024 3072 5120 7424; do
  for S in 0 1; do
    echo "=== M=$M TT_GEMM_STAGED=$S"
    TT_GEMM_STAGED=$S timeout 300 ./tools/gemm_bench_diag $M 50 | grep -v "^fp8\|small"
  done
done
for T in 192 256 384; do
  echo "=== M=3072 TT_GEMM_STAGED_T=$T (split planes: staged below T tiles of 256x256)"
  TT_GEMM_STAGED_T=$T timeout 300 ./tools/gemm_bench_diag 3072 50 | grep "^x3"
  echo "=== M=7424 TT_GEMM_STAGED_T=$T"
  TT_GEMM_STAGED_T=$T timeout 300 ./tools/gemm_bench_diag 7424 50 | grep "^x3"
done
} 2>&1 | tee gpurun_out/r06_staged_ab.log
