#!/bin/bash
# round 6: the staged (four-stage) 128x128 split-plane GEMM kernel against what it replaces, per shape at a lone caller's row counts
# (diagnostic library: TT_GEMM_STAGED=0 = the 256x256 split-plane kernel everywhere), after the parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_x3_gpu.py tests/test_f16_gpu.py -x -q -m gpu 2>&1 | tail -6
for M in 1024 2048 3072 4096 5120 7424; do
  for S in 0 1; do
    echo "=== M=$M TT_GEMM_STAGED=$S"
    TT_GEMM_STAGED=$S timeout 300 ./tools/gemm_bench_diag $M 50 | grep "^x3\|^gemm_bench"
  done
done
for M in 5120 7424; do
  echo "=== M=$M TT_GEMM_STAGED_MAX=512 (two rounds of 128x128 tiles instead of the 256x256 kernel)"
  TT_GEMM_STAGED_MAX=512 timeout 300 ./tools/gemm_bench_diag $M 50 | grep "^x3"
done
} 2>&1 | tee gpurun_out/r06_staged_ab.log
