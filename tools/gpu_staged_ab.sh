#!/bin/bash
# round 6: the staged (four-stage) 128x128 GEMM kernel against what it replaces, per shape at a lone caller's row counts (diagnostic
# library: TT_GEMM_STAGED=0 = the 256x256 split-plane kernel / the two-stage 128x128 kernel everywhere; TT_GEMM_STAGED_MAX / _STAGED16 =
# the 128x128-tile count up to which the staged kernel is taken, default one per CU), after the parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_x3_gpu.py tests/test_f16_gpu.py -x -q -m gpu 2>&1 | tail -6
for M in 1024 2048 3072 4096 5120 7424; do
  echo "=== M=$M TT_GEMM_STAGED=0"
  TT_GEMM_STAGED=0 timeout 300 ./tools/gemm_bench_diag $M 50 | grep -v "^fp8\|small\|shape, bias"
  echo "=== M=$M TT_GEMM_STAGED=1 (default: split planes up to two rounds, 16-bit one round)"
  timeout 300 ./tools/gemm_bench_diag $M 50 | grep -v "^fp8\|small\|shape, bias\|^gemm_bench"
  echo "=== M=$M TT_GEMM_STAGED=1, up to two rounds for the 16-bit form too (TT_GEMM_STAGED16=512)"
  TT_GEMM_STAGED16=512 timeout 300 ./tools/gemm_bench_diag $M 50 | grep -v "^fp8\|small\|shape, bias\|^gemm_bench"
done
} 2>&1 | tee gpurun_out/r06_staged_ab.log
