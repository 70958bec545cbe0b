#!/bin/bash
# round 5: the kernels' A/B switches exist in the diagnostic library only (csrc: make DIAG=1); the product library reads no environment
export TT_LIB_NAME=${TT_LIB_NAME:-libtt_hip_diag.so}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for v in ${VARIANTS:-1 3 1 3}; do echo "== TT_GEMM_VARIANT=$v"; TT_GEMM_VARIANT=$v timeout 120 ./tools/gemm_bench 16384 30; done 2>&1 | tee gpurun_out/gemm_ab.log
TT_GEMM_VARIANT=${TESTV:-3} timeout 900 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q 2>&1 | tail -15
