#!/bin/bash
# round 5: the kernels' A/B switches exist in the diagnostic library only (csrc: make DIAG=1); the product library reads no environment
export TT_LIB_NAME=${TT_LIB_NAME:-libtt_hip_diag.so}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for d in 0 1 2 3; do echo "== TT_GEMM_DEBUG_TRAFFIC=$d (1: all C rows -> row 0, 2: all A rows -> row 0)"; TT_GEMM_DEBUG_TRAFFIC=$d timeout 120 ./tools/gemm_bench_diag 236800 10 | sed -n 6,8p; done
} 2>&1 | tee gpurun_out/gemm_traffic.log
