#!/bin/bash
# round 5: the kernels' A/B switches exist in the diagnostic library only (csrc: make DIAG=1); the product library reads no environment
export TT_LIB_NAME=${TT_LIB_NAME:-libtt_hip_diag.so}
# Skinny-GEMM visit: its tests + single-query latency with and without it.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_encoder_gpu.py tests/test_configs_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q 2>&1 | tail -15
(timeout 300 python tools/probes/single_query_latency.py 1000000 1 2>&1 | tail -6; echo "--- TT_GEMM_SKINNY=0"; TT_GEMM_SKINNY=0 timeout 300 python tools/probes/single_query_latency.py 1000000 1 2>&1 | tail -6) | tee gpurun_out/single_query_latency.log
