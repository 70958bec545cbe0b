#!/bin/bash
# round 5: the kernels' A/B switches exist in the diagnostic library only (csrc: make DIAG=1); the product library reads no environment
export TT_LIB_NAME=${TT_LIB_NAME:-libtt_hip_diag.so}
# Threshold sample of the tiled scan on the contraction kernel (round 4) vs the streaming sample kernel: parity tests, then per-stage times
# of the 256-query batch on the 1.25 M-row shard and on 10 M rows, both ways.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_scan_gpu.py tests/test_scan_sweep_gpu.py tests/test_abi_sweeps_gpu.py tests/test_pipeline_gpu.py tests/test_configs_gpu.py tests/test_two_ranks_gpu.py tests/test_gemm_big_gpu.py -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/pytest_scan.log
out=gpurun_out/scan_sample_ab.log
: > $out
for rows in 1250000 10000000; do
  for s in 1 0 1 0; do
    ROWS=$rows TT_SCAN_GEMM_SAMPLE=$s timeout 300 python tools/probes/scan_gemm_ab.py 2>&1 | grep "^Q=256" | sed "s/^/rows=$rows TT_SCAN_GEMM_SAMPLE=$s /" >> $out
  done
done
cat $out
