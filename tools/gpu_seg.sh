#!/bin/bash
# Segmented-scan visit: its tests + the latency probe.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_scan_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q 2>&1 | tail -15
for cfg in "2000 8 1" "100000 8 1" "1000000 4 1" "100000 8 16"; do timeout 300 python tools/probes/segmented_scan.py $cfg 2>&1 | tail -1; done | tee gpurun_out/segmented_scan.log
