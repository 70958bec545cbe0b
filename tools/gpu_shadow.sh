#!/bin/bash
# round 6: fp8 shadow prefilter -- parity tests, then the lone caller with and without it
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_scan_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q -k "shadow" 2>&1 | tail -15
bash tools/gpu_lone_caller.sh shadow
TT_SCAN_SHADOW=0 bash tools/gpu_lone_caller.sh noshadow | head -8
