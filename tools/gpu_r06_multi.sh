#!/bin/bash
# round 6: first-contact safety on the round-6 tree -- the two new / changed GPU tests, then the 8-rank self-launch at the FULL
# configuration (tools/gpu_eight_ranks_one_gpu.sh) and the two-rank RCCL-refusal fallback
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1700 python -m pytest tests/test_two_ranks_gpu.py tests/test_pipeline_gpu.py tests/test_config5_gpu.py -m gpu -x -q 2>&1 | tail -8
bash tools/gpu_eight_ranks_one_gpu.sh 2>&1 | tail -12
cp gpurun_out/eight_ranks.json gpurun_out/r06_self_launched_eight_ranks_one_gpu.json
