#!/bin/bash
# round 6: the split-plane path at bge-small / MiniLM geometry (hidden 384 = 12 x 32): building blocks, forward vs oracle, C1, MiniLM
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_x3_gpu.py tests/test_configs_gpu.py tests/test_hf_checkpoint_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q 2>&1 | tail -25
