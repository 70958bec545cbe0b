#!/bin/bash
# round 6: relay skinny kernel -- bit-identity tests (alone vs batch, skinny vs tiled) then the lone-caller breakdown
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_encoder_gpu.py tests/test_configs_gpu.py tests/test_x3_gpu.py tests/test_f16_gpu.py -m gpu -x -q 2>&1 | tail -8
bash tools/gpu_lone_caller.sh ${1:-relay}
