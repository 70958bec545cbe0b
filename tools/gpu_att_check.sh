#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q -k "attention or forward or cls" 2>&1 | tail -5
./tools/att_bench 800 292 20; ./tools/att_bench 800 256 20; ./tools/att_bench 400 512 20; ./tools/att_bench 4096 34 20
} 2>&1 | tee gpurun_out/att_check.log
