#!/bin/bash
# the driver's round-end GPU tier on the current tree: pytest -m gpu, then smoke()
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 3300 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r06_pytest_gpu.log
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/r06_smoke.log
