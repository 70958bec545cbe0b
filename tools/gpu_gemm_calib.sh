#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python tools/gemm_calib.py 2>&1 | tee gpurun_out/gemm_calib.log
