#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
./tools/gemm_bench 16384 20 > gpurun_out/gemm_bench.log 2>&1
cat gpurun_out/gemm_bench.log
timeout 1500 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q 2>&1 | tail -40 | tee gpurun_out/pytest_enc.log
