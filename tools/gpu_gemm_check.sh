#!/bin/bash
# GEMM shapes (epilogue-isolating) + encoder parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 120 ./tools/gemm_bench 236800 10 2>&1 | tee gpurun_out/gemm_shapes.log
timeout 900 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q 2>&1 | tail -15
