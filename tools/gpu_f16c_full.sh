#!/bin/bash
# f16c: building-block tests + the full-depth gates (standard and stress fixtures), every reference implementation.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1000 python -m pytest tests/test_f16c_gpu.py tests/test_rank_agreement_gpu.py tests/test_x3_gpu.py -m gpu -q -s --tb=short -k "f16c or f16x3 or stress or bf16x3 or x3_forward or precision_selector or unchanged" 2>&1 | grep -v "^E   *+ \|where <built-in" | tail -${TAIL:-80} | tee gpurun_out/f16c_full.log
