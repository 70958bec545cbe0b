#!/bin/bash
# All gpu-marked tests (or the ones in $TESTS), durations of the slowest, the tests' own report lines, smoke().
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout ${PYTEST_TIMEOUT:-1700} python -m pytest ${TESTS:-tests} -m gpu ${PYTEST_X--x} -q -s --durations=12 > gpurun_out/pytest_gpu_full.log 2>&1
grep -oE "(bf16 @24L|fp16 @24L|fp8 @24L|fp32 @24L|bf16x3 @24L|f16x3 @24L|f16c @24L|stress @24L|config 5 composed|config 5|coalescing:|full depth:|worker ingest).*" gpurun_out/pytest_gpu_full.log | tee gpurun_out/pytest_gpu_report.log
tail -${TAIL:-60} gpurun_out/pytest_gpu_full.log | tee gpurun_out/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.log
