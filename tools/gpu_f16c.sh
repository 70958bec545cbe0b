#!/bin/bash
# f16c bring-up: the scale-semantics probe, then the f16c tests.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 60 tools/probes/mfma_scale_probe | tail -1
timeout 1500 python -m pytest ${TESTS:-tests/test_f16c_gpu.py} -m gpu ${PYTEST_X--x} -q -s --tb=short ${PYTEST_ARGS} 2>&1 | grep -v "^E   *+ \|where <built-in" | tail -${TAIL:-150}
} 2>&1 | tee gpurun_out/f16c.log
