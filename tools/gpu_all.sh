#!/bin/bash
# Full GPU visit: all gpu-marked tests, smoke(), the default bench line, rocprof summary of the bench.
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu ${PYTEST_X--x} -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.log
timeout 900 python bench.py ${BENCH_ARGS} > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -3 gpurun_out/bench.err; cat gpurun_out/bench.json
if [ -n "$PROFILE" ]; then
rm -rf gpurun_out/prof_bench
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
find gpurun_out/prof_bench -name "*kernel_stats*" | head -1 | xargs cat | head -16
fi
