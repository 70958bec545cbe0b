#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --steps 3 --warmup 1 ${BENCH_ARGS} > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -5 gpurun_out/bench.err
cat gpurun_out/bench.json
