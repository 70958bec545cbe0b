#!/bin/bash
# The default bench line (what the driver runs) + optional extra args.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python bench.py ${BENCH_ARGS} > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "rc=$?"; tail -5 gpurun_out/bench.err; cat gpurun_out/bench.json
