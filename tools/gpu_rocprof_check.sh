#!/bin/bash
# rocprofv3 summary of the headline-only bench + agreement check against bench.py's own event timing.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_headline
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_headline -- python3 bench.py --headline-only --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_headline.json 2> gpurun_out/bench_headline.err
tail -2 gpurun_out/bench_headline.err
S=$(find gpurun_out/prof_headline -name "*kernel_stats.csv" | head -1)
cp "$S" gpurun_out/headline_kernel_stats.csv
find gpurun_out/prof_headline -name "*kernel_trace.csv" -delete
python tools/rocprof_vs_bench.py gpurun_out/headline_kernel_stats.csv gpurun_out/bench_headline.json | tee gpurun_out/rocprof_vs_bench.txt
