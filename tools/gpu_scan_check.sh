#!/bin/bash
# One GPU-box visit: scan micro-bench (load modes / ablations), rocprof kernel stats, GPU parity tests.
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
MODES=${MODES:-1,17,33,65}
( SCAN_BENCH_MODES=$MODES ./tools/scan_bench 1000000 1024 64 50 20;
  SCAN_BENCH_MODES=$MODES ./tools/scan_bench 10000000 1024 64 50 5;
  SCAN_BENCH_MODES=1 ./tools/scan_bench 1250000 1024 64 50 10;
  SCAN_BENCH_MODES=1 ./tools/scan_bench 1000000 1024 16 50 10;
  SCAN_BENCH_MODES=1 ./tools/scan_bench 1000000 384 64 10 10 ) > gpurun_out/scan_bench.log 2>&1
cat gpurun_out/scan_bench.log
rm -rf gpurun_out/prof_scan
SCAN_BENCH_MODES=$MODES rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_scan -- ./tools/scan_bench 1000000 1024 64 50 10 > gpurun_out/prof_scan.log 2>&1
find gpurun_out/prof_scan -name "*kernel_stats*" | head -1 | xargs cat | head -20
if [ -z "$SKIP_TESTS" ]; then
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 | tee gpurun_out/pytest_gpu.log
fi
