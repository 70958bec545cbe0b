#!/bin/bash
# One GPU-box visit: scan micro-bench (both load modes), rocprof kernel stats, GPU parity tests.
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rocminfo | grep -E "gfx|Compute Unit" | head -4
( ./tools/scan_bench 1000000 1024 64 50 20 ) > gpurun_out/scan_bench.log 2>&1
cat gpurun_out/scan_bench.log
( ./tools/scan_bench 1000000 1024 16 50 10; ./tools/scan_bench 10000000 1024 64 50 5; ./tools/scan_bench 1000000 384 64 10 10 ) >> gpurun_out/scan_bench.log 2>&1
tail -20 gpurun_out/scan_bench.log
rm -rf gpurun_out/prof_scan
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_scan -- ./tools/scan_bench 1000000 1024 64 50 10 > gpurun_out/prof_scan.log 2>&1
find gpurun_out/prof_scan -name "*kernel_stats*" | head -1 | xargs cat | head -20
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 | tee gpurun_out/pytest_gpu.log
