#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_names
TT_GEMM_TRACE=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_names -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-fp8-leg > gpurun_out/prof_names.log 2>&1
grep "gemm launch" gpurun_out/prof_names.log | sort | uniq -c | sort -rn | head -20
f=$(find gpurun_out/prof_names -name "*kernel_stats*" | head -1)
cut -d, -f1-4 $f | grep -i "gemm" | cut -c1-150
