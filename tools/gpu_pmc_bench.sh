#!/bin/bash
# PMC passes (HBM traffic) + kernel stats of the bench command, for profiles/.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/prof_bench
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fp8-leg > gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fp8-leg > gpurun_out/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp8-leg > gpurun_out/prof_bench.log 2>&1
timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
cat gpurun_out/bench.json
ls gpurun_out/pmc_fetch/*/ | head
