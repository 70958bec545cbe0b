#!/bin/bash
# The three --pmc passes of tools/gpu_pmc_bench_r03.sh alone (HBM traffic: FETCH_SIZE, WRITE_SIZE; matrix-core utilisation) + converters.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
LEGS="--no-cpu-baseline --no-fp8-leg --no-reference-leg --no-fp16-leg --no-surface-leg --no-config5-leg"
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 $LEGS > gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 1 --warmup 1 $LEGS > gpurun_out/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -- python3 bench.py --steps 1 --warmup 1 $LEGS > gpurun_out/pmc_mfma.log 2>&1
python tools/pmc_to_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/r03_pmc_traffic.json
python tools/pmc_to_mfma.py gpurun_out/pmc_mfma gpurun_out/r03_pmc_mfma.json
find gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma -name "*.csv" -delete
