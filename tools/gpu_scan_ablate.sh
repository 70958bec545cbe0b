#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
MODES=${MODES:-1,33,65,129,257}
SCAN_BENCH_MODES=$MODES ./tools/scan_bench 1000000 1024 64 50 20 > gpurun_out/scan_ablate.log 2>&1
cat gpurun_out/scan_ablate.log
rm -rf gpurun_out/pmc1 gpurun_out/pmc2
SCAN_BENCH_MODES=1 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc1 -- ./tools/scan_bench 1000000 1024 64 50 3 > gpurun_out/pmc1.log 2>&1
SCAN_BENCH_MODES=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc2 -- ./tools/scan_bench 1000000 1024 64 50 3 > gpurun_out/pmc2.log 2>&1
ls -R gpurun_out/pmc1 | head; tail -3 gpurun_out/pmc1.log gpurun_out/pmc2.log
