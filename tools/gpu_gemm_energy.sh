#!/bin/bash
# Energy budget of the bf16 GEMM under the power cap (diagnostic library, wrong results by design): the sustained rate with random
# operands when parts of the main loop are removed from the third K-tile on -- what each part costs in energy per flop.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp TT_LIB_NAME=libtt_hip_diag.so
out=gpurun_out/gemm_energy.log
: > $out
run() { timeout 120 python tools/probes/gemm_sustain.py bf16 5 ${DATA:-randn} 2>&1 | tail -1 >> $out; }
for rep in 1 2; do
  TT_GEMM_ENERGY=0 run
  TT_GEMM_ENERGY=1 run
  TT_GEMM_ENERGY=2 run
  TT_GEMM_ENERGY=3 run
  TT_GEMM_DEBUG_A0=1 run
  TT_GEMM_DEBUG_TRAFFIC=1 run
done
DATA=const TT_GEMM_ENERGY=0 run
DATA=const TT_GEMM_ENERGY=3 run
cat $out
