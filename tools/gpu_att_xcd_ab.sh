#!/bin/bash
# round 6: the attention kernel fetches 6.7 GB per launch for 2.9 GB of Q + K + V (profiles/r06_pmc_traffic.json): the XCD-aware workgroup
# order (TT_ATT_XCD=1, diagnostic library) removes the re-fetch; round 2 measured it 1.8 % SLOWER inside the encoder.  Re-measured on this
# tree: headline-only bench, alternating, two runs each.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TT_LIB_NAME=libtt_hip_diag.so
for rep in 1 2; do
  for x in 0 1; do
    TT_ATT_XCD=$x timeout 600 python bench.py --headline-only --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['stage_ms_per_step']
print('TT_ATT_XCD=$x  %.1f q/s  step %.1f ms  attention %.2f ms  gemm %.1f ms  clock %s MHz' % (d['value'], d['ms_per_step'], s['attention'], s['gemm'], d['roofline']['clock']['sclk_mhz_median']))"
  done
done 2>&1 | tee gpurun_out/r06_attention_xcd_ab.log
