#!/bin/bash
# Round 5: clock and power of the attention kernel variants (is the additive compute + traffic time an energy effect?)
export TT_LIB_NAME=libtt_hip_diag.so
cd "$GRAFT_REPO_ROOT" || exit 1
for v in "0 0" "1 0" "1 1" "1 4" "1 7" "3 0"; do set -- $v; TT_ATT_RESIDENT=$1 TT_ATT_RES_ABL=$2 timeout 120 python tools/probes/attention_power.py 2>&1 | tail -1; done
