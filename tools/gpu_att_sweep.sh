#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for nw in 5 4 2; do for bpc in 1 2 3 4 5 6 8 100; do echo -n "NW=$nw BPC=$bpc: "; TT_ATT_NW=$nw TT_ATT_BPC=$bpc timeout 60 ./tools/att_bench 800 292 20; done; done
} 2>&1 | tee gpurun_out/att_sweep.log
