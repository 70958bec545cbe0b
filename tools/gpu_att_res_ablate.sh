#!/bin/bash
# Round 5: what bounds the resident attention kernel (three structures land at the same time)?  Parts removed one at a time (wrong results by design).
export TT_LIB_NAME=libtt_hip_diag.so
cd "$GRAFT_REPO_ROOT" || exit 1
export LD_LIBRARY_PATH=$PWD/tensor-truth_amd:$LD_LIBRARY_PATH
make -C tools att_bench_diag > /dev/null 2>&1
for abl in 0 1 2 4 3 7; do echo "== TT_ATT_RESIDENT=1 TT_ATT_RES_ABL=$abl (1 = no stores, 2 = no Q loads, 4 = no K/V copies): 1600 x 292"; TT_ATT_RESIDENT=1 TT_ATT_RES_ABL=$abl timeout 120 tools/att_bench_diag 1600 292 2>&1 | tail -1; done
for abl in 0 7; do echo "== TT_ATT_RESIDENT=2 (pipelined) TT_ATT_RES_ABL=$abl: 1600 x 292"; TT_ATT_RESIDENT=2 TT_ATT_RES_ABL=$abl timeout 120 tools/att_bench_diag 1600 292 2>&1 | tail -1; done
for len in 258 313; do for v in 1 2; do echo "== TT_ATT_RESIDENT=$v TT_ATT_RES_ABL=7: 1600 x $len"; TT_ATT_RESIDENT=$v TT_ATT_RES_ABL=7 timeout 120 tools/att_bench_diag 1600 $len 2>&1 | tail -1; done; done
