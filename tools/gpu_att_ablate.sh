#!/bin/bash
# round 5: the kernels' A/B switches exist in the diagnostic library only (csrc: make DIAG=1); the product library reads no environment
export TT_LIB_NAME=${TT_LIB_NAME:-libtt_hip_diag.so}
# What each part of the attention key-tile loop costs: diagnostic builds of attention_kernel (TT_ATT_ABLATE, wrong results by
# design) timed stand-alone on the bench shape (1600 sequences x 292 tokens, 16 heads x 64).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export LD_LIBRARY_PATH=$PWD/tensor-truth_amd:$LD_LIBRARY_PATH
make -C tensor-truth_amd/csrc DIAG=1 -j4 > /dev/null 2>&1; make -C tools att_bench_diag > /dev/null 2>&1
{
for r in 1 2; do for a in 0 1 2 3 4; do
  echo "== TT_ATT_ABLATE=$a (round $r)"
  TT_ATT_ABLATE=$a timeout 120 tools/att_bench_diag 1600 292 2>&1 | tail -2
done; done
for a in 0 4; do echo "== TT_ATT_ABLATE=$a with TT_ATT_XCD=1"; TT_ATT_XCD=1 TT_ATT_ABLATE=$a timeout 120 tools/att_bench_diag 1600 292 2>&1 | tail -1; done
} 2>&1 | tee gpurun_out/att_ablate.log
