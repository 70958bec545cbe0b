#!/bin/bash
# round 5: the kernels' A/B switches exist in the diagnostic library only (csrc: make DIAG=1); the product library reads no environment
export TT_LIB_NAME=${TT_LIB_NAME:-libtt_hip_diag.so}
# GEMM write-back A/B (VERDICT r03 item 7 i): the bias epilogue's output head-major (a wave's 32 rows x 128 B = one 4-KiB run) vs row-major
# (32 lines 2 N bytes apart), diagnostic library; with TT_GEMM_DEBUG_TRAFFIC=1 (every C row -> row 0: no write-back at all) as the bound.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export LD_LIBRARY_PATH=$PWD/tensor-truth_amd:$LD_LIBRARY_PATH
make -C tensor-truth_amd/csrc DIAG=1 -j8 > /dev/null 2>&1; make -C tools gemm_bench_diag > /dev/null 2>&1
{
for r in 1 2; do
for v in "TT_GEMM_HEAD_MAJOR=0" "TT_GEMM_HEAD_MAJOR=1" "TT_GEMM_DEBUG_TRAFFIC=1"; do
  echo "== $v (round $r): gemm_bench 473600 10, bias-epilogue shapes"
  env $v timeout 200 tools/gemm_bench_diag 473600 10 | grep -E "^qkv|bias only" | head -4
done; done
} 2>&1 | tee gpurun_out/gemm_writeback_ab.log
