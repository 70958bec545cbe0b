#!/bin/bash
# round 5: the kernels' A/B switches exist in the diagnostic library only (csrc: make DIAG=1); the product library reads no environment
export TT_LIB_NAME=${TT_LIB_NAME:-libtt_hip_diag.so}
# What do the A-operand L2 misses cost the one-tile GEMM, and WHERE in the K-tile?  TT_GEMM_DEBUG_A0=1 makes every tile read
# row-block 0's A rows (wrong results, every A read an L2 hit); stamps of a mid-grid workgroup + whole-kernel times both ways.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export LD_LIBRARY_PATH=$PWD/tensor-truth_amd:$LD_LIBRARY_PATH
make -C tensor-truth_amd/csrc DIAG=1 -j4 > /dev/null 2>&1; make -C tools gemm_stamps gemm_bench_diag > /dev/null 2>&1
{
if [ -z "$SKIP_TIME" ]; then for a0 in 0 1 0 1; do
  echo "== TT_GEMM_DEBUG_A0=$a0: gemm_bench 473600 10"
  TT_GEMM_DEBUG_A0=$a0 timeout 120 tools/gemm_bench_diag 473600 10 | sed -n 2,8p
done; fi
for shape in "118272 3072 1024" "118272 1024 4096"; do for blk in 1000 2000; do for a0 in 0 1; do
  echo "== TT_GEMM_DEBUG_A0=$a0 stamps of workgroup $blk, K-tile 6: M N K = $shape"
  TT_GEMM_DEBUG_A0=$a0 TT_GEMM_STAMP_BLOCK=$blk timeout 120 tools/gemm_stamps $shape | tail -33
done; done; done
} 2>&1 | tee gpurun_out/gemm_a0_stamps.log
