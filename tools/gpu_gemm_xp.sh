#!/bin/bash
# round 5: the kernels' A/B switches exist in the diagnostic library only (csrc: make DIAG=1); the product library reads no environment
export TT_LIB_NAME=${TT_LIB_NAME:-libtt_hip_diag.so}
# A/B of the persistent-kernel experiment switches (TT_GEMM_XP) on the encoder GEMM shapes.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for xp in ${XPS:-0 0x1 0x101 0x201 0x103 0x203 0x100 0x200 0}; do
  echo "== TT_GEMM_XP=$xp"
  TT_GEMM_XP=$xp ./tools/gemm_bench ${M:-473600} ${ITERS:-10} | grep -v "small\|fp8"
done 2>&1 | tee gpurun_out/gemm_xp.log
