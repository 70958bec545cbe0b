#!/bin/bash
# round 5: the kernels' A/B switches exist in the diagnostic library only (csrc: make DIAG=1); the product library reads no environment
export TT_LIB_NAME=${TT_LIB_NAME:-libtt_hip_diag.so}
# Attention tail tiles: live 32-row blocks dealt to different waves per (sequence, head) (TT_ATT_ROTATE, default on) vs plain order.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export LD_LIBRARY_PATH=$PWD/tensor-truth_amd:$LD_LIBRARY_PATH
make -C tools att_bench > /dev/null 2>&1
{
for r in 1 2 3; do for v in 0 1; do
  echo "== TT_ATT_ROTATE=$v (round $r): 1600 x 292 tokens"
  TT_ATT_ROTATE=$v timeout 120 tools/att_bench 1600 292 2>&1 | tail -1
done; done
for len in 160 200 420; do for v in 0 1; do echo "== TT_ATT_ROTATE=$v: 1600 x $len tokens"; TT_ATT_ROTATE=$v timeout 120 tools/att_bench 1600 $len 2>&1 | tail -1; done; done
} 2>&1 | tee gpurun_out/att_rotate.log
timeout 900 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q 2>&1 | tail -3
