#!/bin/bash
# round 5: the kernels' A/B switches exist in the diagnostic library only (csrc: make DIAG=1); the product library reads no environment
export TT_LIB_NAME=${TT_LIB_NAME:-libtt_hip_diag.so}
# Attention: five waves (160 query rows) per workgroup vs four (128), by sequence length; then the parity tests with the default choice.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export LD_LIBRARY_PATH=$PWD/tensor-truth_amd:$LD_LIBRARY_PATH
make -C tools att_bench > /dev/null 2>&1
{
for r in 1 2 3; do for v in ${WAVES:-4 5}; do
  echo "== TT_ATT_WAVES=$v (round $r): 1600 x 292 tokens"
  TT_ATT_WAVES=$v timeout 120 tools/att_bench 1600 292 2>&1 | tail -1
done; done
for len in 64 130 160 200 258 320 420 512; do for v in ${WAVES:-4 5}; do echo "== TT_ATT_WAVES=$v: 1600 x $len tokens"; TT_ATT_WAVES=$v timeout 120 tools/att_bench 1600 $len 2>&1 | tail -1; done; done
} 2>&1 | tee gpurun_out/att_waves.log
timeout 900 python -m pytest tests/test_encoder_gpu.py tests/test_f16_gpu.py tests/test_f16c_gpu.py -m gpu -x -q 2>&1 | tail -3
TT_ATT_WAVES=5 timeout 900 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q 2>&1 | tail -3
