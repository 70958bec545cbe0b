#!/bin/bash
# Round 5: resident attention (whole K / V8 of a (sequence, head) staged once, one workgroup walks all query blocks) vs the streaming
# kernel, by sequence length.  Both sides on the DIAGNOSTIC library (TT_ATT_RESIDENT=0|1|2; 2 = software-pipelined), then the parity suites on the product.
export TT_LIB_NAME=libtt_hip_diag.so
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export LD_LIBRARY_PATH=$PWD/tensor-truth_amd:$LD_LIBRARY_PATH
make -C tools att_bench_diag > /dev/null 2>&1
{
for r in 1 2 3; do for v in ${VARIANTS:-0 1 2}; do
  echo "== TT_ATT_RESIDENT=$v (round $r): 1600 x 292 tokens"
  TT_ATT_RESIDENT=$v timeout 120 tools/att_bench_diag 1600 292 2>&1 | tail -1
done; done
for len in 34 64 100 130 164 200 258 313; do for v in ${VARIANTS:-0 1 2}; do echo "== TT_ATT_RESIDENT=$v: 1600 x $len tokens"; TT_ATT_RESIDENT=$v timeout 120 tools/att_bench_diag 1600 $len 2>&1 | tail -1; done; done
} 2>&1 | tee gpurun_out/r05_attention_resident_ab.log
# parity with the pipelined resident kernel forced (diagnostic library): a row's bits must not depend on the kernel that served it
TT_ATT_RESIDENT=${PARITY_VARIANT:-2} timeout 1200 python -m pytest tests/test_encoder_gpu.py tests/test_f16_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | tail -4
