#!/bin/bash
# Round 5: the tail-tile mask kept as a branch (product library, new) vs if-converted into 67 VALU on every tile (the diagnostic library as built BEFORE the change), streaming kernel.
cd "$GRAFT_REPO_ROOT" || exit 1
export LD_LIBRARY_PATH=$PWD/tensor-truth_amd:$LD_LIBRARY_PATH
make -C tools att_bench att_bench_diag > /dev/null 2>&1
for r in 1 2; do for len in 292 258 164 512; do
  echo "== new (branch), 1600 x $len: $(timeout 120 tools/att_bench 1600 $len 2>&1 | tail -1 | sed 's/.*dh=64: //')"
  echo "== old (selects), 1600 x $len: $(TT_ATT_RESIDENT=0 timeout 120 tools/att_bench_diag 1600 $len 2>&1 | tail -1 | sed 's/.*dh=64: //')"
done; done
timeout 1500 python -m pytest tests/test_encoder_gpu.py tests/test_f16_gpu.py tests/test_x3_gpu.py tests/test_f16c_gpu.py tests/test_rank_agreement_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | tail -3
