#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q -k "attention or forward or cls" 2>&1 | tail -5
for nw in 0 4 8; do for bpc in 0 2; do echo "== TT_ATT_NW=$nw TT_ATT_BPC=$bpc"; TT_ATT_NW=$nw TT_ATT_BPC=$bpc timeout 60 ./tools/att_bench 800 292 20; done; done
for ab in 1 2 3; do echo "== TT_ATT_ABLATE=$ab (1 no softmax, 2 no restaging, 3 no PV)"; TT_ATT_ABLATE=$ab timeout 60 ./tools/att_bench 800 292 20; done
echo "== len 256 / 512 / 34"; ./tools/att_bench 800 256 20; ./tools/att_bench 400 512 20; ./tools/att_bench 4096 34 20
} 2>&1 | tee gpurun_out/att_ablate.log
