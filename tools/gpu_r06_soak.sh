#!/bin/bash
# round 6: the plugin surface under sustained load on the round-6 tree (Unigram tokenizer, pair pool started in the background, per-call token
# sources), the two-rank RCCL-refusal fallback, and the lone caller once more (scan_pass roofline in the breakdown)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
TOKENIZER=unigram-250k timeout 900 python tools/probes/surface_soak.py 6 2>&1 | tail -18 | tee gpurun_out/r06_surface_soak.log
bash tools/gpu_two_rank_fallback.sh 2>&1 | tee gpurun_out/r06_two_ranks_fallback.log
bash tools/gpu_lone_caller.sh final | head -12
