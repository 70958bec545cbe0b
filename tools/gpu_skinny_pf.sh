#!/bin/bash
# Round 5: one query's embedding with the skinny GEMM's prefetch depth / row tiles per wave varied (diagnostic library)
export TT_LIB_NAME=libtt_hip_diag.so
cd "$GRAFT_REPO_ROOT" || exit 1
for r in 1 2; do
echo "default (one row tile per wave, 24 K steps in flight): $(python3 tools/probes/query_embed_only.py 2>&1 | tail -1)"
echo "TT_GEMM_SKINNY_PF=32: $(TT_GEMM_SKINNY_PF=32 python3 tools/probes/query_embed_only.py 2>&1 | tail -1)"
echo "TT_GEMM_SKINNY_MT=2 (16 steps in flight): $(TT_GEMM_SKINNY_MT=2 python3 tools/probes/query_embed_only.py 2>&1 | tail -1)"
done
