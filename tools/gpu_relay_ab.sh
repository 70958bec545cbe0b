#!/bin/bash
# round 6: relay skinny kernel vs the one-wave skinny kernel (diagnostic library, TT_GEMM_RELAY=0) per shape at M = 64 (one query)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for M in 64 256; do
  echo "=== relay (default)"; ./tools/gemm_bench_diag $M 200
  echo "=== one-wave kernel (TT_GEMM_RELAY=0)"; TT_GEMM_RELAY=0 ./tools/gemm_bench_diag $M 200
done 2>&1 | grep -v "^fp8\|small" | tee gpurun_out/r06_relay_ab.log
