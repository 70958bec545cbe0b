#!/bin/bash
# round 6: per-shape GEMM times at the lone caller's sizes (50 x 292-token rerank: M = 14848; the reference's 10 pairs: M = 3072),
# today's kernels (256x256 tiles above 128 tiles, the 128x128 v1 kernel below) and the v1 kernel forced (TT_GEMM_VARIANT=3, diag)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for M in 3072 7424 14848 29696 59392; do
  ./tools/gemm_bench $M 50
  echo "--- v1 (128x128) forced"
  TT_GEMM_VARIANT=3 ./tools/gemm_bench_diag $M 50 | head -8
done 2>&1 | tee gpurun_out/r06_midm_baseline.log
