#!/bin/bash
# A/B of an environment switch on the end-to-end bench, same box, alternating runs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for rep in 1 2; do
for v in "" "$AB_ENV"; do
  echo "== env: [$v]"
  env $v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp8-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms_per_step']['gemm'])"
done; done 2>&1 | tee gpurun_out/bench_ab.log
