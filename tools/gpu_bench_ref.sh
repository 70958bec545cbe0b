#!/bin/bash
# The bench with the reference-precision leg only (+ headline): stage split of the f16c step.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --no-cpu-baseline --no-surface-leg --no-config5-leg --no-fp8-leg --no-fp16-leg ${BENCH_ARGS} > gpurun_out/bench_ref.json 2> gpurun_out/bench_ref.err
tail -3 gpurun_out/bench_ref.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/bench_ref.json"))
c = j["config"]
print("headline", j["value"], "ms/step", j["ms_per_step"], "stage", j["stage_ms_per_step"])
r = c["reference_precision"]
print("reference", r["queries_per_s"], "ms/step", r["ms_per_step"], "stage", r["stage_ms_per_step"], "gemm launches", r["gemm_launches_per_step"])
f = r["fast_variant_f16c"]
print("reference fast (f16c)", f["queries_per_s"], "ms/step", f["ms_per_step"], "stage", f["stage_ms_per_step"])
print("quality", r["score_quality_vs_fp32_path"])
PY
