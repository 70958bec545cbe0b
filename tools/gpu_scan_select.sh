#!/bin/bash
# Scan / select tests + the scan legs of the bench (no encoders' side legs).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_scan_gpu.py tests/test_scan_sweep_gpu.py tests/test_abi_sweeps_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_scan.log
timeout 900 python bench.py --no-cpu-baseline --no-surface-leg --no-config5-leg --no-fp8-leg --no-reference-leg --no-fp16-leg > gpurun_out/bench_scan.json 2> gpurun_out/bench_scan.err
tail -3 gpurun_out/bench_scan.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/bench_scan.json"))
c = j["config"]
print("headline", j["value"], "ms/step", j["ms_per_step"], "stage", j["stage_ms_per_step"])
print("scan_only", {k: v for k, v in c["scan_only"].items() if k != "what"})
print("shard", {k: v for k, v in c["scan_only_shard"].items() if k != "what"})
PY
