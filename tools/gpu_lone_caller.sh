#!/bin/bash
# round 6: lone-caller latency + per-stage breakdown (tools/probes/lone_caller.py) -> gpurun_out/r06_lone_caller*.json
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
TAG=${1:-baseline}
timeout 900 python tools/probes/lone_caller.py 10000000 50 10 bf16,default > gpurun_out/r06_lone_caller_$TAG.json 2> gpurun_out/r06_lone_caller_$TAG.err
tail -3 gpurun_out/r06_lone_caller_$TAG.err
python - <<PY
import json
for line in open("gpurun_out/r06_lone_caller_$TAG.json"):
    d = json.loads(line)
    print(d["mode"], d["precision"], "lone", round(d["single_caller_ms_per_query"], 2), "ms; with leaf ids", d["single_caller_ms_per_query_with_leaf_token_ids"])
    for k, b in (d["lone_caller_breakdown"] or {}).items():
        print(" ", k, "wall", {a: round(v, 2) for a, v in b["wall_ms"].items()})
        print("    host", {a: round(v, 2) for a, v in b["host_ms"].items()})
        print("    gpu ", {ph: {f: (round(v["ms"], 2), v["launches"]) for f, v in fam.items()} for ph, fam in b["gpu_kernel_ms"].items()})
PY
