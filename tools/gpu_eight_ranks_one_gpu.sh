#!/bin/bash
# The driver's N = 8 command on a 1-GPU box: `python bench.py --gpus 8` self-launches 8 ranks, here all on GPU 0 over gloo
# (TT_BENCH_ONE_DEVICE=1), at the FULL configuration (10 M x 1024 rows sharded 8 ways = one GPU's 8-GPU shard each, 24 layers).
# Not a scaling measurement (eight processes share one GPU): it checks that the 8-rank path -- shard bounds, 256 gathered
# queries through the tiled scan per shard, both all-gathers, the merge, the lock-step serving front with 8 x 32 caller threads,
# max-over-ranks timing, the JSON line -- runs end to end.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TT_BENCH_ONE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python bench.py --gpus 8 --steps 2 --warmup 1 --no-cpu-baseline --no-fp8-leg --no-reference-leg --no-fp16-leg --surface-leg \
  > gpurun_out/eight_ranks.json 2> gpurun_out/eight_ranks.err
echo "rc=$?"; tail -4 gpurun_out/eight_ranks.err
python - <<PY
import json
d = json.loads(open("gpurun_out/eight_ranks.json").readline())
c = d["config"]
print({k: d[k] for k in ("value", "n_gpus", "ms_per_step", "scaling")})
print("workload:", c["workload"])
print("ranks_share_one_device:", c["ranks_share_one_device"], "| ranks:", d.get("ranks"), "| collective_backend:", c.get("collective_backend"))
print("surface:", {k: v for k, v in (c.get("plugin_surface") or {}).items() if k != "what"})
print("roofline_scan:", {k: d["roofline_scan"][k] for k in ("queries_per_launch", "avg_launch_ms", "frac")})
PY
