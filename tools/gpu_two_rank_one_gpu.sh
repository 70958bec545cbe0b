#!/bin/bash
# The N > 1 code path of bench.py on a 1-GPU box: 2 and 4 ranks sharing GPU 0 over gloo (TT_BENCH_ONE_DEVICE=1).
# Not a scaling measurement -- a check that the sharded index, the two all-gathers, the merge and the JSON line work.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TT_BENCH_ONE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 2 4; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) \
    bench.py --gpus $n --steps 2 --warmup 1 --corpus-rows 2000000 --layers 4 --no-cpu-baseline --no-fp8-leg 2> gpurun_out/two_rank_$n.err | tail -1 > gpurun_out/two_rank_$n.json
  echo "ranks=$n rc=$?"; tail -3 gpurun_out/two_rank_$n.err; python - <<PY
import json
d = json.loads(open("gpurun_out/two_rank_$n.json").read())
print({k: d[k] for k in ("value", "n_gpus", "ms_per_step")}, d["config"]["scan_only"], d["roofline_scan"]["queries_per_launch"], d["roofline_scan"]["reread_factor"])
PY
done
