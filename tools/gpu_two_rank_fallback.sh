#!/bin/bash
# bench.py's N > 1 start on a 1-GPU box: `python bench.py --gpus 2` (self-launched ranks sharing GPU 0).
#  (a) TT_BENCH_TRY_NCCL=1: the ranks attempt the RCCL data plane -- RCCL refuses two ranks on one device -- agree on the failure and run the
#      step's collectives over gloo: the fallback path on real hardware, `collective_backend` says why;
#  (b) plain TT_BENCH_ONE_DEVICE=1: gloo from the start.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TT_BENCH_ONE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for mode in try_nccl gloo; do
  if [ $mode = try_nccl ]; then export TT_BENCH_TRY_NCCL=1; else unset TT_BENCH_TRY_NCCL; fi
  timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-fp8-leg --no-reference-leg --no-fp16-leg --no-config5-leg \
      2> gpurun_out/two_rank_$mode.err | tail -1 > gpurun_out/two_rank_$mode.json
  echo "mode=$mode rc=${PIPESTATUS[0]}"
  grep -E "data plane|pre-flight|retry|Error|error" gpurun_out/two_rank_$mode.err | head -6 | cut -c1-400
  python - <<PY
import json
d = json.loads(open("gpurun_out/two_rank_$mode.json").read())
print({k: d[k] for k in ("value", "n_gpus", "ms_per_step")}, {k: d["config"].get(k) for k in ("ranks", "collective_backend", "ranks_share_one_device")})
PY
done
