#!/bin/bash
# build_index_sharded end to end on the real kernels: 1, 2 and 3 ranks sharing GPU 0 over gloo (a 1-GPU box).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 TT_ONE_DEVICE=1
{
timeout 300 python tools/probes/sharded_ingest_ranks.py ${DOCS:-384} 4
for n in 2 3; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) \
    tools/probes/sharded_ingest_ranks.py ${DOCS:-384} 4
  echo "ranks=$n rc=$?"
done
} 2>&1 | grep -v "amdgpu.ids\|^W\|^\*\*\*\*\|OMP_NUM" | tee gpurun_out/sharded_ingest.log
