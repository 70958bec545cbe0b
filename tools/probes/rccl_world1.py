"""RCCL sanity on the 1-GPU box: the collectives bench.py / sharded.py use, on a one-rank NCCL(=RCCL) group.
(The multi-rank exchange itself cannot run here; this checks library load, communicator creation and dtypes.)"""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

sys.path.insert(0, ".")
from tensor_truth_amd.sharded import ShardedCorpus, gather_partials, gather_queries  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
q = torch.nn.functional.normalize(torch.randn(8, 1024, device=dev), dim=1).to(torch.bfloat16)
out = torch.empty_like(q)
dist.all_gather_into_tensor(out, q)                       # bf16 payload through RCCL itself
packed = torch.arange(8 * 100, dtype=torch.int32, device=dev).view(8, 100)
o2 = torch.empty((1, 8, 100), dtype=torch.int32, device=dev)
dist.all_gather_into_tensor(o2.view(8, 100), packed)
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(out, q) and torch.equal(o2[0], packed) and float(t.item()) == 1.25
corpus = torch.nn.functional.normalize(torch.randn(50000, 1024, device=dev), dim=1).to(torch.bfloat16)
s, i = ShardedCorpus(corpus, 0, 50000).search(gather_queries(q), 10)
assert s.shape == (8, 10) and (i >= 0).all()
gs, gi = gather_partials(s, i)
assert torch.equal(gs, s) and torch.equal(gi, i)
dist.destroy_process_group()
print("rccl world=1 ok:", torch.version.hip, "nccl", torch.cuda.nccl.version())
