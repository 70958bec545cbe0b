"""bench.open_data_plane on REAL RCCL with the one rank a 1-GPU box has: gloo default group, an RCCL group created in the attempt
thread (device bound there), the value-checked pre-flight on it, the agreement over gloo -> ("nccl" group, "nccl", False)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
import bench

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", bench._free_port())
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("gloo", rank=0, world_size=1)
g, label, hung = bench.open_data_plane(dist, torch, dev, 0, 1, backend="nccl", deadline_s=120.0)
print("data plane:", "own group" if g is not None else "default gloo group", "| label:", label, "| hung:", hung)
assert g is not None and label == "nccl" and not hung
x = torch.arange(8, device=dev, dtype=torch.float32)
out = torch.empty(8, device=dev)
dist.all_gather_into_tensor(out, x, group=g)
assert torch.equal(out, x)
print("RCCL", torch.cuda.nccl.version(), "one-rank group: all_gather_into_tensor through the data-plane group ok")
dist.destroy_process_group()
