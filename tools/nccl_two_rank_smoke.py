#!/usr/bin/env python3
"""Two (or N) ranks over RCCL: the collectives the sharded index uses, checked against what they must return.

    python tools/nccl_two_rank_smoke.py [--ranks 2]           # starts its own ranks (child process, one per GPU)

Each rank r holds a [Q, K] block of packed (score, global row) partials as `ShardedHipVectorIndex.search` does; the
ranks all-gather them (`dist.all_gather_into_tensor`, backend "nccl" = RCCL) and every rank checks every block;
then a barrier and an all-reduce MAX (the bench's max-over-ranks timing).  Any failure prints RCCL's own error text
and exits non-zero -- nothing is caught and papered over.  On a box with fewer GPUs than ranks it says so and exits 3
(RCCL cannot place two ranks on one device); `TT_BENCH_ONE_DEVICE=1 python bench.py --gpus 2` is the gloo stand-in there.
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: this pool's driver has no legacy IPC handles


def parse(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    return ap.parse_args(argv)


def launcher(args):
    import socket

    import torch      # device_count() does not initialise the GPU on this image (counting is safe before spawning)

    n_dev = torch.cuda.device_count()
    if n_dev < args.ranks:
        print(f"nccl smoke: {args.ranks} ranks need {args.ranks} GPUs, this box has {n_dev}: RCCL cannot share a device between ranks")
        raise SystemExit(3)

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), "--ranks", str(args.ranks)]
    raise SystemExit(subprocess.run(cmd).returncode)


def rank_main(args):
    import torch
    import torch.distributed as dist

    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    n_dev = torch.cuda.device_count()
    if n_dev < world:
        if rank == 0:
            print(f"nccl smoke: {world} ranks need {world} GPUs, this box has {n_dev}: RCCL cannot share a device between ranks")
        raise SystemExit(3)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)           # an RCCL failure raises here or in the first collective
    Q, K = 256, 50
    mine = torch.empty((Q, K, 2), dtype=torch.float32, device=dev)
    mine[..., 0] = torch.arange(Q * K, device=dev, dtype=torch.float32).view(Q, K) + 1000.0 * rank
    mine[..., 1] = float(rank)
    out = torch.empty((world, Q, K, 2), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(out.view(-1), mine.view(-1))
    torch.cuda.synchronize(dev)
    for r in range(world):
        want = torch.arange(Q * K, device=dev, dtype=torch.float32).view(Q, K) + 1000.0 * r
        assert torch.equal(out[r, ..., 0], want) and bool((out[r, ..., 1] == float(r)).all()), f"rank {rank}: block {r} wrong"
    t = torch.tensor([float(rank + 1)], dtype=torch.float64, device=dev)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == float(world)
    # ragged query counts, as the tick-driven serving front gathers them
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([rank * 3 + 1], dtype=torch.int64, device=dev))
    assert [int(c.item()) for c in counts] == [r * 3 + 1 for r in range(world)]
    ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    if rank == 0:
        print(f"nccl smoke ok: {world} ranks, RCCL {ver}, all_gather_into_tensor of {Q}x{K} packed partials, barrier, all_reduce MAX, all_gather")
    dist.destroy_process_group()


if __name__ == "__main__":
    a = parse(sys.argv[1:])
    if "RANK" in os.environ:
        rank_main(a)
    else:
        launcher(a)
