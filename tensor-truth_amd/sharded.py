"""Row-sharded exact search across the GPUs of one node (one process per GPU).

The reference has no distributed path; this is the MI355X-native
equivalent BASELINE.json config 4 asks for: GPU r holds rows
``[row_offset_r, row_offset_r + n_r)`` of the corpus matrix, every rank scans its shard for
the whole query batch, the per-shard partial top-k ``(score fp32, global idx int32)`` are
exchanged with ONE all-gather (RCCL over xGMI; ``Q*K*8`` bytes per rank -- latency-bound,
so queries are batched per collective), and every rank merges the ``world*K`` candidates of
its queries with the same exact top-k kernel the scan uses (ties: score desc, global index
asc -- identical to a single-GPU scan of the concatenated matrix).

Works with any initialised ``torch.distributed`` process group: ``nccl`` (= RCCL) on the
GPUs, ``gloo`` for the CPU tests of the merge protocol (the merge function is injected
there, because the product merge kernel needs a GPU).
"""
from __future__ import annotations

from typing import Callable, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced row ranges: the first ``n_total % world`` ranks get one extra row."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_partials(scores: torch.Tensor, idx: torch.Tensor, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """All-gather per-shard partial top-k lists.  scores [Q,K] fp32, idx [Q,K] int32 (global
    row ids, -1 = padding) -> ([Q, world*K], [Q, world*K]) with rank-major candidate order."""
    world = dist.get_world_size(group)
    if world == 1:
        return scores, idx
    q, k = scores.shape
    # one collective: pack both arrays into a single int32 buffer
    packed = torch.cat([scores.contiguous().view(torch.int32), idx.contiguous()], dim=1)  # [Q, 2K]
    out = torch.empty((world, q, 2 * k), dtype=torch.int32, device=packed.device)
    dist.all_gather_into_tensor(out.view(world * q, 2 * k), packed, group=group)
    all_s = out[:, :, :k].permute(1, 0, 2).reshape(q, world * k).contiguous().view(torch.float32)
    all_i = out[:, :, k:].permute(1, 0, 2).reshape(q, world * k).contiguous()
    return all_s, all_i


def sharded_topk(local_scan: Callable[[torch.Tensor, int], Tuple[torch.Tensor, torch.Tensor]],
                 merge: Callable[[torch.Tensor, torch.Tensor, int], Tuple[torch.Tensor, torch.Tensor]],
                 queries: torch.Tensor, k: int, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """queries: the SAME [Q, D] batch on every rank.  ``local_scan(queries, k)`` returns this
    rank's partial top-k with global indices; ``merge(scores, idx, k)`` reduces candidate lists."""
    s, i = local_scan(queries, k)
    all_s, all_i = gather_partials(s, i, group)
    if all_s.shape[1] == k:
        return all_s, all_i
    return merge(all_s, all_i, k)


class ShardedCorpus:
    """This rank's shard of a row-major bf16 corpus matrix resident in HBM."""

    def __init__(self, shard: torch.Tensor, row_offset: int, n_total: int, group=None):
        if shard.dtype != torch.bfloat16 or shard.dim() != 2 or not shard.is_contiguous():
            raise ValueError("shard must be a contiguous [n, D] bfloat16 matrix")
        if not shard.is_cuda:
            raise RuntimeError("ShardedCorpus needs a HIP device; tensor_truth_amd has no CPU path")
        self.shard = shard
        self.row_offset = int(row_offset)
        self.n_total = int(n_total)
        self.group = group

    def local_topk(self, queries: torch.Tensor, k: int):
        from . import scan

        return scan.scan_topk(self.shard, queries, k, idx_base=self.row_offset)

    def search(self, queries: torch.Tensor, k: int):
        """Exact global top-k for ``queries`` (identical on all ranks)."""
        from . import scan

        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return self.local_topk(queries, k)
        return sharded_topk(self.local_topk, scan.topk_merge, queries, k, self.group)


def gather_queries(local_q: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather each rank's [q, D] query embeddings into the [world*q, D] batch every shard scans."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_q
    world = dist.get_world_size(group)
    out = torch.empty((world * local_q.shape[0], local_q.shape[1]), dtype=local_q.dtype, device=local_q.device)
    dist.all_gather_into_tensor(out, local_q.contiguous(), group=group)
    return out
