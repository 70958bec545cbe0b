"""CPU, 2 processes over gloo: the row-sharded search protocol (shard bounds, one all-gather of
packed partial top-k, merge with global indices) equals a single exact scan of the whole corpus.
The local scan and the merge are the CPU oracle here (the product kernels need a GPU); what is
under test is tensor_truth_amd.sharded's exchange + index bookkeeping."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import scan as osc


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, d, q, k, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tensor_truth_amd import sharded

        corpus = osc.synth_corpus(n_total, d, seed=99)
        queries, _ = osc.synth_queries(corpus, q, seed=5)
        lo, hi = sharded.shard_bounds(n_total, world, rank)

        def local_scan(qs, kk):
            v, i, _ = osc.scan_topk(corpus[lo:hi], qs, kk)
            gi = torch.where(i >= 0, i + lo, i).to(torch.int32)
            return v, gi

        def merge(vals, idx, kk):
            v, i = osc.merge_topk(vals, idx.to(torch.int64), kk)
            return v, i.to(torch.int32)

        s, i = sharded.sharded_topk(local_scan, merge, queries, k)
        # gather_queries: each rank contributes its own slice of the batch
        mine = queries[rank::world].contiguous()
        allq = sharded.gather_queries(mine)
        want_q = torch.cat([queries[r::world] for r in range(world)], 0)
        ok_q = torch.equal(allq.view(torch.int16), want_q.view(torch.int16))
        if rank == 0:
            ret.put((s, i, ok_q))
        else:
            ret.put((None, None, ok_q))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total,k", [(1001, 10), (37, 50)])
def test_two_rank_sharded_search_equals_global(n_total, k):
    world, d, q = 2, 128, 6
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, d, q, k, ret)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [ret.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(o[2] for o in outs)
    s, i = next((o[0], o[1]) for o in outs if o[0] is not None)
    corpus = osc.synth_corpus(n_total, d, seed=99)
    queries, _ = osc.synth_queries(corpus, q, seed=5)
    want_s, want_i, gap = osc.scan_topk(corpus, queries, k)
    tie_free = gap > 1e-6
    assert torch.equal(i.to(torch.int64)[tie_free], want_i[tie_free])
    fin = torch.isfinite(want_s)
    assert torch.allclose(s[fin], want_s[fin], rtol=1e-5, atol=1e-6)
    assert (i[~fin] == -1).all()


def test_shard_bounds_cover_everything():
    from tensor_truth_amd.sharded import shard_bounds

    for n, w in ((10_000_000, 8), (7, 3), (5, 8), (0, 2)):
        spans = [shard_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
