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
            # numpy, not torch: a tensor travels through the queue as a handle into THIS process's sharing server, and
            # the parent may unpickle it after this process has exited (FileNotFoundError, seen once in a CPU-suite run)
            ret.put((s.numpy(), i.numpy(), ok_q))
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
    s, i = next((torch.from_numpy(o[0]), torch.from_numpy(o[1])) for o in outs if o[0] is not None)
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


def _retriever_worker(rank, world, port, n_total, d, k, mode, ret):
    """Drive the protocol THROUGH the plugin surface: ShardedHipVectorIndex.as_retriever().retrieve() on every rank
    (row e2 of the scope table: the reference builds this object at rag_engine.py:626-645).  The scan and merge kernels
    are replaced by the CPU oracle (they need a GPU); sharding, both query modes, the exchange, the global-row -> node
    mapping and the score mapping are the product's."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tensor_truth_amd.schema import QueryBundle, TextNode
        from tensor_truth_amd.sharded import shard_bounds
        from tensor_truth_amd.sharded_index import ShardedHipVectorIndex

        corpus = osc.synth_corpus(n_total, d, seed=7)
        queries, _ = osc.synth_queries(corpus, 2 * world, seed=9)
        lo, hi = shard_bounds(n_total, world, rank)
        leaf_ids = [f"leaf{j}" for j in range(n_total)]
        docstore = {nid: TextNode(text=f"text {j}", id_=nid, metadata={"row": j}) for j, nid in enumerate(leaf_ids)}

        def scan_fn(rows, q16, kk, base):
            v, i, _ = osc.scan_topk(rows, q16, kk)
            return v, torch.where(i >= 0, i + base, i).to(torch.int32)

        def merge_fn(vals, idx, kk):
            v, i = osc.merge_topk(vals, idx.to(torch.int64), kk)
            return v, i.to(torch.int32)

        index = ShardedHipVectorIndex(d, corpus[lo:hi].contiguous(), lo, n_total, leaf_ids, docstore, score_mode="cosine",
                                      logical_shards=3, queries=mode, scan_fn=scan_fn, merge_fn=merge_fn)
        # replicated: every rank asks the same two questions; partitioned: rank r asks questions 2r, 2r+1
        mine = [0, 1] if mode == "replicated" else [2 * rank, 2 * rank + 1]
        if mode == "partitioned":    # (direct search() rounds: before the retriever's lock-step front owns the collectives)
            # ranks may bring DIFFERENT numbers of queries to one collective round (rank 0: three, rank 1: one)
            emb = queries[[0, 1, 2]] if rank == 0 else queries[[3]]
            s_r, i_r = index.search(emb.float(), k)
            want = [0, 1, 2] if rank == 0 else [3]
            assert s_r.shape == (len(want), k)
            ws, wi, wgap = osc.scan_topk(corpus, queries[want], k)
            for j in range(len(want)):
                if wgap[j] > 1e-6:
                    assert i_r[j].tolist() == wi[j].tolist()
        retr = index.as_retriever(similarity_top_k=k)
        if mode == "partitioned":
            with pytest.raises(RuntimeError):
                index.search(queries[[0]].float(), k)          # the front owns the collectives now
        out = []
        for qi in mine:
            hits = retr.retrieve(QueryBundle(query_str=f"q{qi}", embedding=queries[qi].float().tolist()))
            out.append((qi, [(h.node.id_, h.score, h.node.metadata["row"]) for h in hits]))
        retr.close(timeout=120)
        ret.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["replicated", "partitioned"])
def test_two_rank_retrieve_through_the_plugin_surface(mode):
    world, n_total, d, k = 2, 777, 128, 12
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_retriever_worker, args=(r, world, port, n_total, d, k, mode, ret)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict(ret.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    corpus = osc.synth_corpus(n_total, d, seed=7)
    queries, _ = osc.synth_queries(corpus, 2 * world, seed=9)
    want_s, want_i, gap = osc.scan_topk(corpus, queries, k)
    seen = set()
    for rank, res in outs.items():
        for qi, hits in res:
            seen.add((rank, qi))
            assert len(hits) == k
            assert [h[1] for h in hits] == sorted((h[1] for h in hits), reverse=True)
            assert all(h[0] == f"leaf{h[2]}" for h in hits)                   # node looked up by GLOBAL row
            if gap[qi] > 1e-6:
                assert [h[2] for h in hits] == want_i[qi].tolist()
            assert torch.allclose(torch.tensor([h[1] for h in hits]), want_s[qi], rtol=1e-5, atol=1e-6)
    want_seen = {(r, q) for r in range(world) for q in ([0, 1] if mode == "replicated" else [2 * r, 2 * r + 1])}
    assert seen == want_seen
    if mode == "replicated":      # identical answers on every rank
        assert outs[0] == outs[1]


def _from_local_worker(rank, world, port, sizes, d, k, ret):
    """Replica-parallel ingest (BASELINE config 5 on N GPUs): every rank holds ONLY the rows and nodes of the documents it
    ingested (ragged shard sizes, one of them empty); ShardedHipVectorIndex.from_local stitches them into one global
    index -- counts + host side tables exchanged once, no matrix row moves.  Scan / merge = CPU oracle stand-ins."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tensor_truth_amd.schema import QueryBundle, TextNode
        from tensor_truth_amd.sharded_index import ShardedHipVectorIndex

        n_total = sum(sizes)
        corpus = osc.synth_corpus(n_total, d, seed=21)
        queries, _ = osc.synth_queries(corpus, 4, seed=22)
        lo = sum(sizes[:rank])
        hi = lo + sizes[rank]
        # this rank's own leaves and a parent node per 5 leaves (hierarchy links must survive the exchange)
        leaf_ids = [f"r{rank}-leaf{j}" for j in range(lo, hi)]
        docstore = {}
        for j, nid in zip(range(lo, hi), leaf_ids):
            nd = TextNode(text=f"text {j}", id_=nid, metadata={"row": j, "file_name": f"f{j}.md"})
            nd.excluded_embed_metadata_keys = ["file_name"]
            nd.parent_id = f"r{rank}-parent{j // 5}"
            docstore[nid] = nd
            par = docstore.setdefault(nd.parent_id, TextNode(text=f"parent of {j // 5}", id_=nd.parent_id, metadata={}))
            par.child_ids = list(getattr(par, "child_ids", None) or []) + [nid]

        def scan_fn(rows, q16, kk, base):
            v, i, _ = osc.scan_topk(rows, q16, kk)
            return v, torch.where(i >= 0, i + base, i).to(torch.int32)

        def merge_fn(vals, idx, kk):
            v, i = osc.merge_topk(vals, idx.to(torch.int64), kk)
            return v, i.to(torch.int32)

        index = ShardedHipVectorIndex.from_local(d, corpus[lo:hi].contiguous(), leaf_ids, docstore, score_mode="cosine",
                                                 scan_fn=scan_fn, merge_fn=merge_fn)
        assert index.n_total == n_total and index.row_lo == lo and len(index.leaf_ids) == n_total
        retr = index.as_retriever(similarity_top_k=k)
        out = []
        for qi in range(4):
            hits = retr.retrieve(QueryBundle(query_str=f"q{qi}", embedding=queries[qi].float().tolist()))
            out.append([(h.node.id_, h.node.metadata["row"], h.node.text, h.node.parent_id,
                         list(h.node.excluded_embed_metadata_keys), h.score) for h in hits])
        # a parent that lives on ANOTHER rank resolves here too (what AutoMergingRetriever walks)
        other = (rank + 1) % world
        while sizes[other] == 0:
            other = (other + 1) % world
        other_lo = sum(sizes[:other])
        par = index.docstore[f"r{other}-parent{other_lo // 5}"]
        ret.put((rank, out, len(par.child_ids) >= 1))
    finally:
        dist.destroy_process_group()


def test_three_rank_index_from_rank_local_ingest():
    world, sizes, d, k = 3, [203, 0, 120], 128, 9
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_from_local_worker, args=(r, world, port, sizes, d, k, ret)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [ret.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_total = sum(sizes)
    corpus = osc.synth_corpus(n_total, d, seed=21)
    queries, _ = osc.synth_queries(corpus, 4, seed=22)
    want_s, want_i, gap = osc.scan_topk(corpus, queries, k)
    owner = lambda row: 0 if row < sizes[0] else 2      # noqa: E731  (rank 1 ingested nothing)
    assert all(o[2] for o in outs)
    assert outs[0][1] == outs[1][1] == outs[2][1]        # replicated queries: identical answers on every rank
    for qi, hits in enumerate(outs[0][1]):
        assert len(hits) == k
        for nid, row, text, parent, excl, _score in hits:
            assert nid == f"r{owner(row)}-leaf{row}" and text == f"text {row}"
            assert parent == f"r{owner(row)}-parent{row // 5}" and excl == ["file_name"]
        if gap[qi] > 1e-6:
            assert [h[1] for h in hits] == want_i[qi].tolist()
        assert torch.allclose(torch.tensor([h[5] for h in hits]), want_s[qi], rtol=1e-5, atol=1e-6)


def _tick_worker(rank, world, port, n_total, d, k, ret):
    """The multi-rank serving front: every rank has ITS OWN request threads (different counts, different questions, one
    rank idle for a while, one malformed query), the ranks' collective rounds stay aligned by the tick protocol, and
    every caller gets the serial answer.  Scan / merge = CPU oracle stand-ins (the kernels need a GPU)."""
    import threading
    import time

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tensor_truth_amd.schema import QueryBundle, TextNode
        from tensor_truth_amd.sharded import shard_bounds
        from tensor_truth_amd.sharded_index import ShardedHipVectorIndex

        corpus = osc.synth_corpus(n_total, d, seed=7)
        queries, _ = osc.synth_queries(corpus, 24, seed=9)
        lo, hi = shard_bounds(n_total, world, rank)
        leaf_ids = [f"leaf{j}" for j in range(n_total)]
        docstore = {nid: TextNode(text=f"text {j}", id_=nid, metadata={"row": j}) for j, nid in enumerate(leaf_ids)}

        def scan_fn(rows, q16, kk, base):
            v, i, _ = osc.scan_topk(rows, q16, kk)
            return v, torch.where(i >= 0, i + base, i).to(torch.int32)

        def merge_fn(vals, idx, kk):
            v, i = osc.merge_topk(vals, idx.to(torch.int64), kk)
            return v, i.to(torch.int32)

        index = ShardedHipVectorIndex(d, corpus[lo:hi].contiguous(), lo, n_total, leaf_ids, docstore, score_mode="cosine",
                                      queries="partitioned", scan_fn=scan_fn, merge_fn=merge_fn)
        retr = index.as_retriever(similarity_top_k=k, max_batch=4)
        assert retr._tick is not None                      # coalescing stays ON at world > 1
        # rank 0: 5 threads x 3 questions (0..14); rank 1: idle for a moment, then 2 threads x 2 questions (15..18) + one malformed
        mine = {0: [[3 * t + j for j in range(3)] for t in range(5)], 1: [[15, 16], [17, 18]]}[rank]
        out, errs = {}, []
        lock = threading.Lock()

        def caller(qs):
            for qi in qs:
                hits = retr.retrieve(QueryBundle(query_str=f"q{qi}", embedding=queries[qi].float().tolist()))
                with lock:
                    out[qi] = [(h.node.id_, h.score, h.node.metadata["row"]) for h in hits]

        def bad_caller():
            try:
                retr.retrieve(QueryBundle(query_str="no embedding and the index has no embed_model"))
            except Exception as exc:  # noqa: BLE001
                errs.append(type(exc).__name__)

        threads = [threading.Thread(target=caller, args=(qs,)) for qs in mine]
        if rank == 1:
            time.sleep(0.3)                                   # rank 0's rounds run with rank 1 bringing no queries
            threads.append(threading.Thread(target=bad_caller))
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=120)
            assert not t.is_alive()
        front = retr._tick
        retr.close(timeout=120)                               # rank 1 closes early and keeps scanning for rank 0 until it closes too
        assert not front._thread.is_alive()
        with pytest.raises(RuntimeError):
            retr.retrieve(QueryBundle(query_str="late", embedding=queries[0].float().tolist()))
        ret.put((rank, out, errs, front.rounds, front.items))
    finally:
        dist.destroy_process_group()


def test_two_rank_tick_front_serves_different_callers_per_rank():
    world, n_total, d, k = 2, 555, 128, 7
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tick_worker, args=(r, world, port, n_total, d, k, ret)) for r in range(world)]
    for p in procs:
        p.start()
    outs = {o[0]: o[1:] for o in (ret.get(timeout=240) for _ in procs)}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    corpus = osc.synth_corpus(n_total, d, seed=7)
    queries, _ = osc.synth_queries(corpus, 24, seed=9)
    want_s, want_i, gap = osc.scan_topk(corpus, queries, k)
    assert set(outs[0][0]) == set(range(15)) and set(outs[1][0]) == {15, 16, 17, 18}     # every caller answered, on its own rank
    for rank in (0, 1):
        for qi, hits in outs[rank][0].items():
            assert len(hits) == k and all(h[0] == f"leaf{h[2]}" for h in hits)
            if gap[qi] > 1e-6:
                assert [h[2] for h in hits] == want_i[qi].tolist()
            assert torch.allclose(torch.tensor([h[1] for h in hits]), want_s[qi], rtol=1e-5, atol=1e-6)
    assert outs[0][1] == [] and outs[1][1] == ["ValueError"]          # the malformed query failed alone, before any collective
    assert outs[0][2] == outs[1][2]                                    # both ranks ran the same collective rounds
    assert outs[0][3] == 15 and outs[1][3] == 4 and outs[0][2] < 19    # rounds were shared by concurrent callers


def _poison_worker(rank, world, port, n_total, d, k, ret):
    """A shard scan that fails inside a collective round: the failing rank completes the round with a poisoned partial list, the
    round's callers on EVERY rank get ShardRoundError, nobody hangs in a collective, and the front serves the next queries."""
    import threading

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tensor_truth_amd.schema import QueryBundle, TextNode
        from tensor_truth_amd.sharded import shard_bounds
        from tensor_truth_amd.sharded_index import ShardedHipVectorIndex, ShardRoundError

        corpus = osc.synth_corpus(n_total, d, seed=17)
        queries, _ = osc.synth_queries(corpus, 12, seed=19)
        lo, hi = shard_bounds(n_total, world, rank)
        leaf_ids = [f"leaf{j}" for j in range(n_total)]
        docstore = {nid: TextNode(text=f"text {j}", id_=nid, metadata={"row": j}) for j, nid in enumerate(leaf_ids)}
        calls = {"n": 0}

        def scan_fn(rows, q16, kk, base):
            calls["n"] += 1
            if rank == 1 and calls["n"] == 2:
                raise RuntimeError("boom: device error in the shard scan")
            v, i, _ = osc.scan_topk(rows, q16, kk)
            return v, torch.where(i >= 0, i + base, i).to(torch.int32)

        def merge_fn(vals, idx, kk):
            v, i = osc.merge_topk(vals, idx.to(torch.int64), kk)
            return v, i.to(torch.int32)

        index = ShardedHipVectorIndex(d, corpus[lo:hi].contiguous(), lo, n_total, leaf_ids, docstore, score_mode="cosine",
                                      queries="partitioned", scan_fn=scan_fn, merge_fn=merge_fn)
        retr = index.as_retriever(similarity_top_k=k, max_batch=4)
        out, errs = {}, []

        def caller(qs):
            for qi in qs:
                try:
                    hits = retr.retrieve(QueryBundle(query_str=f"q{qi}", embedding=queries[qi].float().tolist()))
                    out[qi] = [h.node.metadata["row"] for h in hits]
                except ShardRoundError as exc:
                    errs.append((qi, type(exc).__name__, repr(exc.__cause__)))

        t = threading.Thread(target=caller, args=([6 * rank + j for j in range(6)],))
        t.start()
        t.join(timeout=120)
        assert not t.is_alive()
        front = retr._tick
        assert front._dead is None                      # the front survived the failed round ...
        late = retr.retrieve(QueryBundle(query_str="late", embedding=queries[rank].float().tolist()))   # ... and still serves
        retr.close(timeout=120)
        assert not front._thread.is_alive()
        ret.put((rank, out, errs, [h.node.metadata["row"] for h in late], front.rounds, calls["n"]))
    finally:
        dist.destroy_process_group()


def test_two_rank_tick_front_survives_a_failed_shard_scan():
    world, n_total, d, k = 2, 400, 64, 5
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_poison_worker, args=(r, world, port, n_total, d, k, ret)) for r in range(world)]
    for p in procs:
        p.start()
    outs = {o[0]: o[1:] for o in (ret.get(timeout=240) for _ in procs)}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    corpus = osc.synth_corpus(n_total, d, seed=17)
    queries, _ = osc.synth_queries(corpus, 12, seed=19)
    _, want_i, gap = osc.scan_topk(corpus, queries, k)
    n_err = 0
    for rank in (0, 1):
        out, errs, late, rounds, n_scans = outs[rank]
        n_err += len(errs)
        assert set(out) | {e[0] for e in errs} == set(range(6 * rank, 6 * rank + 6))        # every call returned or raised
        for qi, rows in out.items():
            if gap[qi] > 1e-6:
                assert rows == want_i[qi].tolist()
        if gap[rank] > 1e-6:
            assert late == want_i[rank].tolist()
        assert all(e[1] == "ShardRoundError" for e in errs)
    assert 1 <= n_err <= 2                                   # the poisoned round held one or two callers (one per rank at most)
    assert any("boom" in e[2] for e in outs[1][1]) or not outs[1][1]                        # rank 1's own callers see the cause
    assert outs[0][3] == outs[1][3]                          # the ranks stayed in lock step
