"""GPU, 2 processes over gloo sharing the one device: the HIP path under a process group.

``tests/test_sharded_gloo.py`` checks the multi-rank protocol on CPU with oracle stand-ins for the scan and the merge; here
the SAME protocol runs with nothing injected -- each rank holds its row range of the corpus in HBM, embeds its own callers'
questions (``tt_encoder_forward_cls``), scans (``tt_scan_topk``), merges the gathered partial lists (``tt_topk_merge``) and
reranks its own callers' pairs (``tt_rerank_head``) -- and every caller must get what one process gets alone over the whole
corpus.  Both query modes of SURVEY.md section 8e: "replicated" (the front end hands every rank the same request) and
"partitioned" behind the lock-step tick front (every rank serves different callers).  RCCL cannot run two ranks on one
device, so the transport here is gloo; what this covers is kernels + bookkeeping + collectives composed, not xGMI."""
import os
import socket
import threading

import pytest
import torch

from oracle import scan as osc

pytestmark = pytest.mark.gpu

SMALL = dict(arch="bert", vocab_size=3000, hidden=384, layers=2, heads=12, ffn=1536, max_pos=128, type_vocab=2,
             pad_id=0, ln_eps=1e-12)
XENC = dict(arch="xlmr", vocab_size=3000, hidden=256, layers=2, heads=4, ffn=512, max_pos=130, type_vocab=1,
            pad_id=1, ln_eps=1e-5, num_labels=1)
WORDS = ["tensor", "kernel", "wave", "matrix", "retrieval", "index", "corpus", "query", "rerank", "chunk", "gradient",
         "vector", "cache", "stream", "shard", "token", "layer", "norm", "attention", "softmax", "lattice", "quorum"]
N_ROWS, TOP_K, TOP_N, N_Q = 20_000, 12, 5, 22


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _questions():
    g = torch.Generator().manual_seed(17)
    out = []
    for i in range(N_Q):
        k = int(torch.randint(3, 12, (1,), generator=g))
        out.append(" ".join(WORDS[j] for j in torch.randint(0, len(WORDS), (k,), generator=g).tolist()) + f" q{i}")
    return out


def _node_text(j):
    return " ".join(WORDS[(j * 7 + t * 3) % len(WORDS)] for t in range(4 + j % 9)) + f" chunk{j}"


def _models():
    from tensor_truth_amd.embedding import HipHuggingFaceEmbedding
    from tensor_truth_amd.encoder import EncoderConfig
    from tensor_truth_amd.rerank import HipSentenceTransformerRerank

    emb = HipHuggingFaceEmbedding("test/bge-small-shaped", device="cuda",
                                  model_kwargs={"encoder_config": EncoderConfig(**SMALL), "synthetic_seed": 41})
    rr = HipSentenceTransformerRerank(model="test/xenc", top_n=TOP_N, device="cuda",
                                      model_kwargs={"encoder_config": EncoderConfig(**XENC), "synthetic_seed": 42})
    return emb, rr


def _corpus_and_nodes():
    from tensor_truth_amd.schema import TextNode

    corpus = osc.synth_corpus(N_ROWS, SMALL["hidden"], seed=123)
    leaf_ids = [f"c{j}" for j in range(N_ROWS)]
    docstore = {nid: TextNode(text=_node_text(j), id_=nid, metadata={"row": j}) for j, nid in enumerate(leaf_ids)}
    return corpus, leaf_ids, docstore


def _plain_index(emb):
    from tensor_truth_amd.vector_index import HipVectorIndex

    corpus, leaf_ids, docstore = _corpus_and_nodes()
    plain = HipVectorIndex(SMALL["hidden"], embed_model=emb, score_mode="cosine")
    plain.add([docstore[i] for i in leaf_ids], embeddings=corpus.float())
    return plain


def _answer(retr, rr, question):
    from tensor_truth_amd.schema import QueryBundle

    hits = retr.retrieve(question)
    top = rr.postprocess_nodes(hits, query_bundle=QueryBundle(query_str=question))
    return ([(h.node.id_, h.score) for h in hits], [(h.node.id_, h.score) for h in top])


def _rank_worker(rank, world, port, mode, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from tensor_truth_amd import _lib
        from tensor_truth_amd.sharded import shard_bounds
        from tensor_truth_amd.sharded_index import ShardedHipVectorIndex

        _lib.load_library()                                   # the product library or nothing: no stand-ins in this test
        emb, rr = _models()
        index = ShardedHipVectorIndex.from_index(_plain_index(emb), queries=mode)      # this rank keeps ITS row range only
        lo, hi = shard_bounds(N_ROWS, world, rank)
        assert index.row_lo == lo and sum(s[0].shape[0] for s in index._shards) == hi - lo
        questions = _questions()
        out = {}
        if mode == "replicated":
            # every rank is handed every request, in the same order (one collective round per call)
            retr = index.as_retriever(similarity_top_k=TOP_K)
            assert retr._tick is None
            for qi, q in enumerate(questions[:8]):
                out[qi] = _answer(retr, rr, q)
            rounds = None
        else:
            retr = index.as_retriever(similarity_top_k=TOP_K, max_batch=8)
            assert retr._tick is not None
            # rank 0: 4 threads x 3 questions; rank 1: 2 threads x 5 questions -- different callers, different counts
            mine = ([[3 * t + j for j in range(3)] for t in range(4)] if rank == 0
                    else [[12 + 5 * t + j for j in range(5)] for t in range(2)])
            lock, errs = threading.Lock(), []

            def caller(qs):
                try:
                    for qi in qs:
                        a = _answer(retr, rr, questions[qi])
                        with lock:
                            out[qi] = a
                except BaseException as exc:  # noqa: BLE001
                    errs.append(repr(exc))

            threads = [threading.Thread(target=caller, args=(qs,)) for qs in mine]
            for t in threads:
                t.start()
            for t in threads:
                t.join(timeout=300)
                assert not t.is_alive()
            assert errs == []
            front = retr._tick
            retr.close(timeout=300)
            rounds = (front.rounds, front.items)
        torch.cuda.synchronize()
        ret.put((rank, out, rounds))
    finally:
        dist.destroy_process_group()


def _serial_answers(question_ids):
    """One process, the whole corpus in one matrix, one caller: what every caller of the two-rank runs must receive."""
    emb, rr = _models()
    retr = _plain_index(emb).as_retriever(TOP_K, coalesce=False)
    questions = _questions()
    return {qi: _answer(retr, rr, questions[qi]) for qi in question_ids}


def _same(got, want, qi):
    (g_hits, g_top), (w_hits, w_top) = got, want
    # the scan: a row's cosine is the same bf16 dot product whichever shard holds it -> identical scores; ids identical
    # wherever neighbouring scores differ (equal scores may swap between a shard-wise and a whole-corpus selection)
    assert [s for _, s in g_hits] == [s for _, s in w_hits], f"question {qi}: retrieved scores differ"
    ws = [s for _, s in w_hits]
    for r, ((gid, _), (wid, _)) in enumerate(zip(g_hits, w_hits)):
        tied = (r > 0 and ws[r - 1] == ws[r]) or (r + 1 < len(ws) and ws[r + 1] == ws[r])
        assert gid == wid or tied, f"question {qi}: rank {r} holds {gid}, serial run {wid}"
    if [i for i, _ in g_hits] == [i for i, _ in w_hits]:
        assert g_top == w_top, f"question {qi}: reranked list differs"      # same pairs -> bit-identical reranker scores


@pytest.mark.parametrize("mode", ["replicated", "partitioned"])
def test_two_ranks_with_the_hip_kernels_equal_one_process(dev, built_lib, mode):
    import torch.multiprocessing as mp

    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_worker, args=(r, world, port, mode, ret)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        outs = {o[0]: o[1:] for o in (ret.get(timeout=600) for _ in procs)}
    finally:
        for p in procs:
            p.join(timeout=120)
    assert all(p.exitcode == 0 for p in procs)
    if mode == "replicated":
        want = _serial_answers(range(8))
        assert set(outs[0][0]) == set(outs[1][0]) == set(range(8))
        for qi in range(8):
            assert outs[0][0][qi] == outs[1][0][qi]                     # both ranks hold the same merged answer
            _same(outs[0][0][qi], want[qi], qi)
    else:
        want = _serial_answers(range(N_Q))
        assert set(outs[0][0]) == set(range(12)) and set(outs[1][0]) == set(range(12, 22))   # each caller answered on its own rank
        for rank in (0, 1):
            for qi, got in outs[rank][0].items():
                _same(got, want[qi], qi)
        assert outs[0][1][0] == outs[1][1][0]                             # the same collective rounds on both ranks
        assert outs[0][1][1] == 12 and outs[1][1][1] == 10                # ... each embedding / serving only its own callers
        assert outs[0][1][0] < 22                                         # rounds were shared by concurrent callers


def test_bench_self_launches_two_ranks_and_falls_back_from_rccl_by_agreement(built_lib):
    """`python bench.py --gpus 2` as the driver calls it at N > 1, on this 1-GPU box: the parent starts two ranks (a child process,
    never exec), both share GPU 0 (TT_BENCH_ONE_DEVICE=1) and ATTEMPT the RCCL data plane (TT_BENCH_TRY_NCCL=1) -- RCCL refuses two
    ranks on one device, the ranks agree on the failure over gloo and the step's collectives run on the gloo group: one JSON line,
    exit 0, `ranks: 2`, `n_gpus: 1`, the backend that carried the step named with RCCL's own error text.  (Reduced model depth and
    corpus: this checks the start-up, the sharded step and the line, not a rate.)"""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TT_BENCH_ONE_DEVICE="1", TT_BENCH_TRY_NCCL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--layers", "2",
                        "--corpus-rows", "600000", "--headline-only", "--no-cpu-baseline"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["ranks"] == 2 and d["n_gpus"] == 1 and d["value"] > 0 and d["scaling"] == "weak"
    cb = d["config"]["collective_backend"]
    assert cb.startswith("gloo (nccl pre-flight failed") and "Duplicate GPU" in cb, cb
    assert d["config"]["ranks_share_one_device"] is True
    assert "nccl data plane unusable" in r.stderr


def test_bench_self_launches_eight_ranks_on_one_device(built_lib):
    """the driver's N = 8 command -- `python bench.py --gpus 8` -- has never met an 8-GPU node, so a regression in
    the launcher, the shard bounds, the 8-way gathered scan batch or the merge must be caught HERE: eight self-launched ranks share
    GPU 0 over gloo (TT_BENCH_ONE_DEVICE=1), 800 k rows (100 k per shard), 2 layers, one JSON line with `ranks: 8`.  Checks start-up,
    the sharded step and the line, not a rate."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TT_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "TT_BENCH_TRY_NCCL"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1", "--layers", "2",
                        "--corpus-rows", "800000", "--queries-per-gpu", "8", "--headline-only", "--no-cpu-baseline"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["ranks"] == 8 and d["n_gpus"] == 1 and d["value"] > 0 and d["scaling"] == "weak" and d["steps"] == 1
    assert d["config"]["ranks_share_one_device"] is True and d["config"]["collective_backend"].startswith("gloo")
    assert d["config"]["parallelism"].startswith("corpus row-sharded x8")
