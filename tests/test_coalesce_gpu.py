"""GPU: the query-coalescing front of the plugin surface.

The reference calls ``retrieve()`` / ``postprocess_nodes()`` one query at a time from executor threads
(``rag_engine.py:418-424``, ``api/routes/chat.py:367-374``, ``services/orchestrator_tool_wrappers.py:238-247``); here
32 request threads go through the SAME calls, the front merges them into shared embed / scan / rerank batches, and
every caller gets exactly -- bit for bit -- what it gets alone.  Also: the row-sharded index behind the same surface
 on one GPU as eight logical shards."""
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


def _texts(n, seed=3):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n):
        k = int(torch.randint(5, 40, (1,), generator=g))
        out.append(" ".join(WORDS[j] for j in torch.randint(0, len(WORDS), (k,), generator=g).tolist()) + f" doc{i}")
    return out


def _models():
    from tensor_truth_amd.embedding import HipHuggingFaceEmbedding
    from tensor_truth_amd.encoder import EncoderConfig
    from tensor_truth_amd.rerank import HipSentenceTransformerRerank

    emb = HipHuggingFaceEmbedding("test/bge-small-shaped", device="cuda",
                                  model_kwargs={"encoder_config": EncoderConfig(**SMALL), "synthetic_seed": 41})
    rr = HipSentenceTransformerRerank(model="test/xenc", top_n=5, device="cuda",
                                      model_kwargs={"encoder_config": EncoderConfig(**XENC), "synthetic_seed": 42})
    return emb, rr


def test_32_threads_through_the_surface_equal_serial_calls(dev, built_lib):
    from tensor_truth_amd.schema import QueryBundle, TextNode
    from tensor_truth_amd.vector_index import HipVectorIndex

    emb, rr = _models()
    texts = _texts(2000)
    index = HipVectorIndex(SMALL["hidden"], embed_model=emb)
    index.add([TextNode(text=t, id_=f"n{j}", metadata={"file_name": f"f{j % 9}.md", "title": f"t{j}"}) for j, t in enumerate(texts)])
    retr = index.as_retriever(similarity_top_k=12)                      # coalescing on (the default)
    lone = index.as_retriever(similarity_top_k=12, coalesce=False)
    queries = [" ".join(texts[11 * i + 5].split()[:7]) for i in range(96)]

    def run(r, q):
        nodes = r.retrieve(q)
        ranked = rr.postprocess_nodes(nodes, query_bundle=QueryBundle(query_str=q))
        return [(n.node.id_, n.score) for n in nodes], [(n.node.id_, n.score) for n in ranked]

    # serial reference: no front at all (coalesce=False retriever, a reranker front that only ever sees one caller)
    serial = [run(lone, q) for q in queries]
    b0, i0 = rr._front.batches, rr._front.items
    assert b0 == i0 == len(queries)                                    # one caller at a time -> batches of one
    got, errs = [None] * len(queries), []

    def work(t):
        try:
            for i in range(t, len(queries), 32):
                got[i] = run(retr, queries[i])
        except Exception as exc:  # noqa: BLE001
            errs.append(exc)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(32)]
    [t.start() for t in threads]
    [t.join(timeout=300) for t in threads]
    assert not errs, errs
    assert got == serial                                               # same ids, bitwise the same scores
    assert all(len(r[0]) == 12 and len(r[1]) == 5 for r in got)
    # the front did merge callers: fewer batches than calls, on both stages
    assert retr._front.items == len(queries) and retr._front.batches < len(queries)
    assert rr._front.items - i0 == len(queries) and rr._front.batches - b0 < len(queries)
    print(f"coalescing: {len(queries)} retrieve calls from 32 threads in {retr._front.batches} scan batches, "
          f"{rr._front.batches - b0} rerank batches")

    # a failing caller (empty query bundle string is fine; a missing bundle is not) does not poison its batch mates
    with pytest.raises(ValueError):
        rr.postprocess_nodes([], query_bundle=None)
    assert run(retr, queries[0]) == serial[0]


def test_sharded_index_behind_the_retriever_surface_on_one_gpu(dev, built_lib, tmp_path):
    """Row e2: ``ShardedHipVectorIndex.as_retriever().retrieve()`` with the device's rows cut into 8 logical shards
    (scan per shard -> partial top-k with global rows -> tt_topk_merge) returns what the single-matrix retriever
    returns, from memory and from a persisted directory; auto-merging works on top of it unchanged."""
    from tensor_truth_amd.retrievers import AutoMergingRetriever
    from tensor_truth_amd.schema import QueryBundle, TextNode
    from tensor_truth_amd.sharded_index import ShardedHipVectorIndex
    from tensor_truth_amd.vector_index import HipVectorIndex

    emb, _ = _models()
    n, d = 50_000, SMALL["hidden"]
    corpus = osc.synth_corpus(n, d, seed=77).to(dev)
    nodes = [TextNode(text=f"chunk {j}", id_=f"c{j}", metadata={"row": j}) for j in range(n)]
    plain = HipVectorIndex(d, embed_model=emb, score_mode="cosine")
    plain.add(nodes, embeddings=corpus.float())
    sharded = ShardedHipVectorIndex.from_index(plain, logical_shards=8)
    assert len(sharded._shards) == 8 and sum(s[0].shape[0] for s in sharded._shards) == n
    mat = plain.matrix.cpu()                                   # what the index stores (re-normalised, bf16)
    queries, planted = osc.synth_queries(mat, 24, seed=5)
    q_used = torch.nn.functional.normalize(queries.float(), dim=1).to(torch.bfloat16)   # the retriever's query rounding
    want_s, want_i, gap = osc.scan_topk(mat, q_used, 20)
    r_plain, r_shard = plain.as_retriever(20), sharded.as_retriever(20)
    for qi in range(24):
        qb = QueryBundle(query_str=f"q{qi}", embedding=queries[qi].float().tolist())
        a, b = r_plain.retrieve(qb), r_shard.retrieve(qb)
        assert [(h.node.id_, h.score) for h in a] == [(h.node.id_, h.score) for h in b]
        if gap[qi] > 1e-6:
            assert [h.node.metadata["row"] for h in b] == want_i[qi].tolist()      # bit-exact vs the oracle
    # text queries go through the embedder exactly as on the plain retriever
    assert [h.node.id_ for h in r_shard.retrieve("tensor kernel wave")] == [h.node.id_ for h in r_plain.retrieve("tensor kernel wave")]
    # persisted directory -> every rank loads only its rows (here: all of them), same answers
    plain.persist(str(tmp_path / "ix"))
    loaded = ShardedHipVectorIndex.load(str(tmp_path / "ix"), embed_model=emb, score_mode="cosine", logical_shards=8)
    qb = QueryBundle(query_str="q", embedding=queries[3].float().tolist())
    assert [(h.node.id_, h.score) for h in loaded.as_retriever(20).retrieve(qb)] == [(h.node.id_, h.score) for h in r_plain.retrieve(qb)]
    # the reference's wrapper on top (rag_engine.py:641-643)
    merged = AutoMergingRetriever(loaded.as_retriever(20), loaded.docstore).retrieve(qb)
    assert len(merged) == 20 and merged[0].score >= merged[-1].score
