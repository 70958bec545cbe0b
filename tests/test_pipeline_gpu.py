"""GPU: the plugin surface end to end (embedding model -> index/retriever -> rerank postprocessor
-> retrieval service) against the CPU oracle on the same token ids."""
import math

import pytest
import torch

from oracle import encoder as oe
from oracle import scan as osc

pytestmark = pytest.mark.gpu

SMALL = dict(arch="bert", vocab_size=3000, hidden=384, layers=2, heads=12, ffn=1536, max_pos=128, type_vocab=2,
             pad_id=0, ln_eps=1e-12)
XENC = dict(arch="xlmr", vocab_size=3000, hidden=256, layers=2, heads=4, ffn=512, max_pos=130, type_vocab=1,
            pad_id=1, ln_eps=1e-5, num_labels=1)


def _texts(n):
    words = ["tensor", "kernel", "wave", "matrix", "retrieval", "index", "corpus", "query", "rerank", "chunk",
             "gradient", "vector", "cache", "stream", "shard", "token", "layer", "norm", "attention", "softmax"]
    g = torch.Generator().manual_seed(3)
    out = []
    for i in range(n):
        k = int(torch.randint(5, 40, (1,), generator=g))
        idx = torch.randint(0, len(words), (k,), generator=g).tolist()
        out.append(" ".join(words[j] for j in idx) + f" doc{i}")
    return out


def _pad(seqs, pad):
    L = max(len(s) for s in seqs)
    ids = torch.full((len(seqs), L), pad, dtype=torch.int64)
    mask = torch.zeros(len(seqs), L, dtype=torch.int64)
    for b, s in enumerate(seqs):
        ids[b, : len(s)] = torch.tensor(s)
        mask[b, : len(s)] = 1
    return ids, mask


def test_embedding_index_retriever_roundtrip(dev, built_lib, tmp_path):
    from tensor_truth_amd.embedding import HipHuggingFaceEmbedding
    from tensor_truth_amd.encoder import EncoderConfig
    from tensor_truth_amd.schema import TextNode
    from tensor_truth_amd.vector_index import HipVectorIndex

    cfg = EncoderConfig(**SMALL)
    ocfg = oe.EncoderConfig(**SMALL)
    emb = HipHuggingFaceEmbedding("test/bge-small-shaped", device="cuda",
                                  model_kwargs={"encoder_config": cfg, "synthetic_seed": 21}, embed_batch_size=64)
    assert emb.embed_batch_size == 64 and emb.query_instruction == ""
    texts = _texts(300)
    nodes = [TextNode(text=t, id_=f"n{i}", metadata={"filename": f"f{i % 7}.md", "doc_type": "library"})
             for i, t in enumerate(texts)]
    for nd in nodes:
        nd.excluded_embed_metadata_keys = ["filename", "doc_type"]
    index = HipVectorIndex(cfg.hidden, embed_model=emb, score_mode="cosine")
    index.add(nodes)
    assert index.n == 300

    # oracle embeddings of the very same token ids (bf16 weights, bf16 rounding emulated)
    W = {k: v.to(torch.bfloat16) for k, v in oe.synth_weights(ocfg, seed=21).items()}
    seqs = [emb._tokenizer.encode(t, emb.max_length) for t in texts]
    ids, mask = _pad(seqs, cfg.pad_id)
    want = oe.embed(ids, mask, W, ocfg, emulate_bf16=True)
    got = torch.tensor(emb.get_text_embedding_batch(texts))
    assert ((got * want).sum(1) >= 0.9995).all()
    one = torch.tensor(emb.get_text_embedding(texts[5]))
    assert torch.allclose(one, got[5], atol=2e-3)
    assert abs(HipHuggingFaceEmbedding.similarity(got[0].tolist(), got[0].tolist()) - 1.0) < 1e-5

    # retrieval: same neighbours as the oracle scan over the oracle's embeddings wherever tie-free
    query = texts[17]
    res = index.as_retriever(similarity_top_k=10).retrieve(query)
    assert len(res) == 10 and res[0].node.id_ == "n17" and res[0].score > 0.99
    assert [r.score for r in res] == sorted((r.score for r in res), reverse=True)
    qv = got[17:18].to(torch.bfloat16)
    w_s, w_i, gap = osc.scan_topk(index.matrix.cpu(), qv, 10)
    if gap[0] > 1e-4:
        assert [r.node.id_ for r in res] == [f"n{int(j)}" for j in w_i[0]]
    res[0].node.metadata["_source_index"] = 3            # mutable per-hit metadata, not shared
    assert "_source_index" not in index.docstore["n17"].metadata

    # persist / load / delete
    index.persist(str(tmp_path / "indexes" / "bge-small" / "library_x"), embedding_model="test/bge-small-shaped")
    loaded = HipVectorIndex.load(str(tmp_path / "indexes" / "bge-small" / "library_x"), embed_model=emb,
                                 score_mode="cosine")
    assert loaded.n == 300 and torch.equal(loaded.matrix.cpu().view(torch.int16), index.matrix.cpu().view(torch.int16))
    assert [r.node.id_ for r in loaded.as_retriever(10).retrieve(query)] == [r.node.id_ for r in res]
    assert loaded.delete(["n17"]) == 1 and loaded.num_live == 299
    assert loaded.as_retriever(10).retrieve(query)[0].node.id_ != "n17"
    # chroma-style score mapping
    chroma = HipVectorIndex.load(str(tmp_path / "indexes" / "bge-small" / "library_x"), embed_model=emb)
    top = chroma.as_retriever(3).retrieve(query)[0]
    assert top.score == pytest.approx(math.exp(-(2 - 2 * res[0].score)), rel=1e-5)


def test_rerank_postprocessor_and_service(dev, built_lib):
    from tensor_truth_amd import model_manager as mm
    from tensor_truth_amd.encoder import EncoderConfig
    from tensor_truth_amd.rerank import HipSentenceTransformerRerank
    from tensor_truth_amd.retrieval_service import build_retrieval_service
    from tensor_truth_amd.schema import NodeWithScore, QueryBundle, TextNode
    from tensor_truth_amd.vector_index import HipVectorIndex

    cfg = EncoderConfig(**XENC)
    ocfg = oe.EncoderConfig(**XENC)
    rr = HipSentenceTransformerRerank(model="test/xenc", top_n=3, device="cuda",
                                      model_kwargs={"encoder_config": cfg, "synthetic_seed": 31})
    texts = _texts(12)
    nodes = [NodeWithScore(node=TextNode(text=t, id_=f"n{i}"), score=0.5) for i, t in enumerate(texts)]
    query = "which kernel streams the corpus"
    out = rr.postprocess_nodes(list(nodes), query_bundle=QueryBundle(query_str=query))
    assert len(out) == 3 and all(isinstance(n.score, float) for n in out)
    assert [n.score for n in out] == sorted((n.score for n in out), reverse=True)
    assert rr.postprocess_nodes([], QueryBundle(query_str=query)) == []           # positional bundle, empty input
    with pytest.raises(ValueError):
        rr.postprocess_nodes(list(nodes))
    # scores vs the oracle on the same token ids
    W = {k: v.to(torch.bfloat16) for k, v in oe.synth_weights(ocfg, seed=31).items()}
    pair_ids = [rr._tokenizer.encode_pair(query, t, 512)[0] for t in texts]
    ids, mask = _pad(pair_ids, cfg.pad_id)
    want = oe.rerank_scores(ids, mask, W, ocfg, emulate_bf16=True)
    got = torch.tensor(rr.predict([(query, t) for t in texts]))
    assert (got - want).abs().max().item() < 1.5e-2
    ranked = rr.rerank(query, texts, top_n=4)
    assert len(ranked) == 4 and ranked[0]["relevance_score"] == pytest.approx(got.max().item())
    # ordering / top-3 membership wherever the oracle separates candidates by more than twice the score bound:
    # checked on every separable pair, never skipped (rank_checks.py)
    from rank_checks import assert_order_on_separable, assert_topn_on_separable
    assert_order_on_separable(want.numpy(), got.numpy(), 3e-2, "postprocess_nodes order")
    assert_topn_on_separable(want.numpy(), got.numpy(), 3, 3e-2, "postprocess_nodes top-3")
    by_id = {int(n.node.id_[1:]): n.score for n in out}
    assert sorted(by_id, key=lambda i: -got[i].item())[:3] == [int(n.node.id_[1:]) for n in out]   # returned = top-3 of the scores
    assert all(by_id[i] == pytest.approx(got[i].item(), abs=1e-6) for i in by_id)

    # whole service through the ModelManager (weights resolved from per-model overrides)
    mm.ModelManager.reset_instance()
    mgr = mm.ModelManager.get_instance()
    small = EncoderConfig(**SMALL)
    mgr.model_kwargs_overrides["test/bge-small-shaped"] = {"encoder_config": small, "synthetic_seed": 21}
    mgr.model_kwargs_overrides["test/xenc"] = {"encoder_config": cfg, "synthetic_seed": 31}
    emb = mgr.get_embedder("test/bge-small-shaped", "cuda")
    assert mgr.get_embedder("test/bge-small-shaped", "cuda") is emb
    indexes = []
    for part in range(2):
        ix = HipVectorIndex(small.hidden, embed_model=emb)
        ix.add([TextNode(text=t, id_=f"i{part}_{j}", metadata={"filename": f"p{part}.md"})
                for j, t in enumerate(_texts(60)[part * 30:(part + 1) * 30])])
        indexes.append(ix)
    params = {"reranker_model": "test/xenc", "reranker_top_n": 4, "confidence_cutoff": 0.35,
              "confidence_cutoff_hard": 0.05, "balance_strategy": "top_k_per_index"}
    svc = build_retrieval_service(indexes, params, device="cuda", manager=mgr)
    res = svc.retrieve("attention softmax kernel")
    assert 0 < res.num_sources <= 4 and res.confidence_level in ("normal", "low")
    assert res.metrics["coverage"]["total_chunks"] == res.num_sources
    assert {n.node.metadata["_source_index"] for n in res.source_nodes} <= {0, 1}
    usage = mgr.get_memory_usage()
    assert usage["embedder_bytes"] > 0 and usage["reranker_bytes"] > 0
    mm.ModelManager.reset_instance()


def test_multi_index_embeds_the_query_once(dev, built_lib):
    """MultiIndexRetriever over indexes that share one embed model: one encoder pass per query instead of one per
    index (reference behaviour), same nodes and scores either way."""
    from tensor_truth_amd.embedding import HipHuggingFaceEmbedding
    from tensor_truth_amd.encoder import EncoderConfig
    from tensor_truth_amd.retrievers import AutoMergingRetriever, MultiIndexRetriever
    from tensor_truth_amd.schema import TextNode
    from tensor_truth_amd.vector_index import HipVectorIndex

    cfg = EncoderConfig(**SMALL)
    emb = HipHuggingFaceEmbedding("test/bge-small-shaped", device="cuda",
                                  model_kwargs={"encoder_config": cfg, "synthetic_seed": 21})
    texts = _texts(90)
    retrievers = []
    for part in range(3):
        ix = HipVectorIndex(cfg.hidden, embed_model=emb)
        ix.add([TextNode(text=t, id_=f"i{part}_{j}") for j, t in enumerate(texts[part * 30:(part + 1) * 30])])
        retrievers.append(AutoMergingRetriever(ix.as_retriever(similarity_top_k=5), ix.docstore))
    calls = {"n": 0}
    orig = emb.get_agg_embedding_from_queries

    def counting(qs):
        calls["n"] += 1
        return orig(qs)

    emb.get_agg_embedding_from_queries = counting
    dev_calls = {"n": 0}
    orig_dev = emb.query_embedding_device

    def counting_dev(qs):
        dev_calls["n"] += 1
        return orig_dev(qs)

    emb.query_embedding_device = counting_dev
    shared = MultiIndexRetriever(retrievers, enable_cache=False).retrieve("matrix kernel wave")
    assert calls["n"] == 1 and dev_calls["n"] == 0
    separate = MultiIndexRetriever(retrievers, enable_cache=False, share_query_embedding=False).retrieve("matrix kernel wave")
    assert dev_calls["n"] == 3
    key = lambda n: (n.node.metadata["_source_index"], n.node.id_)  # noqa: E731
    assert sorted(map(key, shared)) == sorted(map(key, separate))
    by_id = {key(n): n.score for n in separate}
    assert all(abs(n.score - by_id[key(n)]) < 1e-5 for n in shared)


def test_multi_index_single_pass_scan_equals_per_index_searches(dev, built_lib, monkeypatch):
    """All modules packed into one matrix and searched with ONE segmented pass
    returns what one search per module returns -- through auto-merging, tagging and balancing -- including an
    empty module, unequal sizes, different top-k per module, and a module mutated after packing."""
    from tensor_truth_amd import scan as tscan
    from tensor_truth_amd.retrievers import AutoMergingRetriever, MultiIndexRetriever
    from tensor_truth_amd.schema import QueryBundle, TextNode
    from tensor_truth_amd.vector_index import HipVectorIndex

    dim = 256
    g = torch.Generator().manual_seed(17)

    class FixedEmbed:  # stands in for the embedding model: deterministic vectors, no encoder needed here
        model_name = "fixed"

        def get_agg_embedding_from_queries(self, qs):
            v = torch.randn(dim, generator=torch.Generator().manual_seed(len(qs[0])))
            return (v / v.norm()).tolist()

    emb = FixedEmbed()
    sizes, topk = [37, 0, 400, 9], [5, 5, 12, 20]
    retrievers, indexes = [], []
    for part, n in enumerate(sizes):
        ix = HipVectorIndex(dim, embed_model=emb, score_mode="chroma")
        leaves = [TextNode(text=f"m{part} leaf {j}", id_=f"m{part}_{j}") for j in range(n)]
        parents = []
        for p0 in range(0, n, 4):  # parents of 4 consecutive leaves, prev/next chain: auto-merge has work to do
            kids = leaves[p0:p0 + 4]
            par = TextNode(text=f"m{part} parent {p0 // 4}", id_=f"m{part}_p{p0 // 4}")
            par.child_ids = [c.id_ for c in kids]
            for a, c in enumerate(kids):
                c.parent_id = par.id_
                c.prev_id = kids[a - 1].id_ if a else None
                c.next_id = kids[a + 1].id_ if a + 1 < len(kids) else None
            parents.append(par)
        ix.add_to_docstore(parents)
        if n:
            base = torch.randn(max(1, n // 4 + 1), dim, generator=g)   # siblings share a direction: merges happen
            vecs = base.repeat_interleave(4, 0)[:n] + 0.15 * torch.randn(n, dim, generator=g)
            ix.add(leaves, embeddings=vecs)
        indexes.append(ix)
        retrievers.append(AutoMergingRetriever(ix.as_retriever(similarity_top_k=topk[part]), ix.docstore))

    calls = {"seg": 0, "one": 0}
    orig_seg, orig_one = tscan.scan_topk_segmented, tscan.scan_topk
    monkeypatch.setattr(tscan, "scan_topk_segmented", lambda *a, **k: (calls.__setitem__("seg", calls["seg"] + 1), orig_seg(*a, **k))[1])
    monkeypatch.setattr(tscan, "scan_topk", lambda *a, **k: (calls.__setitem__("one", calls["one"] + 1), orig_one(*a, **k))[1])

    def canon(nodes):
        return [(n.node.id_, n.node.metadata.get("_source_index"), round(n.score, 6)) for n in nodes]

    for query in ("short", "a somewhat longer query string"):
        calls.update(seg=0, one=0)
        fast = MultiIndexRetriever(retrievers, enable_cache=False).retrieve(query)
        assert calls == {"seg": 1, "one": 0}
        slow = MultiIndexRetriever(retrievers, enable_cache=False, single_pass=False).retrieve(query)
        assert calls["one"] == 3                                  # the empty module never reaches the scan
        assert sorted(canon(fast)) == sorted(canon(slow)) and len(fast) > 0
        assert [n.score for n in fast] == sorted((n.score for n in fast), reverse=True)
        assert any(n.node.id_.split("_")[1].startswith("p") for n in fast), "expected at least one merged parent"

    # mutate a packed module: the group repacks, the member keeps working on its own too
    mir = MultiIndexRetriever(retrievers, enable_cache=False)
    before = mir.retrieve("short")
    qv = torch.tensor(emb.get_agg_embedding_from_queries(["short"]))
    indexes[3].add([TextNode(text="planted", id_="m3_planted")], embeddings=qv[None])
    after = mir.retrieve("short")
    assert after[0].node.id_ == "m3_planted" and abs(after[0].score - 1.0) < 1e-2
    alone = indexes[3].as_retriever(similarity_top_k=1).retrieve(QueryBundle(query_str="short", embedding=qv.tolist()))
    assert alone[0].node.id_ == "m3_planted"
    indexes[3].delete(["m3_planted"])
    again = mir.retrieve("short")
    assert canon(again) == canon(before)


def test_text_ingest_pipeline_equals_single_shot(dev, built_lib, tmp_path):
    """Long inputs are embedded as a pipeline (background tokenization of the next windows, pinned staging, forward
    passes enqueued asynchronously): same embeddings, bit for bit, as tokenizing
    everything first and embedding it in one go -- with the HF tokenizer adapter and with the hash tokenizer."""
    from tensor_truth_amd.embedding import HipHuggingFaceEmbedding
    from tensor_truth_amd.encoder import EncoderConfig
    from tensor_truth_amd.tokenization import HFTokenizer

    from test_host_logic import _write_wordlevel_tokenizer

    words = ["tensor", "kernel", "wave", "matrix", "retrieval", "index", "corpus", "query", "rerank", "chunk",
             "gradient", "vector", "cache", "stream", "shard", "token", "layer", "norm", "attention", "softmax"]
    _write_wordlevel_tokenizer(tmp_path, words + [f"doc{i}" for i in range(700)])
    cfg = EncoderConfig(**SMALL)
    texts = _texts(700)
    for tok in (HFTokenizer(str(tmp_path / "tokenizer.json"), "bert"), None):
        kw = {"encoder_config": cfg, "synthetic_seed": 5, "pipeline_window": 96, "forward_tokens": 1500}   # several forwards per window
        if tok is not None:
            kw["tokenizer"] = tok
        emb = HipHuggingFaceEmbedding("test/bge-small-shaped", device="cuda", embed_batch_size=64, model_kwargs=kw)
        single = emb.embed_token_batches(emb._tokenize(texts, ""))
        piped = emb._embed_texts(texts, "")
        assert piped.shape == (700, cfg.hidden) and torch.equal(single, piped)
        lists = emb.get_text_embedding_batch(texts[:130])
        assert torch.equal(torch.tensor(lists), single[:130].cpu())


def test_retrieve_and_rerank_from_eight_threads(dev, built_lib):
    """The reference calls the retrievers from up to 8 worker threads and the postprocessor from executor threads
    (rag_engine.py:392,420): scratch buffers, pinned staging and error text are per thread, the
    library keeps no global mutable state -- concurrent calls return exactly what serial calls return."""
    import threading

    from tensor_truth_amd.embedding import HipHuggingFaceEmbedding
    from tensor_truth_amd.encoder import EncoderConfig
    from tensor_truth_amd.rerank import HipSentenceTransformerRerank
    from tensor_truth_amd.retrievers import AutoMergingRetriever, MultiIndexRetriever
    from tensor_truth_amd.schema import QueryBundle, TextNode
    from tensor_truth_amd.vector_index import HipVectorIndex

    cfg = EncoderConfig(**SMALL)
    emb = HipHuggingFaceEmbedding("test/bge-small-shaped", device="cuda",
                                  model_kwargs={"encoder_config": cfg, "synthetic_seed": 31})
    rr = HipSentenceTransformerRerank(model="test/xenc", top_n=4, device="cuda",
                                      model_kwargs={"encoder_config": EncoderConfig(**XENC), "synthetic_seed": 32})
    texts = _texts(240)
    retrievers = []
    for part in range(3):
        ix = HipVectorIndex(cfg.hidden, embed_model=emb)
        ix.add([TextNode(text=t, id_=f"i{part}_{j}") for j, t in enumerate(texts[part * 80:(part + 1) * 80])])
        retrievers.append(AutoMergingRetriever(ix.as_retriever(similarity_top_k=8), ix.docstore))
    queries = [" ".join(texts[7 * i + 3].split()[:6]) for i in range(16)]

    def run(mir, q):
        nodes = mir.retrieve(q)
        ranked = rr.postprocess_nodes(nodes, query_bundle=QueryBundle(query_str=q))
        return [(n.node.id_, n.node.metadata["_source_index"], n.score) for n in ranked]

    for single_pass in (True, False):
        mir = MultiIndexRetriever(retrievers, enable_cache=False, single_pass=single_pass)
        serial = [run(mir, q) for q in queries]
        got, errs = [None] * len(queries), []

        def work(t):
            try:
                for i in range(t, len(queries), 8):
                    got[i] = run(mir, queries[i])
            except Exception as exc:  # noqa: BLE001
                errs.append(exc)

        threads = [threading.Thread(target=work, args=(t,)) for t in range(8)]
        [t.start() for t in threads]
        [t.join() for t in threads]
        assert not errs, errs
        assert got == serial and all(len(r) == 4 for r in got)
        # one pass over the packed modules per BATCH of concurrent callers (round 6): every query went through the scan front
        front = mir._scan_front
        assert (front.items, front.batches <= front.items) == ((32, True) if single_pass else (0, True))


def test_deleted_rows_are_tombstones_that_never_rank(dev, built_lib, tmp_path):
    """delete() NaN-fills the rows (document_index.py:568 remove path): the scan needs no
    mask, results equal those of an index that never held the deleted nodes -- through the filter scan (20k rows), the
    dense scan (small index), the segmented multi-index scan, compaction and persist/load."""
    from tensor_truth_amd.retrievers import MultiIndexRetriever
    from tensor_truth_amd.schema import QueryBundle, TextNode
    from tensor_truth_amd.vector_index import HipVectorIndex

    dim = 256
    g = torch.Generator().manual_seed(23)
    for n, n_del in ((20000, 3000), (500, 200)):
        vecs = torch.randn(n, dim, generator=g)
        nodes = [TextNode(text=f"t{i}", id_=f"d{i}") for i in range(n)]
        full = HipVectorIndex(dim, score_mode="cosine")
        full.add(nodes, embeddings=vecs)
        qv = torch.nn.functional.normalize(vecs[:40] + 0.3 * torch.randn(40, dim, generator=g), dim=1)
        dead = set(torch.randperm(n, generator=g)[:n_del].tolist()) | set(range(0, 40, 2))   # incl. the best matches
        keep = [i for i in range(n) if i not in dead]
        clean = HipVectorIndex(dim, score_mode="cosine")
        clean.add([nodes[i] for i in keep], embeddings=vecs[keep])
        assert full.delete([f"d{i}" for i in dead] + ["not-there"]) == len(dead)
        assert full.num_live == len(keep) and full.n == n          # below the compaction threshold: tombstones stay
        for q in range(40):
            qb = QueryBundle(query_str="q", embedding=qv[q].tolist())
            a = full.as_retriever(similarity_top_k=25).retrieve(qb)
            b = clean.as_retriever(similarity_top_k=25).retrieve(qb)
            assert [(x.node.id_, x.score) for x in a] == [(x.node.id_, x.score) for x in b] and len(a) == 25
            assert not any(int(x.node.id_[1:]) in dead for x in a)
        # a deleted id can come back as a fresh row; the segmented path sees tombstones the same way
        full.add([nodes[0]], embeddings=vecs[:1])
        other = HipVectorIndex(dim, score_mode="cosine")
        other.add([TextNode(text="o", id_=f"o{i}") for i in range(64)], embeddings=torch.randn(64, dim, generator=g))
        class E:  # noqa: E701 - embed model stub shared by both indexes
            def get_agg_embedding_from_queries(self, qs):
                return qv[1].tolist()
        full.embed_model = other.embed_model = E()
        hits = MultiIndexRetriever([full.as_retriever(similarity_top_k=10), other.as_retriever(similarity_top_k=10)],
                                   enable_cache=False).retrieve("anything")
        ids = [h.node.id_ for h in hits]
        assert not any(i.startswith("d") and int(i[1:]) in dead - {0} for i in ids) and len(hits) == 20
        full.persist(str(tmp_path / f"ix{n}"))
        again = HipVectorIndex.load(str(tmp_path / f"ix{n}"), score_mode="cosine")
        assert again.n == len(keep) + 1 and not torch.isnan(again.matrix.float()).any() and None not in again.leaf_ids
        if n == 20000:   # persist() compacted; now more than a quarter of the rows (and more than 1024) die: compaction again
            assert full.n == full.num_live == len(keep) + 1
            more = [i for i in keep if i % 3 == 1][:4500]
            clean.delete([f"d{i}" for i in more])
            assert full.delete([f"d{i}" for i in more]) == 4500 and full.n == full.num_live == len(keep) + 1 - 4500
            qb = QueryBundle(query_str="q", embedding=qv[3].tolist())
            a = full.as_retriever(similarity_top_k=25).retrieve(qb)
            b = clean.as_retriever(similarity_top_k=25).retrieve(qb)
            assert [(x.node.id_, x.score) for x in a if x.node.id_ != "d0"][:24] == [(x.node.id_, x.score) for x in b][:24]


def test_devices_other_than_hip_are_refused(built_lib):
    from tensor_truth_amd.embedding import HipHuggingFaceEmbedding

    with pytest.raises(RuntimeError, match="HIP devices only"):
        HipHuggingFaceEmbedding("BAAI/bge-m3", device="cpu", model_kwargs={"synthetic_seed": 1})


def test_semantic_splitter_distances_and_cuts(dev, built_lib):
    from tensor_truth_amd.semantic import SemanticSplitter, adjacent_distances

    g = torch.Generator().manual_seed(4)
    e = torch.randn(37, 1024, generator=g)
    want = 1 - torch.nn.functional.cosine_similarity(e[:-1], e[1:], dim=1)
    got = adjacent_distances(e.to(dev)).cpu()
    assert torch.allclose(got, want, atol=1e-5)
    assert adjacent_distances(e[:1].to(dev)).numel() == 0

    class TopicEmbedder:  # two topics: embeddings cluster by the word "kernel" vs "sauce"
        text_instruction = ""

        def _embed_texts(self, texts, prefix):
            out = torch.zeros(len(texts), 128)
            for i, t in enumerate(texts):
                out[i, 0] = t.count("kernel") + 0.01 * i
                out[i, 1] = t.count("sauce")
            return torch.nn.functional.normalize(out + 1e-3, dim=1).to(dev)

    text = "The kernel streams rows. The kernel uses LDS. The kernel is fast. The sauce needs basil. The sauce simmers."
    chunks = SemanticSplitter(TopicEmbedder(), buffer_size=0, breakpoint_percentile_threshold=70).split_text(text)
    assert len(chunks) == 2 and "kernel is fast" in chunks[0] and chunks[1].startswith("The sauce")


def test_semantic_splitter_batches_documents_without_changing_the_cuts(dev, built_lib):
    """All documents' sentence groups go through ONE embedding call and ONE adjacent-cosine launch; the cuts are
    those of one call per document (the reference's SemanticSplitterNodeParser loop, builder.py:393-407)."""
    from tensor_truth_amd.embedding import HipHuggingFaceEmbedding
    from tensor_truth_amd.encoder import EncoderConfig
    from tensor_truth_amd.schema import TextNode
    from tensor_truth_amd.semantic import SemanticSplitter

    cfg = EncoderConfig(**SMALL)
    emb = HipHuggingFaceEmbedding("test/bge-small-shaped", device="cuda", embed_batch_size=32,
                                  model_kwargs={"encoder_config": cfg, "synthetic_seed": 8, "pipeline_window": 48})
    base = _texts(120)
    docs = [". ".join(base[i * 9:(i + 1) * 9 + (i % 4)]) + "." for i in range(12)] + ["one sentence only", "", "A. B."]
    sp = SemanticSplitter(emb, buffer_size=1, breakpoint_percentile_threshold=80)
    calls = {"n": 0}
    orig = emb._embed_texts

    def counting(texts, prefix):
        calls["n"] += 1
        return orig(texts, prefix)

    emb._embed_texts = counting
    together = sp.split_texts(docs)
    assert calls["n"] == 1
    one_by_one = [sp.split_text(d) for d in docs]
    assert together == one_by_one
    assert together[12] == ["one sentence only"] and together[13] == [] and sum(len(c) > 1 for c in together) >= 8
    nodes = sp.get_nodes_from_documents([TextNode(text=d, metadata={"doc": i}) for i, d in enumerate(docs)],
                                        max_groups_per_call=20)
    assert [n.text for n in nodes] == [c for chunks in together for c in chunks]
    by_doc = {}
    for n in nodes:
        by_doc.setdefault(n.metadata["doc"], []).append(n)
    for chain in by_doc.values():       # prev/next links stay inside a document
        assert chain[0].prev_id is None and chain[-1].next_id is None
        assert all(a.next_id == b.id_ and b.prev_id == a.id_ for a, b in zip(chain, chain[1:]))


@pytest.mark.parametrize("strategy", ["hierarchical", "semantic", "semantic_hierarchical"])
def test_build_index_strategies_persist_and_retrieve(dev, built_lib, tmp_path, strategy):
    """The core of the reference's build_module (indexing/builder.py:376-453) on the HIP embedder: parse with the chosen
    ChunkingStrategy, embed the leaves, docstore + index, index_metadata.json; reload and retrieve with auto-merging."""
    import json

    from tensor_truth_amd.embedding import HipHuggingFaceEmbedding
    from tensor_truth_amd.encoder import EncoderConfig
    from tensor_truth_amd.index_builder import build_index
    from tensor_truth_amd.node_parser import get_leaf_nodes
    from tensor_truth_amd.retrievers import AutoMergingRetriever
    from tensor_truth_amd.schema import QueryBundle, TextNode
    from tensor_truth_amd.vector_index import HipVectorIndex

    cfg = EncoderConfig(**SMALL)
    emb = HipHuggingFaceEmbedding("test/bge-small-shaped", device="cuda", embed_batch_size=64,
                                  model_kwargs={"encoder_config": cfg, "synthetic_seed": 12})
    base = _texts(160)
    docs = [TextNode(text=". ".join(base[i * 20:(i + 1) * 20]) + ".", metadata={"title": f"doc {i}"}) for i in range(8)]
    stages = []
    index = build_index(docs, emb, persist_dir=str(tmp_path / "m"), chunking_strategy=strategy, chunk_sizes=[128, 48, 24],
                        chunk_overlap=6, semantic_breakpoint_threshold=75,
                        progress_callback=lambda st, cur, tot: stages.append((st, cur, tot)))
    leaves = get_leaf_nodes(index.docstore.values())
    assert index.n == len(leaves) > 8 and stages[0] == ("parsing", 0, 8) and stages[-1] == ("embedding", index.n, index.n)
    if strategy != "semantic":
        assert len(index.docstore) > index.n           # parents are in the docstore, only leaves in the matrix
    meta = json.load(open(tmp_path / "m" / "index_metadata.json"))
    assert meta["embedding_model"] == "test/bge-small-shaped" and meta["embedding_model_id"] == "bge-small-shaped"
    assert meta["chunk_sizes"] == [128, 48, 24] and meta["chunk_overlap"] == 6 and meta["chunking_strategy"] == strategy
    assert meta["index_version"] == "1.0" and meta["num_vectors"] == index.n and "created_at" in meta
    again = HipVectorIndex.load(str(tmp_path / "m"), embed_model=emb)
    assert again.n == index.n and torch.equal(again.matrix, index.matrix) and set(again.docstore) == set(index.docstore)
    # query with a leaf's own stored vector (the synthetic random-weight model maps all texts close together, so a text
    # query is not a sharp probe): the leaf -- or the parent it was merged into -- comes back first
    row = index.n // 2
    probe_id = again.leaf_ids[row]
    qb = QueryBundle(query_str="probe", embedding=again.matrix[row].float().cpu().tolist())
    hits = AutoMergingRetriever(again.as_retriever(similarity_top_k=6), again.docstore).retrieve(qb)
    assert hits and hits[0].score > 0.97
    ids = [h.node.id_ for h in hits]
    assert probe_id in ids or any(probe_id in _descendants(again.docstore, i) for i in ids)
    assert [h.score for h in hits] == sorted((h.score for h in hits), reverse=True)
    with pytest.raises(ValueError):
        build_index(docs, emb, chunking_strategy="fixed")


def test_build_index_sharded_in_one_process_equals_build_index(dev, built_lib):
    """``build_index_sharded`` (BASELINE config 5 on N GPUs: rank-local ingest stitched into one sharded index; the N-rank
    exchange is covered over gloo in test_sharded_gloo.py) with no process group = the single-device build: same rows in
    the same order, same hits through the auto-merging retriever."""
    from tensor_truth_amd.embedding import HipHuggingFaceEmbedding
    from tensor_truth_amd.encoder import EncoderConfig
    from tensor_truth_amd.index_builder import build_index, build_index_sharded
    from tensor_truth_amd.retrievers import AutoMergingRetriever
    from tensor_truth_amd.schema import QueryBundle, TextNode

    cfg = EncoderConfig(**SMALL)
    emb = HipHuggingFaceEmbedding("test/bge-small-shaped", device="cuda", embed_batch_size=64,
                                  model_kwargs={"encoder_config": cfg, "synthetic_seed": 12})
    base = _texts(160)
    docs = [TextNode(text=". ".join(base[i * 20:(i + 1) * 20]) + ".", id_=f"doc{i}", metadata={"title": f"doc {i}"}) for i in range(8)]
    kw = dict(chunking_strategy="hierarchical", chunk_sizes=[128, 48, 24], chunk_overlap=6)
    one = build_index(docs, emb, **kw)
    sharded = build_index_sharded(docs, emb, **kw)
    assert sharded.n_total == one.n and sharded.row_lo == 0
    # node ids are fresh uuids per build: compare through the row order (texts) and the scores
    texts_one = [one.docstore[i].text for i in one.leaf_ids]
    texts_sh = [sharded.docstore[i].text for i in sharded.leaf_ids]
    assert texts_one == texts_sh
    qb = QueryBundle(query_str="probe", embedding=one.matrix[one.n // 3].float().cpu().tolist())
    h1 = AutoMergingRetriever(one.as_retriever(similarity_top_k=6), one.docstore).retrieve(qb)
    h2 = AutoMergingRetriever(sharded.as_retriever(similarity_top_k=6), sharded.docstore).retrieve(qb)
    assert [h.node.text for h in h1] == [h.node.text for h in h2]
    assert [h.score for h in h1] == [h.score for h in h2]
    with pytest.raises(ValueError):
        build_index_sharded(docs, emb, persist_dir="/tmp/x", **kw)


def test_document_index_add_remove_roundtrip(dev, built_lib, tmp_path):
    """add_documents / remove_document / counters of the reference's DocumentIndexBuilder (document_index.py:427-581)."""
    from tensor_truth_amd.embedding import HipHuggingFaceEmbedding
    from tensor_truth_amd.encoder import EncoderConfig
    from tensor_truth_amd.index_builder import HipDocumentIndex
    from tensor_truth_amd.schema import QueryBundle, TextNode

    cfg = EncoderConfig(**SMALL)
    emb = HipHuggingFaceEmbedding("test/bge-small-shaped", device="cuda",
                                  model_kwargs={"encoder_config": cfg, "synthetic_seed": 14})
    base = _texts(90)
    docs = [TextNode(text=". ".join(base[i * 30:(i + 1) * 30]) + ".", metadata={"title": f"d{i}"}) for i in range(3)]
    di = HipDocumentIndex(emb, index_dir=str(tmp_path / "session"))
    assert not di.index_exists() and di.get_document_count() == 0
    stages = []
    di.add_documents(docs[:2], ["pdf_a", "pdf_b"], [64, 24], progress_callback=lambda *a: stages.append(a[0]))
    assert di.get_indexed_doc_ids() == {"pdf_a", "pdf_b"} and di.get_index_size() > 4 and stages[-1] == "Complete"
    size_ab = di.get_index_size()
    di.add_documents(docs[2:], ["pdf_c"], [64, 24])
    assert di.get_document_count() == 3 and di.get_index_size() > size_ab
    # a fresh process sees the same index
    again = HipDocumentIndex(emb, index_dir=str(tmp_path / "session"))
    assert again.get_indexed_doc_ids() == {"pdf_a", "pdf_b", "pdf_c"} and again.get_index_size() == di.get_index_size()
    b_nodes = set(again.index.ref_docs["pdf_b"])
    row = next(i for i, nid in enumerate(again.index.leaf_ids) if nid in b_nodes)
    qb = QueryBundle(query_str="q", embedding=again.index.matrix[row].float().cpu().tolist())
    assert again.as_retriever(3).retrieve(qb)[0].node.id_ in b_nodes
    assert again.remove_document("pdf_b") and not again.remove_document("pdf_b") and not again.remove_document("nope")
    assert again.get_indexed_doc_ids() == {"pdf_a", "pdf_c"} and not (b_nodes & set(again.index.docstore))
    hits = again.as_retriever(10).retrieve(qb)
    assert hits and not any(h.node.id_ in b_nodes for h in hits)
    third = HipDocumentIndex(emb, index_dir=str(tmp_path / "session"))       # the removal was persisted
    assert third.get_indexed_doc_ids() == {"pdf_a", "pdf_c"} and third.get_index_size() == again.get_index_size()
    third.add_documents([docs[0]], ["pdf_a"], [64, 24])                      # re-adding replaces
    assert third.get_document_count() == 2 and third.get_index_size() == again.get_index_size()


def _descendants(docstore, node_id):
    out, todo = set(), [node_id]
    while todo:
        for c in getattr(docstore[todo.pop()], "child_ids", []) or []:
            out.add(c)
            todo.append(c)
    return out


def test_profiling_hooks(dev, built_lib):
    import ctypes

    from tensor_truth_amd import _lib, scan as tscan

    lib = _lib.load_library()
    c = osc.synth_corpus(70_000, 128, seed=1).to(dev)
    q = c[:4].contiguous()
    lib.tt_prof_enable(1)
    for _ in range(3):
        tscan.scan_topk(c, q, 5)
    ms, n = ctypes.c_double(0), ctypes.c_int(0)
    lib.tt_prof_read(1, ctypes.byref(ms), ctypes.byref(n))
    assert n.value == 3 and 0 < ms.value < 100
    lib.tt_prof_read(3, ctypes.byref(ms), ctypes.byref(n))
    assert n.value == 6
    lib.tt_prof_enable(0)
    tscan.scan_topk(c, q, 5)
    lib.tt_prof_read(1, ctypes.byref(ms), ctypes.byref(n))
    assert n.value == 0


def test_import_reference_index_with_chroma_stub(dev, built_lib, tmp_path, monkeypatch):
    """A reference-built index directory (Chroma collection "data" + LlamaIndex docstore.json) becomes a
    HipVectorIndex: same leaves, hierarchy intact, auto-merging works on it.  chromadb itself is not installed here,
    so a stub with the three calls the importer makes stands in for it."""
    import json
    import sys
    import types

    from tensor_truth_amd.chroma_import import import_reference_index
    from tensor_truth_amd.retrievers import AutoMergingRetriever
    from tensor_truth_amd.schema import QueryBundle

    def rel(nid):
        return {"node_id": nid, "node_type": "1", "metadata": {}, "hash": "h"}

    def entry(nid, text, rels):
        return {"__data__": {"id_": nid, "metadata": {"file_name": "doc.md"}, "excluded_embed_metadata_keys": [],
                             "relationships": rels, "text": text}, "__type__": "1"}

    data = {"P": entry("P", "parent", {"5": [rel("a"), rel("b"), rel("c")]})}
    for j, nid in enumerate("abc"):
        data[nid] = entry(nid, f"leaf {nid}", {"4": rel("P")})
    data["z"] = entry("z", "unrelated leaf", {})
    (tmp_path / "docstore.json").write_text(json.dumps({"docstore/data": data}))

    g = torch.Generator().manual_seed(0)
    base = torch.nn.functional.normalize(torch.randn(128, generator=g), dim=0)
    vecs = {nid: torch.nn.functional.normalize(base + 0.05 * torch.randn(128, generator=g), dim=0) for nid in "abc"}
    vecs["z"] = torch.nn.functional.normalize(torch.randn(128, generator=g), dim=0)
    order = ["a", "z", "b", "c"]

    class Collection:
        def count(self):
            return len(order)

        def get(self, limit, offset, include):
            ids = order[offset:offset + limit]
            return {"ids": ids, "embeddings": [vecs[i].tolist() for i in ids], "documents": [f"leaf {i}" for i in ids],
                    "metadatas": [{"file_name": "doc.md", "_node_content": "{}"} for _ in ids]}

    class Client:
        def __init__(self, path):
            assert path == str(tmp_path)

        def get_collection(self, name):
            assert name == "data"
            return Collection()

    monkeypatch.setitem(sys.modules, "chromadb", types.SimpleNamespace(PersistentClient=Client))
    index = import_reference_index(str(tmp_path), score_mode="cosine")
    assert index.n == 4 and index.leaf_ids == order and set(index.docstore) == {"P", "a", "b", "c", "z"}
    q = QueryBundle(query_str="q", embedding=base.tolist())
    hits = index.as_retriever(similarity_top_k=3).retrieve(q)
    assert [h.node.id_ for h in hits] == sorted("abc", key=lambda n: -float(vecs[n] @ base))
    merged = AutoMergingRetriever(index.as_retriever(similarity_top_k=3), index.docstore).retrieve(q)
    assert [m.node.id_ for m in merged] == ["P"]                 # all three children hit -> merged into the parent


def test_overlapping_persists_and_loads_never_pair_a_matrix_with_foreign_node_tables(dev, built_lib, tmp_path):
    """HipDocumentIndex persists after every add / remove: persists of one directory may overlap each other and loads.
    Six threads persist states of different row counts while four threads load: every load sees ONE consistent
    generation (rows = ids, row r still maps to id r's vector), no ENOENT, and the directory ends with one matrix."""
    import os
    import threading

    from tensor_truth_amd.schema import TextNode
    from tensor_truth_amd.vector_index import HipVectorIndex

    D = 128
    d = str(tmp_path / "ix")

    def make(n, tag):
        ix = HipVectorIndex(D, dev, None, "cosine")
        emb = torch.zeros((n, D))
        emb[torch.arange(n), torch.arange(n) % D] = 1.0
        emb[:, 0] += float(tag) / 100.0                    # rows of different states differ
        ix.add([TextNode(text=f"{tag}-{i}", id_=f"{tag}-{i}") for i in range(n)], emb)
        return ix

    states = [make(40 + 7 * t, t) for t in range(6)]
    states[0].persist(d)
    errors = []

    def persister(t):
        try:
            for _ in range(6):
                states[t].persist(d)
        except Exception as exc:  # noqa: BLE001
            errors.append(("persist", repr(exc)))

    def loader():
        try:
            for _ in range(12):
                ix = HipVectorIndex.load(d, dev, None, "cosine")
                tag = int(ix.leaf_ids[0].split("-")[0])
                assert ix.n == 40 + 7 * tag == len(ix.leaf_ids)
                assert all(i.split("-")[0] == str(tag) for i in ix.leaf_ids)
                assert torch.equal(ix.matrix.cpu(), states[tag].matrix.cpu())
        except Exception as exc:  # noqa: BLE001
            errors.append(("load", repr(exc)))

    ts = [threading.Thread(target=persister, args=(t,)) for t in range(6)] + [threading.Thread(target=loader) for _ in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
        assert not t.is_alive()
    assert not errors, errors
    mats = [f for f in os.listdir(d) if f.startswith("corpus.") and f.endswith(".bf16")]
    assert len(mats) == 1


def test_leaf_token_ids_travel_with_the_tokenizer_they_were_made_with(dev, built_lib, tmp_path):
    """``leaf_tokens.<generation>.npz`` is only usable by the tokenizer (and text instruction) that produced it.  The index
    records both when ids are stored, persists them in nodes.json, and ``load`` drops the ids -- with a log line -- when the embedder it
    is loaded next to tokenises differently; ``token_source()`` reports the STORED origin, never the loader's; a node re-added
    without ids loses its stale ones."""
    import numpy as np

    from tensor_truth_amd.schema import TextNode
    from tensor_truth_amd.tokenization import HashTokenizer, tokenizer_signature
    from tensor_truth_amd.vector_index import HipVectorIndex

    class Emb:                                  # what the index reads of an embedder: tokenizer, instruction, length limit
        def __init__(self, vocab, instruction=""):
            self._tokenizer, self.text_instruction, self.max_length, self.model_name = HashTokenizer("xlmr", vocab), instruction, 64, "fake/emb"

    em = Emb(5000)
    nodes = [TextNode(text=f"alpha beta gamma {i}", id_=f"n{i}", metadata={}) for i in range(6)]
    toks = [em._tokenizer.encode(n.text, 64) for n in nodes]
    vecs = torch.randn(6, 128, generator=torch.Generator().manual_seed(5))
    idx = HipVectorIndex(128, dev, em, "cosine")
    idx.add(nodes, embeddings=vecs, token_ids=toks)
    src, sig, instr = idx.token_source()
    assert sig == tokenizer_signature(em._tokenizer) and instr == "" and src("n2").tolist() == toks[2][1:-1]
    pdir = str(tmp_path / "ix")
    idx.persist(pdir)
    same = HipVectorIndex.load(pdir, dev, Emb(5000), "cosine")
    assert same.token_source()[1] == sig and set(same.leaf_token_ids) == {f"n{i}" for i in range(6)}
    bare = HipVectorIndex.load(pdir, dev, None, "cosine")                       # no embedder at load: the stored origin is reported
    assert bare.token_source()[1:] == (sig, "") and np.array_equal(bare.leaf_token_ids["n4"], idx.leaf_token_ids["n4"])
    other = HipVectorIndex.load(pdir, dev, Emb(7000), "cosine")                 # another tokenizer
    assert other.leaf_token_ids is None and other.token_source() is None
    prefixed = HipVectorIndex.load(pdir, dev, Emb(5000, "passage: "), "cosine")   # same tokenizer, but texts get an instruction prefix now
    assert prefixed.leaf_token_ids is None and prefixed.token_source() is None
    # a node re-added WITHOUT ids (insert_nodes, in-process build): its old ids must not survive
    same.delete(["n3"])
    same.add([TextNode(text="entirely different words", id_="n3", metadata={})], embeddings=vecs[:1])
    assert "n3" not in same.leaf_token_ids and same.token_source()[0]("n3") is None and same.token_source()[0]("n2") is not None
    # ids made with another tokenizer than the ones already kept: the table is dropped rather than mixed
    mixed = HipVectorIndex(128, dev, em, "cosine")
    mixed.add(nodes[:3], embeddings=vecs[:3], token_ids=toks[:3])
    mixed.embed_model = Emb(7000)
    mixed.add(nodes[3:], embeddings=vecs[3:], token_ids=toks[3:])
    assert set(mixed.leaf_token_ids) == {"n3", "n4", "n5"} and mixed.token_source()[1] == tokenizer_signature(mixed.embed_model._tokenizer)


def test_reference_session_defaults_over_three_modules(dev, built_lib, monkeypatch):
    """the call an unchanged application issues -- the reference's session defaults
    (services/session_service.py:76-90: reranker_top_n = 5, confidence_cutoff_hard = 0.05, balance "top_k_per_index";
    rag_engine.py:592-593: similarity_top_k = max(5, 2 top_n) = 10 PER index) over three index modules, built by
    ``build_retrieval_service`` as ``load_engine_for_modules`` builds it.  Checked: ONE segmented scan for the three modules; every
    module's 10 candidates are the CPU oracle's exact top-10 of that module (indices bit-exact on tie-free queries, Chroma-style
    scores exp(-(2 - 2 cos)) within 1e-3 relative); the balance is the reference's arithmetic (rag_engine.py:463-507: limit =
    max(1, total // n_indexes), first `limit` per index, sorted by score descending -- restated here independently); the service
    returns the 5 best of those 30 by the reranker's own scores, none below the hard cutoff."""
    import math

    from tensor_truth_amd import model_manager as mm
    from tensor_truth_amd import scan as tscan
    from tensor_truth_amd.encoder import EncoderConfig
    from tensor_truth_amd.retrieval_service import build_retrieval_service
    from tensor_truth_amd.retrievers import similarity_top_k_for
    from tensor_truth_amd.schema import TextNode
    from tensor_truth_amd.vector_index import HipVectorIndex

    dim, sizes = 256, [5000, 1200, 9000]
    texts = _texts(64)

    class FixedEmbed:                       # deterministic query vectors: the scan and the host logic are what is under test
        model_name = "fixed"

        def get_agg_embedding_from_queries(self, qs):
            v = torch.randn(dim, generator=torch.Generator().manual_seed(1000 + len(qs[0])))
            return (v / v.norm()).tolist()

    emb = FixedEmbed()
    mm.ModelManager.reset_instance()
    mgr = mm.ModelManager.get_instance()
    mgr.model_kwargs_overrides["test/xenc"] = {"encoder_config": EncoderConfig(**XENC), "synthetic_seed": 31}
    corpora, indexes = [], []
    for part, n in enumerate(sizes):
        c = osc.synth_corpus(n, dim, seed=500 + part)
        ix = HipVectorIndex(dim, embed_model=emb)                    # score_mode "chroma", the store the reference queries
        ix.add([TextNode(text=texts[(7 * part + j) % 64] + f" m{part} leaf {j}", id_=f"m{part}_{j}", metadata={}) for j in range(n)],
               embeddings=c.float())
        corpora.append(c)
        indexes.append(ix)
    params = {"reranker_model": "test/xenc", "reranker_top_n": 5, "confidence_cutoff": 0.35, "confidence_cutoff_hard": 0.05,
              "balance_strategy": "top_k_per_index"}
    K = similarity_top_k_for(5)
    assert K == 10
    svc = build_retrieval_service(indexes, params, device="cuda", manager=mgr)
    rr = mgr.get_reranker("test/xenc", top_n=5, device="cuda")
    calls = {"seg": 0, "one": 0}
    orig_seg, orig_one = tscan.scan_topk_segmented, tscan.scan_topk
    monkeypatch.setattr(tscan, "scan_topk_segmented", lambda *a, **k: (calls.__setitem__("seg", calls["seg"] + 1), orig_seg(*a, **k))[1])
    monkeypatch.setattr(tscan, "scan_topk", lambda *a, **k: (calls.__setitem__("one", calls["one"] + 1), orig_one(*a, **k))[1])
    checked_modules = 0
    for query in ("which kernel streams the corpus", "softmax", "a somewhat longer question about retrieval and ranking"):
        calls.update(seg=0, one=0)
        balanced = svc._retriever.retrieve(query)
        assert calls == {"seg": 1, "one": 0}
        qv = torch.tensor([emb.get_agg_embedding_from_queries([query])]).to(torch.bfloat16)
        by_mod = {i: [n for n in balanced if n.node.metadata["_source_index"] == i] for i in range(3)}
        # the reference's balance over 3 x 10 candidates: limit = max(1, 30 // 3) = 10 -> every module keeps its 10, sorted desc
        assert all(len(v) == K for v in by_mod.values()) and len(balanced) == 3 * K
        assert [n.score for n in balanced] == sorted((n.score for n in balanced), reverse=True)
        for i, c in enumerate(corpora):
            want_s, want_i, gap = osc.scan_topk(c, qv, K)
            got = sorted(by_mod[i], key=lambda n: -n.score)
            want_scores = [math.exp(-(2.0 - 2.0 * float(s))) for s in want_s[0]]
            assert all(abs(g.score - w) <= 1e-3 * w for g, w in zip(got, want_scores))
            if float(gap[0]) > 1e-6:
                assert [int(g.node.id_.split("_")[1]) for g in got] == want_i[0].tolist()
                checked_modules += 1
        # the whole service: rerank the 30, keep 5, drop anything under the hard cutoff
        res = svc.retrieve(query)
        pairs = [(query, n.node.get_content()) for n in balanced]
        scores = rr.predict(pairs)
        order = sorted(range(len(pairs)), key=lambda j: -scores[j])[:5]
        want_nodes = [(balanced[j].node.id_, scores[j]) for j in order if scores[j] >= 0.05]
        assert [(n.node.id_, n.score) for n in res.source_nodes] == want_nodes
        assert res.num_sources == len(want_nodes) and res.confidence_level in ("normal", "low", "none")
        assert res.metrics["configuration"]["configured_top_n"] == 5
    assert checked_modules >= 6
    # unequal candidate counts: the reference's integer arithmetic (a module with 4 rows contributes 4; limit = 24 // 3 = 8)
    tiny = HipVectorIndex(dim, embed_model=emb)
    tiny.add([TextNode(text=f"tiny {j}", id_=f"t_{j}", metadata={}) for j in range(4)], embeddings=osc.synth_corpus(4, dim, seed=9).float())
    svc2 = build_retrieval_service([indexes[0], tiny, indexes[2]], params, device="cuda", manager=mgr)
    bal = svc2._retriever.retrieve("softmax")
    counts = [sum(1 for n in bal if n.node.metadata["_source_index"] == i) for i in range(3)]
    assert counts == [8, 4, 8] and [n.score for n in bal] == sorted((n.score for n in bal), reverse=True)
    mm.ModelManager.reset_instance()


def test_index_keeps_its_fp8_shadow_in_step_with_adds_and_deletes(dev, built_lib, monkeypatch):
    """Round 6: ``HipVectorIndex`` serves batches of <= 4 queries through the fp8 shadow prefilter (scan.ScanShadow; exact, bit-identical
    results).  The shadow must follow the matrix: built on the first lone search, EXTENDED when rows are appended, dropped when rows are
    rewritten in place (tombstones) or the matrix is replaced (growth, compaction) -- at every step the index with the shadow returns
    exactly what the same index returns with the shadow switched off."""
    from tensor_truth_amd import scan as tscan
    from tensor_truth_amd.schema import TextNode
    from tensor_truth_amd.vector_index import HipVectorIndex

    monkeypatch.setattr(tscan.ScanShadow, "MIN_ROWS", 20_000)      # (product: 1 M rows; the test keeps the node tables small)
    dim, k = 256, 10
    g = torch.Generator().manual_seed(41)
    vecs = torch.randn(60_000, dim, generator=g)
    q = torch.randn(3, dim, generator=g)
    nodes = [TextNode(text=f"t{i}", id_=f"n{i}", metadata={}) for i in range(60_000)]
    a, b = HipVectorIndex(dim, dev, None, "cosine"), HipVectorIndex(dim, dev, None, "cosine")
    b.fp8_shadow = False
    calls = {"shadow": 0}
    orig = tscan.scan_topk

    def counting(*args, **kw):
        if kw.get("shadow") is not None and kw["shadow"].serves(args[0].shape[0], args[1].shape[0], args[2]):
            calls["shadow"] += 1
        return orig(*args, **kw)

    monkeypatch.setattr(tscan, "scan_topk", counting)

    def same(nq):
        sa, ia = a.search(q[:nq], k)
        sb, ib = b.search(q[:nq], k)
        torch.cuda.synchronize()
        assert torch.equal(ia, ib) and torch.equal(sa.view(torch.int32), sb.view(torch.int32))

    for ix in (a, b):
        ix.add(nodes[:30_000], embeddings=vecs[:30_000])
    same(1)
    assert calls["shadow"] == 1 and a._shadow is not None and a._shadow.rows == 30_000 and b._shadow is None
    first = a._shadow
    for ix in (a, b):
        ix.add(nodes[30_000:31_000], embeddings=vecs[30_000:31_000])      # the matrix is re-allocated (capacity doubles): a new shadow
    same(2)
    assert a._shadow is not first and a._shadow.rows == 31_000 and a._shadow.cap_rows == a._mat.shape[0] >= 60_000
    second = a._shadow
    for ix in (a, b):
        ix.add(nodes[31_000:], embeddings=vecs[31_000:])                  # fits the matrix's capacity: the shadow is EXTENDED
    same(3)
    assert a._shadow is second and a._shadow.rows == 60_000
    same(4)
    n_before = calls["shadow"]
    sa, ia = a.search(torch.randn(5, dim, generator=g), k)                # 5 queries: the bf16 pass
    assert calls["shadow"] == n_before
    hit = int(a.search(q[:1], k)[1][0, 0])
    for ix in (a, b):
        ix.delete([f"n{hit}"])                                            # a tombstone: rows rewritten in place
    assert a._shadow is None
    same(1)
    assert a._shadow is not None and int(a.search(q[:1], k)[1][0, 0]) != hit
    for ix in (a, b):
        ix.delete([f"n{i}" for i in range(0, 60_000, 3)])                 # a third of the rows: compaction replaces the matrix
    same(2)
    hs, hi = a.search_host(q[:1], k)
    sb, ib = b.search(q[:1], k)
    assert torch.equal(hi, ib.cpu()) and torch.equal(hs, sb.cpu())


def test_index_group_scans_large_modules_through_their_shadows(dev, built_lib, monkeypatch):
    """Round 6: a lone caller (<= 4 queries) over a ``HipIndexGroup`` with large modules takes one pass per module -- through the module's
    fp8 shadow where it has one, the plain exact scan for the small modules -- and returns exactly ``tt_scan_topk_segmented``'s scores,
    module-local rows and padding (the reference's call: one query over the session's modules, rag_engine.py:420-424).  The shadows
    belong to a packing: a mutated member repacks the group and the shadows are rebuilt."""
    from tensor_truth_amd import scan as tscan
    from tensor_truth_amd.schema import TextNode
    from tensor_truth_amd.vector_index import HipIndexGroup, HipVectorIndex

    monkeypatch.setattr(tscan.ScanShadow, "MIN_ROWS", 20_000)
    dim, k = 256, 10
    g = torch.Generator().manual_seed(43)
    sizes = [30_000, 6, 25_000, 0]                                        # large, fewer rows than k, large, empty
    members = []
    for m, n in enumerate(sizes):
        ix = HipVectorIndex(dim, dev, None, "cosine")
        if n:
            ix.add([TextNode(text=f"m{m} t{i}", id_=f"m{m}_n{i}", metadata={}) for i in range(n)], embeddings=torch.randn(n, dim, generator=g))
        members.append(ix)
    group = HipIndexGroup(members)
    used = []
    orig = tscan.scan_topk

    def counting(*args, **kw):
        sh = kw.get("shadow")
        used.append(sh is not None and sh.serves(args[0].shape[0], args[1].shape[0], args[2]))
        return orig(*args, **kw)

    monkeypatch.setattr(tscan, "scan_topk", counting)

    def same(nq):
        q = torch.randn(nq, dim, generator=g)
        s, r, ids = group.search(q, k, return_snapshot=True)
        qn = (q.to(dev) / q.to(dev).norm(dim=1, keepdim=True).clamp_min(1e-12)).to(torch.bfloat16).contiguous()
        ws, wr = tscan.scan_topk_segmented(group._mat, qn, k, list(group.offsets))
        torch.cuda.synchronize()
        assert s.shape == ws.shape == (nq, len(sizes), k) and r.dtype == wr.dtype
        assert torch.equal(r, wr) and torch.equal(s.view(torch.int32), ws.view(torch.int32))
        hs, hr, hids = group.search_host(q, k)                            # what MultiIndexRetriever calls: one copy back
        assert not hs.is_cuda and torch.equal(hr, wr.cpu()) and torch.equal(hs.view(torch.int32), ws.cpu().view(torch.int32)) and hids == ids
        return s, r

    s, r = same(1)
    assert used == [True, False, True, False] * 2 and sorted(group._seg_shadows) == [0, 2]
    assert int((r[0, 1] >= 0).sum()) == 6 and int((r[0, 3] >= 0).sum()) == 0 and bool(torch.isinf(s[0, 3]).all())
    same(4)
    first = group._seg_shadows[0]
    del used[:]
    same(5)                                                               # 5 queries: the dense segmented pass, no per-module calls
    assert used == []
    hit = int(r[0, 2, 0])
    members[2].delete([f"m2_n{hit}"])                                     # a member changes: the group repacks, the shadows with it
    s2, r2 = same(2)
    assert group._seg_shadows[0] is not first and hit not in r2[:, 2].tolist()[0]
    group.fp8_shadow = False
    del used[:]
    same(1)
    assert used == []


def test_index_group_module_whose_survivor_lists_overflow_comes_back_exact(dev, built_lib, monkeypatch):
    """A module of identical rows: every row reaches the shadow pass's threshold, the survivor lists overflow, the module's status word
    is raised -- and ``HipIndexGroup`` scans that module again through the dense exact path (``_unpack_modules``), leaving the other
    modules' shadow results as they are: still ``tt_scan_topk_segmented``'s bits, on the device and through the one-copy host form."""
    from tensor_truth_amd import scan as tscan
    from tensor_truth_amd.schema import TextNode
    from tensor_truth_amd.vector_index import HipIndexGroup, HipVectorIndex

    monkeypatch.setattr(tscan.ScanShadow, "MIN_ROWS", 20_000)
    dim, k = 256, 10
    g = torch.Generator().manual_seed(47)
    one = torch.randn(1, dim, generator=g)
    embs = [torch.randn(24_000, dim, generator=g), one.repeat(80_000, 1), torch.randn(300, dim, generator=g)]   # (list capacity >= 65 536 rows)
    members = []
    for m, e in enumerate(embs):
        ix = HipVectorIndex(dim, dev, None, "cosine")
        ix.add([TextNode(text=f"m{m} t{i}", id_=f"m{m}_n{i}", metadata={}) for i in range(e.shape[0])], embeddings=e)
        members.append(ix)
    group = HipIndexGroup(members)
    reruns = []
    orig = tscan.scan_topk

    def counting(*args, **kw):
        if kw.get("exact_dense"):
            reruns.append(args[0].shape[0])
        return orig(*args, **kw)

    monkeypatch.setattr(tscan, "scan_topk", counting)
    q = torch.cat([one + 0.3 * torch.randn(1, dim, generator=g), torch.randn(1, dim, generator=g)])
    qn = (q.to(dev) / q.to(dev).norm(dim=1, keepdim=True)).to(torch.bfloat16).contiguous()
    s, r = group.search(q, k)
    ws, wr = tscan.scan_topk_segmented(group._mat, qn, k, list(group.offsets))
    torch.cuda.synchronize()
    assert reruns == [80_000]                                             # the flagged module, once
    assert torch.equal(r, wr) and torch.equal(s.view(torch.int32), ws.view(torch.int32))
    assert r[0, 1].tolist() == list(range(k))                             # equal scores: rows in ascending order
    del reruns[:]
    hs, hr, _ = group.search_host(q, k)
    assert reruns == [80_000] and torch.equal(hr, wr.cpu()) and torch.equal(hs.view(torch.int32), ws.cpu().view(torch.int32))
