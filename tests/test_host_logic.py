"""CPU: host-side logic of the plugin surface (no GPU, no HIP calls).

Mirrors the behaviours the reference's own unit tests pin with mocks
(``tests/unit/test_rag_engine.py``, ``tests/unit/test_rag_service.py``,
``tests/unit/test_embedding_model_selection.py``,
``tests/unit/services/test_retrieval_metrics.py``) plus the golden outputs of the reference's
``compute_retrieval_metrics`` (tests/golden/metrics_golden.json).
"""
import json
import math
import os
from unittest.mock import MagicMock, patch

import numpy as np
import pytest

from tensor_truth_amd import model_manager as mm
from tensor_truth_amd import quality_metrics as qm
from tensor_truth_amd import retrievers as rt
from tensor_truth_amd.retrieval_service import RetrievalService
from tensor_truth_amd.schema import NodeWithScore, QueryBundle, TextNode


def _nws(score, text="t", **md):
    return NodeWithScore(node=TextNode(text=text, metadata=dict(md)), score=score)


# ---- retrieval metrics ------------------------------------------------------------------------
def test_metrics_match_reference_golden(golden_dir):
    gold = json.load(open(os.path.join(golden_dir, "metrics_golden.json")))
    for name, case in gold.items():
        if name == "entropy":
            for e in case:
                assert qm.calculate_entropy(e["counts"]) == pytest.approx(e["expected"], abs=1e-12)
            continue
        nodes = [_nws(s, t, **m) for s, t, m in case["nodes"]]
        got = qm.compute_retrieval_metrics(nodes).to_dict()
        want = case["expected"]
        for section in want:
            for key, val in want[section].items():
                if isinstance(val, float):
                    assert got[section][key] == pytest.approx(val, rel=1e-12, abs=1e-12), (name, section, key)
                else:
                    assert got[section][key] == val, (name, section, key)


def test_metrics_known_answers():
    # known answers restated from the reference's tests/unit/services/test_retrieval_metrics.py
    m = qm.compute_retrieval_metrics([_nws(s) for s in (0.9, 0.8, 0.7, 0.6)])
    assert m.score_mean == pytest.approx(0.75) and m.score_median == pytest.approx(0.75)
    assert m.score_q1 == pytest.approx(0.65) and m.score_q3 == pytest.approx(0.85)
    assert m.high_confidence_ratio == pytest.approx(0.75) and m.low_confidence_ratio == 0.0
    assert qm.calculate_entropy([1, 1]) == pytest.approx(1.0)
    assert qm.calculate_entropy([2, 2, 2, 2]) == pytest.approx(2.0)
    assert qm.compute_retrieval_metrics([]).to_dict()["coverage"]["total_chunks"] == 0


# ---- similarity_top_k, multi-index retriever ------------------------------------------------------
def test_similarity_top_k_formula():
    assert [rt.similarity_top_k_for(n) for n in (1, 2, 3, 5, 10)] == [5, 5, 6, 10, 20]


def test_multi_index_balancing_and_tagging():
    n1 = [MagicMock(score=s, metadata={}) for s in (0.9, 0.8, 0.7)]
    n2 = [MagicMock(score=s, metadata={}) for s in (0.6, 0.5, 0.4)]
    r1, r2 = MagicMock(), MagicMock()
    r1.retrieve.return_value, r2.retrieve.return_value = n1, n2
    multi = rt.MultiIndexRetriever([r1, r2], balance_strategy="top_k_per_index")
    res = multi._retrieve(QueryBundle(query_str="test"))
    assert all("_source_index" in n.metadata for n in res)
    assert sum(n.metadata["_source_index"] == 0 for n in res) == 3
    assert sum(n.metadata["_source_index"] == 1 for n in res) == 3
    assert [n.score for n in res] == sorted((n.score for n in res), reverse=True)


def test_multi_index_uneven_limit_and_sort():
    # 5 + 1 nodes over 2 indexes -> per-index limit max(1, 6 // 2) = 3
    a = [_nws(s) for s in (0.9, 0.8, 0.7, 0.6, 0.5)]
    b = [_nws(0.95)]
    r1, r2 = MagicMock(), MagicMock()
    r1.retrieve.return_value, r2.retrieve.return_value = a, b
    res = rt.MultiIndexRetriever([r1, r2]).retrieve("q")
    assert [n.score for n in res] == [0.95, 0.9, 0.8, 0.7]
    assert [n.node.metadata["_source_index"] for n in res] == [1, 0, 0, 0]


def test_multi_index_no_balance_single_and_failure(capsys):
    r1, r2, bad = MagicMock(), MagicMock(), MagicMock()
    r1.retrieve.return_value = [_nws(0.9)]
    r2.retrieve.return_value = [_nws(0.6)]
    bad.retrieve.side_effect = RuntimeError("index offline")
    assert len(rt.MultiIndexRetriever([r1, r2], balance_strategy="none").retrieve("q")) == 2
    assert len(rt.MultiIndexRetriever([r1]).retrieve("q")) == 1
    res = rt.MultiIndexRetriever([r1, bad], balance_strategy="none").retrieve("q")
    assert len(res) == 1 and "Retriever failed" in capsys.readouterr().out


def test_group_scan_front_batches_callers_and_hands_each_its_own_rows():
    """``MultiIndexRetriever._group_scan_many``: concurrent callers of the packed-module scan go out as ONE ``search_host`` per group
    with the largest k asked for; every caller gets its own query's rows (a longer exact top-k list contains every shorter one as its
    prefix, so callers slice their own k) and the id lists of the snapshot that was scanned."""
    import torch

    class FakeGroup:
        def __init__(self, tag):
            self.tag, self.calls = tag, []

        def search_host(self, q, k):
            self.calls.append((tuple(q.shape), k))
            b = q.shape[0]
            scores = q[:, :1].reshape(b, 1, 1) - torch.arange(k, dtype=torch.float32).reshape(1, 1, k).repeat(b, 2, 1)
            rows = torch.arange(k, dtype=torch.int32).reshape(1, 1, k).repeat(b, 2, 1) + (q[:, 1].to(torch.int32) * 100).reshape(b, 1, 1)
            return scores, rows, [f"{self.tag}-ids0", f"{self.tag}-ids1"]

    ga, gb = FakeGroup("a"), FakeGroup("b")
    items = [(ga, [0.5, 1.0, 9.0], 3), (gb, [0.25, 7.0, 9.0], 2), (ga, [0.75, 2.0, 9.0], 5)]
    out = rt.MultiIndexRetriever._group_scan_many(items)
    assert ga.calls == [((2, 3), 5)] and gb.calls == [((1, 3), 2)]                       # one pass per group, the largest k
    (s0, r0, i0), (s1, r1, i1), (s2, r2, i2) = out
    assert i0 == i2 == ["a-ids0", "a-ids1"] and i1 == ["b-ids0", "b-ids1"]
    assert r0[0][:3] == [100, 101, 102] and r2[1] == [200, 201, 202, 203, 204] and r1[0] == [700, 701]
    assert s0[0][:3] == [0.5, -0.5, -1.5] and s2[0][0] == 0.75 and s1[1] == [0.25, -0.75]


def test_multi_index_cache_and_clear():
    r = MagicMock()
    r.retrieve.return_value = [_nws(0.5)]
    multi = rt.MultiIndexRetriever([r], enable_cache=True)
    multi.retrieve("same")
    multi.retrieve("same")
    assert r.retrieve.call_count == 1
    multi.clear_cache()
    multi.clear_cache()
    multi.retrieve("same")
    assert r.retrieve.call_count == 2
    nocache = rt.MultiIndexRetriever([r], enable_cache=False)
    nocache.retrieve("x")
    nocache.retrieve("x")
    assert r.retrieve.call_count == 4


# ---- auto-merging -------------------------------------------------------------------------------------
def _hierarchy():
    parent = TextNode(text="P", id_="p", child_ids=["c0", "c1", "c2", "c3"])
    kids = [TextNode(text=f"c{i}", id_=f"c{i}", parent_id="p",
                     prev_id=f"c{i-1}" if i else None, next_id=f"c{i+1}" if i < 3 else None) for i in range(4)]
    lone = TextNode(text="x", id_="x")
    ds = {n.id_: n for n in [parent, lone] + kids}
    return ds, kids, lone


def test_auto_merge_replaces_children_by_parent():
    ds, kids, lone = _hierarchy()
    base = MagicMock()
    base.retrieve.return_value = [NodeWithScore(kids[0], 0.9), NodeWithScore(kids[1], 0.7), NodeWithScore(kids[2], 0.5),
                                  NodeWithScore(lone, 0.6)]
    out = rt.AutoMergingRetriever(base, ds).retrieve("q")
    assert [n.node.id_ for n in out] == ["p", "x"]
    assert out[0].score == pytest.approx((0.9 + 0.7 + 0.5) / 3)


def test_auto_merge_threshold_is_strict_and_fills_gaps():
    ds, kids, lone = _hierarchy()
    base = MagicMock()
    # 2 of 4 children: ratio 0.5 is NOT > 0.5 -> no merge
    base.retrieve.return_value = [NodeWithScore(kids[0], 0.9), NodeWithScore(kids[3], 0.5)]
    out = rt.AutoMergingRetriever(base, ds).retrieve("q")
    assert sorted(n.node.id_ for n in out) == ["c0", "c3"]
    # c0 and c2 retrieved: the single gap c1 is filled with the mean score, then 3/4 > 0.5 merges
    base.retrieve.return_value = [NodeWithScore(kids[0], 0.8), NodeWithScore(kids[2], 0.4)]
    out = rt.AutoMergingRetriever(base, ds).retrieve("q")
    assert [n.node.id_ for n in out] == ["p"]
    assert out[0].score == pytest.approx((0.8 + 0.6 + 0.4) / 3)


def test_similarity_postprocessor():
    nodes = [_nws(0.9), _nws(0.04), _nws(None)]
    assert [n.score for n in rt.SimilarityPostprocessor(0.05).postprocess_nodes(nodes)] == [0.9]
    assert len(rt.SimilarityPostprocessor(None).postprocess_nodes(nodes)) == 3


# ---- retrieval service ---------------------------------------------------------------------------------
def test_retrieve_contract_confidence_and_progress():
    retr = MagicMock()
    retr.retrieve.return_value = [_nws(0.2, "aaaa", filename="a"), _nws(0.3, "bb", filename="b"), _nws(0.1)]
    rer = MagicMock()
    rer.postprocess_nodes.side_effect = lambda nodes, query_bundle=None: sorted(nodes, key=lambda n: -n.score)
    svc = RetrievalService(retr, [rer], {"reranker_top_n": 2, "confidence_cutoff": 0.35})
    phases = []
    res = svc.retrieve("what?", progress_callback=lambda p: phases.append(p.phase))
    assert phases == ["retrieving", "reranking"]
    assert res.num_sources == 2 and [n.score for n in res.source_nodes] == [0.3, 0.2]
    assert res.confidence_level == "low" and res.condensed_query == "what?"
    assert res.metrics["configuration"]["configured_top_n"] == 2
    assert rer.postprocess_nodes.call_args.kwargs["query_bundle"].query_str == "what?"
    svc2 = RetrievalService(retr, [rer], {"reranker_top_n": 2, "confidence_cutoff": 0.25})
    assert svc2.retrieve("q").confidence_level == "normal"
    retr.retrieve.return_value = []
    assert svc.retrieve("q").confidence_level == "none"
    assert RetrievalService().retrieve("q").confidence_level == "none"


def test_retrieve_postprocessor_failure_keeps_unprocessed_nodes():
    retr = MagicMock()
    retr.retrieve.return_value = [_nws(0.5), _nws(0.6)]
    boom = MagicMock()
    boom.postprocess_nodes.side_effect = RuntimeError("reranker down")
    res = RetrievalService(retr, [boom], {}).retrieve("q")
    assert [n.score for n in res.source_nodes] == [0.5, 0.6]


# ---- model manager ------------------------------------------------------------------------------------------
def test_model_id_helpers():
    assert mm.sanitize_model_id("BAAI/bge-m3") == "bge-m3"
    assert mm.sanitize_model_id("sentence-transformers/all-MiniLM-L6-v2") == "all-minilm-l6-v2"
    assert mm.sanitize_model_id("Org/My Model!!v2") == "my-model-v2"
    assert mm.resolve_embedding_model_name("bge-m3") == "BAAI/bge-m3"
    assert mm.resolve_embedding_model_name("BAAI/bge-m3") == "BAAI/bge-m3"
    assert mm.resolve_embedding_model_name("unknown-model") == "unknown-model"


def test_model_manager_swap_semantics():
    mm.ModelManager.reset_instance()
    mgr = mm.ModelManager.get_instance()
    assert mgr is mm.ModelManager()
    made = []

    def fake_emb(**kw):
        made.append(kw)
        return MagicMock(name="emb", **{"model_name": kw["model_name"]})

    with patch("tensor_truth_amd.embedding.HipHuggingFaceEmbedding", side_effect=fake_emb):
        e1 = mgr.get_embedder("BAAI/bge-m3", "cuda")
        assert mgr.get_embedder("bge-m3", "cuda") is e1              # healed name, same (model, device) -> reuse
        assert made[0]["embed_batch_size"] == 128 and made[0]["tokenizer_kwargs"] is None
        e2 = mgr.get_embedder("BAAI/bge-small-en-v1.5", "cuda")      # model change -> reload
        assert e2 is not e1 and len(made) == 2
        mgr.get_embedder("BAAI/bge-small-en-v1.5", "cpu")            # device change -> reload, CPU batch size
        assert len(made) == 3 and made[2]["embed_batch_size"] == 16
    rr = []

    def fake_rr(**kw):
        rr.append(kw)
        return MagicMock(name="rr")

    with patch("tensor_truth_amd.rerank.HipSentenceTransformerRerank", side_effect=fake_rr):
        a = mgr.get_reranker(top_n=3, device="cuda")
        assert rr[0]["model"] == "BAAI/bge-reranker-v2-m3" and rr[0]["top_n"] == 3
        assert mgr.get_reranker(top_n=3, device="cuda") is a
        assert mgr.get_reranker(top_n=5, device="cuda") is not a     # top_n change -> reload
        assert len(rr) == 2
    st = mgr.get_status()
    assert st["embedder"]["loaded"] and st["reranker"]["top_n"] == 5
    mgr.unload_all()
    assert not mgr.get_status()["embedder"]["loaded"]
    with patch("tensor_truth_amd.embedding.HipHuggingFaceEmbedding", side_effect=OSError("no weights")):
        with pytest.raises(RuntimeError, match="Failed to load embedding model"):
            mgr.get_embedder("BAAI/bge-m3", "cuda")
        assert mgr._embedder is None and mgr._embedder_model_name is None
    mm.ModelManager.reset_instance()


# ---- token packing / tokenizer --------------------------------------------------------------------------------------
def test_pack_tokens_layout():
    from tensor_truth_amd.encoder import BGE_M3, BGE_SMALL_EN_V15, pack_tokens

    b = pack_tokens([[0, 5, 6, 2], [0, 9, 2], list(range(4, 24))], BGE_M3)
    assert b.n_rows == 64 and (b.seq_start % 8 == 0).all()      # up to 256 rows: multiples of 64 (skinny GEMMs)
    assert b.seq_len.tolist() == [4, 3, 20] and b.seq_start.tolist() == [0, 8, 16]
    assert b.pos[:4].tolist() == [2, 3, 4, 5]                 # XLM-R positions start at pad_id + 1
    assert b.ids[4] == 1 and b.n_tokens == 27 and b.max_len == 20
    s = pack_tokens([[101, 7, 102]], BGE_SMALL_EN_V15, type_ids=[[0, 0, 0]])
    assert s.pos[:3].tolist() == [0, 1, 2] and s.types is not None
    long = pack_tokens([list(range(4, 4 + 9000))], BGE_M3)
    assert long.seq_len[0] == 8192 and long.n_rows % 256 == 0  # truncated to the model limit; 256-row GEMM tiles
    with pytest.raises(ValueError):
        pack_tokens([[]], BGE_M3)
    with pytest.raises(ValueError):
        pack_tokens([[0, 999999, 2]], BGE_M3)


def test_hash_tokenizer_pairs():
    from tensor_truth_amd.tokenization import HashTokenizer

    tk = HashTokenizer("xlmr", 250002)
    a = tk.encode("hello world")
    assert a[0] == 0 and a[-1] == 2 and len(a) == 4 and tk.encode("hello world") == a
    ids, types = tk.encode_pair("q " * 10, "p " * 600, max_length=512)
    assert len(ids) == 512 and ids[0] == 0 and ids[-1] == 2 and ids[11:13] == [2, 2] and set(types) == {0}
    bt = HashTokenizer("bert", 30522)
    ids, types = bt.encode_pair("a b", "c d e")
    assert ids[0] == 101 and ids[-1] == 102 and types == [0, 0, 0, 0, 1, 1, 1, 1]


def _write_wordlevel_tokenizer(path, words):
    from tokenizers import Tokenizer, models, pre_tokenizers, processors

    vocab = {"<s>": 0, "<pad>": 1, "</s>": 2, "<unk>": 3}
    for w in words:
        vocab.setdefault(w, len(vocab))
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.Whitespace()
    tk.post_processor = processors.TemplateProcessing(single="<s> $A </s>", pair="<s> $A </s> </s> $B </s>",
                                                      special_tokens=[("<s>", 0), ("</s>", 2)])
    tk.save(str(path / "tokenizer.json"))
    return vocab


def test_hf_tokenizer_adapter_batches_pairs_and_threads(tmp_path):
    """tokenizer.json next to the weights -> HFTokenizer: batch == one-by-one, single texts are
    cut at max_length keeping </s>, pairs are truncated longest-first to max_length, and single / pair encoding
    can run from several threads at once (the ingest pipeline tokenizes on background threads)."""
    import threading

    from tensor_truth_amd.tokenization import HFTokenizer, load_tokenizer

    words = [f"w{i}" for i in range(50)]
    vocab = _write_wordlevel_tokenizer(tmp_path, words)
    tk = load_tokenizer(str(tmp_path), "xlmr", len(vocab))
    assert isinstance(tk, HFTokenizer)
    texts = [" ".join(words[(i * 7 + j) % 50] for j in range(3 + i % 40)) for i in range(64)]
    one = [tk.encode(t, 16) for t in texts]
    assert tk.encode_batch(texts, 16) == one
    assert all(s[0] == 0 and s[-1] == 2 and len(s) <= 16 for s in one) and max(map(len, one)) == 16
    assert tk.encode("w1 w2 nope", None) == [0, vocab["w1"], vocab["w2"], 3, 2]
    ids, types = tk.encode_pair("w1 w2", "w3 w4 w5", 512)
    assert ids == [0, vocab["w1"], vocab["w2"], 2, 2, vocab["w3"], vocab["w4"], vocab["w5"], 2] and len(types) == len(ids)
    pairs = [(texts[i], texts[63 - i]) for i in range(64)]
    pb = tk.encode_pair_batch(pairs, 24)
    # (XLM-R has one token type: the batch form does not build the all-zero segment lists, round 5)
    assert [p[0] for p in pb] == [tk.encode_pair(a, b, 24)[0] for a, b in pairs] and max(len(p[0]) for p in pb) == 24
    assert all(p[1] is None for p in pb)
    got, errs = {}, []

    def work(i):
        try:
            got[i] = tk.encode_batch(texts, 16) if i % 2 == 0 else tk.encode_pair_batch(pairs, 24)
        except Exception as exc:  # noqa: BLE001
            errs.append(exc)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errs and got[0] == got[2] == got[4] == one and got[1] == got[3] == got[5] == pb


def test_pack_token_matrix_equals_pack_tokens():
    from tensor_truth_amd.encoder import BGE_M3, BGE_SMALL_EN_V15, pack_token_matrix, pack_tokens

    rng = np.random.default_rng(0)
    for cfg, length in ((BGE_M3, 292), (BGE_SMALL_EN_V15, 17)):
        a = rng.integers(4, 1000, size=(11, length), dtype=np.int32)
        x, y = pack_token_matrix(a, cfg), pack_tokens([r for r in a], cfg)
        for f in ("ids", "pos", "seq_start", "seq_len"):
            assert (getattr(x, f) == getattr(y, f)).all(), f
        assert (x.n_rows, x.max_len, x.n_tokens) == (y.n_rows, y.max_len, y.n_tokens)


def test_semantic_splitter_host_logic():
    from tensor_truth_amd.semantic import breakpoints_from_distances, split_sentences

    assert split_sentences("One. Two two! Three?\nFour.") == ["One.", " Two two!", " Three?\n", "Four."]
    assert split_sentences("   ") == []
    # np.percentile(linear): 95th percentile of [.1,.2,.9,.15,.8,.1] = 0.875 -> only 0.9 exceeds it
    assert breakpoints_from_distances([0.1, 0.2, 0.9, 0.15, 0.8, 0.1], 95) == [2]
    assert breakpoints_from_distances([0.1, 0.2, 0.9, 0.15, 0.8, 0.1], 60) == [2, 4]
    assert breakpoints_from_distances([], 95) == []


# ---- fp8 emulation in the oracle (what the HIP fp8 mode is checked against) ------------------------------------------
def test_oracle_e4m3_quantisation_properties():
    import torch

    from oracle import encoder as oe

    g = torch.Generator().manual_seed(0)
    x = torch.randn(64, 512, generator=g) * torch.logspace(-3, 2, 64).unsqueeze(1)
    x[5] = 0
    q, s = oe.quantize_rows_e4m3(x)
    assert q.abs().max().item() == 448.0 and s[5].item() == 1.0 and (q[5] == 0).all()
    assert torch.equal(q, q.to(torch.float8_e4m3fn).float())                       # values ARE e4m3 numbers
    back = q * s
    amax = x.abs().amax(1, keepdim=True)
    # 3 mantissa bits: relative error <= 2^-4 for normal values, absolute <= half a subnormal step (2^-10 in scaled units)
    err = (back - x).abs()
    assert (err <= x.abs() * 2.0 ** -4 + amax / 448 * 2.0 ** -10 + 1e-30).all()
    # per-token x per-channel fp8 linear stays within a few percent of the fp32 one
    w = torch.randn(128, 512, generator=g) * 0.05
    b = torch.randn(128, generator=g)
    y8, y = oe.linear_fp8(x[:8], w, b), x[:8] @ w.T + b
    rel = (y8 - y).norm(dim=1) / y.norm(dim=1)
    assert rel.max().item() < 0.06


def test_oracle_fp8_forward_close_to_fp32():
    import torch

    from oracle import encoder as oe

    cfg = oe.EncoderConfig(arch="xlmr", vocab_size=500, hidden=256, layers=2, heads=4, ffn=512, max_pos=80, type_vocab=1,
                           pad_id=1, ln_eps=1e-5, num_labels=1)
    W = oe.synth_weights(cfg, seed=4)
    ids, mask = oe.synth_tokens(6, 40, cfg, seed=1, lengths=[40, 33, 12, 40, 25, 7])
    ref = oe.rerank_scores(ids, mask, W, cfg)
    f8 = oe.rerank_scores(ids, mask, W, cfg, emulate_bf16=True, emulate_fp8=True)
    f8_all = oe.rerank_scores(ids, mask, W, cfg, emulate_bf16=True, emulate_fp8=True, ffn_act_scales=[0.05, 0.05])
    assert (f8 - ref).abs().max().item() < 5e-2 and (f8_all - ref).abs().max().item() < 5e-2
    assert not torch.equal(f8, f8_all)


# ---- importer for reference-built indexes ---------------------------------------------------------------------------
def _llamaindex_docstore_blob():
    def rel(nid, ntype="1"):
        return {"node_id": nid, "node_type": ntype, "metadata": {}, "hash": "h", "class_name": "RelatedNodeInfo"}

    def node(nid, text, rels, meta=None):
        return {"__data__": {"id_": nid, "embedding": None, "metadata": meta or {"file_name": "a.md"},
                             "excluded_embed_metadata_keys": ["file_name"], "excluded_llm_metadata_keys": [],
                             "relationships": rels, "text": text, "start_char_idx": 0, "end_char_idx": len(text),
                             "class_name": "TextNode"}, "__type__": "1"}

    return {
        "docstore/metadata": {"p": {"doc_hash": "x"}},
        "docstore/data": {
            "p": node("p", "parent text", {"1": rel("doc", "4"), "5": [rel("c1"), rel("c2")]}),
            "c1": node("c1", "child one", {"1": rel("doc", "4"), "4": rel("p"), "3": rel("c2")}),
            "c2": node("c2", "child two", {"1": rel("doc", "4"), "4": rel("p"), "2": rel("c1")}),
        },
        "docstore/ref_doc_info": {"doc": {"node_ids": ["p", "c1", "c2"], "metadata": {}}},
    }


def test_llamaindex_docstore_json_is_parsed_without_llamaindex(tmp_path):
    import json

    from tensor_truth_amd.chroma_import import load_llamaindex_docstore, read_chroma_collection

    path = tmp_path / "docstore.json"
    path.write_text(json.dumps(_llamaindex_docstore_blob()))
    nodes = load_llamaindex_docstore(str(path))
    assert set(nodes) == {"p", "c1", "c2"}
    assert nodes["p"].child_ids == ["c1", "c2"] and nodes["p"].parent_id is None
    assert nodes["c1"].parent_id == "p" and nodes["c1"].next_id == "c2" and nodes["c1"].prev_id is None
    assert nodes["c2"].prev_id == "c1" and nodes["c2"].text == "child two"
    assert nodes["c1"].metadata == {"file_name": "a.md"} and nodes["c1"].excluded_embed_metadata_keys == ["file_name"]
    with pytest.raises(ImportError, match="chromadb"):          # not installed here: a clear message, no fallback
        read_chroma_collection(str(tmp_path))


# ---- hierarchical node parser / index metadata (host side of the ingest path) ----------------------------------------
def test_hierarchical_node_parser_structure():
    """Levels, links and overlap semantics of the parser the reference builds at indexing/builder.py:385-388."""
    from tensor_truth_amd.node_parser import (HierarchicalNodeParser, SentenceSplitter, count_tokens, get_leaf_nodes,
                                              get_root_nodes)
    from tensor_truth_amd.schema import TextNode

    sents = [f"Sentence number {i} talks about topic {i % 7} at some length." for i in range(120)]
    doc = TextNode(text=" ".join(sents[:60]) + "\n\n" + " ".join(sents[60:]), metadata={"title": "T", "doc_type": "book"})
    doc.excluded_embed_metadata_keys = ["doc_type"]
    parser = HierarchicalNodeParser.from_defaults(chunk_sizes=[256, 64, 32], chunk_overlap=8)
    nodes = parser.get_nodes_from_documents([doc, TextNode(text="   ")])
    by_id = {n.id_: n for n in nodes}
    roots, leaves = get_root_nodes(nodes), get_leaf_nodes(nodes)
    assert len(roots) >= 4 and len(leaves) > len(roots) and len(by_id) == len(nodes)
    depth = {}
    for n in nodes:                                   # parents precede children; three levels exactly
        depth[n.id_] = 0 if n.parent_id is None else depth[n.parent_id] + 1
        assert n.metadata == {"title": "T", "doc_type": "book"} and n.excluded_embed_metadata_keys == ["doc_type"]
    assert set(depth.values()) == {0, 1, 2}
    limits = {0: 256, 1: 64, 2: 32}
    for n in nodes:
        assert 0 < count_tokens(n.text) <= limits[depth[n.id_]]
        assert (depth[n.id_] == 2) == (not n.child_ids)
        for cid in n.child_ids:
            assert by_id[cid].parent_id == n.id_ and by_id[cid].text.split()[0] in n.text
        if n.child_ids:                               # children of one parent are chained prev/next, in order
            kids = [by_id[c] for c in n.child_ids]
            assert kids[0].prev_id is None and kids[-1].next_id is None
            assert all(a.next_id == b.id_ and b.prev_id == a.id_ for a, b in zip(kids, kids[1:]))
            covered = " ".join(k.text for k in kids)  # the children cover the parent's text (with overlap)
            assert all(w in covered for w in n.text.split()[:20])
    # overlap: consecutive chunks share trailing sentences worth <= chunk_overlap tokens
    sp = SentenceSplitter(chunk_size=40, chunk_overlap=12)
    chunks = sp.split_text(" ".join(sents[:12]))
    assert len(chunks) >= 3
    for a, b in zip(chunks, chunks[1:]):
        first_sentence = b.split(".")[0]
        assert first_sentence in a and count_tokens(b) <= 40
    assert SentenceSplitter(16, 0).split_text("word " * 100)[0].count("word") <= 16   # oversized sentence cut at words
    with pytest.raises(ValueError):
        HierarchicalNodeParser.from_defaults(chunk_sizes=[64, 256])
    with pytest.raises(ValueError):
        SentenceSplitter(10, 10)


def test_sanitize_model_id_known_answers():
    """The two examples of the reference's docstring (indexing/metadata.py:34-38) + edge cases of its regex."""
    from tensor_truth_amd.vector_index import sanitize_model_id

    assert sanitize_model_id("BAAI/bge-m3") == "bge-m3"
    assert sanitize_model_id("sentence-transformers/all-MiniLM-L6-v2") == "all-minilm-l6-v2"
    assert sanitize_model_id("org/My Model@@v1.5_x") == "my-model-v1.5_x"
    assert sanitize_model_id("--weird//name--") == "name"


# ---- round 2: advisor findings + coalescing front ---------------------------------------------------------------

def test_hits_keep_what_embed_content_depends_on(tmp_path):
    """A retrieved node must give the reranker the text that was embedded: ``get_content(EMBED)`` of a hit equals the
    stored node's (the reference's leaves inherit SimpleDirectoryReader's excluded keys; rerank.py scores
    ``get_content(EMBED)`` of the RETRIEVED nodes, services/rag_service.py:617-620)."""
    from tensor_truth_amd.schema import MetadataMode, TextNode
    from tensor_truth_amd.vector_index import HipVectorRetriever, _node_from_dict, _node_to_dict

    src = TextNode(text="body", id_="a", metadata={"file_name": "x.md", "file_size": 123, "title": "T"})
    src.excluded_embed_metadata_keys = ["file_name", "file_size"]
    src.parent_id, src.next_id = "p", "b"
    assert src.get_content(MetadataMode.EMBED) == "title: T\n\nbody"

    class FakeIndex:                      # the attributes nodes_from_hits reads
        docstore = {"a": src}
        leaf_ids = ["a", None]
        num_live = 1

        @staticmethod
        def node_score(c):
            return c

    retr = HipVectorRetriever(FakeIndex(), 5, coalesce=False)
    hits = retr.nodes_from_hits([0.9, 0.8, 0.7], [0, 1, -1])       # row 1 is a tombstone, -1 is padding
    assert len(hits) == 1
    hit = hits[0].node
    assert hit.get_content(MetadataMode.EMBED) == src.get_content(MetadataMode.EMBED) == "title: T\n\nbody"
    assert hit.get_content(MetadataMode.ALL) == src.get_content(MetadataMode.ALL)
    assert hit.parent_id == "p" and hit.next_id == "b" and hit.metadata is not src.metadata
    hit.excluded_embed_metadata_keys.append("title")               # a copy, not the stored node's list
    assert src.excluded_embed_metadata_keys == ["file_name", "file_size"]
    # ... and survives persist / load (nodes.json)
    back = _node_from_dict("a", _node_to_dict(src))
    assert back.get_content(MetadataMode.EMBED) == "title: T\n\nbody" and back.parent_id == "p" and back.child_ids == []
    # a snapshot's id list wins over the index's current one (rows found before a compaction)
    assert retr.nodes_from_hits([0.5], [1], leaf_ids=["zzz", "a"])[0].node.id_ == "a"


def test_persisted_layout_is_generation_stamped(tmp_path):
    """``nodes.json`` names the matrix file it belongs to; a reader never pairs it with another generation's rows."""
    import json

    import numpy as np

    from tensor_truth_amd.vector_index import _next_generation, _read_persisted

    d = tmp_path / "ix"
    d.mkdir()
    assert _next_generation(str(d)) == 1
    rows = np.arange(3 * 128, dtype=np.int16).reshape(3, 128)
    rows.tofile(d / "corpus.4.bf16")
    (d / "corpus.3.bf16").write_bytes(b"stale generation, wrong size")
    json.dump({"dim": 128, "leaf_ids": ["a", "b", "c"], "nodes": {}, "corpus_file": "corpus.4.bf16", "generation": 4},
              open(d / "nodes.json", "w"))
    blob, raw = _read_persisted(str(d))
    assert raw.shape == (3, 128) and np.array_equal(np.asarray(raw), rows) and _next_generation(str(d)) == 5
    _, part = _read_persisted(str(d), rows=lambda n: slice(1, n))
    assert np.array_equal(np.asarray(part), rows[1:])
    json.dump({"dim": 128, "leaf_ids": ["a", "b"], "nodes": {}, "corpus_file": "corpus.4.bf16", "generation": 4},
              open(d / "nodes.json", "w"))
    with pytest.raises(ValueError):
        _read_persisted(str(d))
    # round-1 layout (no corpus_file key) still loads
    rows.tofile(d / "corpus.bf16")
    json.dump({"dim": 128, "leaf_ids": ["a", "b", "c"], "nodes": {}}, open(d / "nodes.json", "w"))
    assert _read_persisted(str(d))[1].shape == (3, 128)


def test_real_weights_are_never_paired_with_the_hashing_tokenizer(tmp_path):
    from tensor_truth_amd.tokenization import HashTokenizer, load_tokenizer

    assert isinstance(load_tokenizer(None, "xlmr", 1000), HashTokenizer)          # synthetic / state_dict weights
    (tmp_path / "model.safetensors").write_bytes(b"")
    with pytest.raises(FileNotFoundError):
        load_tokenizer(str(tmp_path), "xlmr", 1000)                                # weights, but no tokenizer files
    (tmp_path / "vocab.txt").write_text("\n".join(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "hello", "world"]))
    try:
        tk = load_tokenizer(str(tmp_path), "bert", 7)                              # WordPiece vocab -> converted
    except FileNotFoundError as exc:                                               # (no config.json for AutoTokenizer)
        assert "converting" in str(exc)
    else:
        assert tk.encode("hello world")[1:3] == [5, 6]


def test_coalescer_batches_concurrent_callers_and_preserves_results():
    import threading
    import time

    from tensor_truth_amd.coalesce import Coalescer

    sizes = []

    def run(items):
        sizes.append(len(items))
        time.sleep(0.01)
        return [i * i for i in items]

    c = Coalescer(run, max_batch=8)
    assert c.submit(3) == 9 and sizes == [1]                      # a lone caller runs at once, alone
    out = {}

    def worker(i):
        out[i] = c.submit(i)

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(40)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=30)
    assert out == {i: i * i for i in range(40)}
    assert max(sizes) == 8 and sum(sizes) == 41 and len(sizes) < 41 and c.items == 41 and c.batches == len(sizes)

    # a failing call raises, later calls are unaffected
    state = {"fail": True}

    def flaky(items):
        if state["fail"]:
            raise ValueError("boom")
        return items

    f = Coalescer(flaky)
    with pytest.raises(ValueError):
        f.submit(1)
    state["fail"] = False
    assert f.submit(2) == 2
    with pytest.raises(RuntimeError):
        Coalescer(lambda items: []).submit(1)                   # wrong result count is an error, not a hang


def test_two_phase_coalescer_prepares_the_next_batch_while_one_executes():
    import threading
    import time

    from tensor_truth_amd.coalesce import Coalescer

    state = {"executing": False, "overlap": 0, "order": []}

    def prepare(items):
        if state["executing"]:
            state["overlap"] += 1                      # a batch is being prepared while another one executes
        time.sleep(0.004)
        return list(items)

    def execute(prepared):
        assert not state["executing"]                  # device phases never overlap each other
        state["executing"] = True
        state["order"].append(prepared[0])
        time.sleep(0.02)
        state["executing"] = False
        return [i + 100 for i in prepared]

    c = Coalescer(prepare, max_batch=8, execute=execute)
    out = {}

    def worker(i):
        out[i] = c.submit(i)

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(64)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=30)
    assert all(not t.is_alive() for t in ts)
    assert out == {i: i + 100 for i in range(64)}
    assert state["overlap"] >= 1 and c.items == 64 and c.batches < 64
    # failures in either phase reach exactly the callers of that batch; the front keeps working
    boom = Coalescer(prepare, execute=lambda p: (_ for _ in ()).throw(ValueError("device")))
    with pytest.raises(ValueError):
        boom.submit(1)
    ok = Coalescer(lambda items: (_ for _ in ()).throw(KeyError("host")), execute=execute)
    with pytest.raises(KeyError):
        ok.submit(1)
    assert c.submit(7) == 107


def test_pack_tokens_vectorised_equals_the_per_sequence_loop():
    """pack_tokens (one flat copy + one scatter) against the straightforward per-sequence restatement: ragged lengths,
    truncation at the model limit and at max_len, token-type ids, list and ndarray rows, both position conventions."""
    from tensor_truth_amd import encoder as enc

    def reference(seqs, cfg, type_ids, max_len):
        limit = cfg.max_seq_len if max_len is None else min(max_len, cfg.max_seq_len)
        lens = [min(len(s), limit) for s in seqs]
        starts, off = [], 0
        for n in lens:
            starts.append(off)
            off += (n + enc._PACK_ALIGN - 1) // enc._PACK_ALIGN * enc._PACK_ALIGN
        n_rows = enc._round_rows(off)
        ids = np.full(n_rows, cfg.pad_id, dtype=np.int32)
        pos = np.zeros(n_rows, dtype=np.int32)
        types = np.zeros(n_rows, dtype=np.int32) if type_ids is not None else None
        pos_off = cfg.pad_id + 1 if cfg.arch == "xlmr" else 0
        for i, s in enumerate(seqs):
            n, st = lens[i], starts[i]
            ids[st:st + n] = np.asarray(s[:n], dtype=np.int32)
            pos[st:st + n] = np.arange(n, dtype=np.int32) + pos_off
            if types is not None:
                types[st:st + n] = np.asarray(type_ids[i][:n], dtype=np.int32)
        return ids, pos, types, np.array(starts, np.int32), np.array(lens, np.int32), n_rows

    rng = np.random.default_rng(3)
    xlmr = enc.EncoderConfig(arch="xlmr", vocab_size=500, hidden=128, layers=1, heads=2, ffn=256, max_pos=130, type_vocab=1,
                             pad_id=1, ln_eps=1e-5)
    bert = enc.EncoderConfig(arch="bert", vocab_size=700, hidden=128, layers=1, heads=2, ffn=256, max_pos=64, type_vocab=2,
                             pad_id=0, ln_eps=1e-12)
    for trial in range(40):
        cfg = xlmr if trial % 2 else bert
        n_seq = int(rng.integers(1, 70))
        lens = rng.integers(1, 200, size=n_seq)
        as_arrays = trial % 3 == 0
        seqs = [rng.integers(0, cfg.vocab_size, size=int(n), dtype=np.int32) for n in lens]
        tys = [rng.integers(0, 2, size=int(n) + (trial % 4 == 0), dtype=np.int32) for n in lens] if cfg.arch == "bert" else None
        if not as_arrays:
            seqs = [s.tolist() for s in seqs]
            tys = [t.tolist() for t in tys] if tys is not None else None
        max_len = None if trial % 5 else int(rng.integers(4, 50))
        got = enc.pack_tokens(seqs, cfg, tys, max_len)
        ids, pos, types, starts, ln, n_rows = reference(seqs, cfg, tys, max_len)
        assert got.n_rows == n_rows and got.max_len == int(ln.max()) and got.n_tokens == int(ln.sum())
        assert np.array_equal(got.ids, ids) and np.array_equal(got.pos, pos)
        assert np.array_equal(got.seq_start, starts) and np.array_equal(got.seq_len, ln)
        assert (got.types is None) == (types is None) and (types is None or np.array_equal(got.types, types))
        assert got.ids.dtype == np.int32 and got.pos.dtype == np.int32 and got.seq_start.dtype == np.int32
    with pytest.raises(ValueError):
        enc.pack_tokens([[1, 2], []], xlmr)
    with pytest.raises(ValueError):
        enc.pack_tokens([[1, 2, 9999]], xlmr)


def test_sentence_splitter_counts_once_but_cuts_as_before():
    """The splitter counts each piece once and carries the counts (the counter was its inner loop: 3-4 regex passes per
    sentence and level); the chunks must be the ones the recount-everything formulation yields."""
    import re

    from tensor_truth_amd.node_parser import _PARAGRAPH, _SENTENCE, SentenceSplitter, count_tokens

    def old_split(text, chunk_size, chunk_overlap):
        out = []
        for para in _PARAGRAPH.split(text):
            for m in _SENTENCE.finditer(para):
                s = m.group(0)
                if not s.strip():
                    continue
                if count_tokens(s) <= chunk_size:
                    out.append(s)
                    continue
                words, cur, n = re.findall(r"\S+\s*", s), [], 0
                for w in words:
                    c = count_tokens(w)
                    if cur and n + c > chunk_size:
                        out.append("".join(cur))
                        cur, n = [], 0
                    cur.append(w)
                    n += c
                if cur:
                    out.append("".join(cur))
            if out and not out[-1].endswith("\n"):
                out[-1] = out[-1] + "\n"
        pieces = out
        sizes = [count_tokens(p) for p in pieces]
        chunks, cur, n, i = [], [], 0, 0
        while i < len(pieces):
            if cur and n + sizes[i] > chunk_size:
                chunks.append("".join(cur).strip())
                keep, kn = [], 0
                for j in range(len(cur) - 1, -1, -1):
                    c = count_tokens(cur[j])
                    if kn + c > chunk_overlap or kn + c + sizes[i] > chunk_size:
                        break
                    keep.insert(0, cur[j])
                    kn += c
                cur, n = keep, kn
                continue
            cur.append(pieces[i])
            n += sizes[i]
            i += 1
        if cur and "".join(cur).strip():
            chunks.append("".join(cur).strip())
        return [c for c in chunks if c]

    rng = np.random.default_rng(8)
    words = ["alpha", "beta,", "gamma", "delta;", "x", "supercalifragilistic", "e.g", "42", "(note)", "end"]
    for trial in range(60):
        sents = []
        for _ in range(int(rng.integers(1, 40))):
            n = int(rng.integers(1, 60 if trial % 7 else 300))
            sents.append(" ".join(words[int(j)] for j in rng.integers(0, len(words), size=n)) + str(rng.choice([".", "!", "?", ""])))
        text = ""
        for s in sents:
            text += s + str(rng.choice([" ", "  ", "\n", "\n\n", " \n \n"]))
        size = int(rng.integers(8, 120))
        overlap = int(rng.integers(0, size))
        assert SentenceSplitter(size, overlap).split_text(text) == old_split(text, size, overlap), (trial, size, overlap)


@pytest.mark.parametrize("two_phase", [False, True])
def test_coalescer_random_stress_every_caller_gets_its_own_answer_or_its_batch_s_error(two_phase):
    """24 threads x 40 calls with random think times; one batch in ~12 fails (in a random phase; the member-by-member
    re-run that follows may fail again): every caller gets exactly its own result or an error -- never another caller's
    value, never a hang."""
    import random
    import threading
    import time

    from tensor_truth_amd.coalesce import Coalescer

    rnd = random.Random(5)
    lock = threading.Lock()
    seen_batches = []

    def maybe_fail(tag, items):
        with lock:
            bad = rnd.random() < 0.03
        if bad:
            raise ValueError(f"{tag} failed for a batch of {len(items)}")

    def prepare(items):
        maybe_fail("prepare", items)
        time.sleep(0.0005)
        with lock:
            seen_batches.append(len(items))
        return list(items)

    def execute(prepared):
        maybe_fail("execute", prepared)
        time.sleep(0.001)
        return prepared

    def finish(pending):
        maybe_fail("finish", pending)
        return [x * 2 + 1 for x in pending]

    if two_phase:
        c = Coalescer(prepare, max_batch=7, execute=execute, finish=finish)
    else:
        c = Coalescer(lambda items: finish(execute(prepare(items))), max_batch=7)
    outcomes = []

    def worker(tid):
        r = random.Random(tid)
        for j in range(40):
            x = tid * 1000 + j
            try:
                got = c.submit(x)
                ok = got == x * 2 + 1
                with lock:
                    outcomes.append(("ok" if ok else "WRONG", x, got))
            except ValueError as e:
                with lock:
                    outcomes.append(("err", x, str(e)))
            time.sleep(r.random() * 0.002)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(24)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
        assert not t.is_alive(), "a caller is stuck"
    assert len(outcomes) == 24 * 40
    assert not [o for o in outcomes if o[0] == "WRONG"]
    n_err = sum(1 for o in outcomes if o[0] == "err")
    assert 0 < n_err < len(outcomes) // 2              # failures happened and were delivered, most calls succeeded
    assert max(seen_batches) > 1 and max(seen_batches) <= 7
    assert c.items == 24 * 40


@pytest.mark.parametrize("two_phase", [False, True])
def test_coalescer_isolates_a_poisoned_item_from_the_callers_it_was_batched_with(two_phase):
    """24 threads, one poisoned query among them: the offender alone gets the exception, the other 23 get exactly their
    serial results although they rode in a failing batch (the reference isolates failures per call:
    rag_engine.py:453-455, services/rag_service.py:347-350)."""
    import threading
    import time

    from tensor_truth_amd.coalesce import Coalescer

    POISON = 13
    seen = []

    def prepare(items):
        seen.append(list(items))
        time.sleep(0.01)
        return list(items)

    def execute(prepared):
        if POISON in prepared:
            raise ValueError(f"malformed query {POISON}")
        return prepared

    def finish(pending):
        return [x * 3 for x in pending]

    if two_phase:
        c = Coalescer(prepare, max_batch=32, execute=execute, finish=finish)
    else:
        c = Coalescer(lambda items: finish(execute(prepare(items))), max_batch=32)
    gate = threading.Event()
    got, errs = {}, {}

    def hold():                       # a first caller keeps the front busy so that the other 24 pile up into one batch
        c.submit(1000)

    def worker(i):
        gate.wait()
        try:
            got[i] = c.submit(i)
        except ValueError as exc:
            errs[i] = str(exc)

    h = threading.Thread(target=hold)
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(24)]
    h.start()
    for t in ts:
        t.start()
    gate.set()
    for t in ts + [h]:
        t.join(timeout=60)
        assert not t.is_alive()
    assert errs == {POISON: f"malformed query {POISON}"}
    assert got == {i: 3 * i for i in range(24) if i != POISON}
    assert any(POISON in b and len(b) > 1 for b in seen), "the poisoned item never shared a batch: the test proves nothing"
    assert c.isolated >= 1 and c.items == 25
    assert c.submit(5) == 15                                          # the front keeps working


def test_coalescer_survives_a_base_exception_in_the_leader():
    """A BaseException in the leader outside the guarded calls (here: inside the collection window's sleep) still
    answers its batch, releases its execute turn and hands leadership on: later callers are served, nobody hangs."""
    import threading
    import time

    from tensor_truth_amd import coalesce
    from tensor_truth_amd.coalesce import Coalescer

    class Boom(BaseException):
        pass

    calls = {"n": 0}
    real_sleep = time.sleep

    def bad_sleep(dt):
        calls["n"] += 1
        if calls["n"] == 1:
            raise Boom()
        real_sleep(dt)

    c = Coalescer(lambda items: list(items), max_batch=4, max_wait_s=0.002, execute=lambda p: p, finish=lambda p: [x + 1 for x in p])
    coalesce.time.sleep = bad_sleep
    try:
        with pytest.raises(Boom):
            c.submit(1)
    finally:
        coalesce.time.sleep = real_sleep
    out = {}
    ts = [threading.Thread(target=lambda i=i: out.__setitem__(i, c.submit(i))) for i in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=30)
        assert not t.is_alive(), "the front is stuck after a leader died"
    assert out == {i: i + 1 for i in range(8)}


def test_persist_guard_serialises_writers_of_one_directory(tmp_path):
    """persist()'s critical section (snapshot, generation choice, writes, clean-up) is exclusive per directory -- across
    threads (lock per real path) and processes (flock) -- and the next generation is above every file lying around, so
    two overlapping persists can never pick the same corpus.<gen>.bf16."""
    import threading
    import time

    from tensor_truth_amd.vector_index import _generation_of, _next_generation, _persist_guard

    d = tmp_path / "ix"
    d.mkdir()
    inside, overlaps, gens = [0], [0], []

    def writer():
        with _persist_guard(str(d)):
            inside[0] += 1
            overlaps[0] += inside[0] > 1
            g = _next_generation(str(d))
            gens.append(g)
            time.sleep(0.005)
            (d / f"corpus.{g}.bf16").write_bytes(b"x")       # published matrix, nodes.json not yet moved
            inside[0] -= 1

    ts = [threading.Thread(target=writer) for _ in range(12)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=30)
    assert overlaps[0] == 0 and sorted(gens) == list(range(1, 13))
    with _persist_guard(str(tmp_path / "ix" / ".." / "ix")):    # same real path -> same lock object
        assert not _persist_guard(str(d)).lock.acquire(blocking=False)
    assert _generation_of("corpus.17.bf16") == 17 and _generation_of("corpus.bf16") is None and _generation_of("nodes.json") is None


def test_precision_resolution_order():
    """precision.resolve(): explicit model_kwargs beat the ModelManager key (which arrives as model_kwargs["precision"]
    only when nothing explicit is there), which beats TT_PRECISION, which beats the default -- the reference's own fp32 semantics (round 4); torch_dtype float32 is
    the reference's own spelling of "reference" (config_schema.py:66-76)."""
    import torch

    from tensor_truth_amd import model_manager as mm
    from tensor_truth_amd import precision as P

    assert P.resolve(None, {}) == "reference" == P.DEFAULT_MODE and P.resolve({}, {"TT_PRECISION": "bf16"}) == "bf16"
    assert P.resolve({"torch_dtype": "float32"}, {}) == "reference" and P.resolve({"torch_dtype": torch.float32}, {}) == "reference"
    assert P.resolve({"torch_dtype": "bfloat16"}, {"TT_PRECISION": "reference"}) == "bf16"
    assert P.resolve({"gemm_dtype": "fp8"}, {"TT_PRECISION": "reference"}) == "fp8"
    assert P.resolve({"precision": "bf16", "torch_dtype": "float32"}, {}) == "bf16"
    assert P.resolve({}, {"TT_PRECISION": "fp32"}) == "reference" and P.canonical("bf16x3") == "reference"
    with pytest.raises(ValueError):
        P.resolve({}, {"TT_PRECISION": "int4"})
    mm.ModelManager.reset_instance()
    mgr = mm.ModelManager.get_instance()
    assert mgr._with_precision(None) is None
    mgr.precision = "reference"
    assert mgr._with_precision({"trust_remote_code": True}) == {"trust_remote_code": True, "precision": "reference"}
    assert mgr._with_precision({"torch_dtype": "bfloat16"}) == {"torch_dtype": "bfloat16"}      # the per-model config wins
    assert mgr._with_precision({"gemm_dtype": "fp8"}) == {"gemm_dtype": "fp8"}
    with pytest.raises(ValueError):
        mgr.set_precision("fp64")
    mm.ModelManager.reset_instance()


def test_reranker_family_host_side(tmp_path):
    """The reference's three out-of-the-box rerankers (app_utils/config_schema.py:83-87): known configs, the BERT
    sequence-classification head under the XLM-R head's names, the activation CrossEncoder.predict applies, BERT pair
    token types."""
    import json

    import torch

    from tensor_truth_amd import weights
    from tensor_truth_amd.encoder import KNOWN_CONFIGS, _strip_prefix
    from tensor_truth_amd.tokenization import HashTokenizer

    base, mini = KNOWN_CONFIGS["BAAI/bge-reranker-base"], KNOWN_CONFIGS["cross-encoder/ms-marco-MiniLM-L-6-v2"]
    assert (base.arch, base.hidden, base.layers, base.heads, base.ffn, base.num_labels, base.max_seq_len) == ("xlmr", 768, 12, 12, 3072, 1, 512)
    assert (mini.arch, mini.hidden, mini.layers, mini.heads, mini.type_vocab, mini.num_labels, mini.max_seq_len) == ("bert", 384, 6, 12, 2, 1, 512)
    sd = {"bert.pooler.dense.weight": torch.ones(4, 4), "bert.pooler.dense.bias": torch.zeros(4),
          "classifier.weight": torch.full((1, 4), 2.0), "classifier.bias": torch.zeros(1),
          "bert.embeddings.word_embeddings.weight": torch.zeros(8, 4)}
    out = _strip_prefix(sd)
    assert out["classifier.dense.weight"] is sd["bert.pooler.dense.weight"] and out["classifier.out_proj.weight"] is sd["classifier.weight"]
    assert "embeddings.word_embeddings.weight" in out
    xl = _strip_prefix({"roberta.embeddings.word_embeddings.weight": torch.zeros(1), "classifier.dense.weight": torch.zeros(1)})
    assert "classifier.out_proj.weight" not in xl                                    # an XLM-R head is left as it is
    # activation: explicit override > config.json > the ms-marco name > sigmoid
    assert weights.head_activation("BAAI/bge-reranker-v2-m3", None, None) == "sigmoid"
    assert weights.head_activation("cross-encoder/ms-marco-MiniLM-L-6-v2", None, {"synthetic_seed": 1}) == "identity"
    assert weights.head_activation("cross-encoder/ms-marco-MiniLM-L-6-v2", None, {"activation": "sigmoid"}) == "sigmoid"
    d = tmp_path / "m"
    d.mkdir()
    (d / "config.json").write_text(json.dumps({"sbert_ce_default_activation_function": "torch.nn.modules.linear.Identity"}))
    assert weights.head_activation("x/y", str(d), None) == "identity"
    (d / "config.json").write_text(json.dumps({"sentence_transformers": {"activation_fn": "torch.nn.modules.activation.Sigmoid"}}))
    assert weights.head_activation("x/y", str(d), None) == "sigmoid"
    (d / "config.json").write_text(json.dumps({"sbert_ce_default_activation_function": "torch.nn.modules.activation.Tanh"}))
    with pytest.raises(ValueError):
        weights.head_activation("x/y", str(d), None)
    with pytest.raises(ValueError):
        weights.head_activation("x/y", None, {"activation": "softmax"})
    ids, types = HashTokenizer("bert", 3000).encode_pair("alpha beta", "gamma delta epsilon", 64)
    assert len(ids) == 8 and types == [0, 0, 0, 0, 1, 1, 1, 1]                       # [CLS] a b [SEP] | c d e [SEP]
    ids, types = HashTokenizer("xlmr", 3000).encode_pair("alpha beta", "gamma delta epsilon", 64)
    assert len(ids) == 9 and set(types) == {0}                                       # <s> a b </s></s> c d e </s>


def test_pooling_mode_of_a_checkpoint_directory(tmp_path):
    """An embedding checkpoint that declares anything but CLS pooling must not be embedded with CLS pooling silently."""
    import json

    from tensor_truth_amd import weights

    assert weights.pooling_mode(None) == "cls" and weights.pooling_mode(str(tmp_path)) == "cls"
    (tmp_path / "1_Pooling").mkdir()
    conf = {"word_embedding_dimension": 384, "pooling_mode_cls_token": True, "pooling_mode_mean_tokens": False,
            "pooling_mode_max_tokens": False, "include_prompt": True}
    (tmp_path / "1_Pooling" / "config.json").write_text(json.dumps(conf))
    assert weights.pooling_mode(str(tmp_path)) == "cls"
    conf.update(pooling_mode_cls_token=False, pooling_mode_mean_tokens=True)
    (tmp_path / "1_Pooling" / "config.json").write_text(json.dumps(conf))
    assert weights.pooling_mode(str(tmp_path)) == "mean_tokens"


def test_oracle_mean_pooling_matches_the_formula_on_ragged_masks():
    """oracle.encoder.mean_pool_normalize: masked mean over the sequence, then L2 norm -- padding rows must not count."""
    import torch

    from oracle import encoder as oe

    g = torch.Generator().manual_seed(0)
    h = torch.randn(3, 7, 16, generator=g)
    mask = torch.tensor([[1, 1, 1, 1, 1, 1, 1], [1, 1, 1, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0]])
    got = oe.mean_pool_normalize(h, mask)
    for b, n in enumerate((7, 3, 1)):
        want = h[b, :n].mean(0)
        want = want / want.norm()
        assert torch.allclose(got[b], want, atol=1e-6)
    assert torch.allclose(got[2], oe.cls_pool_normalize(h)[2], atol=1e-6)      # one token: mean pooling = CLS pooling


def test_ingest_ramps_cover_every_input_once_in_order():
    """The two ramps of the ingest pipeline (small first pieces so the GPU starts early): the semantic pass's embedding calls
    (``semantic.embedding_calls``) and the hierarchical pass's pieces (``index_builder.iter_parsed``) partition their inputs
    contiguously, in order, without loss, and grow towards their bound."""
    from tensor_truth_amd.semantic import embedding_calls

    texts = ["a. b. c." * (1 + i % 7) for i in range(1000)]
    for first, cap in ((64, 4096), (1, 8), (10 ** 9, 10 ** 9), (2048, 65536)):
        calls = embedding_calls(texts, cap, first)
        assert calls[0][0] == 0 and calls[-1][1] == len(texts)
        assert all(a[1] == b[0] for a, b in zip(calls, calls[1:])) and all(lo < hi for lo, hi in calls)
        sizes = [hi - lo for lo, hi in calls]
        assert all(a <= b + 7 for a, b in zip(sizes[:-1], sizes[1:-1]))      # growing budget (a text holds up to 7 x 3 groups)
        if (first, cap) == (64, 4096):
            assert sizes[0] < sizes[2] < sizes[4]
    assert embedding_calls([], 100, 10) == []
    assert embedding_calls(["no sentence end"], 100, 10) == [(0, 1)]

    class _Doc:
        def __init__(self, i):
            self.text, self.metadata, self.id_ = f"doc {i}. " * 3, {}, f"d{i}"

        def get_content(self, *a, **k):
            return self.text

    class _Hier:                                              # records the pieces it is handed
        def __init__(self):
            self.seen = []

        def get_nodes_from_documents(self, piece):
            self.seen.append(len(piece))
            return list(piece)

    from tensor_truth_amd.index_builder import iter_parsed

    docs = [_Doc(i) for i in range(1000)]
    h = _Hier()
    out = [n for nodes in iter_parsed(docs, None, "hierarchical", node_parser=h, sub_batch=512) for n in nodes]
    assert [d.id_ for d in out] == [d.id_ for d in docs]
    assert h.seen[:3] == [128, 256, 512] and sum(h.seen) == 1000 and max(h.seen) == 512


def test_ingest_worker_processes_build_the_same_nodes_in_order(tmp_path):
    """ingest_workers.IngestWorkers (the host side of build_index in worker processes): document chunks go through the two
    phases on their own worker and come back in document order -- same node texts, same hierarchy links, same leaf token ids
    as the in-process host steps (the GPU steps are replaced by deterministic numpy stand-ins: no GPU here)."""
    import numpy as np

    from tensor_truth_amd import ingest_workers as iw
    from tensor_truth_amd.schema import TextNode

    spec = {"tokenizer": ("hash", "xlmr", 250002), "max_length": 64, "text_instruction": "passage: ", "buffer_size": 1, "percentile": 90,
            "chunk_sizes": [128, 32, 16], "chunk_overlap": 4}
    rng = np.random.default_rng(3)
    docs = []
    for i in range(70):
        n_sent = 1 if i % 17 == 0 else int(rng.integers(20, 50))          # single-sentence documents take the no-split branch
        text = " ".join(" ".join(f"w{rng.integers(0, 400)}" for _ in range(rng.integers(6, 18))) + "." for _ in range(n_sent))
        d = TextNode(text="" if i == 33 else text, metadata={"title": f"doc {i}", "secret": "x"})
        d.excluded_embed_metadata_keys = ["secret"]
        docs.append(d)

    def embed_tokens(seqs):
        return np.stack([np.array([np.sum(s) % 97 + 1, len(s), (int(s[0]) * 7 + int(s[-1])) % 31 + 1, np.sum(s[::2]) % 13 + 1], dtype=np.float64) for s in seqs])

    def distances(e):
        e = e / np.linalg.norm(e, axis=1, keepdims=True)
        return (1 - (e[:-1] * e[1:]).sum(1)).astype(np.float32), (lambda block=False: True)

    for semantic in (True, False):
        got = []
        pool = iw.get_workers(spec, 3)
        pool.run(docs, semantic, embed_tokens, distances, lambda nodes, pos, emb: got.append((nodes, pos, emb)), chunk_docs=16)
        host = iw._Host(spec)
        recs = [iw._doc_record(d) for d in docs]
        if semantic:
            _, flat, lens = host.split(0, recs)
            nodes, leaf_pos, lf, ll = host.cut(0, distances(embed_tokens(iw.unflatten(flat, lens)))[0])
        else:
            nodes, leaf_pos, lf, ll = host.parse(recs)
        mine = [n for g in got for n in g[0]]
        assert [n.text for n in mine] == [n.text for n in nodes] and len(mine) > 300
        assert [n.metadata for n in mine] == [n.metadata for n in nodes]
        # links: same shape (ids differ: uuid4), expressed as positions
        def links(ns):
            pos = {n.id_: i for i, n in enumerate(ns)}
            return [(pos.get(n.parent_id), [pos[c] for c in n.child_ids], pos.get(n.prev_id), pos.get(n.next_id)) for n in ns]
        assert links(mine) == links(nodes)
        # leaf embeddings arrive with their chunk, computed from the same token ids (EMBED content: "secret" excluded, prefix added)
        want = embed_tokens(iw.unflatten(lf, ll))
        assert np.array_equal(np.concatenate([g[2] for g in got if g[2] is not None]), want)
        off = 0
        for g_nodes, g_pos, _ in got:
            assert all(not g_nodes[i].child_ids for i in g_pos)
            off += len(g_nodes)
    # a worker that cannot even start (an invalid hierarchy) is an error of the caller's, not a hang
    with pytest.raises((EOFError, RuntimeError)):
        iw.IngestWorkers({**spec, "chunk_sizes": [16, 32]}, 1)


def test_failed_build_discards_the_worker_pool():
    """A build that raises half-way (here: the index callback) leaves chunks in flight and replies unread in the pipes.  The pool
    must not be reused: it is killed, and the next build of the same configuration gets fresh workers and the right nodes."""
    import numpy as np

    from tensor_truth_amd import ingest_workers as iw
    from tensor_truth_amd.schema import TextNode

    spec = {"tokenizer": ("hash", "xlmr", 250002), "max_length": 64, "text_instruction": "", "buffer_size": 1, "percentile": 90,
            "chunk_sizes": [128, 32, 16], "chunk_overlap": 4}
    rng = np.random.default_rng(5)
    docs = [TextNode(text=" ".join(" ".join(f"w{rng.integers(0, 300)}" for _ in range(10)) + "." for _ in range(25)), metadata={"title": str(i)})
            for i in range(80)]
    embed = lambda seqs: np.stack([np.array([len(s), int(s[0]) % 7 + 1.0]) for s in seqs])                      # noqa: E731
    dist = lambda e: (np.abs(np.diff(e[:, 0])).astype(np.float32), (lambda block=False: True))                     # noqa: E731
    pool = iw.get_workers(spec, 2)
    pids = [p.pid for p in pool.procs]
    calls = []

    def failing(nodes, pos, emb):
        calls.append(len(nodes))
        if len(calls) == 2:
            raise ValueError("index refused the rows")

    with pytest.raises(ValueError, match="index refused"):
        pool.run(docs, True, embed, dist, failing, chunk_docs=8)
    assert not pool.alive() and pool.procs == []
    pool2 = iw.get_workers(spec, 2)
    assert pool2 is not pool and pool2.alive() and not set(pids) & {p.pid for p in pool2.procs}
    got = []
    pool2.run(docs, True, embed, dist, lambda nodes, pos, emb: got.append(nodes), chunk_docs=8)
    host = iw._Host(spec)
    recs = [iw._doc_record(d) for d in docs]
    _, flat, lens = host.split(0, recs)
    nodes, _, _, _ = host.cut(0, dist(embed(iw.unflatten(flat, lens)))[0])
    assert [n.text for g in got for n in g] == [n.text for n in nodes]


def test_two_threads_building_with_the_same_configuration_do_not_share_a_pool():
    """get_workers() hands every caller the process's one pool per configuration, and both builds number their
    chunks from 0 -- two threads calling build_index at once read each other's replies.  lease_workers() gives the second builder a
    private pool (closed after its build); run() holds the pool's lock, so even a direct get_workers() user is serialised.  Two
    concurrent builds over DIFFERENT documents must each return exactly their own nodes, in document order."""
    import threading

    import numpy as np

    from tensor_truth_amd import ingest_workers as iw
    from tensor_truth_amd.schema import TextNode

    spec = {"tokenizer": ("hash", "xlmr", 250002), "max_length": 64, "text_instruction": "", "buffer_size": 1, "percentile": 90,
            "chunk_sizes": [128, 32, 16], "chunk_overlap": 4}
    embed = lambda seqs: np.stack([np.array([len(s), int(s[0]) % 7 + 1.0]) for s in seqs])                      # noqa: E731
    dist = lambda e: (np.abs(np.diff(e[:, 0])).astype(np.float32), (lambda block=False: True))                     # noqa: E731

    def corpus(seed):
        rng = np.random.default_rng(seed)
        return [TextNode(text=" ".join(" ".join(f"s{seed}w{rng.integers(0, 300)}" for _ in range(10)) + "." for _ in range(25)),
                         metadata={"title": f"{seed}-{i}"}) for i in range(72)]

    def expected(docs):
        host = iw._Host(spec)
        recs = [iw._doc_record(d) for d in docs]
        _, flat, lens = host.split(0, recs)
        return [n.text for n in host.cut(0, dist(embed(iw.unflatten(flat, lens)))[0])[0]]

    results, errors, pools = {}, [], {}
    gate = threading.Barrier(3)

    def build(seed, use_lease):
        try:
            docs, got = corpus(seed), []
            gate.wait(timeout=60)
            if use_lease:
                with iw.lease_workers(spec, 2) as pool:
                    pools[seed] = pool
                    pool.run(docs, True, embed, dist, lambda nodes, pos, emb: got.append(nodes), chunk_docs=8)
            else:
                pool = pools[seed] = iw.get_workers(spec, 2)
                pool.run(docs, True, embed, dist, lambda nodes, pos, emb: got.append(nodes), chunk_docs=8)
            results[seed] = [n.text for g in got for n in g]
        except BaseException as exc:  # noqa: BLE001
            errors.append((seed, exc))

    threads = [threading.Thread(target=build, args=(11, True)), threading.Thread(target=build, args=(12, True)),
               threading.Thread(target=build, args=(13, False))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    for seed in (11, 12, 13):
        assert results[seed] == expected(corpus(seed)), f"build {seed} returned another build's nodes"
    # at least one of the leased builds ran on a private pool, which is closed again; the shared pool lives on
    shared = iw.get_workers(spec, 2)
    assert shared.alive()
    private = [p for p in pools.values() if p is not shared]
    assert all(not p.alive() for p in private)


def test_unigram_fixture_and_subword_counted_hierarchy():
    """Round 5: the trained 250 002-piece Unigram tokenizer fixture (tools/synth_text.py) has XLM-R's layout -- specials 0-3, single
    and pair templates -- and the hierarchy parser, told to count chunk sizes with the embedder's tokenizer (the offline stand-in
    for llama-index's tiktoken count), produces leaves within the reference's default leaf size in SUB-WORD tokens, identically
    in process and in a worker process."""
    import sys

    import numpy as np

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import synth_text as st
    from tensor_truth_amd import ingest_workers as iw
    from tensor_truth_amd.node_parser import HierarchicalNodeParser, get_leaf_nodes, tokenizer_counter
    from tensor_truth_amd.schema import TextNode

    tk = st.unigram_tokenizer()
    assert tk.tk.get_vocab_size() == 250_002
    assert [tk.tk.token_to_id(t) for t in ("<s>", "<pad>", "</s>", "<unk>")] == [0, 1, 2, 3]
    a, b = st.zipf_text(1, 12), st.zipf_text(2, 40)
    ia, ib = tk.encode(a), tk.encode(b)
    assert ia[0] == 0 and ia[-1] == 2 and 3 not in ia and len(ia) > 14            # more pieces than words, no <unk>
    ids, types = tk.encode_pair(a, b, 512)
    assert ids == ia + [2] + ib[1:]                                                # <s> A </s></s> B </s>
    ids, _ = tk.encode_pair(a, st.zipf_text(3, 600), 512)
    assert len(ids) == 512 and ids[-1] == 2                                        # longest-first truncation keeps the specials
    # hierarchy in sub-word tokens
    count = tokenizer_counter(tk)
    doc = TextNode(text=" ".join(st.zipf_text(100 + i, 14) + "." for i in range(300)), metadata={})
    nodes = HierarchicalNodeParser.from_defaults(chunk_sizes=[2048, 512, 256], chunk_overlap=64, tokenizer=count).get_nodes_from_documents([doc])
    leaves = get_leaf_nodes(nodes)
    sizes = [count(n.text) for n in leaves]
    assert max(sizes) <= 256 and np.mean(sizes) > 150 and len(leaves) > 20
    words = [len(n.text.split()) for n in leaves]
    assert np.mean(words) < 140                                                   # ~2 pieces per word: a 256-token leaf is ~110 words
    # the worker processes honour the same counter
    spec = {"tokenizer": iw.tokenizer_spec(tk), "max_length": 8192, "text_instruction": "", "buffer_size": 1, "percentile": 95,
            "chunk_sizes": [2048, 512, 256], "chunk_overlap": 64, "token_counter": "embedder"}
    host = iw._Host(spec)
    got, leaf_pos, flat, lens = host.parse([iw._doc_record(doc)])
    assert [n.text for n in got] == [n.text for n in nodes]
    assert int(lens.max()) <= 256 + 2


def test_pair_tokenizer_pool_returns_the_in_process_ids(monkeypatch):
    """ingest_workers.PairTokenizerPool: the pairs of a coalesced rerank batch tokenised in worker processes are, id for id and in
    order, what HFTokenizer.encode_pair_batch returns in process -- including longest-first truncation at 512 and ragged slices."""
    import sys

    import numpy as np

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import synth_text as st
    from tensor_truth_amd import ingest_workers as iw

    tk = st.unigram_tokenizer()
    pairs = [(st.zipf_text(900 + i, 5 + i % 20), st.zipf_text(5000 + i, 30 + (i * 37) % 400)) for i in range(131)]
    want = tk.encode_pair_batch(pairs, 512)
    monkeypatch.setenv("TT_PAIR_WORKERS", "3")
    assert iw.get_pair_pool(tk) is None                       # a request thread never starts the pool itself: it sets a background start off
    pool = iw.get_pair_pool(tk, wait=True)                     # ... which this call waits for (the pool comes back warm: one job per worker)
    assert pool is not None and pool is iw.get_pair_pool(tk)
    ids, types = pool.encode(pairs, 512)
    assert len(ids) == len(want) and max(len(x) for x in ids) == 512
    assert all(np.array_equal(a, np.asarray(w[0], dtype=np.int32)) for a, w in zip(ids, want))
    assert all(t is None for t in types)                       # XLM-R: one token type
    ids2, _ = pool.encode(pairs[:2], 512)                      # fewer pairs than workers
    assert [a.tolist() for a in ids2] == [w[0] for w in want[:2]]
    monkeypatch.setenv("TT_PAIR_WORKERS", "0")
    assert iw.get_pair_pool(tk) is None


def test_pair_tokenizer_pool_failures_fall_back_instead_of_failing_the_request(monkeypatch):
    """nothing about the pair pool may fail or hang a user's rerank request.  A worker that died (EOF on its pipe), a
    pool past its deadline, a pool still being started: ``encode`` / ``get_pair_pool`` hand back None -- the caller tokenises in
    process, identical ids -- the broken pool is killed and the next call gets a fresh one."""
    import sys
    import time

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import synth_text as st
    from tensor_truth_amd import ingest_workers as iw

    tk = st.unigram_tokenizer()
    pairs = [(st.zipf_text(900 + i, 6), st.zipf_text(5000 + i, 40)) for i in range(100)]
    monkeypatch.setenv("TT_PAIR_WORKERS", "2")
    pool = iw.get_pair_pool(tk, wait=True)
    assert pool is not None and pool.encode(pairs, 512) is not None
    # a deadline that has already passed: None, and the pool is gone (not left with half-read frames)
    assert pool.encode(pairs, 512, deadline_s=0.0) is None and not pool.alive()
    fresh = iw.get_pair_pool(tk, wait=True)
    assert fresh is not None and fresh is not pool and fresh.alive()
    # a worker killed under the pool (OOM killer): EOFError inside -> None outside
    fresh.pool.procs[0].kill()
    fresh.pool.procs[0].wait(timeout=10)
    assert fresh.encode(pairs, 512) is None
    # while another thread is starting the pool, a request thread does not wait for it
    import pickle
    key = pickle.dumps((iw.tokenizer_spec(tk), 2))
    with iw._POOLS_LOCK:
        iw._PAIR_POOLS.pop(key, None)
        iw._PAIR_POOLS_STARTING.add(key)
    try:
        t0 = time.perf_counter()
        assert iw.get_pair_pool(tk) is None and time.perf_counter() - t0 < 0.5
    finally:
        with iw._POOLS_LOCK:
            iw._PAIR_POOLS_STARTING.discard(key)
    # no pool and nobody starting one (the last one died): the request thread gets None AT ONCE and a background start is under way
    t0 = time.perf_counter()
    assert iw.get_pair_pool(tk) is None and time.perf_counter() - t0 < 0.5
    with iw._POOLS_LOCK:
        assert key in iw._PAIR_POOLS_STARTING or key in iw._PAIR_POOLS
    last = iw.get_pair_pool(tk, wait=True)                     # waits for that start
    got = last.encode(pairs, 512)
    want = tk.encode_pair_batch(pairs, 512)
    assert got is not None and [a.tolist() for a in got[0]] == [w[0] for w in want]
    monkeypatch.setenv("TT_PAIR_WORKERS", "0")


def test_each_retrieval_service_keeps_its_own_token_source():
    """``ModelManager.get_reranker`` returns ONE cached instance; a service's source of stored leaf ids travels with the
    service (``RerankerWithTokenSource``: a per-call argument), so building a second service over other indexes neither replaces nor
    detaches the first one's, and an index without ids gets the plain reranker."""
    from tensor_truth_amd.rerank import RerankerWithTokenSource
    from tensor_truth_amd.retrieval_service import build_retrieval_service
    from tensor_truth_amd.schema import NodeWithScore, TextNode

    calls = []

    class FakeReranker:
        top_n = 2
        _token_source = None

        def accepts_token_source(self, signature, instruction=""):
            return signature == "sig-A" and not instruction

        def postprocess_nodes(self, nodes, query_bundle=None, query_str=None, token_source=None):
            calls.append(token_source)
            return nodes[: self.top_n]

    class FakeManager:
        def __init__(self):
            self.rr = FakeReranker()

        def get_reranker(self, model=None, top_n=3, device="cuda"):
            return self.rr

    class FakeRetriever:
        def __init__(self, name):
            self.name = name

        def retrieve(self, q):
            return [NodeWithScore(node=TextNode(text=f"{self.name}{i}", id_=f"{self.name}{i}", metadata={}), score=1.0 - 0.1 * i) for i in range(3)]

    class FakeIndex:
        def __init__(self, name, sig):
            self.name, self.sig, self.docstore = name, sig, {}
            self.table = {f"{name}{i}": [i] for i in range(3)}

        def as_retriever(self, similarity_top_k=10):
            return FakeRetriever(self.name)

        def token_source(self):
            return None if self.sig is None else (self.table.get, self.sig, "")

    mgr = FakeManager()
    svc_a = build_retrieval_service([FakeIndex("a", "sig-A")], {"reranker_top_n": 2}, manager=mgr)
    svc_b = build_retrieval_service([FakeIndex("b", "sig-A")], {"reranker_top_n": 2}, manager=mgr)
    svc_c = build_retrieval_service([FakeIndex("c", None)], {"reranker_top_n": 2}, manager=mgr)            # no stored ids
    svc_d = build_retrieval_service([FakeIndex("d", "sig-OTHER")], {"reranker_top_n": 2}, manager=mgr)     # another tokenizer's ids
    va, vb = svc_a._node_postprocessors[0], svc_b._node_postprocessors[0]
    assert isinstance(va, RerankerWithTokenSource) and isinstance(vb, RerankerWithTokenSource) and va._reranker is vb._reranker is mgr.rr
    assert svc_c._node_postprocessors[0] is mgr.rr and svc_d._node_postprocessors[0] is mgr.rr
    assert mgr.rr._token_source is None                                  # the shared instance was never touched
    assert va.top_n == 2                                                 # everything else is the shared reranker's
    svc_a.retrieve("q")
    svc_b.retrieve("q")
    svc_a.retrieve("q2")
    svc_c.retrieve("q")
    assert calls[0]("a1") == [1] and calls[0]("b1") is None              # service A looks up in ITS index only ...
    assert calls[1]("b2") == [2] and calls[1]("a2") is None              # ... B in its own ...
    assert calls[2]("a0") == [0]                                         # ... and A still does after B was built
    assert calls[3] is None


def test_pretokenized_pairs_equal_the_tokenizers_own_pair_encoding():
    """Round 5: leaves are tokenised once, at ingest (HipVectorIndex.leaf_token_ids); the reranker then assembles
    ``<s> q </s></s> passage </s>`` from the stored body ids instead of tokenising the passage again.  ``assemble_pairs`` must return
    exactly what the tokenizer's own pair call returns -- specials, order and the Rust library's longest-first truncation (restated in
    ``truncate_longest_first`` and checked here against the library itself over every regime: no cut, only the longer side cut, both
    sides cut, odd and even budgets) -- for the trained Unigram tokenizer and for the hashing stand-in."""
    import sys

    import numpy as np

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import synth_text as st
    from tensor_truth_amd.tokenization import (HashTokenizer, PreTokenized, assemble_pairs, tokenizer_signature,
                                               truncate_longest_first)

    # the rule against a brute-force reading of "remove from the longer side until it fits; the shorter side keeps up to half"
    for budget in (7, 8, 59, 60, 508):
        for na in range(0, 40, 3):
            for nb in range(0, 700, 37):
                a, b = truncate_longest_first(na, nb, budget)
                assert a <= na and b <= nb and (a + b == min(na + nb, budget) or na + nb <= budget)
                if na + nb > budget and na <= budget // 2:
                    assert a == na and b == budget - na
    tk = st.unigram_tokenizer()
    rng = np.random.default_rng(1)
    for t in range(120):
        nq, npw, ml = int(rng.integers(1, 300)), int(rng.integers(1, 400)), int(rng.choice([23, 24, 64, 129, 512]))
        q, p = st.zipf_text(1000 + t, nq), st.zipf_text(5000 + t, npw)
        body = np.asarray(tk.encode(p, None)[1:-1], dtype=np.int32)
        assert assemble_pairs(tk, [(q, PreTokenized(body))], ml)[0].tolist() == tk.encode_pair(q, p, ml)[0], (nq, npw, ml)
    for arch in ("xlmr", "bert"):
        ht = HashTokenizer(arch, 30000)
        for t in range(60):
            nq, npw, ml = int(rng.integers(1, 200)), int(rng.integers(1, 300)), int(rng.choice([23, 24, 64, 129, 512]))
            q = " ".join(f"w{int(j)}" for j in rng.integers(0, 999, size=nq))
            p = " ".join(f"w{int(j)}" for j in rng.integers(0, 999, size=npw))
            body = np.asarray(ht.encode(p, None)[1:-1], dtype=np.int32)
            assert assemble_pairs(ht, [(q, PreTokenized(body))], ml)[0].tolist() == ht.encode_pair(q, p, ml)[0], (arch, nq, npw, ml)
    assert tokenizer_signature(tk) == tokenizer_signature(st.unigram_tokenizer()) != tokenizer_signature(HashTokenizer("xlmr", 250002))


def test_posted_sends_do_not_block_on_a_busy_peer():
    """ingest_workers._PipeConn.post / pump (round 5): the feeder hands a busy worker its next work unit -- larger than the pipe --
    WITHOUT parking in write(): post() returns at once with bytes pending, pump() moves what the pipe takes, and the peer, once it
    reads, receives the very objects, in order.  (A blocking send here cost a 6000-document build 44 of its 78 seconds.)"""
    import threading
    import time

    import numpy as np

    from tensor_truth_amd import ingest_workers as iw

    r1, w1 = os.pipe()          # feeder -> peer
    r2, w2 = os.pipe()          # peer -> feeder
    feeder = iw._PipeConn(r2, w1, duplex_safe=True)
    peer = iw._PipeConn(r1, w2)
    big = [("split", 7, ["x" * 1000] * 3000), ("cut", 8, np.arange(300_000, dtype=np.float32))]
    t0 = time.perf_counter()
    for m in big:
        feeder.post(m)
    assert time.perf_counter() - t0 < 1.0 and feeder.pending()          # ~4 MB against a 64 KiB pipe: nothing blocked
    got = []

    def slow_peer():
        time.sleep(0.3)                                                   # "busy with its current unit"
        for _ in big:
            got.append(peer.recv())
        peer.send(("done", 1))

    th = threading.Thread(target=slow_peer)
    th.start()
    deadline = time.time() + 30
    while feeder.pending() and time.time() < deadline:
        feeder.pump()
        time.sleep(0.001)
    assert not feeder.pending()
    assert feeder.recv() == ("done", 1)
    th.join(timeout=10)
    assert got[0] == big[0] and got[1][:2] == ("cut", 8) and np.array_equal(got[1][2], big[1][2])
    for fd in (r1, w1, r2, w2):
        os.close(fd)
