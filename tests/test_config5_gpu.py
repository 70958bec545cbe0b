"""GPU: BASELINE.json config 5 as ONE composed workload -- semantic-hierarchical ingest (streaming batch embed) ->
auto-merging retrieval -> fp8 MFMA reranker -- through the same entry points the reference uses
(``indexing/builder.py:376-453`` build_module, ``rag_engine.py:529-738`` load_engine_for_modules,
``services/rag_service.py:594-661`` retrieve), against the same pipeline evaluated with the CPU oracle stage by stage:

  A. semantic splitter: adjacent-cosine distances and the cuts they imply   (builder.py:391-418)
  B. leaf embeddings in the index matrix
  C. top-k retrieval + auto-merge (the host merge logic is shared; its INPUT, the hit list, is what can differ)
  D. fp8 cross-encoder scores -> final top-n ids: rank agreement with the fp32 oracle.

Small model shapes (2 layers) keep the oracle in seconds; every kernel and every host path is the one the full-size
run takes (the full-size timing of this workload is ``bench.py``'s ``config5`` leg)."""
import math

import numpy as np
import pytest
import torch

from oracle import encoder as oe
from oracle import scan as osc
from rank_checks import assert_order_on_separable, kendall_tau, topn_overlap

pytestmark = pytest.mark.gpu

SMALL = dict(arch="bert", vocab_size=3000, hidden=384, layers=2, heads=12, ffn=1536, max_pos=128, type_vocab=2,
             pad_id=0, ln_eps=1e-12)
XENC = dict(arch="xlmr", vocab_size=3000, hidden=256, layers=2, heads=4, ffn=512, max_pos=130, type_vocab=1,
            pad_id=1, ln_eps=1e-5, num_labels=1)
TOPICS = [
    ["kernel", "wave", "lds", "mfma", "tile", "barrier", "register", "occupancy", "prefetch", "swizzle"],
    ["basil", "sauce", "simmer", "garlic", "oven", "dough", "yeast", "salt", "pepper", "broth"],
    ["orbit", "comet", "nebula", "quasar", "planet", "lunar", "solar", "rocket", "gravity", "vacuum"],
    ["ledger", "audit", "invoice", "credit", "debit", "equity", "bond", "yield", "margin", "hedge"],
]
FP8_BOUND_2L = 5e-2      # stated bound of the fp8 mode on a 2-layer model's sigmoid scores


def _pad(seqs, pad):
    L = max(len(s) for s in seqs)
    ids = torch.full((len(seqs), L), pad, dtype=torch.int64)
    mask = torch.zeros(len(seqs), L, dtype=torch.int64)
    for b, s in enumerate(seqs):
        ids[b, : len(s)] = torch.tensor(s)
        mask[b, : len(s)] = 1
    return ids, mask


def _docs(n_docs=24, seed=9):
    """Documents that change topic a few times: the semantic splitter has real breakpoints to find."""
    from tensor_truth_amd.schema import TextNode

    g = torch.Generator().manual_seed(seed)
    docs = []
    for d in range(n_docs):
        sents = []
        for block in range(3):
            words = TOPICS[(d + block) % len(TOPICS)]
            for _ in range(int(torch.randint(4, 8, (1,), generator=g))):
                k = int(torch.randint(6, 14, (1,), generator=g))
                sents.append(" ".join(words[j] for j in torch.randint(0, len(words), (k,), generator=g).tolist()) + ".")
        node = TextNode(text=" ".join(sents), metadata={"title": f"doc {d}", "file_name": f"d{d}.md"})
        node.excluded_embed_metadata_keys = ["file_name"]
        docs.append(node)
    return docs


def test_config5_semantic_hierarchical_ingest_automerge_fp8_rerank(dev, built_lib, tmp_path):
    from tensor_truth_amd import model_manager as mm
    from tensor_truth_amd.encoder import EncoderConfig
    from tensor_truth_amd.index_builder import build_index
    from tensor_truth_amd.node_parser import get_leaf_nodes
    from tensor_truth_amd.retrieval_service import build_retrieval_service
    from tensor_truth_amd.retrievers import AutoMergingRetriever
    from tensor_truth_amd.schema import MetadataMode, NodeWithScore
    from tensor_truth_amd.semantic import SemanticSplitter, adjacent_distances, breakpoints_from_distances, split_sentences

    cfg, xcfg = EncoderConfig(**SMALL), EncoderConfig(**XENC)
    ocfg, oxcfg = oe.EncoderConfig(**SMALL), oe.EncoderConfig(**XENC)
    mm.ModelManager.reset_instance()
    mgr = mm.ModelManager.get_instance()
    mgr.model_kwargs_overrides["test/bge-small-shaped"] = {"encoder_config": cfg, "synthetic_seed": 51, "pipeline_window": 64}
    mgr.model_kwargs_overrides["test/xenc-fp8"] = {"encoder_config": xcfg, "synthetic_seed": 52, "gemm_dtype": "fp8"}
    emb = mgr.get_embedder("test/bge-small-shaped", "cuda")
    W_e = {k: v.to(torch.bfloat16) for k, v in oe.synth_weights(ocfg, seed=51).items()}
    W_x = oe.synth_weights(oxcfg, seed=52)

    def oracle_embed(texts):
        seqs = [emb._tokenizer.encode(t, emb.max_length) for t in texts]
        out = []
        for lo in range(0, len(seqs), 256):
            ids, mask = _pad(seqs[lo:lo + 256], cfg.pad_id)
            out.append(oe.embed(ids, mask, W_e, ocfg, emulate_bf16=True))
        return torch.cat(out)

    docs = _docs()
    # ---- ingest: the reference's build_module with ChunkingStrategy.SEMANTIC_HIERARCHICAL ---------------------------
    index = build_index(docs, emb, persist_dir=str(tmp_path / "m"), chunking_strategy="semantic_hierarchical",
                        chunk_sizes=[96, 40, 20], chunk_overlap=4, semantic_buffer_size=1, semantic_breakpoint_threshold=80)
    leaves = get_leaf_nodes(index.docstore.values())
    assert index.n == len(leaves) > 100 and len(index.docstore) > index.n

    # ---- A. semantic splitter vs oracle ------------------------------------------------------------------------------
    sp = SemanticSplitter(emb, buffer_size=1, breakpoint_percentile_threshold=80)
    checked_docs = same_cuts = 0
    worst = 0.0
    for doc in docs[:8]:
        sents = split_sentences(doc.get_content())
        groups = sp._groups(sents)
        got_d = adjacent_distances(emb._embed_texts(groups, "")).cpu()
        e = oracle_embed(groups)
        want_d = 1 - torch.nn.functional.cosine_similarity(e[:-1], e[1:], dim=1)
        worst = max(worst, (got_d - want_d).abs().max().item())
        thr = float(np.percentile(want_d.numpy().astype(np.float64), 80))
        decisive = bool(((want_d - thr).abs() > 4e-3).all())        # no distance within tolerance of the threshold
        checked_docs += 1
        if breakpoints_from_distances(got_d.tolist(), 80) == breakpoints_from_distances(want_d.tolist(), 80):
            same_cuts += 1
        elif decisive:
            raise AssertionError("semantic cuts differ although every oracle distance is clear of the threshold")
    assert worst < 2e-3, f"adjacent-cosine distance error {worst}"
    assert same_cuts >= checked_docs - 2, (same_cuts, checked_docs)

    # ---- B. leaf embeddings in the matrix vs oracle (text = metadata-prefixed EMBED content, builder.py:437-442) -----
    ids_in_rows = [index.docstore[nid] for nid in index.leaf_ids]
    want_E = oracle_embed([nd.get_content(metadata_mode=MetadataMode.EMBED) for nd in ids_in_rows])
    got_E = index.matrix.float().cpu()
    cos = (got_E * want_E).sum(1)
    assert cos.min().item() >= 0.999, cos.min().item()
    assert "file_name" not in ids_in_rows[0].get_content(metadata_mode=MetadataMode.EMBED)

    # ---- the reference's engine: AutoMerging(index.as_retriever(k)) -> MultiIndex -> [fp8 reranker, cutoff] ----------
    params = {"reranker_model": "test/xenc-fp8", "reranker_top_n": 5, "similarity_top_k": 16,
              "confidence_cutoff": 0.35, "confidence_cutoff_hard": 0.0}
    svc = build_retrieval_service([index], params, device="cuda", manager=mgr)
    rr = mgr.get_reranker("test/xenc-fp8", top_n=5, device="cuda")
    assert rr.model.gemm_dtype == "fp8"
    # self-queries (a leaf's own text: decisive top-1) and topic queries
    probe_rows = [7, len(leaves) // 3, len(leaves) // 2, len(leaves) - 5]
    # (a self-query is the text that was EMBEDDED for the leaf -- metadata lines included -- so its top-1 is decisive)
    queries = [ids_in_rows[r].get_content(metadata_mode=MetadataMode.EMBED) for r in probe_rows] + [" ".join(t[:6]) for t in TOPICS]
    W_q = oracle_embed(queries)            # (bge-small-shaped test model: no query instruction for this name)
    assert emb.query_instruction == ""
    o_s, o_i, o_gap = osc.scan_topk(want_E.to(torch.bfloat16), W_q.to(torch.bfloat16), 16)
    dense = W_q.to(torch.bfloat16).float() @ want_E.to(torch.bfloat16).float().T        # oracle score of EVERY leaf
    row_of = {nid: r for r, nid in enumerate(index.leaf_ids)}
    base = index.as_retriever(similarity_top_k=16)
    amr = AutoMergingRetriever(base, index.docstore)
    overlaps, taus, top5, taus16 = [], [], [], []
    n_sep = n_exact = 0
    SCAN_TOL = 4e-3      # embeddings agree to cos >= 0.999 / 2e-3 per component: scores of unit vectors within ~4e-3
    for qi, q in enumerate(queries):
        # ---- C. retrieval + auto-merge.  The random-weight model crowds all texts together (neighbours at cos > 0.99),
        # so WHICH rows come back is decided inside the tolerance; what must hold for every returned row, always: its
        # score equals the oracle's score of that row, and that score would have made the oracle's top-16 (up to
        # tolerance) -- i.e. the product's list is a valid top-16 of the oracle's scores.  Identical lists where the
        # oracle's own ranking is clear-cut.
        hits = base.retrieve(q)
        got_ids = [h.node.id_ for h in hits]
        want_ids = [index.leaf_ids[int(j)] for j in o_i[qi]]
        overlaps.append(len(set(got_ids) & set(want_ids)) / 16.0)
        assert len(hits) == 16
        for h in hits:
            cos_prod = 1.0 + math.log(h.score) / 2.0                   # chroma mapping score = exp(-(2 - 2 cos))
            cos_orac = float(dense[qi, row_of[h.node.id_]])
            assert abs(cos_prod - cos_orac) <= SCAN_TOL, (qi, cos_prod, cos_orac)
            assert cos_orac >= float(o_s[qi, -1]) - SCAN_TOL, (qi, cos_orac, float(o_s[qi, -1]))
        if qi < len(probe_rows):                                       # a self-query scores ~1 against its own leaf
            assert index.leaf_ids[probe_rows[qi]] in got_ids and hits[0].score > 0.98
        if o_gap[qi] > 2 * SCAN_TOL:
            assert got_ids == want_ids
            n_exact += 1
        merged = amr.retrieve(q)
        # the merge is host logic over (ids, scores): feeding it the product's own hit list must reproduce `merged`
        again = AutoMergingRetriever(base, index.docstore).merge(
            [NodeWithScore(node=h.node, score=h.score) for h in hits])
        assert [m.node.id_ for m in again] == [m.node.id_ for m in merged] and len(merged) <= 16
        # ---- D. fp8 rerank of the merged candidates vs fp32 oracle scores on the same token ids
        res = svc.retrieve(q)
        assert res.num_sources == len(res.source_nodes) <= 5 and res.confidence_level in ("normal", "low")
        texts = [m.node.get_content(metadata_mode=MetadataMode.EMBED) for m in merged]
        pair_ids = [rr._tokenizer.encode_pair(q, t, rr.max_length)[0] for t in texts]
        ids, mask = _pad(pair_ids, xcfg.pad_id)
        want = oe.rerank_scores(ids, mask, W_x, oxcfg)                       # plain fp32
        got = torch.tensor(rr.predict([(q, t) for t in texts]))
        err = (got - want).abs().max().item()
        assert err <= FP8_BOUND_2L, f"query {qi}: fp8 score error {err}"
        # ... and the same candidates through the bf16 reranker (the PRIMARY mode; fp8 is the labelled throughput variant)
        rr.model.set_gemm_dtype("bf16")
        got16 = torch.tensor(rr.predict([(q, t) for t in texts]))
        rr.model.set_gemm_dtype("fp8")
        err16 = (got16 - want).abs().max().item()
        assert err16 <= 2e-2, f"query {qi}: bf16 score error {err16}"
        assert_order_on_separable(want.numpy(), got16.numpy(), max(2 * err16, 1e-4), f"config-5 query {qi} (bf16)")
        taus16.append(kendall_tau(want.numpy(), got16.numpy()))
        # order wherever the oracle separates two candidates by more than twice the error MEASURED on this query
        n_sep += assert_order_on_separable(want.numpy(), got.numpy(), max(2 * err, 1e-4), f"config-5 query {qi}")
        taus.append(kendall_tau(want.numpy(), got.numpy()))
        top5.append(topn_overlap(want.numpy(), got.numpy(), min(5, len(texts))))
        # the service returns the top-n of exactly these scores
        by_score = sorted(range(len(texts)), key=lambda i: -got[i].item())[:5]
        assert [n.node.id_ for n in res.source_nodes] == [merged[i].node.id_ for i in by_score]
    print(f"config 5 composed: {index.n} leaves / {len(index.docstore)} nodes; splitter distance err {worst:.1e}, cuts equal on "
          f"{same_cuts}/{checked_docs} docs; leaf cos min {cos.min().item():.5f}; retrieval: every hit a valid oracle top-16 member within {SCAN_TOL}, overlap@16 mean {np.mean(overlaps):.2f}, {n_exact} clear-cut queries identical; "
          f"fp8 rerank vs fp32: Kendall tau mean {np.mean(taus):.2f}, top-5 overlap mean {np.mean(top5):.2f}, {n_sep} separable pairs ordered; "
          f"bf16 rerank vs fp32: Kendall tau mean {np.mean(taus16):.2f}")
    # (Kendall tau is informational here: the merged candidates of this toy model are near-duplicates whose fp32 scores lie within
    # ~1e-3 of each other -- inside BOTH modes' error -- so neither mode has a ranking to preserve; the gates are the score bounds and
    # the separable-pair order above, and, at full depth, tests/test_rank_agreement_gpu.py)
    # (overlap with the oracle's own top-16 is informational: the lists differ inside the tolerance, see stage C)
    assert n_sep >= 1 and np.mean(overlaps) >= 0.4 and np.mean(top5) >= 0.6 and np.mean(taus) >= 0.5
    mm.ModelManager.reset_instance()


def test_config5_retrieval_stage_is_decidable_in_the_reference_precision(dev, built_lib, tmp_path):
    """Stages A-C again with the embedder in the reference's own precision (``precision="reference"``: fp32 semantics) against
    the plain fp32 oracle.  In bf16 the random-weight model's neighbours sit inside the comparison tolerance, so the bf16 test
    above can only check membership-within-tolerance; here product and oracle embeddings agree to ~1e-5, the bf16 rows of the
    index matrix come out (almost) bit-identical, and the lists themselves must agree: same semantic cuts, same leaves, the
    same ordered top-16 on at least 17 of 20 queries (measured 19; a differing row must score within 1e-4 of the oracle's rank 16),
    the same auto-merged result."""
    from tensor_truth_amd import model_manager as mm
    from tensor_truth_amd.encoder import EncoderConfig
    from tensor_truth_amd.index_builder import build_index
    from tensor_truth_amd.retrievers import AutoMergingRetriever
    from tensor_truth_amd.schema import MetadataMode, NodeWithScore
    from tensor_truth_amd.semantic import SemanticSplitter, adjacent_distances, breakpoints_from_distances, split_sentences

    cfg, ocfg = EncoderConfig(**SMALL), oe.EncoderConfig(**SMALL)
    mm.ModelManager.reset_instance()
    mgr = mm.ModelManager.get_instance()
    mgr.set_precision("reference")                                  # the config key: every model this manager loads
    mgr.model_kwargs_overrides["test/bge-small-shaped"] = {"encoder_config": cfg, "synthetic_seed": 51, "pipeline_window": 64}
    emb = mgr.get_embedder("test/bge-small-shaped", "cuda")
    assert emb.precision.startswith("reference")
    W_e = oe.synth_weights(ocfg, seed=51)                          # plain fp32, no emulation

    def oracle_embed(texts):
        seqs = [emb._tokenizer.encode(t, emb.max_length) for t in texts]
        out = []
        for lo in range(0, len(seqs), 256):
            ids, mask = _pad(seqs[lo:lo + 256], cfg.pad_id)
            out.append(oe.embed(ids, mask, W_e, ocfg))
        return torch.cat(out)

    docs = _docs()
    index = build_index(docs, emb, persist_dir=str(tmp_path / "m"), chunking_strategy="semantic_hierarchical",
                        chunk_sizes=[96, 40, 20], chunk_overlap=4, semantic_buffer_size=1, semantic_breakpoint_threshold=80)
    # A. semantic cuts: identical on every document whose oracle distances are clear of the percentile threshold by 1e-4
    sp = SemanticSplitter(emb, buffer_size=1, breakpoint_percentile_threshold=80)
    worst, same, decisive_docs, gaps_seen = 0.0, 0, 0, []
    for doc in docs[:12]:
        groups = sp._groups(split_sentences(doc.get_content()))
        got_d = adjacent_distances(emb._embed_texts(groups, "")).cpu()
        e = oracle_embed(groups)
        want_d = 1 - torch.nn.functional.cosine_similarity(e[:-1], e[1:], dim=1)
        worst = max(worst, (got_d - want_d).abs().max().item())
        thr = float(np.percentile(want_d.numpy().astype(np.float64), 80))
        equal = breakpoints_from_distances(got_d.tolist(), 80) == breakpoints_from_distances(want_d.tolist(), 80)
        same += equal
        # the percentile threshold interpolates between two of the distances; the cut set is decided unless the product's error
        # can carry a distance across it, i.e. unless the nearest distance on the OTHER side of it is closer than 2e-4
        srt = np.sort(want_d.numpy().astype(np.float64))
        above, below = srt[srt > thr], srt[srt <= thr]
        if len(above) and len(below):
            gaps_seen.append((float(above.min() - below.max()), bool(equal)))
    assert worst < 5e-5, worst
    # (this toy model's distances sit within ~1e-3 of each other: a document counts as decided when the two distances that
    # straddle the threshold are further apart than four times the worst error measured above)
    for gap, equal in gaps_seen:
        if gap > 4 * worst:
            decisive_docs += 1
            assert equal, f"semantic cuts differ although the distances around the threshold are {gap:.1e} apart (error {worst:.1e})"
    assert same >= 11, (same, decisive_docs, gaps_seen)
    # B. leaf embeddings
    rows = [index.docstore[nid] for nid in index.leaf_ids]
    want_E = oracle_embed([nd.get_content(metadata_mode=MetadataMode.EMBED) for nd in rows])
    got_E = index.matrix.float().cpu()
    same_bits = (got_E == want_E.to(torch.bfloat16).float()).float().mean().item()
    assert same_bits > 0.99, same_bits                              # the bf16 rows of the matrix, element for element
    # C. retrieval + auto-merge against the oracle's exact scan of ITS matrix
    probe_rows = [7, len(rows) // 3, len(rows) // 2, len(rows) - 5]
    queries = [rows[r].get_content(metadata_mode=MetadataMode.EMBED) for r in probe_rows] + [" ".join(t[:6]) for t in TOPICS] + \
              [" ".join(TOPICS[i % 4][j] for j in (i % 7, (i + 3) % 10, (2 * i + 1) % 10, 9 - i % 5)) for i in range(12)]
    W_q = oracle_embed(queries).to(torch.bfloat16)
    K = 16
    o_s, o_i, _ = osc.scan_topk(want_E.to(torch.bfloat16), W_q, K + 1)
    base = index.as_retriever(similarity_top_k=K)
    amr = AutoMergingRetriever(base, index.docstore)
    n_identical = 0
    for qi, q in enumerate(queries):
        hits = base.retrieve(q)
        got_ids = [h.node.id_ for h in hits]
        want_ids = [index.leaf_ids[int(j)] for j in o_i[qi, :K]]
        # both sides scan (almost) the same bf16 matrix exactly (bf16 scores tie often: the toy model's neighbours sit at cos > 0.99),
        # so the lists can only differ where one of the few differing matrix elements moves a near-tie
        if got_ids == want_ids:
            n_identical += 1
            merged = amr.retrieve(q)
            oracle_hits = [NodeWithScore(node=index.docstore[i], score=h.score) for i, h in zip(want_ids, hits)]
            again = AutoMergingRetriever(base, index.docstore).merge(oracle_hits)
            assert [m.node.id_ for m in merged] == [m.node.id_ for m in again]
        else:   # every row the product returned instead scores within 1e-4 of the oracle's rank-16 score
            row_of = {nid: r for r, nid in enumerate(index.leaf_ids)}
            dense = W_q[qi].float() @ want_E.to(torch.bfloat16).float().T
            for nid in set(got_ids) - set(want_ids):
                assert float(dense[row_of[nid]]) >= float(o_s[qi, K - 1]) - 1e-4, (qi, nid)
    print(f"config 5, reference precision: splitter distance err {worst:.1e}, cuts equal on {same}/12 docs ({decisive_docs} decisive); "
          f"{same_bits:.4f} of the matrix elements bit-identical to the oracle's bf16 rows; ordered top-{K} identical on "
          f"{n_identical}/{len(queries)} queries")
    assert n_identical >= len(queries) * 17 // 20, n_identical
    mm.ModelManager.reset_instance()


@pytest.mark.parametrize("strategy", ["semantic_hierarchical", "hierarchical"])
def test_worker_process_ingest_builds_the_same_index(dev, built_lib, strategy):
    """build_index with the host work in worker processes (ingest_workers.py: sentence splitting, hierarchy, tokenization off the
    feeding process) against the single-process pipeline: the same leaf texts in the same order and the SAME BITS in the index
    matrix -- cuts come from the same distances (embeddings do not depend on the batch they travel in), leaves from the same
    token ids."""
    from tensor_truth_amd import model_manager as mm
    from tensor_truth_amd.encoder import EncoderConfig
    from tensor_truth_amd.index_builder import build_index

    cfg = EncoderConfig(**SMALL)
    mm.ModelManager.reset_instance()
    mgr = mm.ModelManager.get_instance()
    mgr.model_kwargs_overrides["test/emb"] = {"encoder_config": cfg, "synthetic_seed": 5}
    emb = mgr.get_embedder("test/emb", "cuda")
    docs = _docs(n_docs=150, seed=4)
    kw = dict(chunking_strategy=strategy, chunk_sizes=[64, 32, 16], chunk_overlap=4)
    a = build_index(docs, emb, workers=0, **kw)
    b = build_index(docs, emb, workers=3, **kw)
    torch.cuda.synchronize()
    ta = [a.docstore[i].text for i in a.leaf_ids]
    tb = [b.docstore[i].text for i in b.leaf_ids]
    assert a.n == b.n > 400 and ta == tb
    assert torch.equal(a._mat[: a.n], b._mat[: b.n])
    # every hierarchy level reached the docstore, with the same parent / child structure
    assert len(a.docstore) == len(b.docstore)
    pa = sorted((n.text, len(n.child_ids), n.parent_id is None) for n in a.docstore.values())
    pb = sorted((n.text, len(n.child_ids), n.parent_id is None) for n in b.docstore.values())
    assert pa == pb
    mm.ModelManager.reset_instance()


def test_reference_geometry_with_subword_tokenizer_workers_and_pair_pool(dev, built_lib, monkeypatch):
    """Round 5 (VERDICT r04 items 2 + 3): the reference's own chunk geometry -- build_index with NO chunk sizes = [2048, 512, 256] /
    64 (indexing/builder.py:304-307), sizes counted in sub-word tokens (token_counter="embedder") -- through the trained 250 002-piece
    Unigram tokenizer: (a) leaves hold at most 256 sub-word tokens and the worker-process build equals the in-process build bit for
    bit; (b) a coalesced rerank batch tokenised in the pair-tokeniser worker processes scores exactly as the in-process path, and
    both equal the serial per-call results."""
    import os
    import sys
    import threading

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import synth_text as st
    from tensor_truth_amd import ingest_workers as iw
    from tensor_truth_amd import model_manager as mm
    from tensor_truth_amd.encoder import EncoderConfig
    from tensor_truth_amd.index_builder import DEFAULT_CHUNK_OVERLAP, DEFAULT_CHUNK_SIZES, build_index
    from tensor_truth_amd.schema import NodeWithScore, QueryBundle, TextNode

    assert DEFAULT_CHUNK_SIZES == [2048, 512, 256] and DEFAULT_CHUNK_OVERLAP == 64
    tk = st.unigram_tokenizer()
    ecfg = EncoderConfig(**{**XENC, "vocab_size": 250_002, "max_pos": 2100, "num_labels": 0})
    rcfg = EncoderConfig(**{**XENC, "vocab_size": 250_002, "max_pos": 514})
    mm.ModelManager.reset_instance()
    mgr = mm.ModelManager.get_instance()
    mgr.model_kwargs_overrides["test/emb-u"] = {"encoder_config": ecfg, "synthetic_seed": 5, "torch_dtype": "bfloat16", "tokenizer": tk}
    mgr.model_kwargs_overrides["test/rr-u"] = {"encoder_config": rcfg, "synthetic_seed": 6, "torch_dtype": "bfloat16", "tokenizer": tk}
    emb = mgr.get_embedder("test/emb-u", "cuda")
    rng = np.random.default_rng(8)
    lex = st.lexicon()
    docs = []
    for d in range(70):
        sents = []
        for block in range(3):
            band = int(rng.integers(0, 50)) * 1000
            for _ in range(int(rng.integers(25, 45))):
                sents.append(" ".join(lex[band + int(j)] for j in rng.integers(0, 1000, size=int(rng.integers(8, 20)))) + ".")
        docs.append(TextNode(text=" ".join(sents), metadata={"title": f"doc {d}"}))
    kw = dict(chunking_strategy="semantic_hierarchical", token_counter="embedder")          # no chunk sizes: the reference's defaults
    a = build_index(docs, emb, workers=0, **kw)
    b = build_index(docs, emb, workers=3, **kw)
    torch.cuda.synchronize()
    ta = [a.docstore[i].text for i in a.leaf_ids]
    assert a.n == b.n > 150 and ta == [b.docstore[i].text for i in b.leaf_ids]
    assert torch.equal(a._mat[: a.n], b._mat[: b.n])
    leaf_tokens = [len(x) - 2 for x in tk.encode_batch(ta)]
    assert max(leaf_tokens) <= 256 and np.mean(leaf_tokens) > 120, (max(leaf_tokens), np.mean(leaf_tokens))
    roots = [n for n in a.docstore.values() if n.parent_id is None]
    assert max(len(tk.encode(n.text)) - 2 for n in roots) <= 2048

    # (b) rerank: 6 concurrent callers x 40 candidates through the coalescing front -> one batch of 240 pairs (>= 96: the pool)
    rr = mgr.get_reranker("test/rr-u", top_n=5, device="cuda")
    cands = [[NodeWithScore(node=TextNode(text=ta[int(j)], id_=f"n{int(j)}"), score=0.0) for j in rng.choice(len(ta), 40, replace=False)]
             for _ in range(6)]
    queries = [" ".join(ta[int(rng.integers(0, len(ta)))].replace(".", " ").split()[:12]) for _ in range(6)]

    def run_all():
        out, errs = [None] * 6, []

        def one(i):
            try:
                out[i] = [(x.node.id_, x.score) for x in rr.postprocess_nodes(list(cands[i]), query_bundle=QueryBundle(query_str=queries[i]))]
            except Exception as exc:  # noqa: BLE001
                errs.append(exc)

        gate = threading.Barrier(6)
        threads = [threading.Thread(target=lambda i=i: (gate.wait(), one(i))) for i in range(6)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errs, errs
        return out

    monkeypatch.setenv("TT_PAIR_WORKERS", "3")
    assert iw.get_pair_pool(tk, wait=True, max_length=rr.max_length) is not None      # (request threads tokenise in process until it is up)
    pairs0 = rr.stats["pairs"]
    pooled = run_all()
    pool = iw.get_pair_pool(tk)
    assert pool is not None and rr.stats["pairs"] - pairs0 == 240
    monkeypatch.setenv("TT_PAIR_WORKERS", "0")
    local = run_all()
    serial = [[(x.node.id_, x.score) for x in rr.postprocess_nodes(list(cands[i]), query_bundle=QueryBundle(query_str=queries[i]))] for i in range(6)]
    assert pooled == local == serial
    assert all(len(r) == 5 and r[0][1] >= r[-1][1] for r in serial)
    mm.ModelManager.reset_instance()


def test_leaves_tokenised_once_at_ingest_rerank_from_stored_ids(dev, built_lib, monkeypatch):
    """Round 5: build_index(keep_leaf_token_ids=True) keeps every leaf's token ids (the embedder's tokenizer; bge-m3 and
    bge-reranker-v2-m3 share XLM-R's); build_retrieval_service hands them to the reranker, which then tokenises only the query.
    The composed service must return the same nodes with the SAME score bits as with the reranker tokenising every passage string --
    from several threads (coalesced batches mixing leaves with auto-merged parents, which have no stored ids) and for a lone caller;
    a reranker with another tokenizer refuses the ids."""
    import os
    import sys
    import threading

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import synth_text as st
    from tensor_truth_amd import model_manager as mm
    from tensor_truth_amd.encoder import EncoderConfig
    from tensor_truth_amd.index_builder import build_index
    from tensor_truth_amd.retrieval_service import build_retrieval_service
    from tensor_truth_amd.schema import TextNode
    from tensor_truth_amd.tokenization import HashTokenizer

    monkeypatch.setenv("TT_PAIR_WORKERS", "0")
    tk = st.unigram_tokenizer()
    ecfg = EncoderConfig(**{**XENC, "vocab_size": 250_002, "max_pos": 2100, "num_labels": 0})
    rcfg = EncoderConfig(**{**XENC, "vocab_size": 250_002, "max_pos": 514})
    mm.ModelManager.reset_instance()
    mgr = mm.ModelManager.get_instance()
    mgr.model_kwargs_overrides["BAAI/bge-m3"] = {"encoder_config": ecfg, "synthetic_seed": 5, "torch_dtype": "bfloat16", "tokenizer": tk}
    mgr.model_kwargs_overrides["BAAI/bge-reranker-v2-m3"] = {"encoder_config": rcfg, "synthetic_seed": 6, "torch_dtype": "bfloat16", "tokenizer": tk}
    emb = mgr.get_embedder("BAAI/bge-m3", "cuda")
    rng = np.random.default_rng(18)
    lex = st.lexicon()
    docs = []
    for d in range(80):
        sents = []
        for block in range(3):
            band = int(rng.integers(0, 30)) * 1000
            for _ in range(int(rng.integers(20, 40))):
                sents.append(" ".join(lex[band + int(j)] for j in rng.integers(0, 1000, size=int(rng.integers(8, 20)))) + ".")
        docs.append(TextNode(text=" ".join(sents), metadata={"title": f"doc {d}", "file_name": f"d{d}.md"}))
    index = build_index(docs, emb, chunking_strategy="semantic_hierarchical", chunk_sizes=[256, 64, 32], chunk_overlap=8,
                        token_counter="embedder", workers=3, keep_leaf_token_ids=True)
    torch.cuda.synchronize()
    assert index.leaf_token_ids is not None and len(index.leaf_token_ids) == index.n > 300
    some = index.leaf_ids[7]
    from tensor_truth_amd.schema import MetadataMode

    assert index.leaf_token_ids[some].tolist() == tk.encode(index.docstore[some].get_content(metadata_mode=MetadataMode.EMBED))[1:-1]
    params = {"reranker_top_n": 5, "similarity_top_k": 24, "confidence_cutoff": 0.0}
    svc = build_retrieval_service([index], params, device="cuda", manager=mgr)
    rr = mgr.get_reranker("BAAI/bge-reranker-v2-m3", top_n=5, device="cuda")
    view = svc._node_postprocessors[0]
    #  the source belongs to the SERVICE, handed over per call: ModelManager's cached reranker is not mutated
    assert view.token_source is not None and rr._token_source is None and view._reranker is rr
    live = [x for x in index.leaf_ids if x is not None]
    queries = [" ".join(index.docstore[live[int(rng.integers(0, len(live)))]].text.replace(".", " ").split()[:10]) + f" q{i}" for i in range(24)]

    def run(qs, threads):
        out, errs = [None] * len(qs), []

        def work(t):
            try:
                for i in range(t, len(qs), threads):
                    r = svc.retrieve(qs[i])
                    out[i] = [(n.node.id_, n.score) for n in r.source_nodes]
            except Exception as exc:  # noqa: BLE001
                errs.append(exc)

        ts = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errs, errs
        return out

    pre0 = rr.stats.get("pretokenized", 0)
    with_ids = run(queries, 6)
    assert rr.stats.get("pretokenized", 0) - pre0 > 24 * 5           # most candidates are leaves with stored ids
    lone = run(queries[:3], 1)
    if hasattr(svc._retriever, "clear_cache"):
        svc._retriever.clear_cache()
    view.token_source = None
    from_text = run(queries, 6)
    assert with_ids == from_text and lone == from_text[:3]
    assert all(len(r) == 5 for r in from_text)
    # the ids survive persist / load (leaf_tokens.<generation>.npz beside the matrix), and a deleted leaf's ids go with it
    import tempfile

    from tensor_truth_amd.vector_index import HipVectorIndex

    with tempfile.TemporaryDirectory() as pdir:
        index.persist(pdir)
        again = HipVectorIndex.load(pdir, embed_model=emb)
        assert again.leaf_token_ids is not None and set(again.leaf_token_ids) == set(index.leaf_token_ids)
        assert all(np.array_equal(again.leaf_token_ids[k], index.leaf_token_ids[k]) for k in list(index.leaf_token_ids)[::17])
        assert again.token_source()[1] == index.token_source()[1]
        victim = again.leaf_ids[3]
        again.delete([victim])
        assert victim not in again.leaf_token_ids
        again.persist(pdir)
        third = HipVectorIndex.load(pdir, embed_model=emb)
        assert len(third.leaf_token_ids) == len(index.leaf_token_ids) - 1 and victim not in third.leaf_token_ids
        assert sum(f.startswith("leaf_tokens.") for f in os.listdir(pdir)) == 1          # older generations removed
    # another tokenizer: refused, strings stay the path
    src, sig, instr = index.token_source()
    assert rr.attach_token_source(src, sig, instr) is True
    assert rr.attach_token_source(src, "hash:xlmr:250002", instr) is False and rr._token_source is None
    assert rr.attach_token_source(src, sig, "passage: ") is False
    mm.ModelManager.reset_instance()
