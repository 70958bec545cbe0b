"""GPU parity at the BASELINE.json configurations (full model shapes, full depth).

C1: 1k-chunk toy corpus, bge-small-en-v1.5 shape (BERT 12L x 384), brute-force cosine top-10, no rerank.
C2/C3 (encoder half): bge-m3 / bge-reranker-v2-m3 shapes (XLM-R 24L x 1024) -- embeddings and rerank scores
of full-depth models against the fp32 CPU oracle, i.e. bf16 error accumulated over all 24 layers.
(The 1M x 1024 scan half of C2 is in test_scan_gpu.py::test_full_size_config_c2_against_oracle.)
"""
import numpy as np
import pytest
import torch

from oracle import encoder as oe
from oracle import scan as osc

pytestmark = pytest.mark.gpu


def _pad(seqs, pad):
    L = max(len(s) for s in seqs)
    ids = torch.full((len(seqs), L), pad, dtype=torch.int64)
    mask = torch.zeros(len(seqs), L, dtype=torch.int64)
    for b, s in enumerate(seqs):
        ids[b, : len(s)] = torch.tensor(s)
        mask[b, : len(s)] = 1
    return ids, mask


def test_config_c1_bge_small_toy_corpus_end_to_end(dev, built_lib):
    from tensor_truth_amd.encoder import BGE_SMALL_EN_V15, Encoder, EncoderWeights
    from tensor_truth_amd import scan as tscan

    cfg = BGE_SMALL_EN_V15
    ocfg = oe.EncoderConfig(**cfg.__dict__)
    W = oe.synth_weights(ocfg, seed=3)
    Wb = {k: v.to(torch.bfloat16) for k, v in W.items()}
    g = torch.Generator().manual_seed(777)
    # 1k chunks (ragged 16..128 tokens) + 16 queries; every other query repeats a chunk verbatim (planted hit)
    lens = torch.randint(16, 129, (1000,), generator=g).tolist()
    chunks = [[101] + torch.randint(1000, cfg.vocab_size, (n - 2,), generator=g).tolist() + [102] for n in lens]
    queries = []
    for i in range(16):
        if i % 2 == 0:
            queries.append(list(chunks[37 * i + 5]))
        else:
            n = int(torch.randint(8, 25, (1,), generator=g))
            queries.append([101] + torch.randint(1000, cfg.vocab_size, (n - 2,), generator=g).tolist() + [102])
    enc = Encoder(EncoderWeights(cfg, W, dev))
    c_emb, c16 = enc.embed(chunks)
    q_emb, q16 = enc.embed(queries)
    s, i = tscan.scan_topk(c16, q16, 10)
    torch.cuda.synchronize()
    # oracle: same token ids through the fp32-math oracle with the bf16 rounding points emulated
    ids, mask = _pad(chunks, cfg.pad_id)
    want_c = oe.embed(ids, mask, Wb, ocfg, emulate_bf16=True)
    ids, mask = _pad(queries, cfg.pad_id)
    want_q = oe.embed(ids, mask, Wb, ocfg, emulate_bf16=True)
    assert ((c_emb.cpu() * want_c).sum(1) >= 0.999).all()
    assert ((q_emb.cpu() * want_q).sum(1) >= 0.999).all()
    # scan parity on the embeddings the GPU produced (bit-exact indices on tie-free queries)
    w_s, w_i, gap = osc.scan_topk(c16.cpu(), q16.cpu(), 10)
    tf = gap > 1e-6
    assert tf.sum() >= 6     # random-init embeddings crowd together: several queries have near-ties
    assert torch.equal(i.cpu().to(torch.int64)[tf], w_i[tf])
    assert torch.allclose(s.cpu(), w_s, rtol=1e-3, atol=1e-6)
    # end-to-end vs the all-oracle pipeline (oracle embeddings -> oracle scan).  With random-init weights the
    # embeddings of unrelated random token strings crowd together (cosines within ~1e-2), so bf16-level noise may
    # reorder those near-ties; what must hold: planted hits rank first in both pipelines, and wherever the
    # oracle's own ranking is clear-cut (gap > 2e-3) the two rankings are identical.
    o_s, o_i, o_gap = osc.scan_topk(want_c.to(torch.bfloat16), want_q.to(torch.bfloat16), 10)
    got_i = i.cpu().to(torch.int64)
    for qi in range(0, 16, 2):
        assert int(got_i[qi, 0]) == 37 * qi + 5 == int(o_i[qi, 0])
        assert s[qi, 0].item() > 0.99
    wide = o_gap > 2e-3
    assert torch.equal(got_i[wide], o_i[wide])


@pytest.mark.default_precision
def test_config_c1_bge_small_in_the_default_precision_runs_on_the_split_planes(dev, built_lib):
    """BASELINE config 1 (bge-small-en-v1.5: BERT 12L x 384, 12 heads of 32) with NO dtype named anywhere -- what an
    unchanged reference call asks for -- resolves to the reference precision ON THE MATRIX CORES ("f16x3": split-fp16 planes; rounds
    4-5 fell back to the fp32 MFMA at 1/16 of the bf16 rate for this geometry) and matches the PLAIN fp32 CPU oracle: embeddings to
    5e-5, cosine scores of the scan within 1e-3 relative of the all-oracle pipeline's, top-10 identical wherever the oracle's own
    ranking is clear-cut."""
    from tensor_truth_amd import precision
    from tensor_truth_amd import scan as tscan
    from tensor_truth_amd.encoder import BGE_SMALL_EN_V15
    from tensor_truth_amd.encoder_x3 import EncoderX3

    cfg = BGE_SMALL_EN_V15
    ocfg = oe.EncoderConfig(**cfg.__dict__)
    W = oe.synth_weights(ocfg, seed=3)
    assert precision.resolve(None) == "reference" and precision.reference_impl(cfg) == "f16x3"
    _, enc, desc = precision.build_encoder(cfg, W, dev, None, "config 1 embedder")
    assert isinstance(enc, EncoderX3) and enc.w.gemm_dtype == "f16x3" and "split-fp16" in desc
    g = torch.Generator().manual_seed(777)
    lens = torch.randint(16, 129, (1000,), generator=g).tolist()
    chunks = [[101] + torch.randint(1000, cfg.vocab_size, (n - 2,), generator=g).tolist() + [102] for n in lens]
    queries = []
    for i in range(16):
        if i % 2 == 0:
            queries.append(list(chunks[37 * i + 5]))
        else:
            n = int(torch.randint(8, 25, (1,), generator=g))
            queries.append([101] + torch.randint(1000, cfg.vocab_size, (n - 2,), generator=g).tolist() + [102])
    c_emb, c16 = enc.embed(chunks)
    q_emb, q16 = enc.embed(queries)
    one, _ = enc.embed([queries[3]])                       # one query alone: the skinny / relay GEMMs -- the same bits as in the batch
    s, i = tscan.scan_topk(c16, q16, 10)
    torch.cuda.synchronize()
    assert torch.equal(one.cpu()[0], q_emb.cpu()[3])
    ids, mask = _pad(chunks, cfg.pad_id)
    want_c = oe.embed(ids, mask, W, ocfg)                  # plain fp32 oracle: no rounding point emulated
    ids, mask = _pad(queries, cfg.pad_id)
    want_q = oe.embed(ids, mask, W, ocfg)
    assert (c_emb.cpu() - want_c).abs().max().item() <= 5e-5 and (q_emb.cpu() - want_q).abs().max().item() <= 5e-5
    # the scan on the embeddings the GPU produced: bit-exact indices on tie-free queries
    w_s, w_i, gap = osc.scan_topk(c16.cpu(), q16.cpu(), 10)
    tf = gap > 1e-6
    assert tf.sum() >= 6 and torch.equal(i.cpu().to(torch.int64)[tf], w_i[tf])
    # end to end against the all-oracle pipeline (fp32 oracle embeddings -> bf16 corpus rows -> oracle scan): scores within
    # north_star's 1e-3 relative, planted hits first, identical top-10 wherever the oracle's ranking is clear-cut
    o_s, o_i, o_gap = osc.scan_topk(want_c.to(torch.bfloat16), want_q.to(torch.bfloat16), 10)
    got_i = i.cpu().to(torch.int64)
    same = got_i == o_i
    rel = ((s.cpu() - o_s).abs() / o_s.abs().clamp_min(1e-6))[same].max().item()
    assert rel <= 1e-3, rel
    for qi in range(0, 16, 2):
        assert int(got_i[qi, 0]) == 37 * qi + 5 == int(o_i[qi, 0]) and s[qi, 0].item() > 0.99
    # Random-init embeddings of unrelated token strings crowd together (cosines 0.9995-0.9997, adjacent top-10 scores ~1e-5 apart), and
    # the two pipelines store corpus rows that agree to 5e-5 in bf16: where a row's elements round to different bf16 neighbours (one
    # ulp = 2^-9 relative each) its score moves by ~1e-4.  Rankings must agree up to ties of that size -- every chunk the GPU pipeline
    # returns is, by the ORACLE's own scores, within 5e-4 of the oracle's 10th best, and in non-increasing order within 5e-4
    S_o = want_q.to(torch.bfloat16).float() @ want_c.to(torch.bfloat16).float().T
    picked = torch.gather(S_o, 1, got_i)
    assert (picked >= o_s[:, 9:10] - 5e-4).all()
    assert (picked[:, 1:] <= picked[:, :-1] + 5e-4).all()
    assert same[::2, 0].all()                                          # every planted top-1 identical


def test_full_depth_bge_m3_embeddings_and_reranker_scores(dev, built_lib):
    from tensor_truth_amd.encoder import BGE_RERANKER_V2_M3, Encoder, EncoderWeights

    cfg = BGE_RERANKER_V2_M3                     # same encoder as bge-m3 + the classification head
    ocfg = oe.EncoderConfig(**cfg.__dict__)
    W = oe.synth_weights(ocfg, seed=11)          # 568 M parameters, HF-style init
    g = torch.Generator().manual_seed(5)
    lens = [96, 33, 64, 17, 80, 50]
    seqs = [[0] + torch.randint(4, cfg.vocab_size, (n - 2,), generator=g).tolist() + [2] for n in lens]
    enc = Encoder(EncoderWeights(cfg, W, dev))
    emb, _ = enc.embed(seqs)
    scores, logits = enc.rerank(seqs, want_logits=True)
    torch.cuda.synchronize()
    ids, mask = _pad(seqs, cfg.pad_id)
    torch.set_num_threads(max(1, min(32, torch.get_num_threads())))
    with torch.no_grad():
        hid = oe.encoder_forward(ids, mask, W, ocfg)                    # plain fp32, fp32 weights
        want_e = oe.cls_pool_normalize(hid)
        t = torch.tanh(hid[:, 0] @ W["classifier.dense.weight"].T + W["classifier.dense.bias"])
        want_l = (t @ W["classifier.out_proj.weight"].T + W["classifier.out_proj.bias"])[:, 0]
    cos = (emb.cpu() * want_e).sum(1)
    assert (cos >= 0.999).all(), cos                                     # SURVEY.md 8d embedding gate, 24 layers deep
    assert (emb.cpu() - want_e).abs().max().item() <= 4e-3
    err = (scores.cpu() - torch.sigmoid(want_l)).abs().max().item()
    assert err <= 2e-2, f"rerank score error after 24 bf16 layers: {err}"
    assert np.isfinite(logits.cpu().numpy()).all()

    # the fp8 mode (BASELINE config 5) at full depth against the same fp32 oracle: the stated, looser bound
    from tensor_truth_amd.encoder import pack_tokens

    enc.calibrate_fp8(pack_tokens(seqs, cfg, None, 512))
    enc.w.set_gemm_dtype("fp8")
    emb8, _ = enc.embed(seqs)
    scores8 = enc.rerank(seqs)
    enc.w.set_gemm_dtype("bf16")
    cos8 = (emb8.cpu() * want_e).sum(1)
    assert (cos8 >= 0.99).all(), cos8                                   # e4m3: 3 mantissa bits on every projection operand
    # 96 e4m3 GEMMs deep the logits of this random-init head (spanning about +-0.75) are off by ~0.3, i.e. up to ~0.15
    # on the sigmoid score (bf16: 0.02); tools/probes/fp8_depth.py: the static FFN scale adds nothing to that, it is the
    # 3-bit mantissa of the per-token / per-channel quantised operands
    err8 = (scores8.cpu() - torch.sigmoid(want_l)).abs().max().item()
    assert err8 <= 0.2, f"rerank score error after 24 fp8 layers: {err8}"
    print(f"full depth: bf16 cos min {cos.min().item():.5f} score err {err:.4f}; fp8 cos min {cos8.min().item():.5f} score err {err8:.4f}")


def test_full_size_rerank_is_batch_invariant(dev, built_lib):
    """BASELINE config 3 at full size (24-layer bge-reranker-v2-m3 shape, 16 queries x 50 pairs x 292 tokens): the score
    of a pair must not depend on what else is in the batch -- every kernel computes a token row / a sequence from its
    own data in a fixed order.  Permuting the batch permutes the scores, and half the batch alone scores the same bits.
    (The CPU oracle needs minutes per pair at this depth; depth parity itself is test_full_depth_models_vs_fp32_oracle.)"""
    import numpy as np

    from tensor_truth_amd.encoder import BGE_RERANKER_V2_M3, Encoder, EncoderWeights, pack_token_matrix, synthetic_state_device

    cfg = BGE_RERANKER_V2_M3
    enc = Encoder(EncoderWeights(cfg, synthetic_state_device(cfg, dev, seed=2), dev))
    rng = np.random.default_rng(11)
    pairs = rng.integers(4, cfg.vocab_size, size=(800, 292), dtype=np.int32)
    pairs[:, 0], pairs[:, -1] = 0, 2
    pairs[:, 34:36] = 2                                           # </s></s> between query and passage
    s_all = enc.rerank_packed(pack_token_matrix(pairs, cfg)).cpu()
    assert torch.isfinite(s_all).all() and ((s_all > 0) & (s_all < 1)).all()
    perm = rng.permutation(800)
    s_perm = enc.rerank_packed(pack_token_matrix(pairs[perm], cfg)).cpu()
    assert torch.equal(s_perm, s_all[torch.from_numpy(perm)])
    s_half = enc.rerank_packed(pack_token_matrix(pairs[:400], cfg)).cpu()
    assert torch.equal(s_half, s_all[:400])
    # three pairs alone: their CLS tail (64 rows) runs on the skinny GEMMs, the batch's (1024 rows) on the tiled ones
    s_few = enc.rerank_packed(pack_token_matrix(pairs[5:8], cfg)).cpu()
    assert torch.equal(s_few, s_all[5:8])


@pytest.mark.parametrize("precision", ["bf16", "reference"])
def test_the_references_session_sized_reranks_are_batch_invariant(dev, built_lib, precision):
    """The reference's own operating points (session defaults: 10 candidates per index module, ``rag_engine.py:592-593``) at full
    depth: 10 / 30 pairs x 292 tokens = 12 / 35 row tiles.  Those sizes take other kernels than a large batch does -- in the
    default precision the staged 128x128 split-plane kernel for the N = 1024 projections of 10 pairs, in bf16 the 128x128 kernel --
    and must score every pair with the bits it gets inside a 120-pair batch (all 24 layers: QKV with its transposed V third,
    residual and GELU epilogues, both plane outputs)."""
    import numpy as np

    from tensor_truth_amd.encoder import BGE_RERANKER_V2_M3, Encoder, EncoderWeights, pack_token_matrix, synthetic_state_device

    cfg = BGE_RERANKER_V2_M3
    state = synthetic_state_device(cfg, dev, seed=2)
    if precision == "bf16":
        enc = Encoder(EncoderWeights(cfg, state, dev))
    else:
        from tensor_truth_amd.encoder_x3 import EncoderWeightsX3, EncoderX3

        enc = EncoderX3(EncoderWeightsX3(cfg, state, dev, dtype=torch.float16))
    rng = np.random.default_rng(13)
    pairs = rng.integers(4, cfg.vocab_size, size=(120, 292), dtype=np.int32)
    pairs[:, 0], pairs[:, -1] = 0, 2
    pairs[:, 34:36] = 2
    s_all = enc.rerank_packed(pack_token_matrix(pairs, cfg)).cpu()
    assert torch.isfinite(s_all).all() and ((s_all > 0) & (s_all < 1)).all()
    assert pack_token_matrix(pairs[:30], cfg).n_rows == 8960 and pack_token_matrix(pairs[:10], cfg).n_rows == 3072
    for lo, hi in ((0, 30), (30, 40), (47, 77)):
        s_few = enc.rerank_packed(pack_token_matrix(pairs[lo:hi], cfg)).cpu()
        assert torch.equal(s_few.view(torch.int32), s_all[lo:hi].view(torch.int32)), (precision, lo, hi)


def test_query_embedding_alone_equals_query_embedding_in_a_batch(dev, built_lib):
    """One query on its own (64 token rows: every projection of the 24 layers is a skinny weight-streaming GEMM) gets
    bit for bit the embedding it gets inside a batch of 200 queries (8000 rows: tiled kernels)."""
    import numpy as np

    from tensor_truth_amd.encoder import BGE_M3, Encoder, EncoderWeights, pack_token_matrix, synthetic_state_device

    cfg = BGE_M3
    enc = Encoder(EncoderWeights(cfg, synthetic_state_device(cfg, dev, seed=1), dev))
    rng = np.random.default_rng(5)
    q = rng.integers(4, cfg.vocab_size, size=(200, 34), dtype=np.int32)
    q[:, 0], q[:, -1] = 0, 2
    e_all, _ = enc.embed_packed(pack_token_matrix(q, cfg))
    assert pack_token_matrix(q[:1], cfg).n_rows == 64 and pack_token_matrix(q[:6], cfg).n_rows == 256
    for lo, hi in ((0, 1), (7, 8), (10, 16), (100, 104)):
        e_few, _ = enc.embed_packed(pack_token_matrix(q[lo:hi], cfg))
        assert torch.equal(e_few, e_all[lo:hi]), (lo, hi)
    assert torch.isfinite(e_all).all() and (e_all.norm(dim=1) - 1).abs().max() < 1e-3
