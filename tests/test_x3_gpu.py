"""GPU: reference precision on the bf16 matrix cores (split-bf16, "bf16x3": csrc/x3_path.hip, gemm.hip GemmParams.x3)
against fp64 / the fp32 CPU oracle.  The mode exists to give the reference's unchanged calls -- which carry no dtype,
i.e. fp32 (services/model_manager.py:333-337, app_utils/config_schema.py:66-76) -- north_star's tolerance (scores within
1e-3 relative) at a third of the bf16 rate; full depth (4 x 50 pairs x 24 layers): test_rank_agreement_gpu.py."""
import numpy as np
import pytest
import torch

from oracle import encoder as oe

pytestmark = pytest.mark.gpu

XLMR = dict(arch="xlmr", vocab_size=2000, hidden=256, layers=3, heads=4, ffn=512, max_pos=300, type_vocab=1, pad_id=1,
            ln_eps=1e-5, num_labels=1)
WIDE = dict(arch="xlmr", vocab_size=1000, hidden=1024, layers=2, heads=16, ffn=4096, max_pos=300, type_vocab=1, pad_id=1,
            ln_eps=1e-5, num_labels=1)
# round 6: bge-small-en-v1.5 / ms-marco-MiniLM geometry -- 384 = 1.5 column tiles of 256, 12 heads of 32, ffn 1536
SMALL384 = dict(arch="bert", vocab_size=1500, hidden=384, layers=3, heads=12, ffn=1536, max_pos=300, type_vocab=2, pad_id=0,
                ln_eps=1e-12, num_labels=1)


def _pad(seqs, pad):
    L = max(len(s) for s in seqs)
    ids = torch.full((len(seqs), L), pad, dtype=torch.int64)
    mask = torch.zeros(len(seqs), L, dtype=torch.int64)
    for b, s in enumerate(seqs):
        ids[b, : len(s)] = torch.tensor(s)
        mask[b, : len(s)] = 1
    return ids, mask


def _planes(x):
    from tensor_truth_amd.encoder_x3 import split_planes

    return split_planes(x)


def _join(planes, cols):
    return planes[:, :cols].float() + planes[:, cols:2 * cols].float()


def test_split_planes_kernel_equals_the_torch_formula(dev, built_lib):
    from tensor_truth_amd import _lib

    lib = _lib.load_library()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(777, 320, generator=g) * torch.logspace(-6, 3, 320)
    out = torch.zeros((777, 640), dtype=torch.bfloat16, device=dev)
    _lib.check(lib.tt_split_planes(x.to(dev).data_ptr(), 777, 320, out.data_ptr(), None), "tt_split_planes")
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), _planes(x))
    rel = ((_join(out.cpu(), 320) - x).abs() / x.abs().clamp_min(1e-30)).max().item()
    assert rel <= 2.0 ** -16, rel                       # hi + lo carries x to 16+ bits


@pytest.mark.parametrize("m,n,k", [(256, 256, 128), (512, 768, 256), (256, 1024, 1024), (768, 256, 4096), (1024, 4096, 1024),
                                   (64, 1024, 1024), (192, 4096, 1024), (128, 1024, 4096),       # <= 256 rows: the skinny kernel
                                   # round 6: N = 64 j, the last column tile partial (bge-small: 384 / 1152 / 1536 columns, K = 384 / 1536)
                                   (256, 384, 384), (512, 1536, 384), (768, 384, 1536), (256, 64, 128), (512, 448, 256), (64, 384, 1536),
                                   # round 6, the staged kernel's four-stage ring around its depth: 4, 6 and 10 K-steps of 32 (K is a multiple of 64)
                                   (512, 128, 128), (256, 256, 192), (512, 384, 320)])
def test_gemm_x3_building_block(dev, built_lib, m, n, k):
    """tt_gemm_x3 against fp64 on the operands' own (hi + lo) values: error of an fp32-accumulated product, two orders
    below what one bf16 rounding of an operand costs (2^-9 relative)."""
    from tensor_truth_amd import _lib

    lib = _lib.load_library()
    g = torch.Generator().manual_seed(m + n + k)
    a, w = torch.randn(m, k, generator=g), torch.randn(n, k, generator=g) * 0.05
    bias, res = torch.randn(n, generator=g), torch.randn(m, n, generator=g)
    ap, wp = _planes(a), _planes(w)
    a64, w64 = _join(ap, k).double(), _join(wp, k).double()
    ref = a64 @ w64.T + bias.double()
    mag = (a64.abs() @ w64.abs().T)                                    # sum |a_i w_i|: the scale rounding errors grow with
    wants = {0: ref, 1: 0.5 * ref * (1 + torch.erf(ref / 2 ** 0.5)), 2: ref + res.double()}
    dap, dwp, db, dr = ap.to(dev), wp.to(dev), bias.to(dev), res.to(dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    for epi, want in wants.items():
        cp = torch.full((m, 2 * n), float("nan"), dtype=torch.bfloat16, device=dev)
        c32 = torch.full((m, n), float("nan"), device=dev)
        rc = lib.tt_gemm_x3(dap.data_ptr(), dwp.data_ptr(), db.data_ptr(), dr.data_ptr() if epi == 2 else None,
                            cp.data_ptr() if epi != 2 else None, c32.data_ptr() if epi == 2 else None, m, n, k, epi, st)
        _lib.check(rc, "tt_gemm_x3")
        torch.cuda.synchronize()
        got = c32.cpu().double() if epi == 2 else _join(cp.cpu(), n).double()
        assert torch.isfinite(got).all()
        # dropped lo.lo terms (2^-16 each, incoherent) + fp32 accumulation + (planes output) the 2^-17 split of the result
        bound = mag * (2.0 ** -15 / k ** 0.5 + 2.0 ** -22) + want.abs() * 2.0 ** -16 + 1e-6
        excess = ((got - want).abs() / bound).max().item()
        assert excess <= 1.0, (epi, excess)
        rel = ((got - want).abs().max() / want.abs().max()).item()
        assert rel <= 2e-5, (epi, rel)
    if m > 256:      # (the tiled kernel's column granularity is 64; the skinny kernels take any multiple of 16)
        with pytest.raises(_lib.TTError):
            _lib.check(lib.tt_gemm_x3(dap.data_ptr(), dwp.data_ptr(), db.data_ptr(), None, cp.data_ptr(), None, m, n + 8, k, 0, st), "x")


@pytest.mark.parametrize("planes", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n,k", [(1024, 1024), (1024, 4096), (2048, 1024)])
def test_gemm_x3_rows_do_not_depend_on_the_kernel_that_computed_them(dev, built_lib, n, k, planes):
    """Split planes, three kernels by row count: <= 256 rows the relay kernel, up to two 128x128 tiles per CU the staged 128x128
    kernel (round 6: the reference's own call, 10-20 pairs = 1-8 k rows, was one 256x256 tile's latency on a fraction of the chip),
    above that the 256x256 ping-pong kernel.  Per 32 K elements the products hi.hi, x_hi.w_lo, x_lo.w_hi into ONE accumulator, K ascending, the
    same epilogue operations: the first rows of every launch carry the same bits (planes out, GELU planes out, fp32 residual out)."""
    from tensor_truth_amd import _lib

    lib = _lib.load_library()
    g = torch.Generator().manual_seed(n + k)
    rows = (64, 256, 768, 256 * 128 * 128 // n, 512 * 128 * 128 // n, 16384)   # relay, relay, staged: part of a round, one round, two rounds; 256x256
    big = rows[-1]
    from tensor_truth_amd.encoder_x3 import split_planes

    ap = split_planes(torch.randn(big, k, generator=g), planes).to(dev)
    wp = split_planes(torch.randn(n, k, generator=g) * 0.04, planes).to(dev)
    bias, res = torch.randn(n, generator=g).to(dev), torch.randn(big, n, generator=g).to(dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    gemm = lib.tt_gemm_x3 if planes == torch.bfloat16 else lib.tt_gemm_x3_f16
    for epi in (0, 1, 2):
        outs = []
        for m in rows:
            cp = torch.full((m, 2 * n), float("nan"), dtype=planes, device=dev)
            c32 = torch.full((m, n), float("nan"), device=dev)
            rc = gemm(ap.data_ptr(), wp.data_ptr(), bias.data_ptr(), res.data_ptr() if epi == 2 else None,
                                cp.data_ptr() if epi != 2 else None, c32.data_ptr() if epi == 2 else None, m, n, k, epi, st)
            _lib.check(rc, "tt_gemm_x3")
            outs.append(c32.view(torch.int32) if epi == 2 else cp.view(torch.int16))
        torch.cuda.synchronize()
        for m, c in zip(rows[:-1], outs[:-1]):
            assert torch.equal(c, outs[-1][:m]), (epi, m)
        if epi == 2:          # (and they are the right bits: against fp64 on the planes' own values)
            a64 = (ap[:768, :k].double() + ap[:768, k:].double()).cpu()
            w64 = (wp[:, :k].double() + wp[:, k:].double()).cpu()
            want = a64 @ w64.T + bias.cpu().double() + res[:768].cpu().double()
            got = outs[2].view(torch.float32).cpu().double()
            assert ((got - want).abs().max() / want.abs().max()).item() <= 2e-5


def test_gemm_x3_is_row_permutation_equivariant_bit_for_bit(dev, built_lib):
    from tensor_truth_amd import _lib

    lib = _lib.load_library()
    m, n, k = 2048, 1024, 1024
    g = torch.Generator().manual_seed(9)
    ap, wp = _planes(torch.randn(m, k, generator=g)).to(dev), _planes(torch.randn(n, k, generator=g) * 0.03).to(dev)
    bias = torch.randn(n, generator=g).to(dev)
    perm = torch.randperm(m, generator=g).to(dev)
    outs = []
    for a in (ap, ap[perm].contiguous()):
        c = torch.empty((m, 2 * n), dtype=torch.bfloat16, device=dev)
        _lib.check(lib.tt_gemm_x3(a.data_ptr(), wp.data_ptr(), bias.data_ptr(), None, c.data_ptr(), None, m, n, k, 1, None), "tt_gemm_x3")
        outs.append(c)
    torch.cuda.synchronize()
    assert torch.equal(outs[0][perm].view(torch.int16), outs[1].view(torch.int16))


@pytest.mark.parametrize("lens", [[292] * 7, [1, 8, 64, 65, 127, 128, 129, 300, 17, 33], [512, 31, 257], [1100, 40, 2049]])
def test_attention_x3_against_fp64(dev, built_lib, lens):
    """tt_attention_x3 on ragged sequences (any start row: back-to-back packing shares 8-row token groups between
    neighbours) against an fp64 softmax attention over the operands' (hi + lo) values."""
    from tensor_truth_amd import _lib

    lib = _lib.load_library()
    heads, dh = 4, 64
    H = heads * dh
    g = torch.Generator().manual_seed(sum(lens))
    starts, row = [], 0
    for i, n in enumerate(lens):
        starts.append(row)
        row += n if i % 2 else (n + 7) // 8 * 8          # every other sequence follows its neighbour without a gap
    T = (row + 255) // 256 * 256
    q, k, v = (torch.randn(T, H, generator=g) * s for s in (1.5, 1.5, 1.0))
    qp, kp, vp = _planes(q), _planes(k), _planes(v)
    qk = torch.cat([qp[:, :H], kp[:, :H], qp[:, H:], kp[:, H:]], dim=1).contiguous()           # Q hi | K hi | Q lo | K lo
    def v8(plane):                                                                                # [T][H] -> [T/8][H][8]
        return plane.view(T // 8, 8, H).permute(0, 2, 1).contiguous()
    out = torch.zeros((T, 2 * H), dtype=torch.bfloat16, device=dev)
    ss, sl = torch.tensor(starts, dtype=torch.int32, device=dev), torch.tensor(lens, dtype=torch.int32, device=dev)
    dq, dvh, dvl = qk.to(dev), v8(vp[:, :H]).to(dev), v8(vp[:, H:]).to(dev)
    rc = lib.tt_attention_x3(dq.data_ptr(), 4 * H, 0, H, 2 * H, dvh.data_ptr(), dvl.data_ptr(), 8 * H, out.data_ptr(), 2 * H, H,
                             ss.data_ptr(), sl.data_ptr(), len(lens), heads, max(lens), torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "tt_attention_x3")
    torch.cuda.synchronize()
    got = _join(out.cpu(), H).double()
    q64, k64, v64 = _join(qp, H).double(), _join(kp, H).double(), _join(vp, H).double()
    worst = 0.0
    for s, n in zip(starts, lens):
        for h in range(heads):
            sl_ = slice(h * dh, (h + 1) * dh)
            p = torch.softmax(q64[s:s + n, sl_] @ k64[s:s + n, sl_].T / 8.0, dim=1)
            worst = max(worst, (got[s:s + n, sl_] - p @ v64[s:s + n, sl_]).abs().max().item())
    # scores S ~ N(0, 2.25^2) here: the dropped lo.lo terms of K.Q^T (2^-16 each, 64 incoherent terms of |q||k| ~ 2.25, / 8)
    # leave ~3e-5 in S, i.e. ~3e-5 relative in P and in ctx (|ctx| ~ 1); one bf16 rounding of P alone would be 4e-3
    assert worst <= 1e-4, worst
    used = torch.zeros(T, dtype=torch.bool)
    for s, n in zip(starts, lens):
        used[s:s + n] = True
    assert not out.cpu()[~used].any()                   # rows of no sequence are not written


@pytest.mark.parametrize("planes", [torch.bfloat16, torch.float16], ids=["bf16x3", "f16x3"])
@pytest.mark.parametrize("shape", [XLMR, WIDE, SMALL384], ids=["xlmr256", "wide1024", "small384"])
def test_x3_forward_matches_the_fp32_oracle(dev, built_lib, shape, planes):
    """Both instantiations of the split-plane forward: two bf16 planes per operand ("bf16x3", round 3) and two fp16 planes
    ("f16x3", round 4: x3_path.hip / gemm.hip compiled a second time with the fp16 element helpers; the default implementation
    of the reference precision)."""
    from tensor_truth_amd.encoder import EncoderConfig, pack_tokens
    from tensor_truth_amd.encoder_x3 import EncoderWeightsX3, EncoderX3

    cfg_o, cfg = oe.EncoderConfig(**shape), EncoderConfig(**shape)
    W = oe.synth_weights(cfg_o, seed=29)
    g = torch.Generator().manual_seed(4)
    lens = [n for n in (cfg.max_seq_len, 65, 129, 33, 7, 200, 100, 17)]
    seqs = [[0 if cfg.arch == "xlmr" else 101] + torch.randint(4, cfg.vocab_size, (n - 2,), generator=g).tolist() + [2 if cfg.arch == "xlmr" else 102]
            for n in lens]
    enc = EncoderX3(EncoderWeightsX3(cfg, W, dev, dtype=planes))
    batch = pack_tokens(seqs, cfg)
    hidden, _ = enc.forward_packed(batch)
    emb, emb16 = enc.embed_packed(batch)
    scores, logits = enc.rerank_packed(batch, want_logits=True)
    torch.cuda.synchronize()
    ids, mask = _pad(seqs, cfg.pad_id)
    with torch.no_grad():
        want_h = oe.encoder_forward(ids, mask, W, cfg_o)
        want_s = oe.rerank_scores(ids, mask, W, cfg_o)
    hidden = hidden.cpu()
    worst = max((hidden[int(batch.seq_start[b]):int(batch.seq_start[b]) + len(s)] - want_h[b, : len(s)]).abs().max().item()
                for b, s in enumerate(seqs))
    assert worst <= 5e-4, worst                          # LayerNorm-scale values (|x| ~ 1-5); the bf16 path: ~5e-2
    assert (emb.cpu() - oe.cls_pool_normalize(want_h)).abs().max().item() <= 5e-5
    assert torch.equal(emb16.cpu(), emb.cpu().to(torch.bfloat16))
    rel = ((scores.cpu() - want_s).abs() / want_s.abs()).max().item()
    assert rel <= 1e-3, rel                              # north_star: fp scores within 1e-3 relative
    assert rel <= 2e-4, rel                              # (what this path delivers on a 2-3 layer model)
    assert torch.allclose(scores.cpu(), torch.sigmoid(logits.cpu()), atol=1e-6)
    # a single short sequence (64 token rows: weight-streaming skinny GEMMs over the same virtual K stream) gives the SAME BITS
    # as inside the batch of tiled GEMMs: a text's embedding does not depend on what it was embedded with
    one, _ = enc.embed([seqs[4]])
    assert torch.equal(one.cpu()[0], emb.cpu()[4])
    two, _ = enc.embed([seqs[2], seqs[6]])          # 129 + 100 tokens: 256 rows, skinny as well
    assert torch.equal(two.cpu()[0], emb.cpu()[2]) and torch.equal(two.cpu()[1], emb.cpu()[6])


@pytest.mark.default_precision
def test_precision_selector_reaches_both_surfaces(dev, built_lib, monkeypatch):
    """The UNCHANGED reference calls -- SentenceTransformerRerank(model=, top_n=, device=) carries no dtype
    (model_manager.py:333-337), the embedder none unless the per-model config names one -- get the reference's own fp32
    semantics BY DEFAULT (round 4; precision.DEFAULT_MODE), scores within 1e-3 relative of the fp32 oracle; TT_PRECISION
    (process level), ModelManager.set_precision (config key) and model_kwargs name another mode; TT_REFERENCE_IMPL=fp32
    picks the fp32-MFMA implementation of the reference mode."""
    from tensor_truth_amd import model_manager as mm
    from tensor_truth_amd.encoder import Encoder, EncoderConfig
    from tensor_truth_amd.encoder_f32 import EncoderF32
    from tensor_truth_amd.encoder_x3 import EncoderX3
    from tensor_truth_amd.rerank import HipSentenceTransformerRerank

    cfg, cfg_o = EncoderConfig(**XLMR), oe.EncoderConfig(**XLMR)
    W = oe.synth_weights(cfg_o, seed=31)
    base = {"encoder_config": cfg, "state_dict": W}
    texts = [" ".join(f"w{(7 * i + j) % 50}" for j in range(5 + 3 * i)) for i in range(9)]
    query = "w1 w2 w3 which one"

    def scores_of(rr):
        got = torch.tensor(rr.predict([(query, t) for t in texts]))
        ids, mask = _pad([rr._tokenizer.encode_pair(query, t, rr.max_length)[0] for t in texts], cfg.pad_id)
        want = oe.rerank_scores(ids, mask, W, cfg_o)
        return ((got - want).abs() / want.abs()).max().item()

    monkeypatch.delenv("TT_PRECISION", raising=False)
    monkeypatch.setenv("TT_REFERENCE_IMPL", "bf16x3")      # this file tests the split-bf16 implementation (the default one: test_f16c_gpu.py)
    rr = HipSentenceTransformerRerank(model="test/xenc", top_n=3, device="cuda", model_kwargs=dict(base))
    assert isinstance(rr._encoder, EncoderX3) and rr.precision.startswith("reference") and scores_of(rr) <= 2e-4    # the default
    monkeypatch.setenv("TT_PRECISION", "bf16")
    rr = HipSentenceTransformerRerank(model="test/xenc", top_n=3, device="cuda", model_kwargs=dict(base))
    assert isinstance(rr._encoder, Encoder) and rr.precision.startswith("bf16")          # the process setting names bf16
    rr = HipSentenceTransformerRerank(model="test/xenc", top_n=3, device="cuda", model_kwargs={**base, "torch_dtype": "bfloat16"})
    assert isinstance(rr._encoder, Encoder)                                                # the reference's own config spelling
    monkeypatch.setenv("TT_PRECISION", "reference")
    rr = HipSentenceTransformerRerank(model="test/xenc", top_n=3, device="cuda", model_kwargs=dict(base))
    assert isinstance(rr._encoder, EncoderX3) and scores_of(rr) <= 2e-4
    monkeypatch.setenv("TT_REFERENCE_IMPL", "fp32")
    rr = HipSentenceTransformerRerank(model="test/xenc", top_n=3, device="cuda", model_kwargs=dict(base))
    assert isinstance(rr._encoder, EncoderF32) and scores_of(rr) <= 1e-4
    monkeypatch.setenv("TT_REFERENCE_IMPL", "bf16x3")
    rr = HipSentenceTransformerRerank(model="test/xenc", top_n=3, device="cuda", model_kwargs={**base, "precision": "bf16"})
    assert isinstance(rr._encoder, Encoder)                                                # an explicit kwarg beats the environment
    monkeypatch.delenv("TT_PRECISION")
    # the ModelManager config key, through the reference's own lifecycle calls
    mm.ModelManager.reset_instance()
    mgr = mm.ModelManager.get_instance()
    mgr.model_kwargs_overrides["test/xenc"] = dict(base)
    mgr.model_kwargs_overrides["test/emb"] = {"encoder_config": EncoderConfig(**{**XLMR, "num_labels": 0}), "state_dict": W}
    rr = mgr.get_reranker("test/xenc", top_n=3, device="cuda")                            # nothing named anywhere: the default
    assert isinstance(rr._encoder, EncoderX3) and scores_of(rr) <= 2e-4
    mgr.set_precision("bf16")
    assert isinstance(mgr.get_reranker("test/xenc", top_n=3, device="cuda")._encoder, Encoder)
    assert isinstance(mgr.get_embedder("test/emb", "cuda")._encoder, Encoder)
    mgr.set_precision(None)
    emb = mgr.get_embedder("test/emb", "cuda")
    assert isinstance(emb._encoder, EncoderX3) and emb.precision.startswith("reference")
    e = torch.tensor(emb.get_text_embedding_batch(texts))
    ids, mask = _pad([emb._tokenizer.encode(t, emb.max_length) for t in texts], cfg.pad_id)
    assert (e - oe.embed(ids, mask, W, cfg_o)).abs().max().item() <= 5e-5
    mm.ModelManager.reset_instance()


@pytest.mark.parametrize("shape", [dict(arch="xlmr", vocab_size=2000, hidden=256, layers=3, heads=4, ffn=1024, max_pos=320, type_vocab=1,
                                        pad_id=1, ln_eps=1e-5, num_labels=1),
                                   dict(arch="xlmr", vocab_size=2000, hidden=384, layers=2, heads=12, ffn=1536, max_pos=320, type_vocab=1,
                                        pad_id=1, ln_eps=1e-5, num_labels=1)], ids=["256x4heads64", "384x12heads32"])
def test_cls_only_last_layer_equals_the_full_forward_rows(dev, built_lib, shape):
    """``tt_encoder_forward_x3_cls`` (the last layer for every sequence's first row only) against the first rows of the full
    forward: the GEMM rows are bit-identical whichever kernel computes them, the one-query attention sums its keys in another
    order than the tiled kernel -- fp32 rounding noise apart, the same numbers.  Few sequences (skinny tail GEMMs) and many
    (tiled tail GEMMs); embeddings and rerank scores go through this path."""
    import numpy as np

    from oracle import encoder as oe
    from tensor_truth_amd.encoder import EncoderConfig, pack_tokens
    from tensor_truth_amd.encoder_x3 import EncoderWeightsX3, EncoderX3

    cfg, ocfg = EncoderConfig(**shape), oe.EncoderConfig(**shape)
    W = oe.synth_weights(ocfg, seed=23)
    enc = EncoderX3(EncoderWeightsX3(cfg, W, dev))
    rng = np.random.default_rng(6)
    for n_seq in (5, 70, 300):
        seqs = [[0] + rng.integers(5, 2000, size=int(rng.integers(3, 120))).tolist() + [2] for _ in range(n_seq)]
        batch = pack_tokens(seqs, cfg, None, None)
        full, starts = enc.forward_packed(batch)
        cls, rows = enc.cls_hidden_packed(batch)
        want = full[starts[:n_seq].long()].cpu()
        got = cls[:n_seq].cpu()
        assert torch.isfinite(got).all()
        assert (got - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item()), n_seq
        assert rows.tolist() == list(range(n_seq))
    # and against the fp32 oracle through the public calls (which use the CLS path)
    seqs = [[0] + rng.integers(5, 2000, size=int(rng.integers(3, 200))).tolist() + [2] for _ in range(40)]
    L = max(map(len, seqs))
    ids = torch.full((40, L), 1, dtype=torch.long); mask = torch.zeros((40, L), dtype=torch.long)
    for i, sq in enumerate(seqs):
        ids[i, :len(sq)] = torch.tensor(sq); mask[i, :len(sq)] = 1
    with torch.no_grad():
        ref = oe.rerank_scores(ids, mask, W, ocfg)
    got = enc.rerank(seqs).cpu()
    assert ((got - ref).abs() / ref.abs()).max().item() < 1e-4
