"""GPU: reference precision on two matrix-time units -- the "f16c" path (csrc/f16c_path.hip, gemm.hip GemmParams.xc,
attention.hip's c-planes epilogue; host side ``tensor_truth_amd/encoder_f16c.py``).

What the reference's unchanged calls compute is fp32 (``services/model_manager.py:333-337`` passes no dtype,
``app_utils/config_schema.py:66-76``); the bar is north_star's 1e-3 relative on scores against the plain fp32 CPU oracle.
Building blocks are pinned against ``oracle/f16c.py`` (the operand format restated on the CPU): the quantiser bit for bit,
the GEMM against a fp64 product of the same planes and against the plain fp32 product, the attention epilogue against the fp16
kernel's own output; then the whole forward, the CLS-only tail and the plugin surface against ``oracle/encoder.py``.
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import encoder as oe
from oracle import f16c as of

pytestmark = pytest.mark.gpu

XLMR = dict(arch="xlmr", vocab_size=1000, hidden=256, layers=2, heads=4, ffn=1024, max_pos=300, type_vocab=1, pad_id=1,
            ln_eps=1e-5, num_labels=1)


def _lib_and_stream(dev):
    from tensor_truth_amd import _lib

    return _lib.load_library(), torch.cuda.current_stream(dev).cuda_stream


def _quantize(x, weight, dev):
    from tensor_truth_amd.encoder_f16c import quantize_planes

    p, s = quantize_planes(x.to(dev), weight)
    torch.cuda.synchronize()
    return p, s


@pytest.mark.parametrize("weight", [False, True])
@pytest.mark.parametrize("rows,k", [(256, 256), (300, 1024), (512, 4096)])
def test_quantizer_matches_the_cpu_restatement_bit_for_bit(dev, built_lib, rows, k, weight):
    """tt_f16c_quantize: the three planes and the tiled E8M0 scales, byte for byte -- values spanning 40 binades inside a row,
    zero blocks, fp16-subnormal and fp16-overflowing magnitudes."""
    g = torch.Generator().manual_seed(rows + k + int(weight))
    x = torch.randn(rows, k, generator=g) * torch.exp2(torch.randint(-20, 8, (rows, 1), generator=g).float())
    x[3, 64:128] = 0.0                                        # zero blocks
    x[5, :32] *= 2.0 ** -20                                   # far below the row's other blocks
    x[7, 0] = 1e-7                                            # fp16 subnormal
    x[9, 5] = 7e4 if not weight else 3.0                      # beyond fp16's range (activations: hi saturates, lo8 saturates)
    q = of.quantize(x, weight)
    want_p, want_s = of.planes_bytes(q, weight), of.tiled_scales(q, weight)
    got_p, got_s = _quantize(x, weight, dev)
    got_p, got_s = got_p.cpu().numpy(), got_s.cpu().numpy()
    assert got_s.shape == want_s.shape and np.array_equal(got_s, want_s)
    same = got_p == want_p
    if not same.all():
        r, c = np.argwhere(~same)[0]
        raise AssertionError(f"plane byte differs at row {r}, byte {c} ({'hi' if c < 2 * k else 'e4m3'} plane): {got_p[r, c]} vs {want_p[r, c]}; "
                             f"{int((~same).sum())} bytes differ")


@pytest.mark.parametrize("m,n,k", [(256, 256, 256), (512, 1024, 1024), (256, 1024, 4096), (768, 768, 768)])
def test_gemm_f16c_vs_its_own_planes_and_vs_fp32(dev, built_lib, m, n, k):
    """tt_gemm_f16c (bias epilogue -> fp16, residual epilogue -> fp32): against the fp64 product of the SAME planes (what the
    kernel must compute: its only freedom is fp32 accumulation order) and against the plain fp32 product (what it stands for:
    ~2^-16 of the row's scale).  Per-block scales that differ wildly inside a row exercise the scale strips."""
    lib, st = _lib_and_stream(dev)
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g) * torch.exp2(torch.randint(-6, 6, (m, k // 32), generator=g).float()).repeat_interleave(32, 1)
    w = torch.randn(n, k, generator=g) * 0.05 * torch.exp2(torch.randint(-4, 4, (n, k // 32), generator=g).float()).repeat_interleave(32, 1)
    bias = torch.randn(n, generator=g)
    res = torch.randn(m, n, generator=g)
    ap, asc = _quantize(a, False, dev)
    wp, wsc = _quantize(w, True, dev)
    bias_d, res_d = bias.to(dev), res.to(dev)
    out32 = torch.empty((m, n), dtype=torch.float32, device=dev)
    rc = lib.tt_gemm_f16c(ap.data_ptr(), asc.data_ptr(), wp.data_ptr(), wsc.data_ptr(), bias_d.data_ptr(), res_d.data_ptr(),
                          out32.data_ptr(), None, m, n, k, 2, st)
    assert rc == 0, lib.tt_last_error()
    torch.cuda.synchronize()
    planes = of.matmul(a, w) + bias + res
    exact = (a.double() @ w.double().T).float() + bias + res
    # the magnitude the rounding errors live on: the products' absolute sum (+ the bias / residual the fp32 result is rounded with)
    scale = (a.abs().double() @ w.abs().double().T).float() + bias.abs() + res.abs()
    got = out32.cpu()
    err_planes = ((got - planes).abs() / scale).max().item()
    err_exact = ((got - exact).abs() / scale).max().item()
    assert err_planes < 2e-6, f"vs the fp64 product of the same planes: {err_planes:.2e} of sum |a||w| (fp32 accumulation: ~K 2^-24)"
    assert err_exact < 2.0 ** -15, f"vs the fp32 product: {err_exact:.2e} of sum |a||w| (fp16 alone: 2^-11)"
    # fp16 output (the Q / K projection's epilogue)
    out16 = torch.empty((m, n), dtype=torch.float16, device=dev)
    rc = lib.tt_gemm_f16c(ap.data_ptr(), asc.data_ptr(), wp.data_ptr(), wsc.data_ptr(), bias_d.data_ptr(), None,
                          out16.data_ptr(), None, m, n, k, 0, st)
    assert rc == 0, lib.tt_last_error()
    torch.cuda.synchronize()
    want16 = (of.matmul(a, w) + bias)
    assert ((out16.cpu().float() - want16).abs() <= want16.abs() * 2.0 ** -10 + scale * 2e-6).all()


def test_gemm_f16c_gelu_epilogue_writes_the_next_operand(dev, built_lib):
    """The GELU epilogue's c-planes output IS a valid A operand: its planes and scales equal the quantiser's on the fp32 values
    (to the epilogue's fp32 rounding of the GELU), and feeding it to the next GEMM gives the fp32 two-layer result."""
    lib, st = _lib_and_stream(dev)
    m, k, f = 512, 256, 1024
    g = torch.Generator().manual_seed(5)
    a, w1, b1 = torch.randn(m, k, generator=g), torch.randn(f, k, generator=g) * 0.06, torch.randn(f, generator=g) * 0.1
    w2, b2, res = torch.randn(k, f, generator=g) * 0.03, torch.randn(k, generator=g) * 0.1, torch.randn(m, k, generator=g)
    ap, asc = _quantize(a, False, dev)
    w1p, w1s = _quantize(w1, True, dev)
    w2p, w2s = _quantize(w2, True, dev)
    hp = torch.zeros((m, 4 * f), dtype=torch.uint8, device=dev)
    hs = torch.zeros(int(lib.tt_f16c_scale_bytes(m, f, 0)), dtype=torch.uint8, device=dev)
    b1_d, b2_d, res_d = b1.to(dev), b2.to(dev), res.to(dev)
    rc = lib.tt_gemm_f16c(ap.data_ptr(), asc.data_ptr(), w1p.data_ptr(), w1s.data_ptr(), b1_d.data_ptr(), None, hp.data_ptr(),
                          hs.data_ptr(), m, f, k, 1, st)
    assert rc == 0, lib.tt_last_error()
    out = torch.empty((m, k), dtype=torch.float32, device=dev)
    rc = lib.tt_gemm_f16c(hp.data_ptr(), hs.data_ptr(), w2p.data_ptr(), w2s.data_ptr(), b2_d.data_ptr(), res_d.data_ptr(),
                          out.data_ptr(), None, m, k, f, 2, st)
    assert rc == 0, lib.tt_last_error()
    torch.cuda.synchronize()
    h = oe.gelu_erf((a.double() @ w1.double().T).float() + b1)
    # the hi plane is fp16(h) to the GEMM's own 2^-16: one fp16 ulp at most
    hi = hp.cpu()[:, : 2 * f].contiguous().view(torch.float16).float()
    s1 = (a.abs().double() @ w1.abs().double().T).float() + b1.abs()        # (the pre-activation is right to 2^-15 of this)
    assert ((hi - h).abs() <= h.abs() * 2.0 ** -10 + s1 * 2.0 ** -14 + 1e-6).all()
    # scale bytes: the exponent of each block's absmax - 7 (a block whose absmax sits within 2^-15 of a power of two may differ by one)
    want_s = of.tiled_scales(of.quantize(h, False), False)
    diff = hs.cpu().numpy().astype(np.int32) - want_s.astype(np.int32)
    assert np.abs(diff).max() <= 1 and (diff != 0).mean() < 1e-2
    want = (h.double() @ w2.double().T).float() + b2 + res
    scale = (h.abs().double() @ w2.abs().double().T).float() + 1.0
    assert ((out.cpu() - want).abs() / scale).max().item() < 2.0 ** -14


def test_gemm_f16c_rows_do_not_depend_on_their_tile(dev, built_lib):
    """Row-permutation equivariance, bit for bit: a row's result depends on the row alone, not on where it sits in the tile grid
    (the scale strips are addressed per 256-row block: a wrong strip index shows up here)."""
    lib, st = _lib_and_stream(dev)
    m, n, k = 1024, 512, 1024
    g = torch.Generator().manual_seed(9)
    a, w, bias = torch.randn(m, k, generator=g), torch.randn(n, k, generator=g) * 0.04, torch.randn(n, generator=g)
    perm = torch.randperm(m, generator=g)
    wp, wsc = _quantize(w, True, dev)
    bias_d = bias.to(dev)
    outs = []
    for x in (a, a[perm]):
        ap, asc = _quantize(x, False, dev)
        hp = torch.zeros((m, 4 * n), dtype=torch.uint8, device=dev)
        hs = torch.zeros(int(lib.tt_f16c_scale_bytes(m, n, 0)), dtype=torch.uint8, device=dev)
        rc = lib.tt_gemm_f16c(ap.data_ptr(), asc.data_ptr(), wp.data_ptr(), wsc.data_ptr(), bias_d.data_ptr(), None, hp.data_ptr(),
                              hs.data_ptr(), m, n, k, 1, st)
        assert rc == 0, lib.tt_last_error()
        torch.cuda.synchronize()
        outs.append(hp.cpu())
    assert torch.equal(outs[0][perm], outs[1])


def _forward_case(cfg_kw, lens, seed, dev):
    from tensor_truth_amd.encoder import EncoderConfig
    from tensor_truth_amd.encoder_f16c import EncoderF16C, EncoderWeightsF16C

    cfg, ocfg = EncoderConfig(**cfg_kw), oe.EncoderConfig(**cfg_kw)
    W = oe.synth_weights(ocfg, seed=seed)
    ids, mask = oe.synth_tokens(len(lens), max(lens), ocfg, seed=seed + 1, lengths=lens)
    seqs = [ids[b, : int(mask[b].sum())].tolist() for b in range(len(lens))]
    enc = EncoderF16C(EncoderWeightsF16C(cfg, W, dev))
    return cfg, ocfg, W, ids, mask, seqs, enc


@pytest.mark.parametrize("shape", ["xlmr256", "xlmr1024", "xlmr768"])
def test_forward_f16c_vs_fp32_oracle(dev, built_lib, shape):
    """Embeddings and rerank scores of the whole forward against the PLAIN fp32 oracle (no emulation of anything): north_star's
    1e-3 relative on scores with an order of magnitude to spare; ragged lengths incl. 1 token, a sequence beyond one query tile,
    sequences crossing 256-row blocks."""
    kw = {"xlmr256": XLMR,
          "xlmr1024": {**XLMR, "hidden": 1024, "heads": 16, "ffn": 4096, "layers": 3},
          "xlmr768": {**XLMR, "hidden": 768, "heads": 12, "ffn": 3072}}[shape]
    lens = [24, 1, 130, 64, 65, 7, 200, 33, 129, 257 if shape == "xlmr256" else 40]
    cfg, ocfg, W, ids, mask, seqs, enc = _forward_case({**kw, "max_pos": 300}, lens, 17, dev)
    emb, _ = enc.embed(seqs)
    scores = enc.rerank(seqs)
    hidden, starts = enc.forward_packed(__import__("tensor_truth_amd.encoder", fromlist=["pack_tokens"]).pack_tokens(seqs, cfg))
    torch.cuda.synchronize()
    want_h = oe.encoder_forward(ids, mask, W, ocfg)
    want_e = oe.embed(ids, mask, W, ocfg)
    want_s = oe.rerank_scores(ids, mask, W, ocfg)
    st = starts.cpu().tolist()
    worst_h = max((hidden[st[b]: st[b] + lens[b]].cpu() - want_h[b, : lens[b]]).abs().max().item() for b in range(len(lens)))
    rel_s = ((scores.cpu() - want_s).abs() / want_s.abs()).max().item()
    err_e = (emb.cpu() - want_e).abs().max().item()
    print(f"f16c {shape}: hidden |err| {worst_h:.1e}, embedding |err| {err_e:.1e}, score rel err {rel_s:.1e}")
    # (the 1024-wide case's random head has a gain of ~6 on the logit: 4e-4 relative on scores from 1e-4 on the hidden state)
    assert worst_h < 1e-3 and err_e < 5e-5 and rel_s < 1e-3
    assert (emb.cpu() * want_e).sum(1).min().item() > 0.999999


def test_cls_only_last_layer_equals_the_full_forward_rows(dev, built_lib):
    """``tt_encoder_forward_f16c_cls`` (the last layer for every sequence's first row only) against the first rows of the full
    forward: same GEMM kernel, same rows -> the projections agree bit for bit; the one-query attention sums its keys in another
    order than the tiled kernel (fp32 rounding noise).  Few sequences and more than one 256-row block of them."""
    from tensor_truth_amd.encoder import pack_tokens

    for n_seq in (5, 300):
        lens = [int(x) for x in np.random.default_rng(n_seq).integers(1, 90, size=n_seq)]
        cfg, ocfg, W, ids, mask, seqs, enc = _forward_case({**XLMR, "layers": 3}, lens, 23, dev)
        batch = pack_tokens(seqs, cfg)
        hidden, starts = enc.forward_packed(batch)
        cls, _ = enc.cls_hidden_packed(batch)
        torch.cuda.synchronize()
        full = hidden[starts.long()].cpu()
        assert (cls[:n_seq].cpu() - full).abs().max().item() < 2e-4      # (the tiled kernel rounds P to fp16, the one-query kernel does not)


def test_embedding_does_not_depend_on_the_batch(dev, built_lib):
    """One text embedded alone and inside a large batch: the same bits (every GEMM runs the tiled kernel on 256-row tiles, a row's
    result is a function of the row; the attention of a sequence does not see its neighbours)."""
    lens = [34, 120, 7, 250, 90, 64, 33, 18]
    cfg, ocfg, W, ids, mask, seqs, enc = _forward_case({**XLMR, "hidden": 1024, "heads": 16, "ffn": 4096}, lens, 29, dev)
    emb, _ = enc.embed(seqs)
    one, _ = enc.embed([seqs[4]])
    two, _ = enc.embed([seqs[6], seqs[1]])
    torch.cuda.synchronize()
    assert torch.equal(one.cpu()[0], emb.cpu()[4])
    assert torch.equal(two.cpu()[0], emb.cpu()[6]) and torch.equal(two.cpu()[1], emb.cpu()[1])


@pytest.mark.default_precision
def test_the_unchanged_reference_calls_and_their_implementations(dev, built_lib, monkeypatch):
    """SentenceTransformerRerank(model=, top_n=, device=) and the embedder, with no dtype anywhere (services/model_manager.py:
    333-337, 218-229): the default mode is the reference's own precision, implemented on split-fp16 planes ("f16x3": the one that
    holds 1e-3 on the stress weights too); TT_REFERENCE_IMPL=f16c selects this file's faster path, bf16x3 / fp32 the older ones --
    all within 1e-3 relative of the fp32 oracle here."""
    from tensor_truth_amd.encoder import EncoderConfig
    from tensor_truth_amd.encoder_f16c import EncoderF16C
    from tensor_truth_amd.encoder_f32 import EncoderF32
    from tensor_truth_amd.encoder_x3 import EncoderX3
    from tensor_truth_amd.embedding import HipHuggingFaceEmbedding
    from tensor_truth_amd.rerank import HipSentenceTransformerRerank

    monkeypatch.delenv("TT_REFERENCE_IMPL", raising=False)
    cfg, cfg_o = EncoderConfig(**XLMR), oe.EncoderConfig(**XLMR)
    W = oe.synth_weights(cfg_o, seed=31)
    base = {"encoder_config": cfg, "state_dict": W}
    texts = [" ".join(f"w{(7 * i + j) % 50}" for j in range(5 + 3 * i)) for i in range(9)]
    query = "w1 w2 w3 which one"
    rr = HipSentenceTransformerRerank(model="test/xenc", top_n=3, device="cuda", model_kwargs=dict(base))
    assert isinstance(rr._encoder, EncoderX3) and rr._encoder.w.dtype == torch.float16 and rr.precision.startswith("reference")
    got_default = torch.tensor(rr.predict([(query, t) for t in texts]))
    monkeypatch.setenv("TT_REFERENCE_IMPL", "f16c")
    rr = HipSentenceTransformerRerank(model="test/xenc", top_n=3, device="cuda", model_kwargs=dict(base))
    assert isinstance(rr._encoder, EncoderF16C) and rr.precision.startswith("reference")
    got = torch.tensor(rr.predict([(query, t) for t in texts]))
    toks = [rr._tokenizer.encode_pair(query, t, rr.max_length)[0] for t in texts]
    L = max(len(t) for t in toks)
    ids = torch.full((len(toks), L), cfg.pad_id, dtype=torch.int64)
    mask = torch.zeros((len(toks), L), dtype=torch.int64)
    for i, t in enumerate(toks):
        ids[i, : len(t)] = torch.tensor(t)
        mask[i, : len(t)] = 1
    want = oe.rerank_scores(ids, mask, W, cfg_o)
    assert ((got - want).abs() / want.abs()).max().item() < 1e-4
    assert ((got_default - want).abs() / want.abs()).max().item() < 1e-4
    emb = HipHuggingFaceEmbedding("test/emb", device="cuda",
                                  model_kwargs={"encoder_config": EncoderConfig(**{**XLMR, "num_labels": 0}), "state_dict": W})
    assert isinstance(emb._encoder, EncoderF16C)
    monkeypatch.setenv("TT_REFERENCE_IMPL", "bf16x3")
    rr3 = HipSentenceTransformerRerank(model="test/xenc", top_n=3, device="cuda", model_kwargs=dict(base))
    assert isinstance(rr3._encoder, EncoderX3) and rr3._encoder.w.dtype == torch.bfloat16
    monkeypatch.setenv("TT_REFERENCE_IMPL", "fp32")
    assert isinstance(HipSentenceTransformerRerank(model="test/xenc", top_n=3, device="cuda", model_kwargs=dict(base))._encoder, EncoderF32)
