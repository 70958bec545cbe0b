"""GPU parity of the fp8 (OCP e4m3) mode -- BASELINE.json config 5, "fp8 MFMA reranker".

The Q/K/V and FFN-up projections run on e4m3 operands (weights quantised per output channel, activations per token
inside the LayerNorm kernels), fp32 accumulation, everything else as in the bf16 mode.  Checked (a) against the
oracle's emulation of exactly that arithmetic (tight: the two differ only in fp32 summation order and in bf16
roundings that flip by one ulp), and (b) against the plain fp32 oracle with the looser, stated fp8 bound.
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import encoder as oe

pytestmark = pytest.mark.gpu


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _same_up_to_ties(got_q, want_q, x):
    """Identical e4m3 values, except where x * 448 / absmax lands (to within a few fp32 ulps) ON the midpoint of two
    e4m3 neighbours: v_cvt_pk_fp8_f32 resolves those as ties-to-even, torch's conversion rounds the fp32 value as it
    is.  bf16 inputs hit such midpoints whenever the row maximum is a round number (7e-4 of this data)."""
    amax = x.abs().amax(dim=-1, keepdim=True)
    scaled = x * torch.where(amax > 0, 448.0 / amax, torch.zeros_like(amax))
    bad = got_q != want_q
    assert bad.float().mean().item() < 2e-3
    mid = 0.5 * (got_q + want_q)
    assert ((scaled - mid).abs()[bad] <= 4e-6 * mid.abs()[bad]).all()


def test_quantize_rows_matches_torch_e4m3(dev, built_lib):
    from tensor_truth_amd import _lib

    lib = _lib.load_library()
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(300, 1024, generator=g) * torch.rand(300, 1, generator=g) * 3).to(torch.bfloat16)
    x[7] = 0                                   # all-zero row: scale 1, zeros
    x[9, 5] = 1000.0                           # outlier row
    want_q, want_s = oe.quantize_rows_e4m3(x.float())
    xd = x.to(dev)
    q = torch.empty(300, 1024, dtype=torch.uint8, device=dev)
    s = torch.empty(300, dtype=torch.float32, device=dev)
    _lib.check(lib.tt_quantize_rows_fp8(xd.data_ptr(), 300, 1024, q.data_ptr(), s.data_ptr(), _stream()), "quantize")
    got_q = q.cpu().view(torch.float8_e4m3fn).float()
    assert torch.allclose(s.cpu(), want_s.reshape(-1), rtol=1e-6)
    _same_up_to_ties(got_q, want_q, x.float())


def test_layernorm_emits_its_own_quantisation(dev, built_lib):
    from tensor_truth_amd import _lib

    lib = _lib.load_library()
    g = torch.Generator().manual_seed(6)
    x = torch.randn(256, 1024, generator=g).to(torch.bfloat16).to(dev)
    gamma = (1 + 0.1 * torch.randn(1024, generator=g)).to(dev)
    beta = (0.1 * torch.randn(1024, generator=g)).to(dev)
    out = torch.empty_like(x)
    out2 = torch.empty_like(x)
    q = torch.empty(256, 1024, dtype=torch.uint8, device=dev)
    s = torch.empty(256, dtype=torch.float32, device=dev)
    _lib.check(lib.tt_layernorm_bf16_fp8(x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 256, 1024, 1e-5,
                                         q.data_ptr(), s.data_ptr(), _stream()), "ln fp8")
    _lib.check(lib.tt_layernorm_bf16(x.data_ptr(), out2.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 256, 1024, 1e-5,
                                     _stream()), "ln")
    assert torch.equal(out.view(torch.int16), out2.view(torch.int16))          # the bf16 output is unchanged
    want_q, want_s = oe.quantize_rows_e4m3(out.float().cpu())
    assert torch.allclose(s.cpu(), want_s.reshape(-1), rtol=1e-6)
    _same_up_to_ties(q.cpu().view(torch.float8_e4m3fn).float(), want_q, out.float().cpu())


@pytest.mark.parametrize("m,n,k,epi", [(256, 256, 256, 0), (512, 768, 1024, 0), (768, 1024, 512, 1), (2048, 3072, 1024, 0)])
def test_gemm_fp8_exact_products(dev, built_lib, m, n, k, epi):
    """e4m3 x e4m3 products accumulate exactly in fp32 for these sizes: only the epilogue rounds."""
    from tensor_truth_amd import _lib

    lib = _lib.load_library()
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(m, k, generator=g).to(torch.bfloat16)
    w = (0.05 * torch.randn(n, k, generator=g)).to(torch.bfloat16)
    bias = torch.randn(n, generator=g)
    aq, sa = oe.quantize_rows_e4m3(a.float())
    wq, sw = oe.quantize_rows_e4m3(w.float())
    want = (aq.double() @ wq.double().T).float() * sa * sw.T + bias
    if epi == 1:
        want = oe.gelu_erf(want)
    a8 = aq.to(torch.float8_e4m3fn).view(torch.uint8).to(dev)
    w8 = wq.to(torch.float8_e4m3fn).view(torch.uint8).to(dev)
    sad, swd, bd = sa.reshape(-1).contiguous().to(dev), sw.reshape(-1).contiguous().to(dev), bias.to(dev)
    c = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    _lib.check(lib.tt_gemm_fp8(a8.data_ptr(), sad.data_ptr(), w8.data_ptr(), swd.data_ptr(), bd.data_ptr(), c.data_ptr(),
                               m, n, k, epi, _stream()), "gemm fp8")
    got = c.float().cpu()
    err = (got - want).abs()
    # bf16 output rounding (2^-8 relative) + fp32 accumulation order
    assert (err <= 2.0 ** -7 * want.abs() + 1e-3).all(), (err.max().item(), want.abs().max().item())


def test_gemm_fp8_residual_and_fp8_output(dev, built_lib):
    from tensor_truth_amd import _lib

    lib = _lib.load_library()
    m, n, k = 512, 1024, 4096
    g = torch.Generator().manual_seed(3)
    a = torch.randn(m, k, generator=g).to(torch.bfloat16)
    w = (0.05 * torch.randn(n, k, generator=g)).to(torch.bfloat16)
    bias = torch.randn(n, generator=g)
    res = torch.randn(m, n, generator=g).to(torch.bfloat16)
    aq, sa = oe.quantize_rows_e4m3(a.float())
    wq, sw = oe.quantize_rows_e4m3(w.float())
    a8 = aq.to(torch.float8_e4m3fn).view(torch.uint8).to(dev)
    w8 = wq.to(torch.float8_e4m3fn).view(torch.uint8).to(dev)
    sad, swd, bd, rd = sa.reshape(-1).contiguous().to(dev), sw.reshape(-1).contiguous().to(dev), bias.to(dev), res.to(dev)
    base = (aq.double() @ wq.double().T).float() * sa * sw.T + bias
    # residual epilogue
    c = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    _lib.check(lib.tt_gemm_fp8_ex(a8.data_ptr(), sad.data_ptr(), w8.data_ptr(), swd.data_ptr(), bd.data_ptr(), rd.data_ptr(),
                                  c.data_ptr(), None, 0.0, m, n, k, 2, _stream()), "gemm fp8 res")
    want = base + res.float()
    assert ((c.float().cpu() - want).abs() <= 2.0 ** -7 * want.abs() + 1e-3).all()
    # GELU epilogue with an e4m3 result under a static scale: = e4m3(bf16(gelu) / s), saturating
    sf = 0.02
    c8 = torch.zeros(m, n, dtype=torch.uint8, device=dev)
    _lib.check(lib.tt_gemm_fp8_ex(a8.data_ptr(), sad.data_ptr(), w8.data_ptr(), swd.data_ptr(), bd.data_ptr(), None, None,
                                  c8.data_ptr(), 1.0 / sf, m, n, k, 1, _stream()), "gemm fp8 out8")
    got = c8.cpu().view(torch.float8_e4m3fn).float()
    want8 = (oe.gelu_erf(base).to(torch.bfloat16).float() / sf).clamp(-448, 448).to(torch.float8_e4m3fn).float()
    assert torch.isfinite(got).all()
    # one e4m3 step (2^-3 relative) where the bf16 rounding of the GELU value or an exact tie went the other way, plus
    # the fp32 accumulation error of K = 4096 products (2e-4 on a pre-activation that cancels to ~0) in scaled units
    close = (got - want8).abs() <= 0.125 * want8.abs() + 2e-4 / sf
    assert close.all()
    assert (got != want8).float().mean().item() < 0.02


XENC = dict(arch="xlmr", vocab_size=2000, hidden=256, layers=2, heads=4, ffn=1024, max_pos=300, type_vocab=1,
            pad_id=1, ln_eps=1e-5, num_labels=1)


def _pad(seqs, pad):
    L = max(len(s) for s in seqs)
    ids = torch.full((len(seqs), L), pad, dtype=torch.int64)
    mask = torch.zeros(len(seqs), L, dtype=torch.int64)
    for b, s in enumerate(seqs):
        ids[b, : len(s)] = torch.tensor(s)
        mask[b, : len(s)] = 1
    return ids, mask


def test_fp8_forward_matches_its_emulation_and_stays_near_fp32(dev, built_lib):
    from tensor_truth_amd.encoder import Encoder, EncoderConfig, EncoderWeights

    cfg, ocfg = EncoderConfig(**XENC), oe.EncoderConfig(**XENC)
    W = {k: v.to(torch.bfloat16) for k, v in oe.synth_weights(ocfg, seed=17).items()}
    g = torch.Generator().manual_seed(2)
    seqs = []
    for i in range(24):
        n = int(torch.randint(12, 200, (1,), generator=g))
        seqs.append([0] + torch.randint(4, cfg.vocab_size, (n,), generator=g).tolist() + [2])
    ids, mask = _pad(seqs, cfg.pad_id)

    weights = EncoderWeights(cfg, W, dev)
    enc = Encoder(weights)
    s_bf16 = enc.rerank(seqs).cpu()
    weights.set_gemm_dtype("fp8")
    s_fp8, l_fp8 = enc.rerank(seqs, want_logits=True)
    s_fp8, l_fp8 = s_fp8.cpu(), l_fp8.cpu()
    weights.set_gemm_dtype("bf16")
    assert torch.equal(enc.rerank(seqs).cpu(), s_bf16)                         # switching back restores the bf16 path
    assert not torch.equal(s_fp8, s_bf16)                                      # ... and fp8 really ran

    emu = oe.rerank_scores(ids, mask, W, ocfg, emulate_bf16=True, emulate_fp8=True)
    ref = oe.rerank_scores(ids, mask, W, ocfg)                                 # plain fp32
    # (a) vs the emulation of the same arithmetic: same bound as the bf16 mode has against its emulation
    assert (s_fp8 - emu).abs().max().item() < 1.5e-2
    # (b) vs fp32: the stated fp8 bound (bf16 mode: 2e-2 on 24 layers; e4m3 has 3 mantissa bits)
    assert (s_fp8 - ref).abs().max().item() < 5e-2
    assert torch.isfinite(l_fp8).all()

    # ---- all four projections: calibrate the FFN intermediate's static scales first
    from tensor_truth_amd.encoder import pack_tokens

    scales = enc.calibrate_fp8(pack_tokens(seqs, cfg, None, 512))
    assert len(scales) == cfg.layers and all(s > 0 for s in scales)
    weights.set_gemm_dtype("fp8")
    s_all = enc.rerank(seqs).cpu()
    weights.set_gemm_dtype("bf16")
    assert not torch.equal(s_all, s_fp8)                                       # the FFN output projection changed path
    emu_all = oe.rerank_scores(ids, mask, W, ocfg, emulate_bf16=True, emulate_fp8=True, ffn_act_scales=scales)
    assert (s_all - emu_all).abs().max().item() < 1.5e-2
    assert (s_all - ref).abs().max().item() < 5e-2


def test_fp8_needs_tileable_shapes(dev, built_lib):
    from tensor_truth_amd.encoder import EncoderConfig, EncoderWeights

    small = dict(XENC, hidden=384, heads=12, ffn=1536)
    cfg, ocfg = EncoderConfig(**small), oe.EncoderConfig(**small)
    W = {k: v.to(torch.bfloat16) for k, v in oe.synth_weights(ocfg, seed=1).items()}
    weights = EncoderWeights(cfg, W, dev)
    with pytest.raises(ValueError, match="multiples of 256"):
        weights.set_gemm_dtype("fp8")


@pytest.mark.parametrize("n,k,epi", [(2048, 1024, 0), (4096, 1024, 1), (1024, 4096, 2)])
def test_bench_sized_fp8_gemms_sampled_rows_and_row_equivariance(dev, built_lib, n, k, epi):
    """The e4m3 GEMM kernels at the bench's M = 473 600 (BASELINE config 5's "fp8 MFMA reranker"): rows quantised on the
    device (tt_quantize_rows_fp8), a random sample of rows against exact products of the same e4m3 values on the CPU, and
    bit-exact row-permutation equivariance over the whole output."""
    from tensor_truth_amd import _lib

    lib = _lib.load_library()
    m = 473_600
    g = torch.Generator(device=dev).manual_seed(n + k + epi)
    a = torch.randn(m, k, device=dev, generator=g).to(torch.bfloat16)
    w = (0.05 * torch.randn(n, k, generator=torch.Generator().manual_seed(n + k))).to(torch.bfloat16)
    bias = torch.randn(n, device=dev, generator=g)
    res = torch.randn(m, n, device=dev, generator=g).to(torch.bfloat16) if epi == 2 else None
    wq, sw = oe.quantize_rows_e4m3(w.float())
    w8 = wq.to(torch.float8_e4m3fn).view(torch.uint8).to(dev)
    swd = sw.reshape(-1).contiguous().to(dev)

    def run(a_rows, res_rows):
        a8 = torch.empty(a_rows.shape, dtype=torch.uint8, device=dev)
        sa = torch.empty(a_rows.shape[0], dtype=torch.float32, device=dev)
        _lib.check(lib.tt_quantize_rows_fp8(a_rows.data_ptr(), a_rows.shape[0], k, a8.data_ptr(), sa.data_ptr(), _stream()), "quantize")
        c = torch.empty(a_rows.shape[0], n, dtype=torch.bfloat16, device=dev)
        _lib.check(lib.tt_gemm_fp8_ex(a8.data_ptr(), sa.data_ptr(), w8.data_ptr(), swd.data_ptr(), bias.data_ptr(),
                                      res_rows.data_ptr() if res_rows is not None else None, c.data_ptr(), None, 0.0,
                                      a_rows.shape[0], n, k, epi, _stream()), "gemm fp8")
        torch.cuda.synchronize()
        return c, a8, sa

    got, a8, sa = run(a, res)
    assert bool(got.view(torch.int16).bitwise_and(0x7F80).ne(0x7F80).all()), "non-finite output"
    rows = torch.from_numpy(np.unique(np.concatenate([np.random.default_rng(n).integers(0, m, 1024), np.arange(0, 256),
                                                      np.arange(m - 256, m)]))).to(dev)
    aq = a8[rows].cpu().view(torch.float8_e4m3fn).double()
    want = (aq @ wq.double().T).float() * sa[rows].cpu()[:, None] * sw.T + bias.cpu()
    if epi == 1:
        want = oe.gelu_erf(want)
    elif epi == 2:
        want = want + res[rows].float().cpu()
    err = (got[rows].float().cpu() - want).abs()
    assert (err <= 2.0 ** -7 * want.abs() + 2e-3).all(), (err.max().item(), want.abs().max().item())
    perm = torch.randperm(m, device=dev, generator=g)
    got_p, _, _ = run(a[perm].contiguous(), res[perm].contiguous() if res is not None else None)
    assert torch.equal(got_p, got[perm])
