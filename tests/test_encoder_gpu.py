"""GPU parity: encoder kernels (GEMM, LayerNorm, attention, full forward, pooling, rerank head)
through the C ABI against the CPU oracle.

Tolerances (bf16 compute, fp32 accumulate -- the reference's ``torch_dtype: bfloat16`` mode):
  * single kernels: one bf16 rounding of the fp32 reference -> |err| <= 2^-7 |ref| + 2e-3;
  * whole forward vs the oracle emulating the same bf16 rounding points: hidden within
    2^-5 |ref| + 3e-2 for >= 99.9 % of the elements (a few bf16 ulps), max 0.25, mean abs error < 6e-3, embeddings cosine >= 0.9995;
  * embeddings vs the fp32 oracle / transformers golden: cosine >= 0.999, max-abs <= 2e-3 per
    unit-norm component;
  * rerank sigmoid scores: |err| <= 1.5e-2 absolute vs fp32 (looser bound stated for bf16).
"""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import encoder as oe

pytestmark = pytest.mark.gpu


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _bf(x):
    return x.to(torch.bfloat16)


@pytest.mark.parametrize("m,n,k", [(128, 128, 64), (256, 384, 384), (384, 1152, 384), (128, 1536, 384),
                                   (512, 1024, 1024), (256, 4096, 1024), (256, 1024, 4096),
                                   (64, 1024, 1024), (192, 3072, 1024), (64, 1024, 4096), (64, 1152, 384),   # skinny
                                   (384, 128, 64), (640, 1024, 4096), (1024, 384, 1536),                     # 128x128 tiles
                                   # round 6, the staged kernel's four-stage ring around its depth: 1, 2, 3, 5 and 6 K-steps of 64
                                   (256, 256, 128), (128, 256, 192), (256, 128, 320), (384, 384, 384)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
def test_gemm_epilogues(dev, built_lib, m, n, k, epi):
    from tensor_truth_amd import _lib

    lib = _lib.load_library()
    g = torch.Generator().manual_seed(m + n + k + epi)
    a = _bf(torch.randn(m, k, generator=g))
    w = _bf(torch.randn(n, k, generator=g) * 0.05)
    bias = torch.randn(n, generator=g) * 0.1
    res = _bf(torch.randn(m, n, generator=g))
    ref = a.float() @ w.float().T + bias
    if epi == 1:
        ref = oe.gelu_erf(ref)
    elif epi == 2:
        ref = ref + res.float()
    elif epi == 3:
        ref = torch.tanh(ref)
    a_d, w_d, b_d, r_d = a.to(dev), w.to(dev), bias.to(dev), res.to(dev)
    c_d = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    rc = lib.tt_gemm_bf16(a_d.data_ptr(), w_d.data_ptr(), b_d.data_ptr(), r_d.data_ptr() if epi == 2 else None,
                          c_d.data_ptr(), m, n, k, epi, _stream())
    _lib.check(rc, "tt_gemm_bf16")
    torch.cuda.synchronize()
    got = c_d.float().cpu()
    err = (got - ref).abs()
    assert (err <= 2 ** -7 * ref.abs() + 2e-3).all(), f"max err {err.max().item()}"


@pytest.mark.parametrize("epi", [0, 1, 2, 3])
@pytest.mark.parametrize("n,k", [(1024, 1024), (1024, 4096), (384, 1536)])
def test_skinny_gemm_is_bit_identical_to_the_tiled_kernels(dev, built_lib, n, k, epi):
    """Up to 256 rows the projections run as weight-streaming skinny GEMMs (one wave per 16 columns, no LDS); same MFMA,
    same K order, same epilogue code as the tiled kernels -> the same bits for the same rows, whatever else is in the
    batch.  64 rows alone (skinny) vs the same rows inside 512- and 2048-row GEMMs (the staged 128x128 kernel: up to one tile per
    CU) -- and ``test_rows_do_not_depend_on_the_kernel_that_computed_them`` below with the two-stage and 256x256 kernels as well."""
    from tensor_truth_amd import _lib

    lib = _lib.load_library()
    g = torch.Generator().manual_seed(n + k + epi)
    big = 2048
    a = _bf(torch.randn(big, k, generator=g)).to(dev)
    w = _bf(torch.randn(n, k, generator=g) * 0.05).to(dev)
    bias = (torch.randn(n, generator=g) * 0.1).to(dev)
    res = _bf(torch.randn(big, n, generator=g)).to(dev)
    outs = []
    for m in (64, 192, 512, big):
        c = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        rc = lib.tt_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), res.data_ptr() if epi == 2 else None,
                              c.data_ptr(), m, n, k, epi, _stream())
        _lib.check(rc, "tt_gemm_bf16")
        outs.append(c)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1][:64]) and torch.equal(outs[0], outs[2][:64]) and torch.equal(outs[0], outs[3][:64])
    assert torch.equal(outs[1], outs[3][:192])


@pytest.mark.parametrize("epi", [0, 1, 2])
@pytest.mark.parametrize("n,k", [(1024, 1024), (1024, 4096), (3072, 1024)])
def test_rows_do_not_depend_on_the_kernel_that_computed_them(dev, built_lib, n, k, epi):
    """Which GEMM kernel runs is decided by the row count of the batch a text rides in: <= 256 rows the skinny kernel, up to one
    128x128 tile per CU the staged 128x128 kernel (round 6: a lone caller's 10-pair rerank), fewer than 128 tiles of 256x256 the
    two-stage 128x128 kernel, from there the 256x256 ping-pong kernel (a row count that is not a multiple of 256: the two-stage
    kernel again, on a large grid).  One accumulator per output element, K ascending, the same MFMA and epilogue code in all of
    them: the first rows of every launch carry the same bits."""
    from tensor_truth_amd import _lib

    lib = _lib.load_library()
    g = torch.Generator().manual_seed(7 * n + k + epi)
    rows = {1024: (64, 512, 3072, 5120, 8192, 8320), 3072: (64, 512, 2816, 8320)}[n]   # skinny, staged (x2 / x1), [two-stage,] 256x256, two-stage on a large grid
    big = max(rows)
    a = _bf(torch.randn(big, k, generator=g)).to(dev)
    w = _bf(torch.randn(n, k, generator=g) * 0.05).to(dev)
    bias = (torch.randn(n, generator=g) * 0.1).to(dev)
    res = _bf(torch.randn(big, n, generator=g)).to(dev)
    outs = []
    for m in rows:
        c = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        rc = lib.tt_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), res.data_ptr() if epi == 2 else None,
                              c.data_ptr(), m, n, k, epi, _stream())
        _lib.check(rc, "tt_gemm_bf16")
        outs.append(c)
    torch.cuda.synchronize()
    for m, c in zip(rows[:-1], outs[:-1]):
        assert torch.equal(c.view(torch.int16), outs[-1][:m].view(torch.int16)), m
    ref = a[:512].float() @ w.float().T + bias                      # (and they are the right bits: the product itself)
    if epi == 1:
        ref = oe.gelu_erf(ref.cpu()).to(dev)
    elif epi == 2:
        ref = ref + res[:512].float()
    err = (outs[1].float() - ref).abs()
    assert (err <= 2 ** -7 * ref.abs() + 2e-3).all()


def test_gemm_rejects_bad_shapes(dev, built_lib):
    from tensor_truth_amd import _lib

    lib = _lib.load_library()
    t = torch.zeros(128, 128, dtype=torch.bfloat16, device=dev)
    b = torch.zeros(128, device=dev)
    rc = lib.tt_gemm_bf16(t.data_ptr(), t.data_ptr(), b.data_ptr(), None, t.data_ptr(), 100, 128, 128, 0, _stream())
    assert rc != 0 and b"multiples" in lib.tt_last_error()


@pytest.mark.parametrize("rows,h,eps", [(5, 384, 1e-12), (1000, 1024, 1e-5), (129, 128, 1e-5)])
def test_layernorm(dev, built_lib, rows, h, eps):
    from tensor_truth_amd import _lib

    lib = _lib.load_library()
    g = torch.Generator().manual_seed(rows)
    x = _bf(torch.randn(rows, h, generator=g) * 3 + 0.5)
    gamma = 1 + 0.1 * torch.randn(h, generator=g)
    beta = 0.1 * torch.randn(h, generator=g)
    ref = oe.layer_norm(x.float(), gamma, beta, eps)
    x_d, g_d, b_d = x.to(dev), gamma.to(dev), beta.to(dev)
    y_d = torch.empty_like(x_d)
    _lib.check(lib.tt_layernorm_bf16(x_d.data_ptr(), y_d.data_ptr(), g_d.data_ptr(), b_d.data_ptr(), rows, h, eps,
                                     _stream()), "ln")
    torch.cuda.synchronize()
    err = (y_d.float().cpu() - ref).abs()
    assert (err <= 2 ** -7 * ref.abs() + 2e-3).all(), err.max().item()


def _attention_case(dev, heads, dh, lens, align, q, k, v, max_abs=2e-2, mean_abs=2e-3):
    """q, k, v: [T, heads * dh] bf16 (rows beyond the packed sequences are padding) -> checks every sequence against fp32 softmax."""
    from tensor_truth_amd import _lib

    lib = _lib.load_library()
    H = heads * dh
    starts, off = [], 0
    for n in lens:
        starts.append(off)
        off += (n + align - 1) // align * align      # align 1: sequences share 8-row token groups
    T = q.shape[0]
    assert T % 128 == 0 and T >= off
    qk = torch.cat([q, k], 1).contiguous()
    vt = v.view(T // 8, 8, H).permute(0, 2, 1).contiguous()   # V8 layout [T/8][H][8]
    out = torch.zeros(T, H, dtype=torch.bfloat16, device=dev)
    st_d = torch.tensor(starts, dtype=torch.int32, device=dev)
    ln_d = torch.tensor(lens, dtype=torch.int32, device=dev)
    qk_d, vt_d = qk.to(dev), vt.to(dev)
    _lib.check(lib.tt_attention_varlen(qk_d.data_ptr(), 2 * H, 0, H, vt_d.data_ptr(), 8 * H, out.data_ptr(), H,
                                       st_d.data_ptr(), ln_d.data_ptr(), len(lens), heads, dh, max(lens), _stream()),
               "attention")
    torch.cuda.synchronize()
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    for s0, n in zip(starts, lens):
        qq = q[s0:s0 + n].float().view(n, heads, dh).transpose(0, 1)
        kk = k[s0:s0 + n].float().view(n, heads, dh).transpose(0, 1)
        vv = v[s0:s0 + n].float().view(n, heads, dh).transpose(0, 1)
        p = torch.softmax(qq @ kk.transpose(-1, -2) / math.sqrt(dh), dim=-1)
        ref = (p @ vv).transpose(0, 1).reshape(n, H)
        err = (got[s0:s0 + n] - ref).abs()
        # P is rounded to bf16 before the PV product: error ~ 2^-8 * sum|p v|
        assert err.max().item() < max_abs, f"len {n} at row {s0}: max err {err.max().item()}"
        assert err.mean().item() < mean_abs


def _padded_rows(lens, align):
    off = sum((n + align - 1) // align * align for n in lens)
    return (off + 127) // 128 * 128


@pytest.mark.parametrize("align", [8, 1])
@pytest.mark.parametrize("heads,dh,lens", [(16, 64, [16, 9, 5, 12]), (12, 32, [16, 7, 11, 3]),
                                           (4, 64, [300, 64, 65, 129, 1]), (2, 32, [513, 128]),
                                           (2, 64, [1100])])
def test_attention_varlen(dev, built_lib, heads, dh, lens, align):
    H = heads * dh
    g = torch.Generator().manual_seed(sum(lens) + heads)
    T = _padded_rows(lens, align)
    q, k, v = (_bf(torch.randn(T, H, generator=g)) for _ in range(3))
    _attention_case(dev, heads, dh, lens, align, q, k, v)


@pytest.mark.parametrize("align", [8, 1])
@pytest.mark.parametrize("heads,dh,seed", [(16, 64, 1), (12, 32, 2), (4, 64, 3)])
def test_attention_random_length_mix(dev, built_lib, heads, dh, seed, align):
    """Seeded sweep: 60 sequences of 1..700 tokens packed back to back (every query-tile / key-tile remainder, sequences
    that start inside another sequence's 8-token V group when align = 1), the bench's 292 among them."""
    rng = np.random.default_rng(seed)
    lens = [int(x) for x in rng.integers(1, 700, size=57)] + [292, 292, 1]
    H = heads * dh
    g = torch.Generator().manual_seed(seed)
    T = _padded_rows(lens, align)
    q, k, v = (_bf(torch.randn(T, H, generator=g)) for _ in range(3))
    _attention_case(dev, heads, dh, lens, align, q, k, v)


@pytest.mark.parametrize("where", ["late", "early", "every_tile"])
def test_attention_running_maximum_under_spiked_logits(dev, built_lib, where):
    """The kernel's softmax reference is lazy (it only moves when a tile's maximum exceeds it by 2^8, attention.hip): logits
    that jump by tens of units between key tiles -- a spike key late in the sequence, early, or a bigger one in every tile --
    must still normalise like the fp32 softmax."""
    heads, dh, lens = 4, 64, [292, 320, 64, 129]
    H = heads * dh
    g = torch.Generator().manual_seed(7)
    T = _padded_rows(lens, 8)
    q = torch.randn(T, H, generator=g)
    k = torch.randn(T, H, generator=g) * 0.3
    v = torch.randn(T, H, generator=g)
    starts = np.cumsum([0] + [(n + 7) // 8 * 8 for n in lens[:-1]])
    for s0, n in zip(starts, lens):
        # spike keys: aligned with the mean query direction of the sequence, so EVERY query scores them far above the rest
        d = torch.nn.functional.normalize(q[s0:s0 + n].view(n, heads, dh).mean(0), dim=-1).reshape(H)
        q[s0:s0 + n] += 6.0 * d                       # every query has a large component along d
        spots = {"late": [n - 3], "early": [1], "every_tile": list(range(5, n, 64))}[where]
        for j, pos in enumerate(spots):
            k[s0 + pos] = (4.0 + 3.0 * j) * d          # logits ~ 6 * (4 + 3 j) * 8 / 8 = +24, +42, +60 ... natural units
    _attention_case(dev, heads, dh, lens, 8, _bf(q), _bf(k), _bf(v), max_abs=4e-2, mean_abs=4e-3)


def _load_golden(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    cfg_o = oe.EncoderConfig(**json.loads(str(z["cfg"])))
    return z, cfg_o


def _product_cfg(cfg_o):
    from tensor_truth_amd.encoder import EncoderConfig

    return EncoderConfig(**cfg_o.__dict__)


def _seqs_from_padded(ids, mask):
    return [ids[b, : int(mask[b].sum())].tolist() for b in range(ids.shape[0])]


@pytest.mark.parametrize("name", ["xlmr_encoder.npz", "bert_encoder.npz"])
def test_encoder_forward_vs_golden_and_oracle(dev, built_lib, golden_dir, name):
    from tensor_truth_amd.encoder import Encoder, EncoderWeights, pack_tokens

    z, cfg_o = _load_golden(golden_dir, name)
    W = oe.synth_weights(cfg_o, seed=int(z["seed"]))
    Wb = {k: v.to(torch.bfloat16) for k, v in W.items()}
    ids, mask = torch.from_numpy(z["ids"]), torch.from_numpy(z["mask"])
    type_ids = torch.from_numpy(z["type_ids"]) if "type_ids" in z.files else None
    seqs = _seqs_from_padded(ids, mask)
    types = None if type_ids is None else [type_ids[b, : len(s)].tolist() for b, s in enumerate(seqs)]
    cfg = _product_cfg(cfg_o)
    enc = Encoder(EncoderWeights(cfg, W, dev))
    batch = pack_tokens(seqs, cfg, types)
    hidden, _ = enc.forward_packed(batch)
    emb, emb16 = enc.embed_packed(batch)
    torch.cuda.synchronize()
    hidden, emb = hidden.float().cpu(), emb.cpu()
    # (1) vs the oracle emulating the kernel's bf16 rounding points, bf16 weights
    want_h = oe.encoder_forward(ids, mask, Wb, cfg_o, emulate_bf16=True, type_ids=type_ids)
    for b, s in enumerate(seqs):
        st = int(batch.seq_start[b])
        ref = want_h[b, : len(s)]
        err = (hidden[st:st + len(s)] - ref).abs()
        # a few bf16 ulps of the value (ulp = 2^-8 |x|) plus an absolute floor
        # (a value sitting on a bf16 rounding boundary before a LayerNorm can flip by one
        # ulp of the pre-LN magnitude, so the bound is on the 99.9th percentile + a hard cap)
        bad = (err > 2 ** -5 * ref.abs() + 3e-2).float().mean().item()
        assert bad < 1e-3 and err.max().item() < 0.25, f"seq {b}: {bad:.2e} outliers, max err {err.max().item()}"
        assert err.mean().item() < 6e-3
    want_e = oe.cls_pool_normalize(want_h)
    assert ((emb * want_e).sum(1) >= 0.9995).all()
    # (2) vs the transformers golden (fp32 weights and math)
    gold = torch.from_numpy(z["emb"])
    cos = (emb * gold).sum(1)
    assert (cos >= 0.999).all(), cos
    assert (emb - gold).abs().max().item() <= 2e-3 * math.sqrt(1024 / cfg.hidden) * 2
    assert torch.allclose(emb.norm(dim=1), torch.ones(emb.shape[0]), atol=1e-3)
    assert torch.equal(emb16.cpu(), emb.to(torch.bfloat16))


def test_rerank_head_vs_golden(dev, built_lib, golden_dir):
    from tensor_truth_amd.encoder import Encoder, EncoderWeights, pack_tokens

    z, cfg_o = _load_golden(golden_dir, "xenc_head.npz")
    W = oe.synth_weights(cfg_o, seed=int(z["seed"]))
    ids, mask = torch.from_numpy(z["ids"]), torch.from_numpy(z["mask"])
    seqs = _seqs_from_padded(ids, mask)
    cfg = _product_cfg(cfg_o)
    enc = Encoder(EncoderWeights(cfg, W, dev))
    scores, logits = enc.rerank(seqs, want_logits=True)
    torch.cuda.synchronize()
    scores, logits = scores.cpu(), logits.cpu()
    Wb = {k: v.to(torch.bfloat16) for k, v in W.items()}
    want_emul = oe.rerank_logits(ids, mask, Wb, cfg_o, emulate_bf16=True)
    # the synthetic head (out_proj std 0.2 over 1024 inputs) amplifies bf16 noise ~6x
    assert (logits - want_emul).abs().max().item() < 1e-1
    gold_s = torch.from_numpy(z["scores"])
    assert (scores - gold_s).abs().max().item() < 1.5e-2  # stated bf16 bound
    assert torch.allclose(scores, torch.sigmoid(logits), atol=1e-6)
    # order of the pairs is preserved wherever the fp32 scores are separated by twice the score bound -- checked on
    # every separable pair, unconditionally (rank_checks.py), and the golden batch must contain such pairs
    from rank_checks import assert_order_on_separable
    n_sep = assert_order_on_separable(gold_s.numpy(), scores.numpy(), 3e-2, "xenc_head golden")
    assert n_sep >= 1, "golden scores hold no separable pair: regenerate the fixture with a wider head"


def test_encoder_longer_sequences_and_truncation(dev, built_lib):
    """Ragged 4-layer XLM-R-shaped model with sequences crossing the 64/128 tile sizes."""
    from tensor_truth_amd.encoder import Encoder, EncoderConfig, EncoderWeights, pack_tokens

    cfg_o = oe.EncoderConfig(arch="xlmr", vocab_size=2000, hidden=256, layers=3, heads=4, ffn=512, max_pos=300,
                             type_vocab=1, pad_id=1, ln_eps=1e-5)
    cfg = EncoderConfig(**cfg_o.__dict__)
    W = oe.synth_weights(cfg_o, seed=5)
    Wb = {k: v.to(torch.bfloat16) for k, v in W.items()}
    lens = [200, 65, 1 + 128, 33, 298, 400]  # the last one is truncated to max_seq_len = 298
    g = torch.Generator().manual_seed(1)
    seqs = [torch.randint(4, 2000, (n,), generator=g).tolist() for n in lens]
    enc = Encoder(EncoderWeights(cfg, W, dev))
    emb, _ = enc.embed(seqs)
    torch.cuda.synchronize()
    L = cfg.max_seq_len
    ids = torch.full((len(seqs), L), cfg.pad_id, dtype=torch.int64)
    mask = torch.zeros(len(seqs), L, dtype=torch.int64)
    for b, s in enumerate(seqs):
        n = min(len(s), L)
        ids[b, :n] = torch.tensor(s[:n])
        mask[b, :n] = 1
    want = oe.embed(ids, mask, Wb, cfg_o, emulate_bf16=True)
    cos = (emb.cpu() * want).sum(1)
    assert (cos >= 0.9995).all(), cos


@pytest.mark.parametrize("shape", ["xlmr", "bert"])
def test_cls_only_last_layer_matches_full_forward(dev, built_lib, shape):
    """tt_encoder_forward_cls == the CLS rows of tt_encoder_forward (the single-query attention keeps its
    probabilities in fp32 instead of bf16: agreement to a couple of bf16 ulps)."""
    from tensor_truth_amd.encoder import Encoder, EncoderConfig, EncoderWeights, pack_tokens

    if shape == "xlmr":
        cfg = EncoderConfig(arch="xlmr", vocab_size=2000, hidden=256, layers=3, heads=4, ffn=512, max_pos=600,
                            type_vocab=1, pad_id=1, ln_eps=1e-5)
    else:
        cfg = EncoderConfig(arch="bert", vocab_size=2000, hidden=384, layers=2, heads=12, ffn=1536, max_pos=600,
                            type_vocab=2, pad_id=0, ln_eps=1e-12)
    from tensor_truth_amd.encoder import synthetic_state

    enc = Encoder(EncoderWeights(cfg, synthetic_state(cfg, seed=7), dev))
    g = torch.Generator().manual_seed(2)
    lens = [300, 1, 64, 65, 129, 7, 513, 40]
    seqs = [torch.randint(4, 2000, (n,), generator=g).tolist() for n in lens]
    batch = pack_tokens(seqs, cfg)
    full, starts = enc.forward_packed(batch)
    cls, rows = enc.cls_hidden_packed(batch)
    torch.cuda.synchronize()
    want = full[starts.long()].float().cpu()
    got = cls[: len(seqs)].float().cpu()
    err = (got - want).abs()
    assert (err <= 2 ** -6 * want.abs() + 2e-2).all(), err.max().item()
    assert err.mean().item() < 3e-3


@pytest.mark.parametrize("shape,dtype", [("xlmr256", torch.bfloat16), ("bert384", torch.bfloat16), ("xlmr1024", torch.bfloat16),
                                         ("xlmr1024", torch.float16), ("xlmr1024-one", torch.bfloat16)])
def test_cls_tail_kv_only_projection_is_bit_identical(dev, built_lib, tmp_path, diag_lib_env, shape, dtype):
    """Last layer of the CLS tail: K and V from the big projection, the first rows' queries from a small GEMM over the gathered
    rows (encoder_api.hip) -- the same bits as the full Q,K,V projection it replaces (TT_CLS_KV_ONLY=0), on the mixed-epilogue path
    (small grids), the split path (>= 4096 rows) and the skinny path (one query), bf16 and fp16.  The library reads the switch once
    per process, so the other side runs in a child process (this file as a script) and hands its tensors over in a file."""
    import subprocess
    import sys

    cls_new, emb_new, score_new = _kv_only_case(shape, dtype, dev)
    out = tmp_path / "old.pt"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(diag_lib_env, TT_CLS_KV_ONLY="0", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    subprocess.run([sys.executable, os.path.abspath(__file__), "kv_only_child", shape, str(dtype), str(out)], check=True, env=env, timeout=600)
    cls_old, emb_old, score_old = torch.load(str(out))
    assert torch.isfinite(cls_new.float()).all()
    assert torch.equal(cls_new.cpu(), cls_old) and torch.equal(emb_new.cpu(), emb_old) and torch.equal(score_new.cpu(), score_old)


def _kv_only_case(shape, dtype, dev):
    from tensor_truth_amd.encoder import Encoder, EncoderConfig, EncoderWeights, pack_tokens, synthetic_state

    if shape == "xlmr256":
        cfg = EncoderConfig(arch="xlmr", vocab_size=2000, hidden=256, layers=3, heads=4, ffn=512, max_pos=600, type_vocab=1, pad_id=1,
                            ln_eps=1e-5, num_labels=1)
        lens = [300, 1, 64, 65, 129, 7, 513, 40]
    elif shape == "bert384":
        cfg = EncoderConfig(arch="bert", vocab_size=2000, hidden=384, layers=2, heads=12, ffn=1536, max_pos=600, type_vocab=2, pad_id=0,
                            ln_eps=1e-12, num_labels=1)
        lens = [300, 1, 64, 65, 129, 7, 513, 40]
    else:
        cfg = EncoderConfig(arch="xlmr", vocab_size=2000, hidden=1024, layers=2, heads=16, ffn=4096, max_pos=600, type_vocab=1, pad_id=1,
                            ln_eps=1e-5, num_labels=1)
        lens = [34] if shape.endswith("one") else [200 + 3 * i for i in range(24)]
    enc = Encoder(EncoderWeights(cfg, synthetic_state(cfg, seed=7), dev, dtype=dtype))
    g = torch.Generator().manual_seed(2)
    seqs = [torch.randint(4, 2000, (n,), generator=g).tolist() for n in lens]
    batch = pack_tokens(seqs, cfg)
    cls = enc.cls_hidden_packed(batch)[0][: len(seqs)].clone()
    emb = enc.embed_packed(batch)[0].clone()
    score = enc.rerank_packed(batch).clone()
    torch.cuda.synchronize()
    return cls, emb, score


def test_forward_is_bit_reproducible(dev, built_lib):
    """The GEMM / attention kernels order their LDS-DMA copies with hand-counted waits: a mis-counted one would show up
    as an occasional different bit.  Same batch, same weights -> the same bits every time (bf16 and fp8 modes);
    tools/probes/soak.py is the long version."""
    import numpy as np

    from tensor_truth_amd.encoder import Encoder, EncoderConfig, EncoderWeights, pack_token_matrix

    cfg = EncoderConfig(arch="xlmr", vocab_size=3000, hidden=1024, layers=2, heads=16, ffn=4096, max_pos=300, type_vocab=1,
                        pad_id=1, ln_eps=1e-5, num_labels=1)
    ocfg = oe.EncoderConfig(**cfg.__dict__)
    W = {k: v.to(torch.bfloat16) for k, v in oe.synth_weights(ocfg, seed=9).items()}
    enc = Encoder(EncoderWeights(cfg, W, dev))
    rng = np.random.default_rng(0)
    pairs = rng.integers(4, cfg.vocab_size, size=(300, 292), dtype=np.int32)
    pairs[:, 0], pairs[:, -1] = 0, 2
    batch = pack_token_matrix(pairs, cfg)          # 88800 rows: hundreds of tiles per GEMM, several per CU
    for mode in ("bf16", "fp8"):
        if mode == "fp8":
            enc.calibrate_fp8(pack_token_matrix(pairs[:32], cfg))
            enc.w.set_gemm_dtype("fp8")
        ref = enc.rerank_packed(batch).clone()
        assert torch.isfinite(ref).all()
        for _ in range(10):
            assert torch.equal(enc.rerank_packed(batch), ref), mode


def test_8192_token_sequence_through_the_encoder(dev, built_lib):
    """bge-m3's limit ("L <= 8192 (embed, rare)"; max_position_embeddings 8194): one sequence at
    the model limit, one truncated to it, and short neighbours, through tt_encoder_forward -- 128 key tiles per query
    tile, position ids up to 8193 -- against the oracle on the same ids."""
    from tensor_truth_amd.encoder import Encoder, EncoderConfig, EncoderWeights

    shape = dict(arch="xlmr", vocab_size=4000, hidden=256, layers=2, heads=4, ffn=512, max_pos=8194, type_vocab=1,
                 pad_id=1, ln_eps=1e-5)
    cfg_o, cfg = oe.EncoderConfig(**shape), EncoderConfig(**shape)
    assert cfg.max_seq_len == 8192
    W = oe.synth_weights(cfg_o, seed=23)
    Wb = {k: v.to(torch.bfloat16) for k, v in W.items()}
    g = torch.Generator().manual_seed(2)
    lens = [8192, 9000, 33, 4099]                        # the second one is truncated to 8192
    seqs = [[0] + torch.randint(4, 4000, (n - 2,), generator=g).tolist() + [2] for n in lens]
    enc = Encoder(EncoderWeights(cfg, W, dev))
    emb, _ = enc.embed(seqs)
    from tensor_truth_amd.encoder import pack_tokens
    hidden, _ = enc.forward_packed(pack_tokens(seqs, cfg))
    torch.cuda.synchronize()
    emb, hidden = emb.cpu(), hidden.float().cpu()
    assert torch.isfinite(emb).all()
    starts = pack_tokens(seqs, cfg).seq_start
    for b, s in enumerate(seqs):
        s = s[:8192]
        ids = torch.tensor([s])
        with torch.no_grad():
            want_h = oe.encoder_forward(ids, torch.ones_like(ids), Wb, cfg_o, emulate_bf16=True)[0]
        want_e = torch.nn.functional.normalize(want_h[0], dim=0)
        assert float((emb[b] * want_e).sum()) >= 0.9995, (b, float((emb[b] * want_e).sum()))
        got_h = hidden[int(starts[b]): int(starts[b]) + len(s)]
        err = (got_h - want_h).abs()
        bad = (err > 2 ** -5 * want_h.abs() + 3e-2).float().mean().item()
        assert bad < 1e-3 and err.max().item() < 0.25, (b, bad, err.max().item())
        # the LAST token attends over the whole sequence too: check it separately (tail key tile + position table end)
        assert err[-1].max().item() < 0.1


_SWEEP_SHAPES = {
    "xlmr256": dict(arch="xlmr", vocab_size=2000, hidden=256, layers=3, heads=4, ffn=512, max_pos=300, type_vocab=1, pad_id=1, ln_eps=1e-5, num_labels=1),
    "bert384": dict(arch="bert", vocab_size=3000, hidden=384, layers=2, heads=12, ffn=1536, max_pos=512, type_vocab=2, pad_id=0, ln_eps=1e-12),
    "xlmr1024": dict(arch="xlmr", vocab_size=4000, hidden=1024, layers=2, heads=16, ffn=4096, max_pos=514, type_vocab=1, pad_id=1, ln_eps=1e-5, num_labels=1),
}


@pytest.mark.parametrize("n_seq", [1, 2, 7, 64, 300])
@pytest.mark.parametrize("shape", sorted(_SWEEP_SHAPES))
def test_random_batch_mixes_embed_and_rerank(dev, built_lib, shape, n_seq):
    """Seeded sweep over batch compositions: 1..300 sequences of 1..max tokens in one packed batch (token counts from a
    handful to ~50 k: the skinny, 128x128 and 256x256 GEMM kernels, the CLS tail with many sequences, every attention tile
    remainder), embeddings and cross-encoder scores against the oracle emulating the same bf16 rounding points."""
    from tensor_truth_amd.encoder import Encoder, EncoderConfig, EncoderWeights

    if shape == "xlmr1024" and n_seq == 300:
        n_seq = 128                                       # (the CPU oracle at width 1024 sets this test's run time)
    cfg_o = oe.EncoderConfig(**_SWEEP_SHAPES[shape])
    cfg = EncoderConfig(**cfg_o.__dict__)
    W = oe.synth_weights(cfg_o, seed=11)
    Wb = {k: v.to(torch.bfloat16) for k, v in W.items()}
    rng = np.random.default_rng(n_seq * 7 + len(shape))
    L = cfg.max_seq_len
    lens = [int(x) for x in rng.integers(1, L + 1, size=n_seq)]
    lens[0] = L                                           # the longest admissible sequence is always there
    if n_seq > 2:
        lens[1] = 1                                       # ... and a single-token one
    seqs = [[int(t) for t in rng.integers(4, cfg.vocab_size, size=n)] for n in lens]
    enc = Encoder(EncoderWeights(cfg, W, dev))
    emb, _ = enc.embed(seqs)
    scores = enc.rerank(seqs) if cfg.arch == "xlmr" else None
    torch.cuda.synchronize()
    ids = torch.full((n_seq, L), cfg.pad_id, dtype=torch.int64)
    mask = torch.zeros(n_seq, L, dtype=torch.int64)
    for b, s in enumerate(seqs):
        ids[b, :len(s)] = torch.tensor(s)
        mask[b, :len(s)] = 1
    want = oe.embed(ids, mask, Wb, cfg_o, emulate_bf16=True)
    cos = (emb.cpu() * want).sum(1)
    assert (cos >= 0.9995).all(), (cos.min().item(), int(cos.argmin()), lens[int(cos.argmin())])
    if scores is not None:
        want_s = torch.sigmoid(oe.rerank_logits(ids, mask, Wb, cfg_o, emulate_bf16=True))
        err = (scores.cpu() - want_s).abs()
        assert err.max().item() < 1.5e-2, (err.max().item(), int(err.argmax()), lens[int(err.argmax())])


def test_bench_sized_attention_sampled_sequences_and_order_equivariance(dev, built_lib):
    """The bench's attention launch (1600 pairs x 292 tokens, 16 heads x 64): sampled sequences against the fp32 softmax,
    and a size-independent property over the WHOLE output -- a sequence's context depends on its own rows only, so packing
    the same sequences in another order permutes the output rows, bit for bit, whichever workgroup, CU or XCD a tile lands on."""
    from tensor_truth_amd import _lib

    lib = _lib.load_library()
    heads, dh, n_seq, L = 16, 64, 1600, 292
    H, stride = heads * dh, 296                       # sequences start on 8-row boundaries
    T = (n_seq * stride + 255) // 256 * 256
    g = torch.Generator(device=dev).manual_seed(3)
    q = torch.randn(T, H, device=dev, generator=g).to(torch.bfloat16)
    k = torch.randn(T, H, device=dev, generator=g).to(torch.bfloat16)
    v = torch.randn(T, H, device=dev, generator=g).to(torch.bfloat16)

    def run(q, k, v, starts):
        qk = torch.cat([q, k], 1).contiguous()
        vt = v.view(T // 8, 8, H).permute(0, 2, 1).contiguous()
        out = torch.zeros(T, H, dtype=torch.bfloat16, device=dev)
        st_d = torch.tensor(starts, dtype=torch.int32, device=dev)
        ln_d = torch.full((n_seq,), L, dtype=torch.int32, device=dev)
        _lib.check(lib.tt_attention_varlen(qk.data_ptr(), 2 * H, 0, H, vt.data_ptr(), 8 * H, out.data_ptr(), H, st_d.data_ptr(),
                                           ln_d.data_ptr(), n_seq, heads, dh, L, _stream()), "attention")
        torch.cuda.synchronize()
        return out

    starts = [i * stride for i in range(n_seq)]
    out = run(q, k, v, starts)
    assert bool(out.view(torch.int16).bitwise_and(0x7F80).ne(0x7F80).all()), "non-finite output"
    for s in (0, 1, 777, n_seq - 1):
        s0 = starts[s]
        qq = q[s0:s0 + L].float().cpu().view(L, heads, dh).transpose(0, 1)
        kk = k[s0:s0 + L].float().cpu().view(L, heads, dh).transpose(0, 1)
        vv = v[s0:s0 + L].float().cpu().view(L, heads, dh).transpose(0, 1)
        ref = (torch.softmax(qq @ kk.transpose(-1, -2) / math.sqrt(dh), dim=-1) @ vv).transpose(0, 1).reshape(L, H)
        err = (out[s0:s0 + L].float().cpu() - ref).abs()
        assert err.max().item() < 2e-2 and err.mean().item() < 2e-3, (s, err.max().item())
    # the same sequences packed in a random order
    perm = torch.randperm(n_seq, generator=torch.Generator().manual_seed(4)).tolist()
    rows = torch.cat([torch.arange(starts[p], starts[p] + stride) for p in perm]).to(dev)
    pad = torch.arange(n_seq * stride, T, device=dev)
    rows = torch.cat([rows, pad])
    out_p = run(q[rows].contiguous(), k[rows].contiguous(), v[rows].contiguous(), starts)
    for j in (0, 5, 800, n_seq - 1):
        a = out_p[j * stride: j * stride + L]
        b = out[starts[perm[j]]: starts[perm[j]] + L]
        assert torch.equal(a, b), f"sequence {perm[j]} differs when packed at slot {j}"
    valid = torch.cat([torch.arange(i * stride, i * stride + L) for i in range(n_seq)]).to(dev)
    want = torch.cat([torch.arange(starts[p], starts[p] + L) for p in perm]).to(dev)
    assert torch.equal(out_p[valid], out[want])


if __name__ == "__main__":          # child of test_cls_tail_kv_only_projection_is_bit_identical (TT_CLS_KV_ONLY=0 in its environment)
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if len(sys.argv) == 5 and sys.argv[1] == "kv_only_child":
        dt = {"torch.bfloat16": torch.bfloat16, "torch.float16": torch.float16}[sys.argv[3]]
        torch.save(tuple(t.cpu() for t in _kv_only_case(sys.argv[2], dt, torch.device("cuda:0"))), sys.argv[4])
