"""GPU: the fp16 compute mode (`precision="fp16"` / the reference's `torch_dtype: "float16"`; libtt_hip's `*_f16` entry points).

The mode is the 16-bit encoder path compiled a second time with IEEE fp16 elements and `v_mfma_*_f16` (csrc/common.h, TT_F16):
the same tiles, LDS images, copy schedules and wait counts, another element arithmetic.  What is checked here:
  * the building blocks (`tt_gemm_f16` on the 256-tile, 128-tile and skinny kernels, `tt_layernorm_f16`, `tt_attention_varlen_f16`)
    against fp32 references of the same fp16-rounded operands, within ONE fp16 rounding of the result;
  * saturation instead of infinities at the GEMM output;
  * the whole forward (embed + rerank) against the fp32 CPU oracle and against the oracle emulating the mode's rounding points;
  * a sequence's result does not depend on the batch it travels in (bit for bit), as in the bf16 mode;
  * the precision selector (`torch_dtype: float16` lands here, fp8 projections are refused).
The full-depth rank agreement (24 layers, 4 x 50 x 292) is in tests/test_rank_agreement_gpu.py next to the other modes."""
import numpy as np
import pytest
import torch

from oracle import encoder as oe

pytestmark = pytest.mark.gpu

F16_EPS = 2.0 ** -11            # half an ulp of a normal fp16 value, relative


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _gemm(lib, _lib, a, w, bias, res, epi):
    m, k = a.shape
    n = w.shape[0]
    c = torch.empty(m, n, dtype=torch.float16, device=a.device)
    rc = lib.tt_gemm_f16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), res.data_ptr() if epi == 2 else None, c.data_ptr(),
                         m, n, k, epi, _stream())
    _lib.check(rc, "tt_gemm_f16")
    return c


# 256-tile kernels (one-tile and, for GELU, persistent), the 128-tile kernel of small grids, the weight-streaming skinny kernel
@pytest.mark.parametrize("m,n,k,epi", [(8192, 2048, 1024, 0), (8192, 1024, 1024, 2), (8192 + 256, 4096, 1024, 1), (8192, 1024, 4096, 2),
                                       (256 * 37, 768, 3072, 2), (1024, 1024, 1024, 0), (1280, 4096, 1024, 1), (768, 1024, 4096, 2),
                                       (64, 3072, 1024, 0), (128, 1024, 4096, 2), (64, 4096, 1024, 1), (192, 384, 1536, 2)])
def test_gemm_f16_against_the_cpu_reference(dev, built_lib, m, n, k, epi):
    from tensor_truth_amd import _lib

    lib = _lib.load_library()
    g = torch.Generator().manual_seed(m + n + k + epi)
    a = torch.randn(m, k, generator=g).to(torch.float16)
    w = (torch.randn(n, k, generator=g) * 0.05).to(torch.float16)
    bias = torch.randn(n, generator=g) * 0.1
    res = torch.randn(m, n, generator=g).to(torch.float16)
    got = _gemm(lib, _lib, a.to(dev), w.to(dev), bias.to(dev), res.to(dev), epi).float().cpu()
    ref = a.float() @ w.float().T + bias
    if epi == 1:
        ref = oe.gelu_erf(ref)
    elif epi == 2:
        ref = ref + res.float()
    err = (got - ref).abs()
    bad = err > 2 * F16_EPS * ref.abs() + 2e-4        # one rounding of the result + fp32 summation-order noise
    assert not bad.any(), f"{int(bad.sum())} elements off, max err {err.max().item()}, first at {torch.nonzero(bad)[0].tolist()}"


def test_gemm_f16_saturates_instead_of_overflowing(dev, built_lib):
    from tensor_truth_amd import _lib

    lib = _lib.load_library()
    m, n, k = 512, 256, 256
    a = torch.full((m, k), 60.0, dtype=torch.float16, device=dev)
    w = torch.full((n, k), 30.0, dtype=torch.float16, device=dev)
    w[::2] *= -1                                              # 60 * 30 * 256 = 460 800 >> 65 504, both signs
    bias = torch.zeros(n, device=dev)
    got = _gemm(lib, _lib, a, w, bias, a, 0).float().cpu()
    assert torch.isfinite(got).all()
    assert (got[:, 1::2] == 65504.0).all() and (got[:, 0::2] == -65504.0).all()


def test_layernorm_and_attention_f16(dev, built_lib):
    from tensor_truth_amd import _lib

    lib = _lib.load_library()
    g = torch.Generator().manual_seed(9)
    rows, H = 1000, 1024
    x = (torch.randn(rows, H, generator=g) * 3 + 0.5).to(torch.float16)
    gamma, beta = 1 + 0.1 * torch.randn(H, generator=g), 0.1 * torch.randn(H, generator=g)
    out = torch.empty(rows, H, dtype=torch.float16, device=dev)
    xd, gd, bd = x.to(dev), gamma.to(dev), beta.to(dev)          # (named: a temporary would be freed before the kernel reads it)
    rc = lib.tt_layernorm_f16(xd.data_ptr(), out.data_ptr(), gd.data_ptr(), bd.data_ptr(), rows, H, 1e-5, _stream())
    _lib.check(rc, "tt_layernorm_f16")
    ref = torch.nn.functional.layer_norm(x.float(), (H,), gamma, beta, 1e-5)
    assert (out.float().cpu() - ref).abs().max().item() <= 2 * F16_EPS * ref.abs().max().item() + 1e-4

    # attention: ragged sequences, 16 heads x 64, against softmax(QK^T / 8) V in fp32 on the same fp16 operands
    heads, dh = 16, 64
    lens = [292, 17, 64, 130, 1, 257]
    T = sum(lens)
    Tp = (T + 7) // 8 * 8
    Hd = heads * dh
    qk = (torch.randn(Tp, 2 * Hd, generator=g) * 0.8).to(torch.float16)
    v = torch.randn(Tp, Hd, generator=g).to(torch.float16)
    vt = v.view(Tp // 8, 8, Hd).permute(0, 2, 1).contiguous()        # V8 layout: [row / 8][feature][row % 8]
    starts = np.cumsum([0] + lens[:-1]).astype(np.int32)
    outa = torch.zeros(Tp, Hd, dtype=torch.float16, device=dev)
    qkd, vtd = qk.to(dev), vt.to(dev)
    sd, ld = torch.from_numpy(starts).to(dev), torch.tensor(lens, dtype=torch.int32).to(dev)
    rc = lib.tt_attention_varlen_f16(qkd.data_ptr(), 2 * Hd, 0, Hd, vtd.data_ptr(), 8 * Hd, outa.data_ptr(), Hd,
                                     sd.data_ptr(), ld.data_ptr(), len(lens), heads, dh, max(lens), _stream())
    _lib.check(rc, "tt_attention_varlen_f16")
    got = outa.float().cpu()
    worst = 0.0
    for s0, n in zip(starts, lens):
        q = qk[s0:s0 + n, :Hd].float().view(n, heads, dh).transpose(0, 1)
        kk = qk[s0:s0 + n, Hd:].float().view(n, heads, dh).transpose(0, 1)
        vv = v[s0:s0 + n].float().view(n, heads, dh).transpose(0, 1)
        ref_o = (torch.softmax(q @ kk.transpose(1, 2) / 8.0, -1) @ vv).transpose(0, 1).reshape(n, Hd)
        worst = max(worst, (got[s0:s0 + n] - ref_o).abs().max().item())
    assert worst < 3e-3, worst        # probabilities and the output are rounded to fp16 once each (bf16 mode: 2e-2)


SMALL_X = dict(arch="xlmr", vocab_size=3000, hidden=256, layers=2, heads=4, ffn=1024, max_pos=300, type_vocab=1, pad_id=1,
               ln_eps=1e-5, num_labels=1)
SMALL_B = dict(arch="bert", vocab_size=3000, hidden=384, layers=2, heads=12, ffn=1536, max_pos=256, type_vocab=2, pad_id=0,
               ln_eps=1e-12)


def _ragged(rng, n, lo, hi, vocab, bos, eos):
    return [[bos] + rng.integers(5, vocab, size=int(rng.integers(lo, hi)) - 2).tolist() + [eos] for _ in range(n)]


def _padded(seqs, pad):
    L = max(map(len, seqs))
    ids = torch.full((len(seqs), L), pad, dtype=torch.long)
    mask = torch.zeros((len(seqs), L), dtype=torch.long)
    for i, s in enumerate(seqs):
        ids[i, : len(s)] = torch.tensor(s)
        mask[i, : len(s)] = 1
    return ids, mask


def test_forward_f16_vs_the_oracle_and_batch_independence(dev, built_lib):
    from tensor_truth_amd.encoder import Encoder, EncoderConfig, EncoderWeights

    rng = np.random.default_rng(2)
    # ---- cross-encoder (XLM-R shape): scores vs fp32 oracle and vs the oracle rounding where the fp16 mode rounds
    cfg = EncoderConfig(**SMALL_X)
    ocfg = oe.EncoderConfig(**SMALL_X)
    W = oe.synth_weights(ocfg, seed=5)
    seqs = _ragged(rng, 48, 6, 280, 3000, 0, 2)
    ids, mask = _padded(seqs, 1)
    with torch.no_grad():
        ref = oe.rerank_scores(ids, mask, W, ocfg)
        with oe.rounding_dtype(torch.float16):
            W16 = {k: (v.to(torch.float16).float() if v.dim() == 2 else v) for k, v in W.items()}
            emu = oe.rerank_scores(ids, mask, W16, ocfg, emulate_bf16=True)
    enc16 = Encoder(EncoderWeights(cfg, W, dev, dtype=torch.float16))
    enc_bf = Encoder(EncoderWeights(cfg, W, dev))
    got16, got_bf = enc16.rerank(seqs).cpu(), enc_bf.rerank(seqs).cpu()
    e16, ebf = (got16 - ref).abs().max().item(), (got_bf - ref).abs().max().item()
    assert e16 < 1e-3 and e16 < ebf / 3, (e16, ebf)                # the mode's point: several times closer than bf16
    assert (got16 - emu).abs().max().item() < 6e-4                   # and where the emulation of its rounding points says
    # a pair's score does not depend on the batch it travels in (alone, in the middle of 48): bit for bit
    for j in (0, 17, 47):
        alone = enc16.rerank([seqs[j]]).cpu()
        assert alone[0].item() == got16[j].item()
    # ---- bi-encoder (BERT-small shape, head_dim 32, token types): embeddings
    cfgb, ocfgb = EncoderConfig(**SMALL_B), oe.EncoderConfig(**SMALL_B)
    Wb = oe.synth_weights(ocfgb, seed=6)
    seqsb = _ragged(rng, 40, 3, 200, 3000, 101, 102)
    idsb, maskb = _padded(seqsb, 0)
    with torch.no_grad():
        refb = oe.embed(idsb, maskb, Wb, ocfgb)
    encb = Encoder(EncoderWeights(cfgb, Wb, dev, dtype=torch.float16))
    embf32, emb16 = encb.embed(seqsb)
    assert emb16.dtype == torch.bfloat16                            # the 16-bit copy is a scan query: bf16 like the corpus
    assert torch.equal(emb16, embf32.to(torch.bfloat16))
    err = (embf32.cpu() - refb).abs().max().item()
    cos = (embf32.cpu() * refb).sum(1).min().item()
    assert cos > 0.99999 and err < 1e-3, (cos, err)
    m_out, _ = encb.embed_packed(__import__("tensor_truth_amd.encoder", fromlist=["pack_tokens"]).pack_tokens(seqsb, cfgb, None, None),
                                 pooling="mean")
    with torch.no_grad():
        refm = oe.embed(idsb, maskb, Wb, ocfgb, pooling="mean")
    assert (m_out.cpu() - refm).abs().max().item() < 1e-3


def test_precision_selector_reaches_the_fp16_mode(dev, built_lib):
    from tensor_truth_amd import precision
    from tensor_truth_amd.encoder import EncoderConfig, EncoderWeights
    from tensor_truth_amd.rerank import HipSentenceTransformerRerank
    from tensor_truth_amd.tokenization import HashTokenizer

    assert precision.resolve({"torch_dtype": "float16"}) == "fp16" and precision.resolve({"torch_dtype": torch.float16}) == "fp16"
    assert precision.resolve({"precision": "half"}) == "fp16" and precision.resolve({}, environ={"TT_PRECISION": "fp16"}) == "fp16"
    assert precision.resolve({"torch_dtype": "bfloat16"}) == "bf16"
    cfg = EncoderConfig(**SMALL_X)
    W = oe.synth_weights(oe.EncoderConfig(**SMALL_X), seed=5)
    rr = HipSentenceTransformerRerank(model="test/xenc", top_n=3, device="cuda",
                                      model_kwargs={"encoder_config": cfg, "state_dict": W, "torch_dtype": "float16",
                                                    "tokenizer": HashTokenizer("xlmr", 3000)})
    assert rr.precision.startswith("fp16") and rr.model.dtype == torch.float16
    assert next(iter(rr.model.parameters())).dtype in (torch.float16, torch.float32)     # matrices fp16, vectors fp32
    with pytest.raises(ValueError):
        EncoderWeights(cfg, W, torch.device(dev), dtype=torch.float16).set_gemm_dtype("fp8")
    scores = rr.predict([("alpha beta", "gamma delta epsilon"), ("alpha beta", "zeta eta")])
    assert len(scores) == 2 and all(0.0 < s < 1.0 for s in scores)
    # the embedder through the same selector; the process-level switch (ModelManager) hands the mode to what it loads
    from tensor_truth_amd.embedding import HipHuggingFaceEmbedding
    from tensor_truth_amd.model_manager import ModelManager

    cfgb = EncoderConfig(**SMALL_B)
    Wb = oe.synth_weights(oe.EncoderConfig(**SMALL_B), seed=6)
    kw = {"encoder_config": cfgb, "state_dict": Wb, "tokenizer": HashTokenizer("bert", 3000)}
    e16 = HipHuggingFaceEmbedding("test/bge-small-shaped", device="cuda", model_kwargs={**kw, "torch_dtype": torch.float16})
    ebf = HipHuggingFaceEmbedding("test/bge-small-shaped", device="cuda", model_kwargs=kw)
    assert e16.precision.startswith("fp16") and ebf.precision.startswith("bf16")
    a, b = torch.tensor(e16.get_text_embedding("tensor kernel wave matrix")), torch.tensor(ebf.get_text_embedding("tensor kernel wave matrix"))
    assert abs(a.norm().item() - 1) < 1e-3 and (a * b).sum().item() > 0.9995
    mm = ModelManager.get_instance()
    prev = mm.precision
    try:
        mm.set_precision("fp16")
        assert mm._with_precision({"encoder_config": cfgb})["precision"] == "fp16"
        assert mm._with_precision({"torch_dtype": "float32"}).get("precision") is None      # an explicit per-model dtype wins
    finally:
        mm.set_precision(prev)


def test_forward_f16_with_outlier_features(dev, built_lib):
    """Real XLM-R / BERT checkpoints carry a few "outlier" hidden features (LayerNorm gains of tens, activations of hundreds).
    The fp16 mode must take them: a checkpoint-like perturbation -- two features with LayerNorm gain x40 in every layer and an
    FFN whose intermediate reaches into the thousands -- stays finite and stays closer to the fp32 oracle than bf16 does."""
    from tensor_truth_amd.encoder import Encoder, EncoderConfig, EncoderWeights

    cfg = EncoderConfig(**SMALL_X)
    ocfg = oe.EncoderConfig(**SMALL_X)
    W = oe.synth_weights(ocfg, seed=8)
    for name in list(W):
        if name.endswith("LayerNorm.weight"):
            W[name] = W[name].clone()
            W[name][[7, 133]] *= 40.0                                   # outlier features in the residual stream
        if name.endswith("intermediate.dense.weight"):
            W[name] = W[name] * 6.0                                     # FFN intermediate in the hundreds / thousands
        if name.endswith("output.dense.weight") and "attention" not in name:
            W[name] = W[name] / 6.0
    rng = np.random.default_rng(3)
    seqs = _ragged(rng, 32, 6, 250, 3000, 0, 2)
    ids, mask = _padded(seqs, 1)
    with torch.no_grad():
        ref = oe.rerank_logits(ids, mask, W, ocfg)
        hid = oe.encoder_forward(ids, mask, W, ocfg, layers=1)
    assert hid.abs().max().item() > 30.0                               # the outliers are really there
    got16 = Encoder(EncoderWeights(cfg, W, dev, dtype=torch.float16)).rerank(seqs, want_logits=True)[1].cpu()
    got_bf = Encoder(EncoderWeights(cfg, W, dev)).rerank(seqs, want_logits=True)[1].cpu()
    assert torch.isfinite(got16).all()
    e16, ebf = (got16 - ref).abs().max().item(), (got_bf - ref).abs().max().item()
    assert e16 < 5e-3 and e16 < ebf, (e16, ebf)        # measured 2.2e-3 vs 4.8e-3 (logits of magnitude ~1)
