"""CPU oracle: BERT / XLM-RoBERTa encoder forward, CLS pooling, rerank head.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Pure torch, fp32 on CPU.

What it restates (reference call sites -> un-vendored upstream arithmetic):

* bi-encoder embedding, ``HuggingFaceEmbedding`` built at
  ``src/tensortruth/services/model_manager.py:254-260`` and
  ``src/tensortruth/indexing/builder.py:146-152``; consumed by
  ``VectorStoreIndex(...)`` at ``indexing/builder.py:437-442`` and by
  ``index.as_retriever`` at ``rag_engine.py:639``.  Upstream arithmetic
  (sentence-transformers >=3.0 ``Transformer -> Pooling(cls) -> Normalize`` over
  transformers ``XLMRobertaModel`` / ``BertModel``):
  post-LN encoder, exact-erf GELU, learned absolute positions, XLM-R position
  ids ``cumsum(mask)*mask + padding_idx``, additive -inf key-padding mask,
  softmax(QK^T/sqrt(dh)), CLS token, L2 normalise.
* cross-encoder rerank, ``SentenceTransformerRerank`` built at
  ``model_manager.py:333-337`` and called at ``services/rag_service.py:617-620``:
  ``XLMRobertaForSequenceClassification`` head ``out_proj(tanh(dense(h[:,0])))``
  then sigmoid (CrossEncoder default activation for one label; Appendix A5-A7).

Pinning: ``tests/golden/make_golden.py`` runs this file against transformers
5.15 (the only implementation importable in the build container; outside the
reference's ``<5`` pin but identical layer math) on seeded synthetic weights;
``tests/test_oracle_golden.py`` re-checks the committed fixtures.

``emulate_bf16=True`` rounds activations to bf16 at exactly the points where the
HIP path stores bf16 (one rounding per fused kernel output), so the GPU parity
test can use a tight tolerance; ``emulate_bf16=False`` is the plain fp32 math
that is compared against transformers.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch


@dataclass(frozen=True)
class EncoderConfig:
    """Architecture description (mirrors the HF config fields the path reads)."""

    arch: str = "xlmr"  # "xlmr" (XLM-RoBERTa) or "bert"
    vocab_size: int = 250002
    hidden: int = 1024
    layers: int = 24
    heads: int = 16
    ffn: int = 4096
    max_pos: int = 8194
    type_vocab: int = 1
    pad_id: int = 1
    ln_eps: float = 1e-5
    num_labels: int = 0  # 1 => cross-encoder head present

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads


# BASELINE.md section 2 shapes.
BGE_M3 = EncoderConfig()
BGE_RERANKER_V2_M3 = EncoderConfig(num_labels=1)
BGE_SMALL_EN = EncoderConfig(
    arch="bert", vocab_size=30522, hidden=384, layers=12, heads=12, ffn=1536,
    max_pos=512, type_vocab=2, pad_id=0, ln_eps=1e-12,
)


# element type the "emulate_bf16" rounding points round to: bfloat16 (the HIP path's default mode), or float16 for its fp16
# mode (tests set it around a call with ``rounding_dtype``)
_ROUND_DTYPE = torch.bfloat16


class rounding_dtype:
    """``with rounding_dtype(torch.float16): encoder_forward(..., emulate_bf16=True)`` emulates the fp16 mode's rounding points."""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global _ROUND_DTYPE
        self.prev, _ROUND_DTYPE = _ROUND_DTYPE, self.dtype

    def __exit__(self, *exc):
        global _ROUND_DTYPE
        _ROUND_DTYPE = self.prev


def _rnd(x: torch.Tensor, on: bool) -> torch.Tensor:
    return x.to(_ROUND_DTYPE).to(torch.float32) if on else x


def position_ids(mask: torch.Tensor, cfg: EncoderConfig) -> torch.Tensor:
    """XLM-R: cumsum(mask)*mask + padding_idx (positions start at pad_id+1);
    BERT: 0..L-1.  [UPSTREAM-K A4]"""
    if cfg.arch == "xlmr":
        m = mask.to(torch.int64)
        return torch.cumsum(m, dim=1) * m + cfg.pad_id
    B, L = mask.shape
    return torch.arange(L, dtype=torch.int64).unsqueeze(0).expand(B, L)


def layer_norm(x: torch.Tensor, g: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * g + b


def gelu_erf(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def quantize_rows_e4m3(t: torch.Tensor):
    """Per-row OCP e4m3 quantisation as the HIP fp8 path does it (BASELINE.json config 5, "fp8 MFMA reranker"):
    q = e4m3(t * 448 / absmax_row) (round to nearest even), scale = absmax_row / 448 (1 for an all-zero row).
    Returns (q as fp32 values, scale [..., 1])."""
    t = t.to(torch.float32)
    amax = t.abs().amax(dim=-1, keepdim=True)
    inv = torch.where(amax > 0, 448.0 / amax, torch.zeros_like(amax))
    q = (t * inv).to(torch.float8_e4m3fn).to(torch.float32)
    scale = torch.where(amax > 0, amax * (1.0 / 448.0), torch.ones_like(amax))
    return q, scale


def linear_fp8(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """x @ w.T + b with x quantised per token and w per output channel to e4m3, fp32 accumulation."""
    xq, sx = quantize_rows_e4m3(x)
    wq, sw = quantize_rows_e4m3(w)
    return (xq @ wq.T) * sx * sw.transpose(-1, -2) + b


def encoder_forward(
    ids: torch.Tensor,
    mask: torch.Tensor,
    W: Dict[str, torch.Tensor],
    cfg: EncoderConfig,
    emulate_bf16: bool = False,
    type_ids: Optional[torch.Tensor] = None,
    layers: Optional[int] = None,
    emulate_fp8: bool = False,
    ffn_act_scales=None,
) -> torch.Tensor:
    """ids, mask: [B, L] int.  Returns last hidden state [B, L, H] fp32.

    Weight names follow the HF checkpoints (no ``roberta.``/``bert.`` prefix).
    ``emulate_fp8``: the layer projections run on e4m3 operands (``linear_fp8``: per-token activation and
    per-output-channel weight scales), everything else as with ``emulate_bf16`` -- the HIP fp8 mode.  The FFN
    output projection joins in when ``ffn_act_scales`` (one static scale per layer for the GELU output, as the HIP
    calibration produces) is given; without it that projection stays bf16, as in the HIP path.
    """
    r = lambda t: _rnd(t, emulate_bf16)  # noqa: E731
    f = lambda name: W[name].to(torch.float32)  # noqa: E731
    lin = (lambda x, w, b: linear_fp8(x, w, b)) if emulate_fp8 else (lambda x, w, b: x @ w.T + b)  # noqa: E731
    ids = ids.to(torch.int64)
    B, L = ids.shape
    H, nh, dh = cfg.hidden, cfg.heads, cfg.head_dim
    pos = position_ids(mask, cfg)
    if type_ids is None:
        type_ids = torch.zeros_like(ids)
    x = (
        f("embeddings.word_embeddings.weight")[ids]
        + f("embeddings.position_embeddings.weight")[pos]
        + f("embeddings.token_type_embeddings.weight")[type_ids.to(torch.int64)]
    )
    x = r(layer_norm(x, f("embeddings.LayerNorm.weight"), f("embeddings.LayerNorm.bias"), cfg.ln_eps))

    neg = torch.zeros(B, 1, 1, L, dtype=torch.float32)
    neg.masked_fill_(mask.to(torch.bool).logical_not().view(B, 1, 1, L), float("-inf"))
    scale = 1.0 / math.sqrt(dh)
    n_layers = cfg.layers if layers is None else layers
    for i in range(n_layers):
        p = f"encoder.layer.{i}."
        q = r(lin(x, f(p + "attention.self.query.weight"), f(p + "attention.self.query.bias")))
        k = r(lin(x, f(p + "attention.self.key.weight"), f(p + "attention.self.key.bias")))
        v = r(lin(x, f(p + "attention.self.value.weight"), f(p + "attention.self.value.bias")))
        q = q.view(B, L, nh, dh).transpose(1, 2)
        k = k.view(B, L, nh, dh).transpose(1, 2)
        v = v.view(B, L, nh, dh).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) * scale + neg
        # HIP path: P is normalised after the PV product; un-normalised exp() is
        # what gets rounded to bf16 for the MFMA.  Emulate that.
        m = s.max(dim=-1, keepdim=True).values
        e = torch.exp(s - m)
        denom = e.sum(dim=-1, keepdim=True)
        ctx = (r(e) @ v) / denom
        ctx = r(ctx.transpose(1, 2).reshape(B, L, H))
        a = lin(ctx, f(p + "attention.output.dense.weight"), f(p + "attention.output.dense.bias"))
        x = r(layer_norm(r(a + x), f(p + "attention.output.LayerNorm.weight"),
                         f(p + "attention.output.LayerNorm.bias"), cfg.ln_eps))
        h = r(gelu_erf(lin(x, f(p + "intermediate.dense.weight"), f(p + "intermediate.dense.bias"))))
        if emulate_fp8 and ffn_act_scales is not None:
            sf = float(ffn_act_scales[i])
            hq = (h * (1.0 / sf)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)   # saturating, static scale
            wq, sw = quantize_rows_e4m3(f(p + "output.dense.weight"))
            o = (hq @ wq.T) * sf * sw.transpose(-1, -2) + f(p + "output.dense.bias")
        else:
            o = h @ f(p + "output.dense.weight").T + f(p + "output.dense.bias")
        x = r(layer_norm(r(o + x), f(p + "output.LayerNorm.weight"),
                         f(p + "output.LayerNorm.bias"), cfg.ln_eps))
    return x


def cls_pool_normalize(hidden: torch.Tensor) -> torch.Tensor:
    """sentence-transformers Pooling(cls) + Normalize: h[:,0]/||h[:,0]||_2 (eps 1e-12)."""
    c = hidden[:, 0, :].to(torch.float32)
    return c / c.norm(dim=-1, keepdim=True).clamp_min(1e-12)


def mean_pool_normalize(hidden: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """sentence-transformers Pooling(mean) + Normalize ([UPSTREAM-K] sentence_transformers/models/Pooling.py): the sum of
    the un-masked token states over the number of un-masked tokens (clamped at 1e-9), then h/||h||_2 (eps 1e-12)."""
    m = mask.to(torch.float32).unsqueeze(-1)
    c = (hidden.to(torch.float32) * m).sum(dim=1) / m.sum(dim=1).clamp_min(1e-9)
    return c / c.norm(dim=-1, keepdim=True).clamp_min(1e-12)


def embed(ids, mask, W, cfg, emulate_bf16=False, type_ids=None, emulate_fp8=False, pooling="cls") -> torch.Tensor:
    """Token ids -> L2-normalised embeddings [B, H] fp32 (reference a2/a4); ``pooling`` "cls" (BGE) or "mean"."""
    h = encoder_forward(ids, mask, W, cfg, emulate_bf16, type_ids, emulate_fp8=emulate_fp8)
    return mean_pool_normalize(h, mask) if pooling == "mean" else cls_pool_normalize(h)


def rerank_logits(ids, mask, W, cfg, emulate_bf16=False, emulate_fp8=False, ffn_act_scales=None) -> torch.Tensor:
    """XLMRobertaClassificationHead: out_proj(tanh(dense(h[:,0]))) -> [B] logits."""
    h = encoder_forward(ids, mask, W, cfg, emulate_bf16, emulate_fp8=emulate_fp8, ffn_act_scales=ffn_act_scales)[:, 0, :]
    t = torch.tanh(h @ W["classifier.dense.weight"].float().T + W["classifier.dense.bias"].float())
    return (t @ W["classifier.out_proj.weight"].float().T + W["classifier.out_proj.bias"].float())[:, 0]


def rerank_scores(ids, mask, W, cfg, emulate_bf16=False, emulate_fp8=False, ffn_act_scales=None) -> torch.Tensor:
    """CrossEncoder.predict default activation for num_labels==1: sigmoid."""
    return torch.sigmoid(rerank_logits(ids, mask, W, cfg, emulate_bf16, emulate_fp8, ffn_act_scales))


def synth_weights(cfg: EncoderConfig, seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights: HF init normal(0, 0.02) with
    LayerNorm gamma/beta perturbed so LN bugs show; biases non-zero."""
    g = torch.Generator().manual_seed(seed)
    n = lambda *s, std=0.02: (torch.randn(*s, generator=g) * std)  # noqa: E731
    H, F = cfg.hidden, cfg.ffn
    W = {
        "embeddings.word_embeddings.weight": n(cfg.vocab_size, H),
        "embeddings.position_embeddings.weight": n(cfg.max_pos, H),
        "embeddings.token_type_embeddings.weight": n(cfg.type_vocab, H),
        "embeddings.LayerNorm.weight": 1.0 + n(H, std=0.1),
        "embeddings.LayerNorm.bias": n(H, std=0.05),
    }
    for i in range(cfg.layers):
        p = f"encoder.layer.{i}."
        for nm in ("query", "key", "value"):
            W[p + f"attention.self.{nm}.weight"] = n(H, H)
            W[p + f"attention.self.{nm}.bias"] = n(H)
        W[p + "attention.output.dense.weight"] = n(H, H)
        W[p + "attention.output.dense.bias"] = n(H)
        W[p + "attention.output.LayerNorm.weight"] = 1.0 + n(H, std=0.1)
        W[p + "attention.output.LayerNorm.bias"] = n(H, std=0.05)
        W[p + "intermediate.dense.weight"] = n(F, H)
        W[p + "intermediate.dense.bias"] = n(F)
        W[p + "output.dense.weight"] = n(H, F)
        W[p + "output.dense.bias"] = n(H)
        W[p + "output.LayerNorm.weight"] = 1.0 + n(H, std=0.1)
        W[p + "output.LayerNorm.bias"] = n(H, std=0.05)
    if cfg.num_labels:
        W["classifier.dense.weight"] = n(H, H)
        W["classifier.dense.bias"] = n(H)
        W["classifier.out_proj.weight"] = n(cfg.num_labels, H, std=0.2)
        W["classifier.out_proj.bias"] = n(cfg.num_labels)
    if dtype != torch.float32:
        W = {k: v.to(dtype) for k, v in W.items()}
    return W


def synth_tokens(B: int, L: int, cfg: EncoderConfig, seed: int = 777, lengths=None):
    """BASELINE.md token inputs: ids uniform in [4, vocab), right padded with
    pad_id, <s>=0 first and </s>=2 last (XLM-R) / [CLS]=101,[SEP]=102 (BERT)."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(4, cfg.vocab_size, (B, L), generator=g, dtype=torch.int64)
    if lengths is None:
        lengths = [L] * B
    mask = torch.zeros(B, L, dtype=torch.int64)
    bos, eos = (0, 2) if cfg.arch == "xlmr" else (101, 102)
    for b, n_tok in enumerate(lengths):
        n_tok = max(2, min(L, int(n_tok)))
        mask[b, :n_tok] = 1
        ids[b, 0] = bos
        ids[b, n_tok - 1] = eos
        ids[b, n_tok:] = cfg.pad_id
    return ids, mask
