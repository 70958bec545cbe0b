"""Encoder (bi-encoder / cross-encoder) host side: weight packing, varlen token packing and
the ctypes calls into libtt_hip.so's ``tt_encoder_forward`` / ``tt_embed_pool`` /
``tt_rerank_head``.

Mirrors what the reference obtains from ``HuggingFaceEmbedding`` (built at
``src/tensortruth/services/model_manager.py:254-260``) and ``SentenceTransformerRerank``
(``model_manager.py:333-337``): XLM-R / BERT encoder forward -> CLS pooling + L2
normalisation (embeddings), or -> classification head + sigmoid (rerank scores).
Computation is bf16 with fp32 accumulation (the reference's ``torch_dtype: bfloat16``
option, ``model_manager.py:218-229``); there is no CPU path.
"""
from __future__ import annotations

import ctypes
import itertools
import os
import threading
from ctypes import POINTER, Structure, c_float, c_int32, c_void_p
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib


@dataclass(frozen=True)
class EncoderConfig:
    """The HF config fields the hot path reads."""

    arch: str = "xlmr"  # "xlmr" | "bert"
    vocab_size: int = 250002
    hidden: int = 1024
    layers: int = 24
    heads: int = 16
    ffn: int = 4096
    max_pos: int = 8194
    type_vocab: int = 1
    pad_id: int = 1
    ln_eps: float = 1e-5
    num_labels: int = 0

    @property
    def max_seq_len(self) -> int:
        # XLM-R reserves positions 0..pad_id for padding
        return self.max_pos - (self.pad_id + 1) if self.arch == "xlmr" else self.max_pos


BGE_M3 = EncoderConfig()
BGE_RERANKER_V2_M3 = EncoderConfig(num_labels=1)
BGE_SMALL_EN_V15 = EncoderConfig(arch="bert", vocab_size=30522, hidden=384, layers=12, heads=12, ffn=1536,
                                 max_pos=512, type_vocab=2, pad_id=0, ln_eps=1e-12)

# the other two rerankers the reference offers out of the box (app_utils/config_schema.py:83-87): XLM-R base with the same
# head as v2-m3, and a 6-layer BERT (MiniLM) whose head is BertForSequenceClassification's pooler + classifier
BGE_RERANKER_BASE = EncoderConfig(vocab_size=250002, hidden=768, layers=12, heads=12, ffn=3072, max_pos=514, num_labels=1)
MS_MARCO_MINILM_L6_V2 = EncoderConfig(arch="bert", vocab_size=30522, hidden=384, layers=6, heads=12, ffn=1536, max_pos=512,
                                      type_vocab=2, pad_id=0, ln_eps=1e-12, num_labels=1)

KNOWN_CONFIGS = {
    "BAAI/bge-m3": BGE_M3,
    "BAAI/bge-reranker-v2-m3": BGE_RERANKER_V2_M3,
    "BAAI/bge-small-en-v1.5": BGE_SMALL_EN_V15,
    "BAAI/bge-reranker-base": BGE_RERANKER_BASE,
    "cross-encoder/ms-marco-MiniLM-L-6-v2": MS_MARCO_MINILM_L6_V2,
}


class _LayerW(Structure):
    _fields_ = [(n, c_void_p) for n in ("qkv_w", "qkv_b", "o_w", "o_b", "ln1_g", "ln1_b", "ffn1_w", "ffn1_b",
                                        "ffn2_w", "ffn2_b", "ln2_g", "ln2_b",
                                        "qkv_w8", "qkv_wscale", "ffn1_w8", "ffn1_wscale",
                                        "o_w8", "o_wscale", "ffn2_w8", "ffn2_wscale")] + [("ffn_act_scale", c_float)]


class _EncW(Structure):
    _fields_ = [
        ("hidden", c_int32), ("layers", c_int32), ("heads", c_int32), ("ffn", c_int32), ("vocab", c_int32),
        ("max_pos", c_int32), ("type_vocab", c_int32), ("ln_eps", c_float),
        ("word_emb", c_void_p), ("pos_emb", c_void_p), ("type_emb", c_void_p), ("emb_ln_g", c_void_p),
        ("emb_ln_b", c_void_p), ("layer", POINTER(_LayerW)),
        ("cls_dense_w", c_void_p), ("cls_dense_b", c_void_p), ("cls_out_w", c_void_p), ("cls_out_b", c_void_p),
        ("ffn_absmax_out", c_void_p),
    ]


def _strip_prefix(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in sd.items():
        for pre in ("roberta.", "bert.", "model.", "0.auto_model."):
            if k.startswith(pre):
                k = k[len(pre):]
                break
        out[k] = v
    # BertForSequenceClassification (cross-encoder/ms-marco-MiniLM-L-6-v2, the third of the reference's out-of-the-box
    # rerankers, config_schema.py:83-87): logits = classifier(tanh(pooler.dense(h[CLS]))) -- the same dense -> tanh -> projection
    # as RobertaClassificationHead's classifier.dense / classifier.out_proj, under other names
    if "classifier.dense.weight" not in out and "pooler.dense.weight" in out and "classifier.weight" in out:
        out["classifier.dense.weight"], out["classifier.dense.bias"] = out["pooler.dense.weight"], out["pooler.dense.bias"]
        out["classifier.out_proj.weight"], out["classifier.out_proj.bias"] = out["classifier.weight"], out["classifier.bias"]
    return out


class EncoderWeights:
    """Device-resident weights in the layout libtt_hip.so expects.

    ``state`` maps HF checkpoint names (``embeddings.word_embeddings.weight``,
    ``encoder.layer.{i}.attention.self.query.weight`` ... optional ``roberta.``/``bert.``
    prefix, ``classifier.dense`` / ``classifier.out_proj`` for the reranker) to tensors.
    Matrices are stored bf16, biases and LayerNorm parameters fp32.
    """

    def __init__(self, cfg: EncoderConfig, state: Dict[str, torch.Tensor], device: torch.device,
                 dtype: torch.dtype = torch.bfloat16):
        if device.type != "cuda":
            raise RuntimeError("EncoderWeights need a HIP device; tensor_truth_amd has no CPU path")
        if dtype not in (torch.bfloat16, torch.float16):
            raise ValueError("EncoderWeights: the 16-bit path computes in bfloat16 or float16")
        self.cfg = cfg
        self.device = device
        self.dtype = dtype           # element type of matrices and activations: bf16, or fp16 (libtt_hip's *_f16 entry points)
        sd = _strip_prefix(state)
        self._keep: List[torch.Tensor] = []
        self._named: Dict[str, torch.Tensor] = {}     # HF name -> the resident tensor (state_dict())

        def mat(name):
            t = sd[name].to(device=device, dtype=dtype).contiguous()
            self._keep.append(t)
            self._named[name] = t
            return t

        def vec(name):
            t = sd[name].to(device=device, dtype=torch.float32).contiguous()
            self._keep.append(t)
            self._named[name] = t
            return t

        H = cfg.hidden
        self.word = mat("embeddings.word_embeddings.weight")
        self.pos = mat("embeddings.position_embeddings.weight")
        self.type = mat("embeddings.token_type_embeddings.weight")
        self.emb_g = vec("embeddings.LayerNorm.weight")
        self.emb_b = vec("embeddings.LayerNorm.bias")
        if self.word.shape != (cfg.vocab_size, H) or self.pos.shape != (cfg.max_pos, H):
            raise ValueError(f"embedding tables {tuple(self.word.shape)} / {tuple(self.pos.shape)} do not match {cfg}")
        self._layers = (_LayerW * max(cfg.layers, 1))()
        self._qkv_w: List[torch.Tensor] = []
        self._ffn1_w: List[torch.Tensor] = []
        self._o_w: List[torch.Tensor] = []
        self._ffn2_w: List[torch.Tensor] = []
        self._fp8: List[torch.Tensor] = []
        self.ffn_act_scales: Optional[List[float]] = None   # per layer, from calibrate_fp8()
        self.gemm_dtype = "bf16"
        for i in range(cfg.layers):
            p = f"encoder.layer.{i}."
            qkv_w = torch.cat([sd[p + f"attention.self.{n}.weight"] for n in ("query", "key", "value")], 0)
            qkv_b = torch.cat([sd[p + f"attention.self.{n}.bias"] for n in ("query", "key", "value")], 0)
            qkv_w = qkv_w.to(device=device, dtype=dtype).contiguous()
            qkv_b = qkv_b.to(device=device, dtype=torch.float32).contiguous()
            self._keep += [qkv_w, qkv_b]
            for j, nm in enumerate(("query", "key", "value")):
                self._named[p + f"attention.self.{nm}.weight"] = qkv_w[j * H:(j + 1) * H]
                self._named[p + f"attention.self.{nm}.bias"] = qkv_b[j * H:(j + 1) * H]
            self._qkv_w.append(qkv_w)
            L = self._layers[i]
            L.qkv_w, L.qkv_b = qkv_w.data_ptr(), qkv_b.data_ptr()
            o_w = mat(p + "attention.output.dense.weight")
            self._o_w.append(o_w)
            L.o_w, L.o_b = o_w.data_ptr(), vec(p + "attention.output.dense.bias").data_ptr()
            L.ln1_g = vec(p + "attention.output.LayerNorm.weight").data_ptr()
            L.ln1_b = vec(p + "attention.output.LayerNorm.bias").data_ptr()
            ffn1_w = mat(p + "intermediate.dense.weight")
            self._ffn1_w.append(ffn1_w)
            L.ffn1_w, L.ffn1_b = ffn1_w.data_ptr(), vec(p + "intermediate.dense.bias").data_ptr()
            ffn2_w = mat(p + "output.dense.weight")
            self._ffn2_w.append(ffn2_w)
            L.ffn2_w, L.ffn2_b = ffn2_w.data_ptr(), vec(p + "output.dense.bias").data_ptr()
            L.ln2_g = vec(p + "output.LayerNorm.weight").data_ptr()
            L.ln2_b = vec(p + "output.LayerNorm.bias").data_ptr()
        w = _EncW()
        w.hidden, w.layers, w.heads, w.ffn = H, cfg.layers, cfg.heads, cfg.ffn
        w.vocab, w.max_pos, w.type_vocab, w.ln_eps = cfg.vocab_size, cfg.max_pos, cfg.type_vocab, cfg.ln_eps
        w.word_emb, w.pos_emb, w.type_emb = self.word.data_ptr(), self.pos.data_ptr(), self.type.data_ptr()
        w.emb_ln_g, w.emb_ln_b = self.emb_g.data_ptr(), self.emb_b.data_ptr()
        w.layer = ctypes.cast(self._layers, POINTER(_LayerW))
        if cfg.num_labels:
            if cfg.num_labels != 1:
                raise ValueError("only single-label (sigmoid) cross-encoder heads are supported")
            w.cls_dense_w = mat("classifier.dense.weight").data_ptr()
            w.cls_dense_b = vec("classifier.dense.bias").data_ptr()
            w.cls_out_w = mat("classifier.out_proj.weight").data_ptr()
            w.cls_out_b = vec("classifier.out_proj.bias").data_ptr()
        self.struct = w

    @staticmethod
    def _quantize_rows(w: torch.Tensor):
        """bf16 [out][in] -> (e4m3 bytes [out][in], fp32 scale [out]) with w ~= w8 * scale[out]."""
        wf = w.to(torch.float32)
        amax = wf.abs().amax(dim=1, keepdim=True)
        inv = torch.where(amax > 0, 448.0 / amax, torch.zeros_like(amax))
        w8 = (wf * inv).to(torch.float8_e4m3fn).view(torch.uint8).contiguous()
        scale = torch.where(amax > 0, amax * (1.0 / 448.0), torch.ones_like(amax)).reshape(-1).contiguous()
        return w8, scale

    def set_gemm_dtype(self, dtype: str) -> None:
        """"bf16" (default) or "fp8": the encoder layers' projections run on OCP e4m3 operands with fp32 accumulation
        (BASELINE.json config 5, "fp8 MFMA reranker").  Weights are quantised here per output channel; activations
        per token inside the LayerNorm kernels (Q/K/V, FFN-up inputs) or by a row pass (attention context); the FFN
        intermediate is written as e4m3 by the FFN-up epilogue with one static scale per layer, which needs
        ``calibrate_fp8`` first -- without it the FFN output projection stays bf16.  Needs hidden and ffn to be
        multiples of 256; the bf16 weights stay resident (CLS tail, small batches)."""
        if dtype not in ("bf16", "fp8"):
            raise ValueError(f"gemm dtype {dtype!r} not in ('bf16', 'fp8')")
        if dtype == "fp8" and self.dtype != torch.bfloat16:
            raise ValueError("fp8 projections exist for bf16 encoders only")
        n = self.cfg.layers
        if dtype == "fp8":
            if self.cfg.hidden % 256 or self.cfg.ffn % 256:
                raise ValueError("fp8 GEMMs need hidden and ffn to be multiples of 256")
            if not self._fp8:
                for i in range(n):
                    for w in (self._qkv_w[i], self._ffn1_w[i], self._o_w[i], self._ffn2_w[i]):
                        self._fp8 += list(self._quantize_rows(w))
            for i in range(n):
                L = self._layers[i]
                q8, qs, f8, fs, o8, os_, d8, ds = self._fp8[8 * i: 8 * i + 8]
                L.qkv_w8, L.qkv_wscale, L.ffn1_w8, L.ffn1_wscale = q8.data_ptr(), qs.data_ptr(), f8.data_ptr(), fs.data_ptr()
                L.o_w8, L.o_wscale, L.ffn2_w8, L.ffn2_wscale = o8.data_ptr(), os_.data_ptr(), d8.data_ptr(), ds.data_ptr()
                L.ffn_act_scale = float(self.ffn_act_scales[i]) if self.ffn_act_scales else 0.0
        else:
            for i in range(n):
                L = self._layers[i]
                L.qkv_w8 = L.qkv_wscale = L.ffn1_w8 = L.ffn1_wscale = None
                L.o_w8 = L.o_wscale = L.ffn2_w8 = L.ffn2_wscale = None
                L.ffn_act_scale = 0.0
        self.gemm_dtype = dtype

    def parameters(self) -> Iterable[torch.Tensor]:
        """For ModelManager-style memory accounting (reference model_manager.py:477-507)."""
        return iter(self._keep + self._fp8)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """HF checkpoint name -> the resident (bf16 / fp32) tensor: what these weights ARE after rounding to bf16,
        e.g. to run the same model through the fp32 path (``encoder_f32.EncoderWeightsF32(cfg, w.state_dict(), dev)``)."""
        return dict(self._named)

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self._keep + self._fp8)


def synthetic_state(cfg: EncoderConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded random-init weights of the given architecture (HF init: normal(0, 0.02);
    LayerNorm gamma/beta perturbed).  Same generator sequence as the test oracle's
    ``synth_weights`` so parity tests can rebuild identical weights from a seed."""
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
    return W


def synthetic_state_device(cfg: EncoderConfig, device: torch.device, seed: int = 0,
                           dtype: torch.dtype = torch.bfloat16) -> Dict[str, torch.Tensor]:
    """Random-init weights generated directly on the device (benchmarks: avoids building a
    2.3 GB fp32 model on the host).  Not reproducible against the CPU oracle.  ``dtype=torch.float32`` keeps the
    matrices unrounded (the reference-precision legs: their lo planes must not be all zero)."""
    g = torch.Generator(device=device).manual_seed(seed)
    n = lambda *s, std=0.02: (torch.randn(*s, generator=g, device=device) * std).to(dtype)  # noqa: E731
    H, F = cfg.hidden, cfg.ffn
    W = {
        "embeddings.word_embeddings.weight": n(cfg.vocab_size, H),
        "embeddings.position_embeddings.weight": n(cfg.max_pos, H),
        "embeddings.token_type_embeddings.weight": n(cfg.type_vocab, H),
        "embeddings.LayerNorm.weight": 1.0 + n(H, std=0.1).float(),
        "embeddings.LayerNorm.bias": n(H, std=0.05).float(),
    }
    for i in range(cfg.layers):
        p = f"encoder.layer.{i}."
        for nm in ("query", "key", "value"):
            W[p + f"attention.self.{nm}.weight"] = n(H, H)
            W[p + f"attention.self.{nm}.bias"] = n(H).float()
        W[p + "attention.output.dense.weight"] = n(H, H)
        W[p + "attention.output.dense.bias"] = n(H).float()
        W[p + "attention.output.LayerNorm.weight"] = 1.0 + n(H, std=0.1).float()
        W[p + "attention.output.LayerNorm.bias"] = n(H, std=0.05).float()
        W[p + "intermediate.dense.weight"] = n(F, H)
        W[p + "intermediate.dense.bias"] = n(F).float()
        W[p + "output.dense.weight"] = n(H, F)
        W[p + "output.dense.bias"] = n(H).float()
        W[p + "output.LayerNorm.weight"] = 1.0 + n(H, std=0.1).float()
        W[p + "output.LayerNorm.bias"] = n(H, std=0.05).float()
    if cfg.num_labels:
        W["classifier.dense.weight"] = n(H, H)
        W["classifier.dense.bias"] = n(H).float()
        W["classifier.out_proj.weight"] = n(cfg.num_labels, H, std=0.2)
        W["classifier.out_proj.bias"] = n(cfg.num_labels).float()
    return W


# Rows between sequence starts are rounded up to this.  8 (default): every sequence owns its V8 token groups, so its
# attention tiles -- and therefore its bits -- do not depend on where it sits in the batch (scores are invariant under
# permuting / splitting a batch: tests/test_configs_gpu.py).  1: back to back, no padding rows at all (the attention
# kernels mask the token group two sequences share): -0.8 % step time on 292-token pairs, more on short texts, at the
# price of one-bf16-ulp differences between placements (a different tiling of the same keys).
_PACK_ALIGN = max(1, int(os.environ.get("TT_PACK_ALIGN", "8")))


@dataclass
class PackedBatch:
    """Varlen packed token batch (host arrays): see include/tt_hip.h 'Token layout'."""

    ids: np.ndarray        # [n_rows] int32
    pos: np.ndarray        # [n_rows] int32
    types: Optional[np.ndarray]
    seq_start: np.ndarray  # [B] int32 (any row is legal for the kernels; multiples of _PACK_ALIGN here)
    seq_len: np.ndarray    # [B] int32
    n_rows: int            # multiple of 128
    max_len: int
    n_tokens: int          # real tokens (sum of seq_len)


def _round_rows(used: int) -> int:
    """Token rows of a batch: a multiple of the 256-row GEMM tile; up to 256 rows (one query, a few short texts) a
    multiple of 64 -- the projections then run as weight-streaming skinny GEMMs (csrc/gemm.hip)."""
    # (TT_GEMM_SKINNY=0: the A/B switch of tools/gpu_skinny.sh -- honoured with the diagnostic library only, like the kernels' own read)
    if used <= 256 and not (_lib.is_diag() and os.environ.get("TT_GEMM_SKINNY", "1") == "0"):
        return max(64, (used + 63) // 64 * 64)
    return (used + 255) // 256 * 256


def pack_tokens(seqs: Sequence[Sequence[int]], cfg: EncoderConfig,
                type_ids: Optional[Sequence[Sequence[int]]] = None, max_len: Optional[int] = None) -> PackedBatch:
    """Pack token-id sequences (already carrying their special tokens) without padding tokens: sequence
    starts are aligned to ``_PACK_ALIGN`` rows (8 by default, see above), the total to the 256-row GEMM tile.  Sequences longer
    than ``max_len`` (default: the model's limit) are truncated on the right, as the
    reference's tokenizer call does (``truncation=True``)."""
    limit = cfg.max_seq_len if max_len is None else min(max_len, cfg.max_seq_len)
    n_seq = len(seqs)
    lens = np.fromiter((min(len(s), limit) for s in seqs), dtype=np.int64, count=n_seq)
    if n_seq == 0 or (lens <= 0).any():
        raise ValueError("empty token sequence")
    # no per-sequence Python work below: one flat copy of all tokens and one scatter (a 100k-document ingest packs ~10^7
    # sequences; the per-sequence slice / arange loop was a third of the host time of an ingest batch)
    aligned = (lens + _PACK_ALIGN - 1) // _PACK_ALIGN * _PACK_ALIGN
    starts = np.zeros(n_seq, dtype=np.int64)
    np.cumsum(aligned[:-1], out=starts[1:])
    n_rows = _round_rows(int(aligned.sum()))
    total = int(lens.sum())

    def flat_of(rows) -> np.ndarray:
        if all(isinstance(r, np.ndarray) for r in rows):
            return np.concatenate([np.asarray(r[:n], dtype=np.int32) for r, n in zip(rows, lens)]) if n_seq > 1 else \
                np.asarray(rows[0][: lens[0]], dtype=np.int32)
        return np.fromiter(itertools.chain.from_iterable(r if len(r) == n else r[:n] for r, n in zip(rows, lens)),
                           dtype=np.int32, count=total)

    first = np.zeros(n_seq, dtype=np.int64)
    np.cumsum(lens[:-1], out=first[1:])
    within = np.arange(total, dtype=np.int64) - np.repeat(first, lens)
    dest = np.repeat(starts, lens) + within
    flat = flat_of(seqs)
    if flat.min() < 0 or flat.max() >= cfg.vocab_size:
        raise ValueError("token id outside the vocabulary")
    ids = np.full(n_rows, cfg.pad_id, dtype=np.int32)
    pos = np.zeros(n_rows, dtype=np.int32)
    pos_off = cfg.pad_id + 1 if cfg.arch == "xlmr" else 0
    ids[dest] = flat
    pos[dest] = within + pos_off
    types = None
    if type_ids is not None:
        types = np.zeros(n_rows, dtype=np.int32)
        types[dest] = flat_of(type_ids)
    return PackedBatch(ids, pos, types, starts.astype(np.int32), lens.astype(np.int32), int(n_rows),
                       int(lens.max()), total)


def pack_flat(flat: np.ndarray, first: np.ndarray, lens: np.ndarray, sel: np.ndarray, cfg: EncoderConfig,
              max_len: Optional[int] = None) -> PackedBatch:
    """``pack_tokens([sequence i for i in sel])`` for sequences given as ONE flat int32 array (sequence i = ``flat[first[i] : first[i] +
    lens[i]]``): no per-sequence Python at all -- the ingest feeder packs ~10^7 sequences per 100 000 documents, and the list of
    per-sequence arrays ``pack_tokens`` takes cost it 40 us per sequence (a quarter of a 100 000-document build, round 5)."""
    limit = cfg.max_seq_len if max_len is None else min(max_len, cfg.max_seq_len)
    sel = np.asarray(sel, dtype=np.int64)
    n_seq = len(sel)
    ln = np.minimum(lens[sel].astype(np.int64), limit)
    if n_seq == 0 or (ln <= 0).any():
        raise ValueError("empty token sequence")
    aligned = (ln + _PACK_ALIGN - 1) // _PACK_ALIGN * _PACK_ALIGN
    starts = np.zeros(n_seq, dtype=np.int64)
    np.cumsum(aligned[:-1], out=starts[1:])
    n_rows = _round_rows(int(aligned.sum()))
    total = int(ln.sum())
    local_first = np.zeros(n_seq, dtype=np.int64)
    np.cumsum(ln[:-1], out=local_first[1:])
    within = np.arange(total, dtype=np.int64) - np.repeat(local_first, ln)
    src = np.repeat(first[sel].astype(np.int64), ln) + within
    dest = np.repeat(starts, ln) + within
    vals = flat[src]
    if vals.min() < 0 or vals.max() >= cfg.vocab_size:
        raise ValueError("token id outside the vocabulary")
    ids = np.full(n_rows, cfg.pad_id, dtype=np.int32)
    pos = np.zeros(n_rows, dtype=np.int32)
    pos_off = cfg.pad_id + 1 if cfg.arch == "xlmr" else 0
    ids[dest] = vals
    pos[dest] = within + pos_off
    return PackedBatch(ids, pos, None, starts.astype(np.int32), ln.astype(np.int32), int(n_rows), int(ln.max()), total)


def pack_token_matrix(ids2d: np.ndarray, cfg: EncoderConfig, type_ids2d: Optional[np.ndarray] = None) -> PackedBatch:
    """Vectorised ``pack_tokens`` for sequences of one common length (rows of ``ids2d``): no Python
    loop, so host packing of a few thousand rerank pairs stays in the 100-microsecond range."""
    ids2d = np.ascontiguousarray(ids2d, dtype=np.int32)
    n, length = ids2d.shape
    if n == 0 or length == 0:
        raise ValueError("empty token matrix")
    if length > cfg.max_seq_len:
        ids2d = ids2d[:, : cfg.max_seq_len]
        length = cfg.max_seq_len
    stride = (length + _PACK_ALIGN - 1) // _PACK_ALIGN * _PACK_ALIGN
    n_rows = _round_rows(n * stride)
    ids = np.full(n_rows, cfg.pad_id, dtype=np.int32)
    pos = np.zeros(n_rows, dtype=np.int32)
    pos_off = cfg.pad_id + 1 if cfg.arch == "xlmr" else 0
    view = ids[: n * stride].reshape(n, stride)
    view[:, :length] = ids2d
    pos[: n * stride].reshape(n, stride)[:, :length] = np.arange(length, dtype=np.int32) + pos_off
    types = None
    if type_ids2d is not None:
        types = np.zeros(n_rows, dtype=np.int32)
        types[: n * stride].reshape(n, stride)[:, :length] = np.asarray(type_ids2d, dtype=np.int32)[:, :length]
    if ids2d.min() < 0 or ids2d.max() >= cfg.vocab_size:
        raise ValueError("token id outside the vocabulary")
    starts = (np.arange(n, dtype=np.int64) * stride).astype(np.int32)
    lens = np.full(n, length, dtype=np.int32)
    return PackedBatch(ids, pos, types, starts, lens, int(n_rows), int(length), int(n * length))


class _Scratch:
    """Workspace buffers keyed by (kind, device, HIP stream): launches on one stream execute in order, so every
    forward enqueued on that stream can reuse ONE buffer -- whichever host thread enqueues it -- as long as a whole
    forward is enqueued atomically (``Encoder._enqueue_lock``; two threads interleaving their launches over one
    workspace would corrupt both).  Bounded by the largest batch per stream, not by the number of request threads
    (per-thread buffers of 5-20 GB each would not fit 32 executor threads).  Growing frees the old buffer through
    torch's stream-ordered caching allocator, which is safe for work already enqueued on the same stream."""

    def __init__(self):
        self.bufs = {}
        self.lock = threading.Lock()

    def get(self, key, device, nbytes):
        stream = torch.cuda.current_stream(device).cuda_stream
        k = (key, device.type, device.index, stream)
        with self.lock:
            buf = self.bufs.get(k)
            if buf is None or buf.numel() < nbytes + 256:
                buf = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
                self.bufs[k] = buf
        base = (buf.data_ptr() + 255) // 256 * 256
        return buf, base


_scratch = _Scratch()
_ENQUEUE_LOCKS: Dict[Tuple[str, Optional[int]], threading.Lock] = {}


class _Stager:
    """Ring of pinned host buffers for the token arrays of a batch, shared by all threads of the process.  One buffer
    holds every array of one batch, goes to the device in ONE asynchronous copy on the compute stream, and is reused
    once the event recorded behind that copy has fired -- so the host can pack and enqueue batch i+1 (and tokenize
    batch i+2) while the GPU is still computing batch i.  Process-wide rather than per
    thread: pinning memory costs milliseconds per allocation, and under the coalescing front every request thread
    leads a batch now and then (32 executor threads x their own rings = 128 pinned allocations on the hot path)."""

    SLOTS = 8

    def __init__(self):
        self.slots = []
        self.cursor = 0
        self.lock = threading.Lock()

    def acquire(self, nbytes: int):
        """-> a slot whose ``lock`` is HELD: the caller fills ``buf``, issues the copy, records ``ev`` and releases it."""
        with self.lock:
            if len(self.slots) < self.SLOTS:
                self.slots.append({"buf": None, "ev": None, "lock": threading.Lock()})
                slot = self.slots[-1]
            else:
                slot = self.slots[self.cursor % self.SLOTS]
            self.cursor += 1
        slot["lock"].acquire()
        if slot["ev"] is not None:
            slot["ev"].synchronize()  # eight batches later: long fired
            slot["ev"] = None
        if slot["buf"] is None or slot["buf"].numel() < nbytes:
            slot["buf"] = torch.empty(max(nbytes, 4 << 20), dtype=torch.uint8, pin_memory=True)
        return slot


_stager = _Stager()


class Encoder:
    """Runs the HIP encoder for one set of weights."""

    def __init__(self, weights: EncoderWeights):
        self.w = weights
        self.cfg = weights.cfg
        self.device = weights.device
        self.lib = _lib.load_library()
        # one forward (scratch lookup + every launch of it) is enqueued atomically: the workspace is shared per stream
        self._enqueue_lock = _ENQUEUE_LOCKS.setdefault((self.device.type, self.device.index), threading.Lock())
        self._sfx = "_f16" if weights.dtype == torch.float16 else ""

    def _fn(self, name: str):
        """The entry point for this encoder's element type: ``name`` (bf16) or its fp16 twin ``name_f16``."""
        return getattr(self.lib, name + self._sfx)

    def _upload(self, batch: PackedBatch):
        """Token arrays of a batch -> device int32 views (ids, pos, types | None, seq_start, seq_len): one pinned
        staging buffer, one asynchronous host-to-device copy on the current stream."""
        parts = [batch.ids, batch.pos, batch.types, batch.seq_start, batch.seq_len]
        offs, total = [], 0
        for a in parts:
            offs.append(total)
            if a is not None:
                total += (a.size + 63) // 64 * 64          # 256-byte aligned sub-arrays
        slot = _stager.acquire(total * 4)
        try:
            host = slot["buf"][: total * 4].view(torch.int32)
            hn = host.numpy()
            for a, o in zip(parts, offs):
                if a is not None:
                    hn[o:o + a.size] = a
            with torch.cuda.device(self.device):
                devbuf = torch.empty(total, dtype=torch.int32, device=self.device)
                devbuf.copy_(host, non_blocking=True)
                slot["ev"] = torch.cuda.Event()
                slot["ev"].record(torch.cuda.current_stream(self.device))
        finally:
            slot["lock"].release()
        return tuple(devbuf[o:o + a.size] if a is not None else None for a, o in zip(parts, offs))

    def forward_packed(self, batch: PackedBatch, want_lens: bool = False):
        """-> (hidden [n_rows, H] bf16, cls_rows [B] int32 device tensor[, seq_len [B] int32 device tensor])."""
        lib, dev, H = self.lib, self.device, self.cfg.hidden
        ids, pos, types, starts, lens = self._upload(batch)
        hidden = torch.empty((batch.n_rows, H), dtype=self.w.dtype, device=dev)
        need = self._fn("tt_encoder_workspace_bytes")(ctypes.byref(self.w.struct), batch.n_rows)
        with self._enqueue_lock, torch.cuda.device(dev):
            ws, base = _scratch.get("enc", dev, need)
            rc = self._fn("tt_encoder_forward")(ctypes.byref(self.w.struct), ids.data_ptr(), pos.data_ptr(),
                                        types.data_ptr() if types is not None else None, starts.data_ptr(),
                                        lens.data_ptr(), len(batch.seq_len), batch.n_rows, batch.max_len,
                                        hidden.data_ptr(), base, need, torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, "tt_encoder_forward")
        return (hidden, starts, lens) if want_lens else (hidden, starts)

    def calibrate_fp8(self, batch: PackedBatch, margin: float = 2.0) -> List[float]:
        """One bf16 forward over ``batch`` that records max |GELU output| per layer (``ffn_absmax_out`` hook) and sets
        the static e4m3 scales of the FFN intermediate, ``margin * max / 448`` (values beyond saturate).  Call before
        ``weights.set_gemm_dtype("fp8")`` to move the FFN output projection to fp8 as well."""
        w = self.w
        prev = w.gemm_dtype
        w.set_gemm_dtype("bf16")
        absmax = torch.zeros(max(self.cfg.layers, 1), dtype=torch.float32, device=self.device)
        w.struct.ffn_absmax_out = absmax.data_ptr()
        try:
            self.forward_packed(batch)
            vals = absmax.cpu().tolist()
        finally:
            w.struct.ffn_absmax_out = None
        w.ffn_act_scales = [max(v, 1e-6) * margin / 448.0 for v in vals[: self.cfg.layers]]
        w.set_gemm_dtype(prev)
        return w.ffn_act_scales

    def cls_hidden_packed(self, batch: PackedBatch) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (final hidden state of every sequence's CLS token [round_up(B,256), H] bf16, row ids [B] int32).
        The last layer is evaluated for the CLS rows only (``tt_encoder_forward_cls``)."""
        lib, dev, H = self.lib, self.device, self.cfg.hidden
        if self.cfg.layers == 0:
            hidden, starts = self.forward_packed(batch)
            return hidden, starts
        B = len(batch.seq_len)
        ids, pos, types, starts, lens = self._upload(batch)
        b_pad = (B + 255) // 256 * 256
        cls = torch.empty((b_pad, H), dtype=self.w.dtype, device=dev)
        need = self._fn("tt_encoder_cls_workspace_bytes")(ctypes.byref(self.w.struct), batch.n_rows, B)
        with self._enqueue_lock, torch.cuda.device(dev):
            ws, base = _scratch.get("enc", dev, need)
            rc = self._fn("tt_encoder_forward_cls")(ctypes.byref(self.w.struct), ids.data_ptr(), pos.data_ptr(),
                                            types.data_ptr() if types is not None else None, starts.data_ptr(),
                                            lens.data_ptr(), B, batch.n_rows, batch.max_len, cls.data_ptr(), base, need,
                                            torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, "tt_encoder_forward_cls")
        rows = torch.arange(B, dtype=torch.int32, device=dev)
        return cls, rows

    def embed_packed(self, batch: PackedBatch, pooling: str = "cls") -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (embeddings fp32 [B, H] L2-normalised, same rounded to bf16).  ``pooling``: "cls" (the BGE family: the last layer
        runs for the CLS rows only) or "mean" (sentence-transformers mean pooling over a sequence's tokens: full last layer)."""
        B, H = len(batch.seq_len), self.cfg.hidden
        out = torch.empty((B, H), dtype=torch.float32, device=self.device)
        # the 16-bit copy of an embedding is a SCAN QUERY, i.e. bf16 like the corpus: the bf16 kernels write it themselves, in
        # the fp16 mode it is rounded from the fp32 vector here (round to nearest even either way)
        f16 = self.w.dtype == torch.float16
        out16 = None if f16 else torch.empty((B, H), dtype=torch.bfloat16, device=self.device)
        o16 = None if f16 else out16.data_ptr()
        if pooling == "mean":
            hidden, starts, lens = self.forward_packed(batch, want_lens=True)
            with torch.cuda.device(self.device):
                rc = self._fn("tt_embed_pool_mean")(hidden.data_ptr(), H, starts.data_ptr(), lens.data_ptr(), B, H, out.data_ptr(),
                                                    o16, torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(rc, "tt_embed_pool_mean")
            return out, (out.to(torch.bfloat16) if f16 else out16)
        if pooling != "cls":
            raise ValueError(f"pooling '{pooling}' (supported: 'cls', 'mean')")
        hidden, cls_rows = self.cls_hidden_packed(batch)
        with torch.cuda.device(self.device):
            rc = self._fn("tt_embed_pool")(hidden.data_ptr(), H, cls_rows.data_ptr(), B, H, out.data_ptr(), o16,
                                           torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(rc, "tt_embed_pool")
        return out, (out.to(torch.bfloat16) if f16 else out16)

    def rerank_packed(self, batch: PackedBatch, want_logits: bool = False):
        """-> sigmoid scores fp32 [B] (and logits)."""
        if not self.cfg.num_labels:
            raise RuntimeError("these weights carry no classification head")
        hidden, cls_rows = self.cls_hidden_packed(batch)
        B, H = len(batch.seq_len), self.cfg.hidden
        scores = torch.empty(B, dtype=torch.float32, device=self.device)
        logits = torch.empty(B, dtype=torch.float32, device=self.device) if want_logits else None
        n_pad = (B + 127) // 128 * 128
        need = 2 * ((n_pad * H * 2 + 255) // 256 * 256)
        with self._enqueue_lock, torch.cuda.device(self.device):
            ws, base = _scratch.get("head", self.device, need)
            rc = self._fn("tt_rerank_head")(ctypes.byref(self.w.struct), hidden.data_ptr(), cls_rows.data_ptr(), B,
                                         scores.data_ptr(), logits.data_ptr() if want_logits else None, base, need,
                                         torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(rc, "tt_rerank_head")
        return (scores, logits) if want_logits else scores

    # -- convenience over python lists -------------------------------------------------------
    def embed(self, seqs, type_ids=None, max_len=None):
        return self.embed_packed(pack_tokens(seqs, self.cfg, type_ids, max_len))

    def rerank(self, seqs, max_len: Optional[int] = 512, want_logits: bool = False):
        return self.rerank_packed(pack_tokens(seqs, self.cfg, None, max_len), want_logits)
