"""Reference precision on two matrix-time units: host side of ``tt_encoder_forward_f16c`` (csrc/f16c_path.hip, round 4).

The reference builds its embedder and reranker without a dtype (``services/model_manager.py:218-229,333-337``;
``app_utils/config_schema.py:66-76``: ``torch_dtype: None``), i.e. in fp32, and north_star's score tolerance (1e-3
relative) is an fp32 tolerance.  ``encoder_x3`` (split planes, three products) meets it at a third of the bf16 matrix rate -- on
ordinary AND hostile weights -- and is what the default mode runs; this path is the FAST VARIANT at half the rate
(``TT_REFERENCE_IMPL=f16c``; 9e-5 relative on ordinary weights, 7e-3 on the stress fixture: DESIGN.md section 4.6): every GEMM operand is carried as "c-planes" -- ``hi = fp16(x)`` plus two OCP e4m3 planes (``x`` and ``x - hi``)
with one E8M0 block exponent per 32 elements -- and a product runs as ``hi.hi`` on the fp16 matrix cores plus two
block-scaled e4m3 cross terms at twice the rate (``a.w ~= a_hi.w_hi + e4m3(a).e4m3(w_lo) + e4m3(a_lo).e4m3(w)``; the cross
terms are 2^-12 of the result, the dropped term 2^-24).  Attention on single fp16 products with fp32 softmax; the residual
stream, LayerNorm, exact-erf GELU and the classification head stay fp32.  ``precision.reference_impl()`` selects it only on
request, and only where the model shape fits (hidden a multiple of 256 with 64-wide heads: bge-m3, bge-reranker-v2-m3,
bge-reranker-base); the default is ``f16x3`` (``encoder_x3`` on fp16 planes), ``bf16x3`` / ``fp32`` the older implementations.
Same token packing and surface as ``encoder.Encoder``.
"""
from __future__ import annotations

import ctypes
import dataclasses
import threading
from ctypes import POINTER, Structure, c_float, c_int32, c_void_p
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .encoder import EncoderConfig, PackedBatch, _ENQUEUE_LOCKS, _scratch, _strip_prefix, pack_tokens


class _LayerWC(Structure):
    _fields_ = [(n, c_void_p) for n in ("qkv_w", "qkv_s", "qkv_b", "o_w", "o_s", "o_b", "ln1_g", "ln1_b", "ffn1_w", "ffn1_s",
                                        "ffn1_b", "ffn2_w", "ffn2_s", "ffn2_b", "ln2_g", "ln2_b")]


class _EncWC(Structure):
    _fields_ = [
        ("hidden", c_int32), ("layers", c_int32), ("heads", c_int32), ("ffn", c_int32), ("vocab", c_int32),
        ("max_pos", c_int32), ("type_vocab", c_int32), ("ln_eps", c_float),
        ("word_emb", c_void_p), ("pos_emb", c_void_p), ("type_emb", c_void_p), ("emb_ln_g", c_void_p),
        ("emb_ln_b", c_void_p), ("layer", POINTER(_LayerWC)),
        ("cls_dense_w", c_void_p), ("cls_dense_b", c_void_p), ("cls_out_w", c_void_p), ("cls_out_b", c_void_p),
    ]


def supports(cfg: EncoderConfig) -> bool:
    """Shapes the f16c kernels take (csrc/f16c_path.hip check_weights_c)."""
    return (cfg.hidden % 256 == 0 and cfg.hidden <= 1024 and cfg.hidden == cfg.heads * 64 and cfg.ffn % 256 == 0
            and cfg.layers > 0)


def quantize_planes(x: torch.Tensor, weight: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """fp32 [rows][k] on the device -> (c-planes uint8 [rows][4 k], tiled E8M0 scales uint8) through ``tt_f16c_quantize``."""
    lib = _lib.load_library()
    if x.device.type != "cuda" or x.dim() != 2 or x.shape[1] % 256:
        raise ValueError("quantize_planes takes a device fp32 matrix whose width is a multiple of 256")
    x = x.to(torch.float32).contiguous()
    rows, k = x.shape
    planes = torch.empty((rows, 4 * k), dtype=torch.uint8, device=x.device)
    scales = torch.zeros(int(lib.tt_f16c_scale_bytes(rows, k, 1 if weight else 0)), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.tt_f16c_quantize(x.data_ptr(), rows, k, 1 if weight else 0, planes.data_ptr(), scales.data_ptr(),
                                  torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(rc, "tt_f16c_quantize")
    return planes, scales


class EncoderWeightsF16C:
    """Device-resident f16c weights (HF checkpoint names, see ``encoder.EncoderWeights``): matrices as c-planes (4 bytes per
    element + 1/16 byte of block scales: 2 x 1.13 GB for the 1024-wide models), tables / biases / LayerNorm / head in fp32."""

    gemm_dtype = "f16c"

    def __init__(self, cfg: EncoderConfig, state: Dict[str, torch.Tensor], device: torch.device):
        if device.type != "cuda":
            raise RuntimeError("EncoderWeightsF16C need a HIP device; tensor_truth_amd has no CPU path")
        if not supports(cfg):
            raise ValueError(f"the f16c path takes hidden % 256 == 0 with 64-wide heads and ffn % 256 == 0, not {cfg}")
        self.cfg, self.device = cfg, device
        sd = _strip_prefix(state)
        self._keep: List[torch.Tensor] = []

        def t(x):
            x = x.to(device=device, dtype=torch.float32).contiguous()
            self._keep.append(x)
            return x

        def planes(x):
            p, s = quantize_planes(x.to(device=device, dtype=torch.float32), weight=True)
            self._keep += [p, s]
            return p.data_ptr(), s.data_ptr()

        H = cfg.hidden
        word, pos, typ = t(sd["embeddings.word_embeddings.weight"]), t(sd["embeddings.position_embeddings.weight"]), \
            t(sd["embeddings.token_type_embeddings.weight"])
        if word.shape != (cfg.vocab_size, H) or pos.shape != (cfg.max_pos, H):
            raise ValueError(f"embedding tables {tuple(word.shape)} / {tuple(pos.shape)} do not match {cfg}")
        self._layers = (_LayerWC * max(cfg.layers, 1))()
        for i in range(cfg.layers):
            p = f"encoder.layer.{i}."
            L = self._layers[i]
            L.qkv_w, L.qkv_s = planes(torch.cat([sd[p + f"attention.self.{n}.weight"] for n in ("query", "key", "value")], 0))
            L.qkv_b = t(torch.cat([sd[p + f"attention.self.{n}.bias"] for n in ("query", "key", "value")], 0)).data_ptr()
            L.o_w, L.o_s = planes(sd[p + "attention.output.dense.weight"])
            L.o_b = t(sd[p + "attention.output.dense.bias"]).data_ptr()
            L.ln1_g = t(sd[p + "attention.output.LayerNorm.weight"]).data_ptr()
            L.ln1_b = t(sd[p + "attention.output.LayerNorm.bias"]).data_ptr()
            L.ffn1_w, L.ffn1_s = planes(sd[p + "intermediate.dense.weight"])
            L.ffn1_b = t(sd[p + "intermediate.dense.bias"]).data_ptr()
            L.ffn2_w, L.ffn2_s = planes(sd[p + "output.dense.weight"])
            L.ffn2_b = t(sd[p + "output.dense.bias"]).data_ptr()
            L.ln2_g, L.ln2_b = t(sd[p + "output.LayerNorm.weight"]).data_ptr(), t(sd[p + "output.LayerNorm.bias"]).data_ptr()
        torch.cuda.synchronize(device)          # (the fp32 sources of the planes may be freed from here on)
        w = _EncWC()
        w.hidden, w.layers, w.heads, w.ffn = H, cfg.layers, cfg.heads, cfg.ffn
        w.vocab, w.max_pos, w.type_vocab, w.ln_eps = cfg.vocab_size, cfg.max_pos, cfg.type_vocab, cfg.ln_eps
        w.word_emb, w.pos_emb, w.type_emb = word.data_ptr(), pos.data_ptr(), typ.data_ptr()
        w.emb_ln_g, w.emb_ln_b = t(sd["embeddings.LayerNorm.weight"]).data_ptr(), t(sd["embeddings.LayerNorm.bias"]).data_ptr()
        w.layer = ctypes.cast(self._layers, POINTER(_LayerWC))
        if cfg.num_labels:
            if cfg.num_labels != 1:
                raise ValueError("only single-label (sigmoid) cross-encoder heads are supported")
            w.cls_dense_w, w.cls_dense_b = t(sd["classifier.dense.weight"]).data_ptr(), t(sd["classifier.dense.bias"]).data_ptr()
            w.cls_out_w, w.cls_out_b = t(sd["classifier.out_proj.weight"]).data_ptr(), t(sd["classifier.out_proj.bias"]).data_ptr()
        self.struct = w

    def set_gemm_dtype(self, dtype: str) -> None:
        if dtype not in ("f16c", "reference", "float32", "fp32"):
            raise ValueError(f"f16c weights run in reference precision only (asked for {dtype!r})")

    def parameters(self) -> Iterable[torch.Tensor]:
        return iter(self._keep)

    def nbytes(self) -> int:
        return sum(x.numel() * x.element_size() for x in self._keep)


def _pad_rows(batch: PackedBatch, multiple: int = 256) -> PackedBatch:
    """The f16c GEMMs run whole 256-row tiles (no skinny kernel yet: one query's 64 rows are padded to one tile)."""
    n = (batch.n_rows + multiple - 1) // multiple * multiple
    if n == batch.n_rows:
        return batch

    def pad(a, fill):
        if a is None:
            return None
        out = np.full(n, fill, dtype=a.dtype)
        out[: a.size] = a
        return out

    return dataclasses.replace(batch, ids=pad(batch.ids, batch.ids[-1] if batch.ids.size else 0), pos=pad(batch.pos, 0),
                               types=pad(batch.types, 0), n_rows=n)


class EncoderF16C:
    """``encoder.Encoder``'s interface on the f16c forward."""

    def __init__(self, weights: EncoderWeightsF16C):
        self.w, self.cfg, self.device = weights, weights.cfg, weights.device
        self.lib = _lib.load_library()
        self._enqueue_lock = _ENQUEUE_LOCKS.setdefault((self.device.type, self.device.index), threading.Lock())

    def _upload(self, batch: PackedBatch):
        from .encoder import Encoder

        return Encoder._upload(self, batch)       # same pinned staging ring, one async copy

    def forward_packed(self, batch: PackedBatch, want_lens: bool = False):
        """-> (hidden [n_rows, H] fp32, seq_start [B] int32 device tensor[, seq_len [B] int32 device tensor])."""
        lib, dev, H = self.lib, self.device, self.cfg.hidden
        batch = _pad_rows(batch)
        ids, pos, types, starts, lens = self._upload(batch)
        hidden = torch.empty((batch.n_rows, H), dtype=torch.float32, device=dev)
        need = lib.tt_encoder_f16c_workspace_bytes(ctypes.byref(self.w.struct), batch.n_rows)
        with self._enqueue_lock, torch.cuda.device(dev):
            ws, base = _scratch.get("encf16c", dev, need)
            rc = lib.tt_encoder_forward_f16c(ctypes.byref(self.w.struct), ids.data_ptr(), pos.data_ptr(),
                                             types.data_ptr() if types is not None else None, starts.data_ptr(),
                                             lens.data_ptr(), len(batch.seq_len), batch.n_rows, batch.max_len,
                                             hidden.data_ptr(), base, need, torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, "tt_encoder_forward_f16c")
        return (hidden, starts, lens) if want_lens else (hidden, starts)

    def cls_hidden_packed(self, batch: PackedBatch) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (final hidden state of every sequence's first row [pad(B), H] fp32, row ids [B] int32): the last layer runs for
        those rows only (``tt_encoder_forward_f16c_cls``), as ``Encoder.cls_hidden_packed`` does in the 16-bit modes."""
        lib, dev, H = self.lib, self.device, self.cfg.hidden
        B = len(batch.seq_len)
        batch = _pad_rows(batch)
        ids, pos, types, starts, lens = self._upload(batch)
        b_pad = (B + 255) // 256 * 256
        cls = torch.empty((b_pad, H), dtype=torch.float32, device=dev)
        need = lib.tt_encoder_f16c_cls_workspace_bytes(ctypes.byref(self.w.struct), batch.n_rows, B)
        with self._enqueue_lock, torch.cuda.device(dev):
            ws, base = _scratch.get("encf16c", dev, need)
            rc = lib.tt_encoder_forward_f16c_cls(ctypes.byref(self.w.struct), ids.data_ptr(), pos.data_ptr(),
                                                 types.data_ptr() if types is not None else None, starts.data_ptr(),
                                                 lens.data_ptr(), B, batch.n_rows, batch.max_len, cls.data_ptr(), base, need,
                                                 torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, "tt_encoder_forward_f16c_cls")
        return cls, torch.arange(B, dtype=torch.int32, device=dev)

    def embed_packed(self, batch: PackedBatch, pooling: str = "cls") -> Tuple[torch.Tensor, torch.Tensor]:
        B, H = len(batch.seq_len), self.cfg.hidden
        out = torch.empty((B, H), dtype=torch.float32, device=self.device)
        out16 = torch.empty((B, H), dtype=torch.bfloat16, device=self.device)
        if pooling == "mean":
            hidden, starts, lens = self.forward_packed(batch, want_lens=True)
            with torch.cuda.device(self.device):
                rc = self.lib.tt_embed_pool_mean_f32(hidden.data_ptr(), H, starts.data_ptr(), lens.data_ptr(), B, H, out.data_ptr(),
                                                     out16.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(rc, "tt_embed_pool_mean_f32")
            return out, out16
        if pooling != "cls":
            raise ValueError(f"pooling '{pooling}' (supported: 'cls', 'mean')")
        hidden, rows = self.cls_hidden_packed(batch)
        with torch.cuda.device(self.device):
            rc = self.lib.tt_embed_pool_f32(hidden.data_ptr(), H, rows.data_ptr(), B, H, out.data_ptr(), out16.data_ptr(),
                                            torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(rc, "tt_embed_pool_f32")
        return out, out16

    def rerank_packed(self, batch: PackedBatch, want_logits: bool = False):
        if not self.cfg.num_labels:
            raise RuntimeError("these weights carry no classification head")
        hidden, rows = self.cls_hidden_packed(batch)
        B, H = len(batch.seq_len), self.cfg.hidden
        scores = torch.empty(B, dtype=torch.float32, device=self.device)
        logits = torch.empty(B, dtype=torch.float32, device=self.device) if want_logits else None
        n_pad = (B + 127) // 128 * 128
        need = 2 * ((n_pad * H * 4 + 255) // 256 * 256)
        with self._enqueue_lock, torch.cuda.device(self.device):
            ws, base = _scratch.get("headf16c", self.device, need)
            rc = self.lib.tt_rerank_head_f16c(ctypes.byref(self.w.struct), hidden.data_ptr(), rows.data_ptr(), B,
                                              scores.data_ptr(), logits.data_ptr() if want_logits else None, base, need,
                                              torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(rc, "tt_rerank_head_f16c")
        return (scores, logits) if want_logits else scores

    def calibrate_fp8(self, *_a, **_k):
        raise RuntimeError("fp8 calibration does not apply to the reference-precision path")

    def embed(self, seqs, type_ids=None, max_len=None):
        return self.embed_packed(pack_tokens(seqs, self.cfg, type_ids, max_len))

    def rerank(self, seqs, max_len: Optional[int] = 512, want_logits: bool = False):
        return self.rerank_packed(pack_tokens(seqs, self.cfg, None, max_len), want_logits)
