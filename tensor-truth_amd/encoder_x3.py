"""Reference precision at matrix-core speed: host side of ``tt_encoder_forward_x3`` (csrc/x3_path.hip, "bf16x3").

The reference builds its embedder and reranker without a dtype (``services/model_manager.py:218-229,333-337``;
``app_utils/config_schema.py:66-76``: ``torch_dtype: None``), i.e. in fp32, and north_star's score tolerance (1e-3
relative) is an fp32 tolerance.  The fp32-MFMA path (``encoder_f32.py``) meets it at 1/16 of the bf16 matrix rate; this
one meets it at 1/3: every matrix product runs on the bf16 matrix cores with both operands split into two bf16 planes
(``x = hi + lo``; ``a.b ~= a_hi.b_hi + a_hi.b_lo + a_lo.b_hi``, fp32 accumulate), everything else (residual stream,
LayerNorm, softmax, exact-erf GELU, classification head) stays fp32.  Selected by ``precision.resolve()`` -- the process
setting ``TT_PRECISION=reference``, ``ModelManager``'s ``precision`` config key, or ``model_kwargs={"torch_dtype":
"float32"}`` -- whenever the model shape fits (hidden a multiple of 256 with 64-wide heads; bge-m3 and
bge-reranker-v2-m3 do, bge-small falls back to ``encoder_f32``).  Same token packing and surface as ``encoder.Encoder``.
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


class _LayerWX(Structure):
    _fields_ = [(n, c_void_p) for n in ("qkv_w", "qkv_b", "o_w", "o_b", "ln1_g", "ln1_b", "ffn1_w", "ffn1_b",
                                        "ffn2_w", "ffn2_b", "ln2_g", "ln2_b")]


class _EncWX(Structure):
    _fields_ = [
        ("hidden", c_int32), ("layers", c_int32), ("heads", c_int32), ("ffn", c_int32), ("vocab", c_int32),
        ("max_pos", c_int32), ("type_vocab", c_int32), ("ln_eps", c_float),
        ("word_emb", c_void_p), ("pos_emb", c_void_p), ("type_emb", c_void_p), ("emb_ln_g", c_void_p),
        ("emb_ln_b", c_void_p), ("layer", POINTER(_LayerWX)),
        ("cls_dense_w", c_void_p), ("cls_dense_b", c_void_p), ("cls_out_w", c_void_p), ("cls_out_b", c_void_p),
    ]


def supports(cfg: EncoderConfig) -> bool:
    """Shapes the split-plane kernels take (csrc/x3_path.hip check_weights_x3).  Round 6: hidden a multiple of 128 with 64- or
    32-wide heads -- the 384-wide BERT models the reference names (``BAAI/bge-small-en-v1.5``, BASELINE config 1;
    ``cross-encoder/ms-marco-MiniLM-L-6-v2``, app_utils/config_schema.py:83-87) no longer fall to the fp32 MFMA."""
    return (cfg.hidden % 128 == 0 and cfg.hidden <= 1024 and cfg.hidden in (cfg.heads * 64, cfg.heads * 32) and cfg.ffn % 64 == 0
            and cfg.layers > 0)


def split_planes(w: torch.Tensor, dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    """fp32 [out][in] -> planes [out][2 in] of ``dtype``: hi = dtype(w), lo = dtype(w - hi) side by side (``GemmParams.x3``).
    bfloat16: "bf16x3" (16 significand bits per operand); float16: "f16x3" (22; values beyond +-65504 saturate in hi)."""
    w = w.to(torch.float32)
    if dtype == torch.float16:
        hi = w.clamp(-65504.0, 65504.0).to(torch.float16)
        lo = (w - hi.to(torch.float32)).clamp(-65504.0, 65504.0).to(torch.float16)
    else:
        hi = w.to(torch.bfloat16)
        lo = (w - hi.to(torch.float32)).to(torch.bfloat16)
    return torch.cat([hi, lo], dim=1).contiguous()


class EncoderWeightsX3:
    """Device-resident split-bf16 weights (HF checkpoint names, see ``encoder.EncoderWeights``): matrices as two bf16
    planes (2 x 1.13 GB for the 1024-wide models), tables / biases / LayerNorm / head in fp32."""

    gemm_dtype = "bf16x3"

    def __init__(self, cfg: EncoderConfig, state: Dict[str, torch.Tensor], device: torch.device, round_weights: bool = False,
                 dtype: torch.dtype = torch.bfloat16):
        """``dtype``: the planes' element type -- bfloat16 ("bf16x3", round 3) or float16 ("f16x3", round 4: the default
        implementation of the reference precision; ``EncoderX3`` then calls the library's ``*_f16`` entry points).
        ``round_weights`` (diagnostic, tools/probes/bf16_error_budget.py): every tensor the bf16 path keeps in bf16 -- the
        matrices and the embedding tables -- is rounded to bf16 first, i.e. the weights' share of the bf16 mode's error."""
        if dtype not in (torch.bfloat16, torch.float16):
            raise ValueError("split planes are bfloat16 or float16")
        self.dtype = dtype
        self.gemm_dtype = "f16x3" if dtype == torch.float16 else "bf16x3"
        if device.type != "cuda":
            raise RuntimeError("EncoderWeightsX3 need a HIP device; tensor_truth_amd has no CPU path")
        if not supports(cfg):
            raise ValueError(f"the split-plane path takes hidden % 128 == 0 with 64- or 32-wide heads and ffn % 64 == 0, not {cfg}")
        self.cfg, self.device = cfg, device
        sd = _strip_prefix(state)
        self._keep: List[torch.Tensor] = []

        def t(x):
            x = x.to(device=device, dtype=torch.float32).contiguous()
            self._keep.append(x)
            return x

        def planes(x):
            if round_weights:
                x = x.to(torch.bfloat16)
            x = split_planes(x.to(device=device), dtype)
            self._keep.append(x)
            return x

        H = cfg.hidden
        rt = (lambda x: t(x.to(torch.bfloat16))) if round_weights else t
        word, pos, typ = rt(sd["embeddings.word_embeddings.weight"]), rt(sd["embeddings.position_embeddings.weight"]), \
            rt(sd["embeddings.token_type_embeddings.weight"])
        if word.shape != (cfg.vocab_size, H) or pos.shape != (cfg.max_pos, H):
            raise ValueError(f"embedding tables {tuple(word.shape)} / {tuple(pos.shape)} do not match {cfg}")
        self._layers = (_LayerWX * max(cfg.layers, 1))()
        for i in range(cfg.layers):
            p = f"encoder.layer.{i}."
            L = self._layers[i]
            L.qkv_w = planes(torch.cat([sd[p + f"attention.self.{n}.weight"] for n in ("query", "key", "value")], 0)).data_ptr()
            L.qkv_b = t(torch.cat([sd[p + f"attention.self.{n}.bias"] for n in ("query", "key", "value")], 0)).data_ptr()
            L.o_w, L.o_b = planes(sd[p + "attention.output.dense.weight"]).data_ptr(), t(sd[p + "attention.output.dense.bias"]).data_ptr()
            L.ln1_g = t(sd[p + "attention.output.LayerNorm.weight"]).data_ptr()
            L.ln1_b = t(sd[p + "attention.output.LayerNorm.bias"]).data_ptr()
            L.ffn1_w, L.ffn1_b = planes(sd[p + "intermediate.dense.weight"]).data_ptr(), t(sd[p + "intermediate.dense.bias"]).data_ptr()
            L.ffn2_w, L.ffn2_b = planes(sd[p + "output.dense.weight"]).data_ptr(), t(sd[p + "output.dense.bias"]).data_ptr()
            L.ln2_g, L.ln2_b = t(sd[p + "output.LayerNorm.weight"]).data_ptr(), t(sd[p + "output.LayerNorm.bias"]).data_ptr()
        w = _EncWX()
        w.hidden, w.layers, w.heads, w.ffn = H, cfg.layers, cfg.heads, cfg.ffn
        w.vocab, w.max_pos, w.type_vocab, w.ln_eps = cfg.vocab_size, cfg.max_pos, cfg.type_vocab, cfg.ln_eps
        w.word_emb, w.pos_emb, w.type_emb = word.data_ptr(), pos.data_ptr(), typ.data_ptr()
        w.emb_ln_g, w.emb_ln_b = t(sd["embeddings.LayerNorm.weight"]).data_ptr(), t(sd["embeddings.LayerNorm.bias"]).data_ptr()
        w.layer = ctypes.cast(self._layers, POINTER(_LayerWX))
        if cfg.num_labels:
            if cfg.num_labels != 1:
                raise ValueError("only single-label (sigmoid) cross-encoder heads are supported")
            w.cls_dense_w, w.cls_dense_b = rt(sd["classifier.dense.weight"]).data_ptr(), t(sd["classifier.dense.bias"]).data_ptr()
            w.cls_out_w, w.cls_out_b = rt(sd["classifier.out_proj.weight"]).data_ptr(), t(sd["classifier.out_proj.bias"]).data_ptr()
        self.struct = w

    def set_gemm_dtype(self, dtype: str) -> None:
        if dtype not in ("bf16x3", "f16x3", "reference", "float32", "fp32"):
            raise ValueError(f"split-bf16 weights run in reference precision only (asked for {dtype!r})")

    def parameters(self) -> Iterable[torch.Tensor]:
        return iter(self._keep)

    def nbytes(self) -> int:
        return sum(x.numel() * x.element_size() for x in self._keep)


def _pad_rows(batch: PackedBatch, multiple: int = 256) -> PackedBatch:
    """The tiled split-bf16 GEMMs run whole 256-row tiles; up to 256 rows (one query: 64 rows) the projections run as
    weight-streaming skinny GEMMs on multiples of 64 rows (``encoder._round_rows`` packs exactly that)."""
    if batch.n_rows <= 256 and batch.n_rows % 64 == 0:
        return batch
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


class EncoderX3:
    """``encoder.Encoder``'s interface on the split-bf16 forward."""

    def __init__(self, weights: EncoderWeightsX3):
        self.w, self.cfg, self.device = weights, weights.cfg, weights.device
        self.lib = _lib.load_library()
        self._sfx = "_f16" if getattr(weights, "dtype", torch.bfloat16) == torch.float16 else ""      # fp16 planes: the second instantiation
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
        need = getattr(lib, "tt_encoder_x3_workspace_bytes" + self._sfx)(ctypes.byref(self.w.struct), batch.n_rows)
        with self._enqueue_lock, torch.cuda.device(dev):
            ws, base = _scratch.get("encx3", dev, need)
            rc = getattr(lib, "tt_encoder_forward_x3" + self._sfx)(ctypes.byref(self.w.struct), ids.data_ptr(), pos.data_ptr(),
                                           types.data_ptr() if types is not None else None, starts.data_ptr(),
                                           lens.data_ptr(), len(batch.seq_len), batch.n_rows, batch.max_len,
                                           hidden.data_ptr(), base, need, torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, "tt_encoder_forward_x3")
        return (hidden, starts, lens) if want_lens else (hidden, starts)

    def cls_hidden_packed(self, batch: PackedBatch) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (final hidden state of every sequence's first row [pad(B), H] fp32, row ids [B] int32): the last layer runs for
        those rows only (``tt_encoder_forward_x3_cls``), as ``Encoder.cls_hidden_packed`` does in the 16-bit modes."""
        lib, dev, H = self.lib, self.device, self.cfg.hidden
        B = len(batch.seq_len)
        batch = _pad_rows(batch)
        ids, pos, types, starts, lens = self._upload(batch)
        b_pad = (B + 63) // 64 * 64 if B <= 256 else (B + 255) // 256 * 256
        cls = torch.empty((b_pad, H), dtype=torch.float32, device=dev)
        need = getattr(lib, "tt_encoder_x3_cls_workspace_bytes" + self._sfx)(ctypes.byref(self.w.struct), batch.n_rows, B)
        with self._enqueue_lock, torch.cuda.device(dev):
            ws, base = _scratch.get("encx3", dev, need)
            rc = getattr(lib, "tt_encoder_forward_x3_cls" + self._sfx)(ctypes.byref(self.w.struct), ids.data_ptr(), pos.data_ptr(),
                                               types.data_ptr() if types is not None else None, starts.data_ptr(),
                                               lens.data_ptr(), B, batch.n_rows, batch.max_len, cls.data_ptr(), base, need,
                                               torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, "tt_encoder_forward_x3_cls")
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
            ws, base = _scratch.get("headx3", self.device, need)
            rc = getattr(self.lib, "tt_rerank_head_x3" + self._sfx)(ctypes.byref(self.w.struct), hidden.data_ptr(), rows.data_ptr(), B,
                                            scores.data_ptr(), logits.data_ptr() if want_logits else None, base, need,
                                            torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(rc, "tt_rerank_head_x3")
        return (scores, logits) if want_logits else scores

    def calibrate_fp8(self, *_a, **_k):
        raise RuntimeError("fp8 calibration does not apply to the reference-precision path")

    def embed(self, seqs, type_ids=None, max_len=None):
        return self.embed_packed(pack_tokens(seqs, self.cfg, type_ids, max_len))

    def rerank(self, seqs, max_len: Optional[int] = 512, want_logits: bool = False):
        return self.rerank_packed(pack_tokens(seqs, self.cfg, None, max_len), want_logits)
