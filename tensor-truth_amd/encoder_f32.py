"""Reference-precision (fp32) encoder: host side of ``tt_encoder_forward_f32`` (csrc/f32_path.hip).

The reference's embedder / reranker run in fp32 unless configured otherwise
(``src/tensortruth/app_utils/config_schema.py:66-76``: ``torch_dtype: None``;
``services/model_manager.py:218-229``).  ``model_kwargs={"torch_dtype": "float32"}`` on the HIP embedding model or
rerank postprocessor selects this path: fp32 weights and activations, fp32 MFMA -- scores within 1e-3 relative of the
CPU reference (north_star's tolerance), at about 1/10 of the bf16 path's throughput: meant for the interactive case
(one query's candidate pairs), not for bulk ingest.  Same token packing, same surface as ``encoder.Encoder``.
"""
from __future__ import annotations

import ctypes
import threading
from ctypes import POINTER, Structure, c_float, c_int32, c_void_p
from typing import Dict, Iterable, List, Optional, Tuple

import torch

from . import _lib
from .encoder import EncoderConfig, PackedBatch, _ENQUEUE_LOCKS, _scratch, _strip_prefix, pack_tokens


class _LayerWF(Structure):
    _fields_ = [(n, c_void_p) for n in ("qkv_w", "qkv_b", "o_w", "o_b", "ln1_g", "ln1_b", "ffn1_w", "ffn1_b",
                                        "ffn2_w", "ffn2_b", "ln2_g", "ln2_b")]


class _EncWF(Structure):
    _fields_ = [
        ("hidden", c_int32), ("layers", c_int32), ("heads", c_int32), ("ffn", c_int32), ("vocab", c_int32),
        ("max_pos", c_int32), ("type_vocab", c_int32), ("ln_eps", c_float),
        ("word_emb", c_void_p), ("pos_emb", c_void_p), ("type_emb", c_void_p), ("emb_ln_g", c_void_p),
        ("emb_ln_b", c_void_p), ("layer", POINTER(_LayerWF)),
        ("cls_dense_w", c_void_p), ("cls_dense_b", c_void_p), ("cls_out_w", c_void_p), ("cls_out_b", c_void_p),
    ]


class EncoderWeightsF32:
    """Device-resident fp32 weights (HF checkpoint names, see ``encoder.EncoderWeights``)."""

    gemm_dtype = "float32"

    def __init__(self, cfg: EncoderConfig, state: Dict[str, torch.Tensor], device: torch.device):
        if device.type != "cuda":
            raise RuntimeError("EncoderWeightsF32 need a HIP device; tensor_truth_amd has no CPU path")
        self.cfg, self.device = cfg, device
        sd = _strip_prefix(state)
        self._keep: List[torch.Tensor] = []

        def t(x):
            x = x.to(device=device, dtype=torch.float32).contiguous()
            self._keep.append(x)
            return x

        H = cfg.hidden
        word, pos, typ = t(sd["embeddings.word_embeddings.weight"]), t(sd["embeddings.position_embeddings.weight"]), \
            t(sd["embeddings.token_type_embeddings.weight"])
        if word.shape != (cfg.vocab_size, H) or pos.shape != (cfg.max_pos, H):
            raise ValueError(f"embedding tables {tuple(word.shape)} / {tuple(pos.shape)} do not match {cfg}")
        self._layers = (_LayerWF * max(cfg.layers, 1))()
        for i in range(cfg.layers):
            p = f"encoder.layer.{i}."
            L = self._layers[i]
            L.qkv_w = t(torch.cat([sd[p + f"attention.self.{n}.weight"] for n in ("query", "key", "value")], 0)).data_ptr()
            L.qkv_b = t(torch.cat([sd[p + f"attention.self.{n}.bias"] for n in ("query", "key", "value")], 0)).data_ptr()
            L.o_w, L.o_b = t(sd[p + "attention.output.dense.weight"]).data_ptr(), t(sd[p + "attention.output.dense.bias"]).data_ptr()
            L.ln1_g = t(sd[p + "attention.output.LayerNorm.weight"]).data_ptr()
            L.ln1_b = t(sd[p + "attention.output.LayerNorm.bias"]).data_ptr()
            L.ffn1_w, L.ffn1_b = t(sd[p + "intermediate.dense.weight"]).data_ptr(), t(sd[p + "intermediate.dense.bias"]).data_ptr()
            L.ffn2_w, L.ffn2_b = t(sd[p + "output.dense.weight"]).data_ptr(), t(sd[p + "output.dense.bias"]).data_ptr()
            L.ln2_g, L.ln2_b = t(sd[p + "output.LayerNorm.weight"]).data_ptr(), t(sd[p + "output.LayerNorm.bias"]).data_ptr()
        w = _EncWF()
        w.hidden, w.layers, w.heads, w.ffn = H, cfg.layers, cfg.heads, cfg.ffn
        w.vocab, w.max_pos, w.type_vocab, w.ln_eps = cfg.vocab_size, cfg.max_pos, cfg.type_vocab, cfg.ln_eps
        w.word_emb, w.pos_emb, w.type_emb = word.data_ptr(), pos.data_ptr(), typ.data_ptr()
        w.emb_ln_g, w.emb_ln_b = t(sd["embeddings.LayerNorm.weight"]).data_ptr(), t(sd["embeddings.LayerNorm.bias"]).data_ptr()
        w.layer = ctypes.cast(self._layers, POINTER(_LayerWF))
        if cfg.num_labels:
            if cfg.num_labels != 1:
                raise ValueError("only single-label (sigmoid) cross-encoder heads are supported")
            w.cls_dense_w, w.cls_dense_b = t(sd["classifier.dense.weight"]).data_ptr(), t(sd["classifier.dense.bias"]).data_ptr()
            w.cls_out_w, w.cls_out_b = t(sd["classifier.out_proj.weight"]).data_ptr(), t(sd["classifier.out_proj.bias"]).data_ptr()
        self.struct = w

    def set_gemm_dtype(self, dtype: str) -> None:
        if dtype not in ("float32", "fp32", "bf16"):
            raise ValueError(f"the fp32 weights run in float32 only (asked for {dtype!r})")

    def parameters(self) -> Iterable[torch.Tensor]:
        return iter(self._keep)

    def nbytes(self) -> int:
        return sum(x.numel() * x.element_size() for x in self._keep)


class EncoderF32:
    """``encoder.Encoder``'s interface on the fp32 forward."""

    def __init__(self, weights: EncoderWeightsF32):
        self.w, self.cfg, self.device = weights, weights.cfg, weights.device
        self.lib = _lib.load_library()
        self._enqueue_lock = _ENQUEUE_LOCKS.setdefault((self.device.type, self.device.index), threading.Lock())

    def _upload(self, batch: PackedBatch):
        from .encoder import Encoder

        return Encoder._upload(self, batch)       # same pinned staging ring, one async copy

    def forward_packed(self, batch: PackedBatch, want_lens: bool = False):
        """-> (hidden [n_rows, H] fp32, seq_start [B] int32 device tensor[, seq_len [B] int32 device tensor])."""
        lib, dev, H = self.lib, self.device, self.cfg.hidden
        ids, pos, types, starts, lens = self._upload(batch)
        hidden = torch.empty((batch.n_rows, H), dtype=torch.float32, device=dev)
        need = lib.tt_encoder_f32_workspace_bytes(ctypes.byref(self.w.struct), batch.n_rows)
        with self._enqueue_lock, torch.cuda.device(dev):
            ws, base = _scratch.get("enc32", dev, need)
            rc = lib.tt_encoder_forward_f32(ctypes.byref(self.w.struct), ids.data_ptr(), pos.data_ptr(),
                                            types.data_ptr() if types is not None else None, starts.data_ptr(),
                                            lens.data_ptr(), len(batch.seq_len), batch.n_rows, batch.max_len,
                                            hidden.data_ptr(), base, need, torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, "tt_encoder_forward_f32")
        return (hidden, starts, lens) if want_lens else (hidden, starts)

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
        hidden, rows = self.forward_packed(batch)
        with torch.cuda.device(self.device):
            rc = self.lib.tt_embed_pool_f32(hidden.data_ptr(), H, rows.data_ptr(), B, H, out.data_ptr(), out16.data_ptr(),
                                            torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(rc, "tt_embed_pool_f32")
        return out, out16

    def rerank_packed(self, batch: PackedBatch, want_logits: bool = False):
        if not self.cfg.num_labels:
            raise RuntimeError("these weights carry no classification head")
        hidden, rows = self.forward_packed(batch)
        B, H = len(batch.seq_len), self.cfg.hidden
        scores = torch.empty(B, dtype=torch.float32, device=self.device)
        logits = torch.empty(B, dtype=torch.float32, device=self.device) if want_logits else None
        n_pad = (B + 127) // 128 * 128
        need = 2 * ((n_pad * H * 4 + 255) // 256 * 256)
        with self._enqueue_lock, torch.cuda.device(self.device):
            ws, base = _scratch.get("head32", self.device, need)
            rc = self.lib.tt_rerank_head_f32(ctypes.byref(self.w.struct), hidden.data_ptr(), rows.data_ptr(), B,
                                             scores.data_ptr(), logits.data_ptr() if want_logits else None, base, need,
                                             torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(rc, "tt_rerank_head_f32")
        return (scores, logits) if want_logits else scores

    def calibrate_fp8(self, *_a, **_k):
        raise RuntimeError("fp8 calibration does not apply to the float32 path")

    def embed(self, seqs, type_ids=None, max_len=None):
        return self.embed_packed(pack_tokens(seqs, self.cfg, type_ids, max_len))

    def rerank(self, seqs, max_len: Optional[int] = 512, want_logits: bool = False):
        return self.rerank_packed(pack_tokens(seqs, self.cfg, None, max_len), want_logits)
