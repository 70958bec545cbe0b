"""Embedding model with the ``HuggingFaceEmbedding`` surface the reference consumes.

Stands in for the object built at ``src/tensortruth/services/model_manager.py:254-260``
(and ``indexing/builder.py:146-152``): same constructor kwargs, same methods
(``get_text_embedding`` ... ``get_agg_embedding_from_queries``), fp32 Python lists,
L2-normalised.  Forward pass = libtt_hip.so (``encoder.py``); batching restates
sentence-transformers ``encode`` (sort by length, slices of ``embed_batch_size``,
truncate to the model limit).
"""
from __future__ import annotations

import logging
from concurrent.futures import ThreadPoolExecutor
from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import weights as _weights
from . import precision as _precision
from .encoder import pack_tokens
from .tokenization import load_tokenizer

logger = logging.getLogger(__name__)

# English BGE v1 / v1.5 models get a query instruction; bge-m3 gets none
_BGE_EN_QUERY_INSTRUCTION = "Represent this question for searching relevant passages: "


def query_instruction_for(model_name: str) -> str:
    n = model_name.lower()
    if "bge" in n and "-en" in n and "m3" not in n:
        return _BGE_EN_QUERY_INSTRUCTION
    return ""


def _report_unused_kwargs(model_name: str, model_kwargs, tokenizer_kwargs) -> None:
    """The reference passes these through to sentence-transformers (model_manager.py:214-252); say what happens to
    them here instead of dropping them silently."""
    mk = model_kwargs or {}
    td = mk.get("torch_dtype")
    if td is not None and str(td).replace("torch.", "") not in ("bfloat16", "float16", "float32", "fp32"):
        logger.warning("%s: torch_dtype=%s is not available on the HIP path; computing in bfloat16 "
                       "(fp32 accumulation). Supported: 'float32' (what no torch_dtype means, as in the reference), 'bfloat16', 'float16'.", model_name, td)
    if mk.get("attn_implementation"):
        logger.info("%s: attn_implementation=%s ignored -- attention is always the fused varlen HIP kernel "
                    "(no padding tokens are computed)", model_name, mk["attn_implementation"])
    side = (tokenizer_kwargs or {}).get("padding_side")
    if side:
        logger.info("%s: padding_side=%s has no effect -- sequences are packed without padding and CLS pooling "
                    "reads the first token of every sequence", model_name, side)


class HipHuggingFaceEmbedding:
    def __init__(self, model_name: str = "BAAI/bge-m3", device: Optional[str] = None,
                 model_kwargs: Optional[Dict[str, Any]] = None, tokenizer_kwargs: Optional[Dict[str, Any]] = None,
                 embed_batch_size: int = 128, max_length: Optional[int] = None, normalize: bool = True,
                 query_instruction: Optional[str] = None, text_instruction: Optional[str] = None, **_ignored):
        if not (0 < embed_batch_size <= 2048):
            raise ValueError(f"embed_batch_size {embed_batch_size} not in (0, 2048]")
        dev = torch.device("cuda" if device in (None, "cuda") else device)
        if dev.type != "cuda":
            raise RuntimeError(f"device '{device}': tensor_truth_amd runs on HIP devices only (no CPU path)")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.model_name = model_name
        self.device = dev
        self.embed_batch_size = embed_batch_size
        self.normalize = normalize
        self.tokenizer_kwargs = tokenizer_kwargs
        _report_unused_kwargs(model_name, model_kwargs, tokenizer_kwargs)
        cfg, state, mdir = _weights.resolve(model_name, model_kwargs, dev, want_head=False)
        self.config = cfg
        # sentence-transformers pooling: what the checkpoint directory declares (1_Pooling/config.json), unless the caller says
        # (model_kwargs["pooling"]); "cls" for the BGE family the reference defaults to, "mean" for e5 / all-MiniLM / gte ...
        pooling = (model_kwargs or {}).get("pooling") or _weights.pooling_mode(mdir)
        pooling = {"cls_token": "cls", "mean_tokens": "mean"}.get(pooling, pooling)
        if pooling not in ("cls", "mean"):
            # no silent wrong vectors: anything else (max, weighted mean, last token ...) has no kernel here
            raise NotImplementedError(f"{model_name}: sentence-transformers pooling '{pooling}' is not supported "
                                      f"(CLS and mean pooling, both followed by L2 normalisation, are)")
        self.pooling = pooling
        # precision.resolve(): model_kwargs (torch_dtype float32 = the reference's own default, config_schema.py:66-76),
        # ModelManager.precision, TT_PRECISION; default: the reference's fp32 semantics.  (`_model.parameters()` is read by the memory accounting.)
        self._model, self._encoder, self.precision = _precision.build_encoder(cfg, state, dev, model_kwargs, f"embedder {model_name}")
        self._tokenizer = (model_kwargs or {}).get("tokenizer") or load_tokenizer(mdir, cfg.arch, cfg.vocab_size)
        self.max_length = min(max_length or cfg.max_seq_len, cfg.max_seq_len)
        self.query_instruction = query_instruction_for(model_name) if query_instruction is None else query_instruction
        self.text_instruction = text_instruction or ""
        # texts tokenized per pipeline step (see _embed_texts): one encoder batch -- sequences are packed without
        # padding tokens, so nothing is lost by sorting by length inside a window only
        self.pipeline_window = int((model_kwargs or {}).get("pipeline_window", max(embed_batch_size, 2048)))
        # tokens per forward pass: ``embed_batch_size`` is the reference's unit (texts per padded batch); sequences are
        # packed without padding here, so what the GEMMs see is the TOKEN count -- 128 sentence groups of ~50 tokens are
        # 27 row tiles, a tenth of a CU-wave.  A forward therefore takes ``embed_batch_size`` texts and keeps adding texts
        # (shorter ones: the window is sorted by length) until it holds ``forward_tokens`` tokens; results do not depend
        # on the batching (bit-identical embeddings, tests/test_configs_gpu.py).
        self.forward_tokens = int((model_kwargs or {}).get("forward_tokens", 131072))
        # running totals of what went through the encoder (bench.py prices an ingest against the matrix roofline with them)
        self.stats = {"sequences": 0, "tokens": 0, "sum_len_sq": 0}

    # ---- token-id level (what the kernels see) ------------------------------------------------
    def embed_token_batches(self, seqs: Sequence[Sequence[int]]) -> torch.Tensor:
        """Embeds tokenised sequences -> fp32 [n, H] on the device, original order."""
        order = sorted(range(len(seqs)), key=lambda i: -len(seqs[i]))
        parts = []
        lo = 0
        while lo < len(order):
            hi, tokens = lo, 0
            while hi < len(order) and (hi - lo < self.embed_batch_size or tokens < self.forward_tokens):
                tokens += min(len(seqs[order[hi]]), self.max_length)
                hi += 1
            sel = order[lo:hi]
            lens = [min(len(seqs[i]), self.max_length) for i in sel]
            self.stats["sequences"] += len(sel)
            self.stats["tokens"] += sum(lens)
            self.stats["sum_len_sq"] += sum(n * n for n in lens)
            emb, _ = self._encoder.embed_packed(pack_tokens([seqs[i] for i in sel], self.config, None, self.max_length),
                                                pooling=self.pooling)
            parts.append(emb)
            lo = hi
        out = torch.empty((len(seqs), self.config.hidden), dtype=torch.float32, device=self.device)
        if parts:  # one scatter back to the caller's order (one small index upload per call, not per batch)
            out[torch.tensor(order, dtype=torch.int64).to(self.device, non_blocking=True)] = torch.cat(parts)
        return out

    def embed_flat(self, flat: np.ndarray, lens: np.ndarray) -> torch.Tensor:
        """``embed_token_batches`` for sequences handed over as one flat int32 array + lengths (what the ingest workers send): the
        same batches (sorted by length, ``embed_batch_size`` texts topped up to ``forward_tokens`` tokens), the same embeddings bit for
        bit, no per-sequence Python (argsort, cumulative sums, ``encoder.pack_flat``)."""
        from .encoder import pack_flat

        lens = np.asarray(lens, dtype=np.int64)
        n = len(lens)
        out = torch.empty((n, self.config.hidden), dtype=torch.float32, device=self.device)
        if n == 0:
            return out
        flat = np.asarray(flat, dtype=np.int32)
        first = np.zeros(n, dtype=np.int64)
        np.cumsum(lens[:-1], out=first[1:])
        order = np.argsort(-lens, kind="stable")               # = sorted(range(n), key=lambda i: -len(seqs[i])): stable, longest first
        eff = np.minimum(lens[order], self.max_length)
        csum = np.cumsum(eff)
        parts, lo = [], 0
        while lo < n:
            # texts lo .. hi-1: at least embed_batch_size of them, then more while the tokens BEFORE the next one stay below forward_tokens
            base = int(csum[lo - 1]) if lo else 0
            hi_tok = int(np.searchsorted(csum, base + self.forward_tokens, side="left")) + 1     # first hi with tokens(lo..hi-1) >= forward_tokens
            hi = min(n, max(lo + self.embed_batch_size, hi_tok))
            sel = order[lo:hi]
            self.stats["sequences"] += hi - lo
            self.stats["tokens"] += int(csum[hi - 1]) - base
            self.stats["sum_len_sq"] += int((eff[lo:hi] * eff[lo:hi]).sum())
            emb, _ = self._encoder.embed_packed(pack_flat(flat, first, lens, sel, self.config, self.max_length), pooling=self.pooling)
            parts.append(emb)
            lo = hi
        out[torch.from_numpy(order).to(self.device, non_blocking=True)] = torch.cat(parts)
        return out

    def _tokenize(self, texts: Sequence[str], prefix: str):
        tk = self._tokenizer
        full = [prefix + t for t in texts] if prefix else list(texts)
        if hasattr(tk, "encode_batch"):
            return tk.encode_batch(full, self.max_length)
        return [tk.encode(t, self.max_length) for t in full]

    def _embed_texts(self, texts: Sequence[str], prefix: str) -> torch.Tensor:
        """Strings -> fp32 [n, H] on the device.  Long inputs run as a pipeline over windows of
        ``pipeline_window`` texts: a background thread tokenizes window i+1 while this thread packs window i
        into pinned staging and enqueues its forward passes, which the GPU executes asynchronously -- host
        tokenization overlaps the encoder instead of preceding it.  Batches are
        formed inside a window (sorted by length, ``embed_batch_size`` each); results do not depend on the
        batching (tests/test_configs_gpu.py)."""
        n, win = len(texts), self.pipeline_window
        if n <= win:
            return self.embed_token_batches(self._tokenize(texts, prefix))
        # a short first window puts the GPU to work early; after that two windows are always being tokenized ahead
        bounds = [0, max(64, win // 8)]
        while bounds[-1] < n:
            bounds.append(min(n, bounds[-1] + win))
        spans = list(zip(bounds[:-1], bounds[1:]))
        out = torch.empty((n, self.config.hidden), dtype=torch.float32, device=self.device)
        with ThreadPoolExecutor(max_workers=2) as pool:
            futs = [pool.submit(self._tokenize, texts[a:b], prefix) for a, b in spans[:2]]
            for i, (a, b) in enumerate(spans):
                seqs = futs[i].result()
                if i + 2 < len(spans):
                    futs.append(pool.submit(self._tokenize, texts[spans[i + 2][0]:spans[i + 2][1]], prefix))
                out[a:b] = self.embed_token_batches(seqs)
                futs[i] = None
        return out

    # ---- HuggingFaceEmbedding / BaseEmbedding surface --------------------------------------------
    def get_text_embedding(self, text: str) -> List[float]:
        return self._embed_texts([text], self.text_instruction)[0].cpu().tolist()

    def get_text_embedding_batch(self, texts: List[str], show_progress: bool = False, **_kw) -> List[List[float]]:
        if not texts:
            return []
        return self._embed_texts(texts, self.text_instruction).cpu().tolist()

    def get_query_embedding(self, query: str) -> List[float]:
        return self._embed_texts([query], self.query_instruction)[0].cpu().tolist()

    def get_agg_embedding_from_queries(self, queries: List[str], agg_fn=None) -> List[float]:
        embs = self._embed_texts(list(queries), self.query_instruction)
        if agg_fn is not None:
            return list(agg_fn(embs.cpu().tolist()))
        return embs.mean(dim=0).cpu().tolist()

    def query_embedding_device(self, queries: Sequence[str]) -> torch.Tensor:
        """Device-resident fp32 [n, H] query embeddings (no host round trip) for the retriever."""
        return self._embed_texts(list(queries), self.query_instruction)

    @staticmethod
    def similarity(a: Sequence[float], b: Sequence[float], mode: str = "cosine") -> float:
        # host arithmetic on two Python vectors, as llama-index's BaseEmbedding.similarity ([UPSTREAM-K]): not the hot path
        ta, tb = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        if mode == "dot_product":
            return float(np.dot(ta, tb))
        if mode == "euclidean":
            return float(-np.linalg.norm(ta - tb))
        return float(np.dot(ta, tb) / max(np.linalg.norm(ta) * np.linalg.norm(tb), 1e-30))
