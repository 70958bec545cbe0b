"""Embedding model with the ``HuggingFaceEmbedding`` surface the reference consumes.

Stands in for the object built at ``src/tensortruth/services/model_manager.py:254-260``
(and ``indexing/builder.py:146-152``): same constructor kwargs, same methods
(``get_text_embedding`` ... ``get_agg_embedding_from_queries``), fp32 Python lists,
L2-normalised.  Forward pass = libtt_hip.so (``encoder.py``); batching restates
sentence-transformers ``encode`` (sort by length, slices of ``embed_batch_size``,
truncate to the model limit).
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence

import torch

from . import weights as _weights
from .encoder import Encoder, EncoderWeights, pack_tokens
from .tokenization import load_tokenizer

# English BGE v1 / v1.5 models get a query instruction; bge-m3 gets none
_BGE_EN_QUERY_INSTRUCTION = "Represent this question for searching relevant passages: "


def query_instruction_for(model_name: str) -> str:
    n = model_name.lower()
    if "bge" in n and "-en" in n and "m3" not in n:
        return _BGE_EN_QUERY_INSTRUCTION
    return ""


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
        cfg, state, mdir = _weights.resolve(model_name, model_kwargs, dev, want_head=False)
        self.config = cfg
        self._model = EncoderWeights(cfg, state, dev)       # .parameters() for memory accounting
        self._model.set_gemm_dtype((model_kwargs or {}).get("gemm_dtype", "bf16"))
        self._encoder = Encoder(self._model)
        self._tokenizer = (model_kwargs or {}).get("tokenizer") or load_tokenizer(mdir, cfg.arch, cfg.vocab_size)
        self.max_length = min(max_length or cfg.max_seq_len, cfg.max_seq_len)
        self.query_instruction = query_instruction_for(model_name) if query_instruction is None else query_instruction
        self.text_instruction = text_instruction or ""

    # ---- token-id level (what the kernels see) ------------------------------------------------
    def embed_token_batches(self, seqs: Sequence[Sequence[int]]) -> torch.Tensor:
        """Embeds tokenised sequences -> fp32 [n, H] on the device, original order."""
        order = sorted(range(len(seqs)), key=lambda i: -len(seqs[i]))
        out = torch.empty((len(seqs), self.config.hidden), dtype=torch.float32, device=self.device)
        for lo in range(0, len(order), self.embed_batch_size):
            sel = order[lo:lo + self.embed_batch_size]
            emb, _ = self._encoder.embed_packed(pack_tokens([seqs[i] for i in sel], self.config, None, self.max_length))
            out[torch.tensor(sel, device=self.device)] = emb
        return out

    def _embed_texts(self, texts: Sequence[str], prefix: str) -> torch.Tensor:
        seqs = [self._tokenizer.encode(prefix + t, self.max_length) for t in texts]
        return self.embed_token_batches(seqs)

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
        ta, tb = torch.tensor(a, dtype=torch.float64), torch.tensor(b, dtype=torch.float64)
        if mode == "dot_product":
            return float(ta @ tb)
        if mode == "euclidean":
            return float(-(ta - tb).norm())
        return float((ta @ tb) / (ta.norm() * tb.norm()).clamp_min(1e-30))
