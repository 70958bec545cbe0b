"""Embedder / reranker lifecycle with the reference's ``ModelManager`` semantics
(``src/tensortruth/services/model_manager.py``): a process-wide singleton holding at most
ONE embedder and ONE reranker; ``get_embedder(model_name, device)`` reuses the resident model
and reloads only when (model, device) changes; ``get_reranker(model_name, top_n, device)``
also reloads when ``top_n`` changes; a failed load raises ``RuntimeError`` and leaves the slot
empty; unloading drops the model and returns its HBM to the allocator.  The loaded objects
are the HIP classes of this package instead of the llama-index wrappers.
"""
from __future__ import annotations

import gc
import logging
import threading
from dataclasses import dataclass
from typing import Any, Dict, Optional

logger = logging.getLogger(__name__)

DEFAULT_EMBEDDING_MODEL = "BAAI/bge-m3"
DEFAULT_RERANKER_MODEL = "BAAI/bge-reranker-v2-m3"


@dataclass
class EmbeddingModelConfig:
    """Per-model settings (reference: app_utils/config_schema.py:40-61)."""

    batch_size_cuda: int = 128
    batch_size_cpu: int = 16
    torch_dtype: Optional[str] = None
    padding_side: Optional[str] = None
    flash_attention: bool = False
    trust_remote_code: bool = True


DEFAULT_EMBEDDING_MODEL_CONFIGS: Dict[str, Dict] = {"BAAI/bge-m3": {}}


def sanitize_model_id(model_name: str) -> str:
    """'BAAI/bge-m3' -> 'bge-m3' (directory-safe id; reference: indexing/metadata.py:22-52)."""
    import re

    name = model_name.split("/")[-1].lower()
    name = re.sub(r"[^a-z0-9\-_.]", "-", name)
    return re.sub(r"-+", "-", name).strip("-")


def resolve_embedding_model_name(value: str, extra_known=()) -> str:
    """Heal a sanitized id back to the full hub path when it is a known model."""
    if "/" in value:
        return value
    for full in list(DEFAULT_EMBEDDING_MODEL_CONFIGS) + list(extra_known):
        if sanitize_model_id(full) == value:
            return full
    return value


class ModelManager:
    _instance: Optional["ModelManager"] = None
    _lock = threading.Lock()

    def __new__(cls) -> "ModelManager":
        if cls._instance is None:
            with cls._lock:
                if cls._instance is None:
                    inst = super().__new__(cls)
                    inst._initialized = False
                    cls._instance = inst
        return cls._instance

    def __init__(self) -> None:
        if getattr(self, "_initialized", False):
            return
        self._embedder = None
        self._embedder_model_name: Optional[str] = None
        self._embedder_device: Optional[str] = None
        self._reranker = None
        self._reranker_model_name: Optional[str] = None
        self._reranker_top_n: Optional[int] = None
        self._reranker_device: Optional[str] = None
        self._default_device = "cuda"
        self._model_lock = threading.Lock()
        self.embedding_model_configs: Dict[str, Dict] = dict(DEFAULT_EMBEDDING_MODEL_CONFIGS)
        self.model_kwargs_overrides: Dict[str, Dict[str, Any]] = {}  # model name -> extra model_kwargs
        # Process-level precision of the models this manager loads: None (= TT_PRECISION, else the reference's fp32 semantics), "bf16", "fp16", "fp8" or
        # "reference" -- the reference's own fp32 semantics for its unchanged calls (precision.py).  Set it before the
        # first get_embedder / get_reranker, or call set_precision() (drops the resident models).
        self.precision: Optional[str] = None
        self._initialized = True

    @classmethod
    def get_instance(cls) -> "ModelManager":
        return cls()

    @classmethod
    def reset_instance(cls) -> None:
        with cls._lock:
            if cls._instance is not None:
                cls._instance.unload_all()
                cls._instance._initialized = False
                cls._instance = None

    def set_default_device(self, device: str) -> None:
        self._default_device = device

    def set_precision(self, precision: Optional[str]) -> None:
        """Config key for the arithmetic of every model loaded from now on (see ``precision.py``); resident models are
        dropped so that the next ``get_*`` reloads them in the new mode."""
        from . import precision as _p

        value = None if precision is None else _p.canonical(precision)
        if value != self.precision:
            self.precision = value
            self.unload_all()

    def _with_precision(self, model_kwargs: Optional[Dict[str, Any]]) -> Optional[Dict[str, Any]]:
        mk = dict(model_kwargs or {})
        explicit = mk.get("precision") is not None or mk.get("torch_dtype") is not None or mk.get("gemm_dtype") is not None
        if self.precision is not None and not explicit:
            mk["precision"] = self.precision
        return mk or None

    # ---- embedder ------------------------------------------------------------------------------
    def _embedding_config(self, model_name: str) -> EmbeddingModelConfig:
        return EmbeddingModelConfig(**self.embedding_model_configs.get(model_name, {}))

    def get_embedder(self, model_name: Optional[str] = None, device: Optional[str] = None):
        model_name = resolve_embedding_model_name(model_name or DEFAULT_EMBEDDING_MODEL, self.embedding_model_configs)
        device = device or self._default_device
        with self._model_lock:
            if (self._embedder is None or self._embedder_model_name != model_name
                    or self._embedder_device != device):
                self._unload_embedder()
                self._load_embedder(model_name, device)
            assert self._embedder is not None
            return self._embedder

    def _load_embedder(self, model_name: str, device: str) -> None:
        try:
            from .embedding import HipHuggingFaceEmbedding

            mc = self._embedding_config(model_name)
            batch = mc.batch_size_cuda if device == "cuda" else mc.batch_size_cpu
            model_kwargs: Dict[str, Any] = {"trust_remote_code": mc.trust_remote_code}
            if mc.torch_dtype:
                model_kwargs["torch_dtype"] = mc.torch_dtype
            if mc.flash_attention:   # reference: attn_implementation="flash_attention_2" when flash_attn imports (:232-242)
                logger.info("flash_attention requested for %s: the HIP encoder's attention is always a fused varlen "
                            "kernel, nothing to enable", model_name)
            logger.info("Creating embedding model: %s (batch_size=%d, dtype=%s)", model_name, batch,
                        mc.torch_dtype or "none named: precision.resolve() decides (default: the reference's fp32 semantics)")
            model_kwargs.update(self.model_kwargs_overrides.get(model_name, {}))
            model_kwargs = self._with_precision(model_kwargs)
            tokenizer_kwargs = {"padding_side": mc.padding_side} if mc.padding_side else None
            self._embedder = HipHuggingFaceEmbedding(model_name=model_name, device=device, model_kwargs=model_kwargs,
                                                     tokenizer_kwargs=tokenizer_kwargs, embed_batch_size=batch)
            self._embedder_model_name, self._embedder_device = model_name, device
        except Exception as e:  # noqa: BLE001
            self._embedder = None
            self._embedder_model_name = self._embedder_device = None
            raise RuntimeError(f"Failed to load embedding model '{model_name}' on {device}: {e}") from e

    def _unload_embedder(self) -> None:
        if self._embedder is not None:
            self._embedder = None
            self._clear_gpu_cache()
        self._embedder_model_name = self._embedder_device = None

    # ---- reranker -------------------------------------------------------------------------------
    def get_reranker(self, model_name: Optional[str] = None, top_n: int = 5, device: Optional[str] = None):
        model_name = model_name or DEFAULT_RERANKER_MODEL
        device = device or self._default_device
        with self._model_lock:
            if (self._reranker is None or self._reranker_model_name != model_name
                    or self._reranker_device != device or self._reranker_top_n != top_n):
                self._unload_reranker()
                self._load_reranker(model_name, top_n, device)
            assert self._reranker is not None
            return self._reranker

    def _load_reranker(self, model_name: str, top_n: int, device: str) -> None:
        try:
            from .rerank import HipSentenceTransformerRerank

            self._reranker = HipSentenceTransformerRerank(
                model=model_name, top_n=top_n, device=device,
                model_kwargs=self._with_precision(self.model_kwargs_overrides.get(model_name)))
            self._reranker_model_name, self._reranker_top_n, self._reranker_device = model_name, top_n, device
        except Exception as e:  # noqa: BLE001
            self._reranker = None
            self._reranker_model_name = self._reranker_top_n = self._reranker_device = None
            raise RuntimeError(f"Failed to load reranker model '{model_name}' on {device}: {e}") from e

    def _unload_reranker(self) -> None:
        if self._reranker is not None:
            self._reranker = None
            self._clear_gpu_cache()
        self._reranker_model_name = self._reranker_top_n = self._reranker_device = None

    # ---- housekeeping -----------------------------------------------------------------------------
    @staticmethod
    def _clear_gpu_cache() -> None:
        gc.collect()
        try:
            import torch

            if torch.cuda.is_available():
                torch.cuda.empty_cache()
        except Exception:  # noqa: BLE001
            pass

    def unload_all(self) -> None:
        with self._model_lock:
            self._unload_embedder()
            self._unload_reranker()

    def get_status(self) -> Dict[str, Any]:
        return {
            "embedder": {"loaded": self._embedder is not None, "model_name": self._embedder_model_name,
                         "device": self._embedder_device},
            "reranker": {"loaded": self._reranker is not None, "model_name": self._reranker_model_name,
                         "top_n": self._reranker_top_n, "device": self._reranker_device},
            "default_device": self._default_device,
        }

    def get_memory_usage(self) -> Dict[str, Any]:
        """Bytes held by the resident models (reference reads wrapper._model / wrapper.model parameters)."""

        def nbytes(obj, attr):
            w = getattr(obj, attr, None) if obj is not None else None
            if w is None:
                return 0
            return sum(p.numel() * p.element_size() for p in w.parameters())

        e, r = nbytes(self._embedder, "_model"), nbytes(self._reranker, "model")
        return {"embedder_bytes": e, "reranker_bytes": r, "total_bytes": e + r,
                "embedder_gb": e / 1e9, "reranker_gb": r / 1e9}
