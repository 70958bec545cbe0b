"""Retrieval + rerank + confidence pipeline: the counterpart of the reference's
``RAGService.retrieve`` (``src/tensortruth/services/rag_service.py:518-661``), i.e. everything
between the query string and the ``RAGRetrievalResult`` except the LLM-backed query
condensation (out of scope).  Order of operations restated from the reference:
retrieve -> each postprocessor in turn (a failing postprocessor leaves the un-processed
nodes, ``:612-622``) -> truncate to ``reranker_top_n`` (``:624-627``) -> metrics
(``:631-634``) -> confidence "none" / "low" (best score < ``confidence_cutoff``) / "normal"
(``:636-648``).
"""
from __future__ import annotations

import logging
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, List, Optional

from .quality_metrics import compute_retrieval_metrics
from .schema import QueryBundle

logger = logging.getLogger(__name__)


@dataclass
class ToolProgress:
    tool_id: str
    phase: str
    message: str
    metadata: Dict[str, Any] = field(default_factory=dict)


@dataclass
class RAGRetrievalResult:
    source_nodes: List[Any] = field(default_factory=list)
    confidence_level: str = "normal"
    metrics: Optional[Dict[str, Any]] = None
    condensed_query: str = ""
    num_sources: int = 0


class RetrievalService:
    def __init__(self, retriever=None, postprocessors: Optional[List[Any]] = None,
                 params: Optional[Dict[str, Any]] = None):
        self._retriever = retriever
        self._node_postprocessors = list(postprocessors or [])
        self._current_params = dict(params or {})

    def is_loaded(self) -> bool:
        return self._retriever is not None

    def clear(self) -> None:
        if self._retriever is not None and hasattr(self._retriever, "clear_cache"):
            self._retriever.clear_cache()
        self._retriever = None
        self._node_postprocessors = []
        self._current_params = {}

    def retrieve(self, query: str, params: Optional[Dict[str, Any]] = None,
                 progress_callback: Optional[Callable[[ToolProgress], None]] = None) -> RAGRetrievalResult:
        if self._retriever is None:
            logger.warning("retrieve() called with no retriever loaded")
            return RAGRetrievalResult(confidence_level="none")
        effective = self._current_params or params or {}
        if progress_callback:
            progress_callback(ToolProgress("rag", "retrieving", "Searching knowledge base..."))
        question = query
        nodes = self._retriever.retrieve(question)
        if self._node_postprocessors:
            if progress_callback:
                progress_callback(ToolProgress("rag", "reranking", "Ranking results..."))
            bundle = QueryBundle(query_str=question)
            try:
                for pp in self._node_postprocessors:
                    nodes = pp.postprocess_nodes(nodes, query_bundle=bundle)
            except Exception as e:  # noqa: BLE001 - same degrade as the reference
                logger.warning(f"Postprocessor failed, using unprocessed nodes: {e}")
        top_n = effective.get("reranker_top_n")
        if top_n and len(nodes) > top_n:
            nodes = nodes[:top_n]
        metrics = compute_retrieval_metrics(nodes)
        metrics.configured_top_n = effective.get("reranker_top_n")
        level = "normal"
        if not nodes:
            level = "none"
        else:
            cutoff = effective.get("confidence_cutoff", 0.0)
            if cutoff > 0:
                best = max((n.score for n in nodes if n.score is not None), default=0.0)
                if best < cutoff:
                    level = "low"
        return RAGRetrievalResult(source_nodes=nodes, confidence_level=level, metrics=metrics.to_dict(),
                                  condensed_query=question, num_sources=len(nodes))


def build_retrieval_service(indexes: List[Any], params: Optional[Dict[str, Any]] = None, device: str = "cuda",
                            manager=None) -> RetrievalService:
    """Assemble the pipeline the way ``load_engine_for_modules`` does (rag_engine.py:529-738),
    minus the LLM: per index ``AutoMergingRetriever(index.as_retriever(similarity_top_k))``,
    ``MultiIndexRetriever`` over them, postprocessors ``[reranker, SimilarityPostprocessor?]``."""
    from .model_manager import ModelManager
    from .retrievers import AutoMergingRetriever, MultiIndexRetriever, SimilarityPostprocessor, similarity_top_k_for

    params = dict(params or {})
    mgr = manager or ModelManager.get_instance()
    top_n = params.get("reranker_top_n", 3)
    k = params.get("similarity_top_k") or similarity_top_k_for(top_n)
    retrievers = [AutoMergingRetriever(ix.as_retriever(similarity_top_k=k), ix.docstore) for ix in indexes]
    multi = MultiIndexRetriever(retrievers, balance_strategy=params.get("balance_strategy", "top_k_per_index"))
    rr = mgr.get_reranker(params.get("reranker_model"), top_n=top_n, device=device)
    post: List[Any] = [rr]
    # leaves tokenised once, at ingest (build_index(keep_leaf_token_ids=True)): if the indexes kept their leaves' token ids and the
    # reranker's tokenizer is the embedder's, retrieved leaves reach the reranker as ids (auto-merged parents still as text)
    # The source stays with THIS service (a per-call argument through RerankerWithTokenSource): `rr` is ModelManager's shared cached
    # instance, and a second service over other indexes must neither replace nor detach what this one looks its leaves up in.
    if hasattr(rr, "accepts_token_source"):
        sources = [ts for ts in (ix.token_source() for ix in indexes if hasattr(ix, "token_source")) if ts is not None]
        if sources and len({(sig, instr) for _, sig, instr in sources}) == 1 and rr.accepts_token_source(sources[0][1], sources[0][2]):
            getters = [g for g, _, _ in sources]

            def lookup(node_id, _getters=getters):
                for g in _getters:
                    ids = g(node_id)
                    if ids is not None:
                        return ids
                return None

            from .rerank import RerankerWithTokenSource

            post[0] = RerankerWithTokenSource(rr, lookup if len(getters) > 1 else getters[0])
    hard = params.get("confidence_cutoff_hard", 0.0)
    if hard and hard > 0:
        post.append(SimilarityPostprocessor(similarity_cutoff=hard))
    return RetrievalService(multi, post, params)
