"""Post-rerank retrieval quality metrics (host code, a handful of floats per query).

Restates ``src/tensortruth/services/retrieval_metrics.py:141-261`` so that
``RetrievalService.retrieve`` returns the same ``metrics`` dict: score statistics (mean /
median / min / max / sample std / quartiles as medians of the lower and upper halves / IQR /
range), source diversity (unique files, doc types, Shannon entropy in bits), coverage (chars,
mean chunk length, ``chars // 4`` estimated tokens) and the high (>= 0.7) / low (< 0.4)
confidence ratios.
"""
from __future__ import annotations

import math
import statistics
from collections import Counter
from dataclasses import dataclass
from typing import Any, Dict, List, Optional


def calculate_entropy(counts: List[int]) -> float:
    if not counts or len(counts) == 1:
        return 0.0
    total = sum(counts)
    if total == 0:
        return 0.0
    h = 0.0
    for c in counts:
        if c > 0:
            p = c / total
            h -= p * math.log2(p)
    return h


def _native(v):
    if v is None:
        return None
    if hasattr(v, "item"):
        return v.item()
    return v


@dataclass
class RetrievalMetrics:
    score_mean: Optional[float] = None
    score_median: Optional[float] = None
    score_min: Optional[float] = None
    score_max: Optional[float] = None
    score_std: Optional[float] = None
    score_q1: Optional[float] = None
    score_q3: Optional[float] = None
    score_iqr: Optional[float] = None
    score_range: Optional[float] = None
    unique_sources: int = 0
    source_types: int = 0
    source_entropy: Optional[float] = None
    total_context_chars: int = 0
    avg_chunk_length: float = 0.0
    total_chunks: int = 0
    estimated_tokens: int = 0
    high_confidence_ratio: float = 0.0
    low_confidence_ratio: float = 0.0
    configured_top_n: Optional[int] = None

    def to_dict(self) -> Dict[str, Any]:
        return {
            "score_distribution": {k: _native(getattr(self, "score_" + k)) for k in
                                   ("mean", "median", "min", "max", "std", "q1", "q3", "iqr", "range")},
            "diversity": {"unique_sources": int(self.unique_sources), "source_types": int(self.source_types),
                          "source_entropy": _native(self.source_entropy)},
            "coverage": {"total_context_chars": int(self.total_context_chars),
                         "avg_chunk_length": _native(self.avg_chunk_length),
                         "total_chunks": int(self.total_chunks), "estimated_tokens": int(self.estimated_tokens)},
            "quality": {"high_confidence_ratio": _native(self.high_confidence_ratio),
                        "low_confidence_ratio": _native(self.low_confidence_ratio)},
            "configuration": {"configured_top_n": _native(self.configured_top_n)},
        }


def _metadata_of(node) -> Dict[str, Any]:
    if hasattr(node, "node") and hasattr(node.node, "metadata"):
        return node.node.metadata or {}
    if hasattr(node, "metadata"):
        return node.metadata or {}
    return {}


def _content_of(node) -> str:
    if hasattr(node, "node"):
        if hasattr(node.node, "get_content"):
            return node.node.get_content()
        if hasattr(node.node, "text"):
            return node.node.text
        return ""
    return node.text if hasattr(node, "text") else ""


def compute_retrieval_metrics(source_nodes: List[Any]) -> RetrievalMetrics:
    m = RetrievalMetrics()
    if not source_nodes:
        return m
    scores = []
    for n in source_nodes:
        s = getattr(n, "score", None)
        if s is not None:
            scores.append(float(s.item()) if hasattr(s, "item") else float(s))
    if scores:
        m.score_mean = statistics.mean(scores)
        m.score_median = statistics.median(scores)
        m.score_min, m.score_max = min(scores), max(scores)
        m.score_range = m.score_max - m.score_min
        if len(scores) >= 2:
            m.score_std = statistics.stdev(scores)
            ordered = sorted(scores)
            m.score_q1 = statistics.median(ordered[: len(ordered) // 2])
            m.score_q3 = statistics.median(ordered[(len(ordered) + 1) // 2:])
            m.score_iqr = m.score_q3 - m.score_q1
    files, types = [], []
    for n in source_nodes:
        md = _metadata_of(n)
        files.append(md.get("filename") or md.get("file_name") or md.get("source_url", "unknown"))
        types.append(md.get("doc_type", "unknown"))
    m.unique_sources, m.source_types = len(set(files)), len(set(types))
    if files:
        m.source_entropy = calculate_entropy(list(Counter(files).values()))
    m.total_chunks = len(source_nodes)
    chars = sum(len(_content_of(n)) for n in source_nodes)
    m.total_context_chars = chars
    m.avg_chunk_length = chars / len(source_nodes)
    m.estimated_tokens = chars // 4
    if scores:
        m.high_confidence_ratio = sum(1 for s in scores if s >= 0.7) / len(scores)
        m.low_confidence_ratio = sum(1 for s in scores if s < 0.4) / len(scores)
    return m
