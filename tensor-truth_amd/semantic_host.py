"""The semantic splitter's host-only steps (no torch, no GPU): sentence split, percentile breakpoints, joining.

Split off ``semantic.py`` so that the ingest worker processes (``ingest_workers.py``) can import them without paying for a
torch import each.  Restates llama-index ``SemanticSplitterNodeParser``'s host logic (the reference builds it at
``src/tensortruth/indexing/builder.py:391-418``)."""
from __future__ import annotations

import re
from typing import List, Sequence

import numpy as np

_SENT = re.compile(r"[^.!?\n]+[.!?]*[\n]*|[\n]+")


def split_sentences(text: str) -> List[str]:
    return [s for s in (m.group(0) for m in _SENT.finditer(text)) if s.strip()]


def breakpoints_from_distances(dist: Sequence[float], percentile: float) -> List[int]:
    """Indices i such that a cut falls after sentence i (host logic, known-answer tested)."""
    d = np.asarray(dist, dtype=np.float64)
    if d.size == 0:
        return []
    thr = np.percentile(d, percentile)
    return [int(i) for i in np.nonzero(d > thr)[0]]


def join_chunks(sentences: List[str], cuts: Sequence[int]) -> List[str]:
    chunks, start = [], 0
    for c in cuts:
        chunks.append("".join(sentences[start:c + 1]).strip())
        start = c + 1
    if start < len(sentences):
        chunks.append("".join(sentences[start:]).strip())
    return [c for c in chunks if c]
