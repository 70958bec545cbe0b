"""Semantic splitter: cuts a document where consecutive sentence groups drift apart in embedding space.

Restates the node parser the reference builds for ``ChunkingStrategy.SEMANTIC`` /
``SEMANTIC_HIERARCHICAL`` (``src/tensortruth/indexing/builder.py:391-418`` ->
llama-index ``SemanticSplitterNodeParser``): regex sentence split, each sentence combined
with ``buffer_size`` neighbours on either side, one embedding per group (the GPU embedder), distance
``d_i = 1 - cos(e_i, e_{i+1})`` (``tt_adjacent_cosine``), breakpoints where ``d_i`` exceeds the
``breakpoint_percentile_threshold``-th percentile (numpy linear interpolation), chunks = sentences joined
between breakpoints.
"""
from __future__ import annotations

import re
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .schema import TextNode

_SENT = re.compile(r"[^.!?\n]+[.!?]*[\n]*|[\n]+")


def split_sentences(text: str) -> List[str]:
    return [s for s in (m.group(0) for m in _SENT.finditer(text)) if s.strip()]


def adjacent_distances(emb: torch.Tensor) -> torch.Tensor:
    """emb fp32 [n, H] on a HIP device -> [n-1] distances 1 - cos(e_i, e_{i+1})."""
    if not emb.is_cuda:
        raise RuntimeError("adjacent_distances needs a HIP device tensor; tensor_truth_amd has no CPU path")
    emb = emb.to(torch.float32).contiguous()
    n, h = emb.shape
    out = torch.empty(max(n - 1, 0), dtype=torch.float32, device=emb.device)
    if n > 1:
        with torch.cuda.device(emb.device):
            rc = _lib.load_library().tt_adjacent_cosine(emb.data_ptr(), n, h, out.data_ptr(),
                                                        torch.cuda.current_stream(emb.device).cuda_stream)
        _lib.check(rc, "tt_adjacent_cosine")
    return out


def breakpoints_from_distances(dist: Sequence[float], percentile: float) -> List[int]:
    """Indices i such that a cut falls after sentence i (host logic, known-answer tested)."""
    d = np.asarray(dist, dtype=np.float64)
    if d.size == 0:
        return []
    thr = np.percentile(d, percentile)
    return [int(i) for i in np.nonzero(d > thr)[0]]


class SemanticSplitter:
    def __init__(self, embed_model, buffer_size: int = 1, breakpoint_percentile_threshold: float = 95):
        self.embed_model = embed_model
        self.buffer_size = buffer_size
        self.breakpoint_percentile_threshold = breakpoint_percentile_threshold

    def _groups(self, sentences: List[str]) -> List[str]:
        b = self.buffer_size
        return ["".join(sentences[max(0, i - b): i + b + 1]) for i in range(len(sentences))]

    def split_text(self, text: str) -> List[str]:
        sentences = split_sentences(text)
        if len(sentences) <= 1:
            return [text] if text.strip() else []
        groups = self._groups(sentences)
        if hasattr(self.embed_model, "_embed_texts"):
            emb = self.embed_model._embed_texts(groups, getattr(self.embed_model, "text_instruction", ""))
        else:
            emb = torch.tensor(self.embed_model.get_text_embedding_batch(groups), device="cuda")
        dist = adjacent_distances(emb).cpu().tolist()
        cuts = breakpoints_from_distances(dist, self.breakpoint_percentile_threshold)
        chunks, start = [], 0
        for c in cuts:
            chunks.append("".join(sentences[start:c + 1]).strip())
            start = c + 1
        if start < len(sentences):
            chunks.append("".join(sentences[start:]).strip())
        return [c for c in chunks if c]

    def get_nodes_from_documents(self, documents, show_progress: bool = False) -> List[TextNode]:
        nodes = []
        for doc in documents:
            text = doc.get_content() if hasattr(doc, "get_content") else str(doc)
            meta = dict(getattr(doc, "metadata", {}) or {})
            prev: Optional[TextNode] = None
            for chunk in self.split_text(text):
                nd = TextNode(text=chunk, metadata=dict(meta))
                if prev is not None:
                    try:
                        prev.next_id, nd.prev_id = nd.id_, prev.id_
                    except Exception:  # noqa: BLE001 - llama-index nodes keep links in .relationships
                        pass
                nodes.append(nd)
                prev = nd
        return nodes
