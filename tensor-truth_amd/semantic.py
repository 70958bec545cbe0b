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

import os
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .schema import TextNode
from .semantic_host import breakpoints_from_distances, join_chunks, split_sentences  # noqa: F401  (host-only steps, torch-free)


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




def embedding_calls(texts: Sequence[str], max_groups_per_call: int = 65536, first_call: Optional[int] = None) -> List[tuple]:
    """Documents per embedding call of the semantic pass -> [(lo, hi), ...], contiguous, in order, covering every text once.
    A call is bounded by its (estimated) number of sentence groups -- host memory for the strings.  The first calls are SMALL
    and double up to that bound: before the first call's groups are enqueued the GPU has nothing to do, and a call's documents
    are cut on the host while the GPU embeds the NEXT call's groups -- with one big call per window a kernel trace shows the
    GPU idle through all of the window's sentence splitting and all of its cutting (profiles/r03_ingest_busy.log).  The cuts do
    not depend on the partition: thresholds are per document, embeddings do not depend on the batch they travel in."""
    if first_call is None:
        first_call = int(os.environ.get("TT_SEMANTIC_FIRST_CALL", "2048"))
    calls, lo, cap = [], 0, max(1, min(max_groups_per_call, first_call))
    while lo < len(texts):
        hi, budget = lo, cap
        while hi < len(texts) and (hi == lo or budget > 0):
            budget -= max(1, texts[hi].count(".") + texts[hi].count("\n"))
            hi += 1
        calls.append((lo, hi))
        lo = hi
        cap = min(max_groups_per_call, cap * 2)
    return calls

class SemanticSplitter:
    def __init__(self, embed_model, buffer_size: int = 1, breakpoint_percentile_threshold: float = 95):
        self.embed_model = embed_model
        self.buffer_size = buffer_size
        self.breakpoint_percentile_threshold = breakpoint_percentile_threshold

    def _groups(self, sentences: List[str]) -> List[str]:
        b = self.buffer_size
        return ["".join(sentences[max(0, i - b): i + b + 1]) for i in range(len(sentences))]

    def _embed_groups(self, groups: List[str]) -> torch.Tensor:
        if hasattr(self.embed_model, "_embed_texts"):
            return self.embed_model._embed_texts(groups, getattr(self.embed_model, "text_instruction", ""))
        return torch.tensor(self.embed_model.get_text_embedding_batch(groups), device="cuda")

    _join = staticmethod(join_chunks)

    def split_texts(self, texts: Sequence[str]) -> List[List[str]]:
        """Chunks of every text.  The sentence groups of ALL texts are embedded in one pipelined call (batches of
        ``embed_batch_size`` groups on the GPU, tokenization overlapped) and ONE ``tt_adjacent_cosine`` launch gives
        the distances of the whole concatenation; a document's distances are its slice of that vector (the pair
        straddling two documents is skipped) and its percentile threshold is taken over that slice -- the same
        cuts as one call per document (the reference's behaviour), at batch throughput: a 100k-document ingest
        (BASELINE config 5) is ~10^7 sentence groups, not 10^5 small launches."""
        return self._finish_split(self._begin_split(texts))

    def _begin_split(self, texts: Sequence[str]):
        """Host half + ENQUEUE of the device half (group embeddings, adjacent distances): nothing here waits for the GPU."""
        sents = [split_sentences(t) for t in texts]
        spans, groups = [], []
        for ss in sents:
            if len(ss) > 1:
                spans.append((len(groups), len(ss)))
                groups.extend(self._groups(ss))
            else:
                spans.append(None)
        fetch = None
        if groups:
            # the copy back is enqueued HERE, behind this call's own work, with an event of its own: fetching it later
            # waits for this call only, not for whatever was enqueued after it (the next call's forward passes)
            d = adjacent_distances(self._embed_groups(groups))
            host = torch.empty(d.shape, dtype=d.dtype, pin_memory=True)
            host.copy_(d, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(d.device))
            fetch = (host, ev)
        return texts, sents, spans, fetch

    def _finish_split(self, state) -> List[List[str]]:
        texts, sents, spans, fetch = state
        if fetch is not None:
            fetch[1].synchronize()
            dist = fetch[0].numpy()
        else:
            dist = np.zeros(0, np.float32)
        out = []
        for text, ss, span in zip(texts, sents, spans):
            if span is None:
                out.append([text] if text.strip() else [])
                continue
            lo, n = span
            cuts = breakpoints_from_distances(dist[lo:lo + n - 1], self.breakpoint_percentile_threshold)
            out.append(self._join(ss, cuts))
        return out

    def split_text(self, text: str) -> List[str]:
        return self.split_texts([text])[0]

    def get_nodes_from_documents(self, documents, show_progress: bool = False, max_groups_per_call: int = 65536) -> List[TextNode]:
        docs = list(documents)
        texts = [d.get_content() if hasattr(d, "get_content") else str(d) for d in docs]
        nodes: List[TextNode] = []
        calls = embedding_calls(texts, max_groups_per_call)
        # software pipeline: call i + 1's sentence groups are tokenized and their forward passes enqueued BEFORE call i's
        # distances are fetched, so the GPU embeds them while the host cuts call i's documents and builds its nodes
        pending = self._begin_split(texts[calls[0][0]:calls[0][1]]) if calls else None
        for ci, (lo, hi) in enumerate(calls):
            state = pending
            pending = self._begin_split(texts[calls[ci + 1][0]:calls[ci + 1][1]]) if ci + 1 < len(calls) else None
            for doc, chunks in zip(docs[lo:hi], self._finish_split(state)):
                meta = dict(getattr(doc, "metadata", {}) or {})
                prev: Optional[TextNode] = None
                for chunk in chunks:
                    nd = TextNode(text=chunk, metadata=dict(meta))
                    # llama-index node parsers hand the source document's excluded-metadata keys down to every node
                    # (what MetadataMode.EMBED / LLM content is built from)
                    for key in ("excluded_embed_metadata_keys", "excluded_llm_metadata_keys"):
                        val = getattr(doc, key, None)
                        if val:
                            try:
                                setattr(nd, key, list(val))
                            except Exception:  # noqa: BLE001
                                pass
                    if prev is not None:
                        try:
                            prev.next_id, nd.prev_id = nd.id_, prev.id_
                        except Exception:  # noqa: BLE001 - llama-index nodes keep links in .relationships
                            pass
                    nodes.append(nd)
                    prev = nd
        return nodes
