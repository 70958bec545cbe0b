"""Hierarchical chunking on the host: the node parser in front of the GPU embedder.

Restates what the reference gets from llama-index at ``src/tensortruth/indexing/builder.py:383-420`` and
``document_index.py:300``: ``HierarchicalNodeParser.from_defaults(chunk_sizes, chunk_overlap)`` = one
sentence-aware splitter per level; every chunk of level ``i`` is split again at level ``i+1``; chunks of one parent
are chained PREVIOUS/NEXT and linked PARENT/CHILD; the leaves (``get_leaf_nodes``) are what gets embedded, all nodes
go to the docstore for auto-merging.  When ``llama_index`` is importable the reference's own parser can be used
unchanged (anything with ``get_nodes_from_documents`` plugs into ``index_builder``); this module is for hosts
without it.  Token counts: llama-index counts with its global tokenizer (tiktoken); here the counter is pluggable and
defaults to a word/punctuation count, so chunk BOUNDARIES are not claimed to match llama-index token for token --
the structure (levels, links, overlap semantics, metadata inheritance) is what the retrieval path depends on.
"""
from __future__ import annotations

import functools
import hashlib
import re
from typing import Callable, Iterable, List, Optional, Sequence

from .schema import TextNode

_TOKEN = re.compile(r"\w+|[^\w\s]", re.UNICODE)
_SENTENCE = re.compile(r"[^.!?\n]+[.!?]*\s*|\n+")
_PARAGRAPH = re.compile(r"\n\s*\n")


@functools.lru_cache(maxsize=1 << 16)
def count_tokens(text: str) -> int:
    # cached: every level of the hierarchy re-splits the text of the level above into (mostly) the same sentences
    return len(_TOKEN.findall(text))


def tokenizer_counter(tk) -> Callable[[str], int]:
    """A chunk-size counter that counts with a real sub-word tokenizer (``tk.encode(text)`` minus its two specials) -- what the
    reference's splitters do with llama-index's global tokenizer (tiktoken): chunk sizes are SUB-WORD token counts
    there, so a 256-token leaf is ~256 sub-word tokens, not 256 words.  Memoised like ``count_tokens``."""
    # Memoised on a DIGEST of the text, for short pieces only: the hierarchy's levels re-count the same sentences (short strings,
    # where the cache pays), while whole chunks and documents are counted once -- an lru_cache keyed on the strings themselves pinned
    # up to 65 536 document-sized texts in memory.
    cache: dict = {}
    LIMIT, MAX_CHARS = 1 << 16, 2048

    def count(text: str) -> int:
        if len(text) > MAX_CHARS:
            return max(0, len(tk.encode(text)) - 2)
        key = hashlib.blake2b(text.encode("utf-8", "surrogatepass"), digest_size=16).digest()     # (16 bytes per entry, not the text)
        hit = cache.get(key)
        if hit is not None:
            return hit
        n = max(0, len(tk.encode(text)) - 2)
        if len(cache) >= LIMIT:
            cache.clear()
        cache[key] = n
        return n

    return count


class SentenceSplitter:
    """Greedy sentence packing: consecutive sentences are joined until ``chunk_size`` tokens would be exceeded; the
    next chunk starts with the trailing sentences of the previous one worth at most ``chunk_overlap`` tokens; a
    sentence longer than ``chunk_size`` is cut at word boundaries (llama-index ``SentenceSplitter`` semantics)."""

    def __init__(self, chunk_size: int = 1024, chunk_overlap: int = 200, tokenizer: Optional[Callable[[str], int]] = None):
        if chunk_overlap >= chunk_size:
            raise ValueError(f"chunk_overlap {chunk_overlap} must be smaller than chunk_size {chunk_size}")
        self.chunk_size, self.chunk_overlap = chunk_size, chunk_overlap
        self._count = tokenizer or count_tokens

    def _pieces(self, text: str):
        """-> (pieces, their token counts): every piece is counted ONCE (the counter is the splitter's inner loop)."""
        out, sizes = [], []
        for para in _PARAGRAPH.split(text):
            for m in _SENTENCE.finditer(para):
                s = m.group(0)
                if not s.strip():
                    continue
                c = self._count(s)
                if c <= self.chunk_size:
                    out.append(s)
                    sizes.append(c)
                    continue
                words, cur, n = re.findall(r"\S+\s*", s), [], 0      # oversized sentence: cut at words
                for w in words:
                    c = self._count(w)
                    if cur and n + c > self.chunk_size:
                        out.append("".join(cur))
                        sizes.append(n)
                        cur, n = [], 0
                    cur.append(w)
                    n += c
                if cur:
                    out.append("".join(cur))
                    sizes.append(n)
            if out and not out[-1].endswith("\n"):
                out[-1] = out[-1] + "\n"      # (a newline is white space: the piece's token count is unchanged)
        return out, sizes

    def split_text(self, text: str) -> List[str]:
        pieces, sizes = self._pieces(text)
        chunks, cur, cur_n, n, i = [], [], [], 0, 0
        while i < len(pieces):
            if cur and n + sizes[i] > self.chunk_size:
                chunks.append("".join(cur).strip())
                keep, keep_n, kn = [], [], 0                           # overlap: trailing pieces of the closed chunk
                for j in range(len(cur) - 1, -1, -1):
                    c = cur_n[j]
                    if kn + c > self.chunk_overlap or kn + c + sizes[i] > self.chunk_size:
                        break
                    keep.insert(0, cur[j])
                    keep_n.insert(0, c)
                    kn += c
                cur, cur_n, n = keep, keep_n, kn
                continue
            cur.append(pieces[i])
            cur_n.append(sizes[i])
            n += sizes[i]
            i += 1
        if cur and "".join(cur).strip():
            chunks.append("".join(cur).strip())
        return [c for c in chunks if c]


class HierarchicalNodeParser:
    def __init__(self, chunk_sizes: Sequence[int], chunk_overlap: int = 20, tokenizer: Optional[Callable[[str], int]] = None):
        if not chunk_sizes or list(chunk_sizes) != sorted(chunk_sizes, reverse=True):
            raise ValueError("chunk_sizes must be non-empty and decreasing, e.g. [2048, 512, 256]")
        self.chunk_sizes = list(chunk_sizes)
        self.splitters = [SentenceSplitter(cs, min(chunk_overlap, cs - 1), tokenizer) for cs in chunk_sizes]

    @classmethod
    def from_defaults(cls, chunk_sizes: Optional[Sequence[int]] = None, chunk_overlap: int = 20, **kw):
        return cls(chunk_sizes or [2048, 512, 128], chunk_overlap, kw.get("tokenizer"))

    def _split(self, parent_text: str, meta: dict, level: int, parent: Optional[TextNode], out: List[TextNode],
               excluded: Sequence[str], excluded_llm: Sequence[str] = ()) -> None:
        chunks = self.splitters[level].split_text(parent_text)
        prev: Optional[TextNode] = None
        for c in chunks:
            nd = TextNode(text=c, metadata=dict(meta))
            nd.excluded_embed_metadata_keys = list(excluded)
            if excluded_llm:
                try:
                    nd.excluded_llm_metadata_keys = list(excluded_llm)
                except Exception:  # noqa: BLE001
                    pass
            if parent is not None:
                nd.parent_id = parent.id_
                parent.child_ids.append(nd.id_)
            if prev is not None:
                prev.next_id, nd.prev_id = nd.id_, prev.id_
            out.append(nd)
            prev = nd
            if level + 1 < len(self.splitters):
                self._split(c, meta, level + 1, nd, out, excluded, excluded_llm)

    def get_nodes_from_documents(self, documents: Iterable, show_progress: bool = False) -> List[TextNode]:
        """Documents (or nodes, as the reference feeds the semantic splitter's output back in, builder.py:415-418)
        -> every node of every level, top level first within a document."""
        out: List[TextNode] = []
        for doc in documents:
            text = doc.get_content() if hasattr(doc, "get_content") else str(doc)
            if not text.strip():
                continue
            meta = dict(getattr(doc, "metadata", {}) or {})
            self._split(text, meta, 0, None, out, getattr(doc, "excluded_embed_metadata_keys", []) or [],
                        getattr(doc, "excluded_llm_metadata_keys", []) or [])
        return out


def get_leaf_nodes(nodes: Iterable[TextNode]) -> List[TextNode]:
    return [n for n in nodes if not getattr(n, "child_ids", None)]


def get_root_nodes(nodes: Iterable[TextNode]) -> List[TextNode]:
    return [n for n in nodes if getattr(n, "parent_id", None) is None]
