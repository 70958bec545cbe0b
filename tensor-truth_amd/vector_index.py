"""Vector index + base retriever: the exact-scan stand-in for ChromaVectorStore +
VectorIndexRetriever (reference: ``src/tensortruth/rag_engine.py:628-645``,
``indexing/builder.py:424-444``, ``document_index.py:306-327,478-581``).

Storage: one row-major bf16 matrix in HBM (capacity-doubling, ``n`` live rows) + host side
tables (leaf nodes by row, a docstore of all nodes by id for auto-merging).  On disk, under
the reference's ``indexes/{model_id}/{module}/`` layout: ``corpus.bf16`` (raw rows),
``nodes.json`` (leaf ids in row order + docstore) and the reference's ``index_metadata.json``.
"""
from __future__ import annotations

import json
import logging
import math
import os
import re
import threading
from datetime import datetime, timezone
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from . import scan as _scan
from .coalesce import Coalescer
from .schema import MetadataMode, NodeWithScore, TextNode, as_query_bundle

logger = logging.getLogger(__name__)

INDEX_METADATA_FILENAME = "index_metadata.json"   # reference: indexing/metadata.py:17
INDEX_VERSION = "1.0"


def sanitize_model_id(model_name: str) -> str:
    """"BAAI/bge-m3" -> "bge-m3": the directory level of ``indexes/{model_id}/{module}/`` (reference:
    indexing/metadata.py:22-52: last path component, lower case, anything outside [a-z0-9-_.] -> "-", runs of
    dashes collapsed, leading/trailing dashes dropped)."""
    name = model_name.split("/")[-1].lower()
    name = re.sub(r"[^a-z0-9\-_.]", "-", name)
    return re.sub(r"-+", "-", name).strip("-")


class HipVectorIndex:
    def __init__(self, dim: int, device=None, embed_model=None, score_mode: str = "chroma"):
        dev = torch.device("cuda" if device in (None, "cuda") else device)
        if dev.type != "cuda":
            raise RuntimeError("HipVectorIndex needs a HIP device; tensor_truth_amd has no CPU path")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        if dim % 128 or dim > 1024:
            raise ValueError(f"dim {dim}: the scan kernels take multiples of 128 up to 1024")
        self.dim, self.device = dim, dev
        self.embed_model = embed_model
        self.score_mode = score_mode           # "chroma": exp(-(2-2cos)) like the reference; "cosine"
        self._mat = torch.empty((0, dim), dtype=torch.bfloat16, device=dev)
        self.n = 0
        self.leaf_ids: List[str] = []          # row -> node id
        # node id -> int32 body ids of the leaf's EMBED content as the embedder tokenised it (filled by add(token_ids=...): the
        # worker-process ingest with keep_leaf_token_ids; persisted as leaf_tokens.<generation>.npz)
        self.leaf_token_ids: Optional[Dict[str, np.ndarray]] = None
        # (tokenizer signature, text instruction) the kept ids were produced with: set when the first ids are stored, persisted with
        # them, and what token_source() reports -- never the values of whatever embed_model the index is later loaded next to
        self.leaf_token_origin: Optional[tuple] = None
        self.docstore: Dict[str, TextNode] = {}  # every node (leaves + parents), for auto-merging
        self.ref_docs: Dict[str, List[str]] = {}  # source document id -> ids of all its nodes (docstore ref_doc_info)
        self._lock = threading.RLock()
        self._version = 0                      # bumped by every mutation (HipIndexGroup repacks on change)
        self._dead = 0                         # tombstoned rows (NaN-filled, leaf id None), see delete()
        self._row_of: Optional[Dict[str, int]] = None   # leaf id -> row, built on the first delete
        self._written: Optional[torch.cuda.Event] = None   # recorded behind the last device write to the matrix
        # fp8 shadow of the matrix (scan.ScanShadow): an EXACT prefilter for batches of <= 4 queries over >= 1 M rows -- a lone caller's
        # scan reads half the bytes and returns the same bits.  Built on first use, extended as rows are appended, dropped (and rebuilt
        # on the next search) when rows are rewritten in place or the matrix is replaced.  + 50 % of the matrix's HBM; TT_SCAN_SHADOW=0: off
        self.fp8_shadow = os.environ.get("TT_SCAN_SHADOW", "1") != "0"
        self._shadow = None

    def _mark_written(self) -> None:
        """Searches may run on another stream than the one that wrote the matrix (the retrievers' own stream, below):
        every device write is followed by an event the searching stream waits for."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._written = ev

    # ---- build / mutate ------------------------------------------------------------------------
    def _reserve(self, n_new: int) -> None:
        need = self.n + n_new
        if need > self._mat.shape[0]:
            cap = max(need, 2 * self._mat.shape[0], 1024)
            grown = torch.empty((cap, self.dim), dtype=torch.bfloat16, device=self.device)
            grown[: self.n] = self._mat[: self.n]
            self._mat = grown

    def add(self, nodes: Sequence[TextNode], embeddings=None, show_progress: bool = False, token_ids=None) -> List[str]:
        """Insert leaf nodes (embedding them with ``embed_model`` unless ``embeddings`` is given).
        Mirrors ``VectorStoreIndex(leaf_nodes, ...)`` / ``index.insert_nodes`` (text embedded with
        MetadataMode.EMBED, ``indexing/builder.py:437-442``, ``document_index.py:527``).
        ``token_ids`` (optional, one int array per node, specials included, as the embedder's tokenizer produced them for the node's
        EMBED content): kept in ``leaf_token_ids`` so that a reranker with the same tokenizer never tokenises these passages again
        (``HipSentenceTransformerRerank.attach_token_source``); a sequence that hit the embedder's length limit is not kept."""
        if not nodes:
            return []
        if embeddings is None:
            if self.embed_model is None:
                raise ValueError("no embeddings given and the index has no embed_model")
            texts = [n.get_content(metadata_mode=MetadataMode.EMBED) for n in nodes]
            if hasattr(self.embed_model, "_embed_texts"):
                emb = self.embed_model._embed_texts(texts, getattr(self.embed_model, "text_instruction", ""))
            else:
                emb = torch.tensor(self.embed_model.get_text_embedding_batch(texts, show_progress=show_progress))
        else:
            emb = embeddings if torch.is_tensor(embeddings) else torch.tensor(np.asarray(embeddings, dtype=np.float32))
        if emb.shape != (len(nodes), self.dim):
            raise ValueError(f"embeddings {tuple(emb.shape)} != ({len(nodes)}, {self.dim})")
        emb = emb.to(self.device, dtype=torch.float32)
        emb = emb / emb.norm(dim=1, keepdim=True).clamp_min(1e-12)
        with self._lock:
            self._reserve(len(nodes))
            self._mat[self.n:self.n + len(nodes)] = emb.to(torch.bfloat16)
            self._mark_written()
            for j, nd in enumerate(nodes):
                self.leaf_ids.append(nd.id_)
                self.docstore[nd.id_] = nd
                if self._row_of is not None:
                    self._row_of[nd.id_] = self.n + j
            if token_ids is not None:
                origin = self._embedder_token_origin()
                if self.leaf_token_ids and self.leaf_token_origin is not None and origin != self.leaf_token_origin:
                    # ids of another tokenizer / instruction than the ones already kept: one table cannot describe both
                    logger.warning("index: leaf token ids dropped (the embedder's tokenizer or text instruction changed between adds)")
                    self.leaf_token_ids = None
                if self.leaf_token_ids is None:
                    self.leaf_token_ids = {}
                self.leaf_token_origin = origin
                limit = getattr(self.embed_model, "max_length", None)
                for nd, ids in zip(nodes, token_ids):
                    if limit is None or len(ids) < limit:
                        self.leaf_token_ids[nd.id_] = np.asarray(ids[1:-1], dtype=np.int32)       # body only: the reranker adds its own specials
                    else:
                        self.leaf_token_ids.pop(nd.id_, None)
            elif self.leaf_token_ids:
                # a node (re-)added WITHOUT ids (in-process build, insert_nodes): ids kept for an earlier version of this node id
                # would make the reranker score text the node no longer has
                for nd in nodes:
                    self.leaf_token_ids.pop(nd.id_, None)
            self.n += len(nodes)
            self._version += 1
        return [nd.id_ for nd in nodes]

    def _embedder_token_origin(self):
        """(tokenizer signature, text instruction) of the index's embed_model, or None when it has no tokenizer to ask."""
        em = self.embed_model
        if em is None or not hasattr(em, "_tokenizer"):
            return None
        from .tokenization import tokenizer_signature

        return tokenizer_signature(em._tokenizer), getattr(em, "text_instruction", "") or ""

    def token_source(self):
        """-> (callable node id -> body ids or None, tokenizer signature, text instruction) for ``attach_token_source``, or None.
        Signature and instruction are the ones the ids were PRODUCED with (recorded at ``add``, persisted beside the ids) -- an index
        loaded next to another embedder still reports its ingest-time tokenizer, and ids of unknown origin are never offered."""
        if not self.leaf_token_ids or self.leaf_token_origin is None:
            return None
        return self.leaf_token_ids.get, self.leaf_token_origin[0], self.leaf_token_origin[1]

    def add_to_docstore(self, nodes: Iterable[TextNode]) -> None:
        """Parents of the hierarchy (``storage_context.docstore.add_documents``, builder.py:430)."""
        with self._lock:
            for nd in nodes:
                self.docstore.setdefault(nd.id_, nd)

    def delete(self, node_ids: Iterable[str]) -> int:
        """Remove leaves by id (``document_index.py:568``).  The rows become TOMBSTONES: they are overwritten with
        NaN, so their dot product with any query is NaN, which the scan's threshold compare and the selection's
        key both reject -- a deleted row can never be returned and the scan kernels need no mask.  One small
        scatter per call; rows are compacted (order preserved) once a quarter of the matrix is dead, and before
        persisting."""
        drop = set(node_ids)
        with self._lock:
            if self._row_of is None:
                self._row_of = {nid: i for i, nid in enumerate(self.leaf_ids) if nid is not None}
            rows = [self._row_of.pop(nid) for nid in drop if nid in self._row_of]
            if rows:
                idx = torch.tensor(rows, dtype=torch.long, device=self.device)
                self._mat[: self.n].index_fill_(0, idx, float("nan"))
                self._mark_written()
                self._shadow = None            # rows rewritten in place: the fp8 shadow is rebuilt by the next lone caller's search
                for r in rows:
                    self.leaf_ids[r] = None
                self._dead += len(rows)
                self._version += 1
                if self._dead > max(1024, self.n // 4):
                    self._compact()
            for nid in drop:
                self.docstore.pop(nid, None)
                if self.leaf_token_ids is not None:
                    self.leaf_token_ids.pop(nid, None)
        return len(rows)

    def _compact(self) -> None:
        """Drop the tombstones.  COPY-ON-WRITE: a new matrix and a new ``leaf_ids`` list replace the old ones, which
        are never moved or rewritten in place -- a search that took its snapshot before the compaction (``snapshot()``)
        keeps scanning, and mapping rows through, a consistent pair (also when the old matrix was a view of a
        ``HipIndexGroup``'s packed matrix)."""
        with self._lock:
            if not self._dead:
                return
            keep = [i for i, nid in enumerate(self.leaf_ids) if nid is not None]
            idx = torch.tensor(keep, dtype=torch.long, device=self.device)
            self._mat = self._mat[: self.n].index_select(0, idx) if keep else torch.empty(
                (0, self.dim), dtype=torch.bfloat16, device=self.device)
            self._mark_written()
            self.leaf_ids = [self.leaf_ids[i] for i in keep]
            self.n, self._dead, self._row_of = len(keep), 0, None
            self._version += 1

    @property
    def num_live(self) -> int:
        return self.n - self._dead

    @property
    def matrix(self) -> torch.Tensor:
        return self._mat[: self.n]

    # ---- query -------------------------------------------------------------------------------------
    def snapshot(self):
        """(matrix view of the live rows, leaf_ids list) taken under the lock.  Appends never touch rows below ``n``
        (growth reallocates), deletes only overwrite rows with NaN and set their ``leaf_ids`` entry to None, and
        compaction replaces both objects (copy-on-write), so the pair stays mutually consistent for as long as the
        caller holds it -- the scan and the row -> node mapping run WITHOUT the index lock, and the reference's
        eight retriever threads (rag_engine.py:392,420) search one index concurrently."""
        with self._lock:
            return self._mat[: self.n], self.leaf_ids

    def _unit_bf16(self, query_emb: torch.Tensor) -> torch.Tensor:
        q = query_emb.to(self.device, dtype=torch.float32)
        return (q / q.norm(dim=1, keepdim=True).clamp_min(1e-12)).to(torch.bfloat16).contiguous()

    def _shadow_for(self, mat: torch.Tensor, n_queries: int):
        """The fp8 shadow that mirrors exactly the rows of ``mat`` (a snapshot's matrix view), or None: too few rows, too many
        queries, switched off, or ``mat`` is no longer the index's current matrix (an older snapshot: it takes the bf16 pass)."""
        if not self.fp8_shadow or not (0 < n_queries <= _scan.ScanShadow.MAX_QUERIES) or mat.shape[0] < _scan.ScanShadow.MIN_ROWS:
            return None
        with self._lock:
            if mat.data_ptr() != self._mat.data_ptr() or mat.shape[0] > self.n:
                return None
            sh = self._shadow
            if self._written is not None and (sh is None or sh.rows < self.n):
                # the build below reads rows that may be newer than the snapshot (and than the event the searching stream waited for)
                torch.cuda.current_stream(self.device).wait_event(self._written)
            if sh is None or sh.base_ptr != self._mat.data_ptr() or sh.cap_rows < self.n:
                sh = self._shadow = _scan.ScanShadow(self._mat[: self.n], cap_rows=self._mat.shape[0])
                sh.base_ptr = self._mat.data_ptr()
            elif sh.rows < self.n:
                sh.extend(self._mat, self.n)                     # rows appended since
            return sh if sh.rows >= mat.shape[0] else None

    def search(self, query_emb: torch.Tensor, k: int, snapshot=None):
        """query_emb fp32/bf16 [Q, D] -> (scores [Q,k] fp32 cosine, rows [Q,k] int32)."""
        mat, _ = snapshot if snapshot is not None else self.snapshot()
        return _scan.scan_topk(mat, self._unit_bf16(query_emb), k, shadow=self._shadow_for(mat, query_emb.shape[0]))

    def search_host(self, query_emb: torch.Tensor, k: int, snapshot=None):
        """``search`` with the hits on the host (CPU tensors): one copy back, one sync per scan batch (``scan.scan_topk_host``)."""
        mat, _ = snapshot if snapshot is not None else self.snapshot()
        return _scan.scan_topk_host(mat, self._unit_bf16(query_emb), k, shadow=self._shadow_for(mat, query_emb.shape[0]))

    def node_score(self, cos: float) -> float:
        return math.exp(-(2.0 - 2.0 * cos)) if self.score_mode == "chroma" else cos

    def as_retriever(self, similarity_top_k: int = 10, coalesce: bool = True, max_batch: int = 64,
                     max_wait_s: float = 0.0, **_kw) -> "HipVectorRetriever":
        return HipVectorRetriever(self, similarity_top_k, coalesce=coalesce, max_batch=max_batch, max_wait_s=max_wait_s)

    # ---- persistence ---------------------------------------------------------------------------------
    def persist(self, persist_dir: str, embedding_model: Optional[str] = None, chunk_sizes=None,
                chunking_strategy: Optional[str] = None, chunk_overlap: Optional[int] = None) -> None:
        """Write ``corpus.bf16`` + ``nodes.json`` + ``index_metadata.json``.  Every file is written under a temporary
        name and moved into place with ``os.replace`` (atomic within a directory), ``nodes.json`` LAST: it carries the
        row count and a generation stamp that ``corpus.bf16``'s name is derived from, so a crash or a concurrent
        ``load`` between the writes sees either the previous complete index or the new one, never a row-count
        mismatch (``HipDocumentIndex`` persists after every add / remove, document_index.py:478-581)."""
        os.makedirs(persist_dir, exist_ok=True)
        # One persist per directory at a time -- across the threads of this process (a lock per real path) and across
        # processes (flock on a lock file): the snapshot, the choice of the generation, the writes and the clean-up of
        # older generations happen inside, so two overlapping persists (HipDocumentIndex persists after every add /
        # remove) can neither pick the same generation nor publish an older snapshot over a newer one, and nodes.json
        # never names a matrix another call has removed.
        with _persist_guard(persist_dir):
            with self._lock:
                self._compact()                    # tombstones are not written
                mat_host = self.matrix.cpu().view(torch.int16).numpy()
                leaf_ids = list(self.leaf_ids)
                nodes = {nid: _node_to_dict(nd) for nid, nd in self.docstore.items()}
                ref_docs = {k: list(v) for k, v in self.ref_docs.items()}
                n_rows = self.n
                tok = None
                if self.leaf_token_ids:        # row-aligned CSR of the kept leaf token ids (rows without ids: length 0)
                    arrs = [self.leaf_token_ids.get(nid) for nid in leaf_ids]
                    lens = np.fromiter((0 if a is None else len(a) for a in arrs), dtype=np.int32, count=len(arrs))
                    flat = np.concatenate([a for a in arrs if a is not None and len(a)]) if int(lens.sum()) else np.zeros(0, np.int32)
                    tok = (flat.astype(np.int32), lens)
                tok_origin = self.leaf_token_origin
            gen = _next_generation(persist_dir)
            corpus_name = f"corpus.{gen}.bf16"
            if tok is not None and tok_origin is None:
                logger.warning("index: leaf token ids not persisted (the tokenizer they came from is unknown)")
                tok = None
            tokens_name = f"leaf_tokens.{gen}.npz" if tok is not None else None

            def _atomic(name: str, write) -> None:
                tmp = os.path.join(persist_dir, f".{name}.tmp.{os.getpid()}.{threading.get_ident()}")
                write(tmp)
                os.replace(tmp, os.path.join(persist_dir, name))

            _atomic(corpus_name, lambda t: mat_host.tofile(t))
            if tok is not None:
                def _write_tokens(t):
                    with open(t, "wb") as f:
                        np.savez(f, flat=tok[0], lens=tok[1])
                _atomic(tokens_name, _write_tokens)
            # the reference's index_metadata.json (indexing/metadata.py:103-146) + what this store adds
            model = embedding_model or getattr(self.embed_model, "model_name", None)
            meta = {"embedding_model": model, "embedding_model_id": sanitize_model_id(model) if model else None,
                    "created_at": datetime.now(timezone.utc).isoformat(), "index_version": INDEX_VERSION,
                    "chunk_sizes": list(chunk_sizes) if chunk_sizes is not None else None, "chunk_overlap": chunk_overlap,
                    "chunking_strategy": chunking_strategy,
                    "embedding_dim": self.dim, "num_vectors": n_rows, "store": f"tensor_truth_amd/{corpus_name}"}

            def _dump(obj):
                def w(t):
                    with open(t, "w") as f:
                        json.dump(obj, f)
                return w

            _atomic(INDEX_METADATA_FILENAME, _dump(meta))
            _atomic("nodes.json", _dump({"dim": self.dim, "leaf_ids": leaf_ids, "nodes": nodes, "ref_docs": ref_docs,
                                         "corpus_file": corpus_name, "generation": gen, "leaf_tokens_file": tokens_name,
                                         # what the ids were tokenised with: load() offers them only to the same tokenizer / instruction
                                         "leaf_tokens_tokenizer": tok_origin[0] if tokens_name else None,
                                         "leaf_tokens_instruction": tok_origin[1] if tokens_name else None}))
            for name in os.listdir(persist_dir):     # LOWER generations only: unreferenced once nodes.json has moved
                g = _generation_of(name)
                if g is None and name.startswith("leaf_tokens.") and name.endswith(".npz"):
                    try:
                        g = int(name[len("leaf_tokens."):-len(".npz")])
                    except ValueError:
                        g = None
                if g is not None and g < gen:
                    try:
                        os.remove(os.path.join(persist_dir, name))
                    except OSError:
                        pass

    @classmethod
    def load(cls, persist_dir: str, device=None, embed_model=None, score_mode: str = "chroma") -> "HipVectorIndex":
        blob, raw = _read_persisted(persist_dir)
        idx = cls(blob["dim"], device, embed_model, score_mode)
        idx._mat = torch.from_numpy(np.array(raw, copy=True)).view(torch.bfloat16).to(idx.device).contiguous()
        idx._mark_written()
        idx.n = raw.shape[0]
        idx.leaf_ids = list(blob["leaf_ids"])
        idx.ref_docs = {k: list(v) for k, v in (blob.get("ref_docs") or {}).items()}
        idx.docstore = {nid: _node_from_dict(nid, d) for nid, d in blob["nodes"].items()}
        tname = blob.get("leaf_tokens_file")
        if tname and os.path.exists(os.path.join(persist_dir, tname)):
            # the leaves' token ids kept at ingest (build_index(keep_leaf_token_ids=True)): usable by a reranker whose tokenizer is
            # THIS embed_model's -- token_source() reports the signature of the tokenizer the caller loads the index with
            sig, instr = blob.get("leaf_tokens_tokenizer"), blob.get("leaf_tokens_instruction")
            mine = idx._embedder_token_origin()
            if sig is None or instr is None:
                logger.warning("index %s: leaf token ids ignored (persisted without the tokenizer signature they were made with)", persist_dir)
            elif mine is not None and mine != (sig, instr):
                # another tokenizer (a hash stand-in, another checkpoint) or another text instruction than at ingest: a reranker that
                # matches THIS embedder would score foreign ids
                logger.warning("index %s: leaf token ids ignored (tokenised with %s / instruction %r at ingest, loaded next to %s / %r)",
                               persist_dir, sig, instr, mine[0], mine[1])
            else:
                with np.load(os.path.join(persist_dir, tname)) as z:
                    flat, lens = z["flat"], z["lens"]
                if len(lens) == len(idx.leaf_ids):
                    ends = np.cumsum(lens)
                    idx.leaf_token_ids = {nid: flat[e - n:e] for nid, e, n in zip(idx.leaf_ids, ends, lens) if n > 0 and nid is not None}
                    idx.leaf_token_origin = (sig, instr)
        return idx


_LINK_KEYS = ("parent_id", "child_ids", "prev_id", "next_id")
# what get_content(MetadataMode.EMBED / LLM) depends on besides text and metadata: the reference's leaves inherit
# SimpleDirectoryReader's excluded keys (file_name, file_size, dates ...), and the reranker scores
# ``get_content(EMBED)`` of the RETRIEVED node (rerank.py) -- a hit that lost them would be scored on different text
# than the reference scores, and than what was embedded at index time.
_CONTENT_KEYS = ("excluded_embed_metadata_keys", "excluded_llm_metadata_keys", "metadata_template",
                 "metadata_separator", "text_template")


def _node_to_dict(nd) -> dict:
    d = {"text": nd.text, "metadata": nd.metadata}
    for key in _LINK_KEYS:
        v = getattr(nd, key, None)
        d[key] = list(v or []) if key == "child_ids" else v
    for key in _CONTENT_KEYS:
        v = getattr(nd, key, None)
        if v:
            d[key] = list(v) if isinstance(v, (list, tuple)) else v
    return d


def _copy_node_attrs(src, dst, keys) -> None:
    for key in keys:
        if hasattr(src, key):
            v = getattr(src, key)
            try:
                setattr(dst, key, list(v) if isinstance(v, (list, tuple)) else v)
            except Exception:  # noqa: BLE001 - LlamaIndex nodes keep links in .relationships
                pass


def _node_from_dict(nid: str, d: dict):
    nd = TextNode(text=d["text"], id_=nid, metadata=d["metadata"])
    for key in _LINK_KEYS + _CONTENT_KEYS:
        if key in d or key in _LINK_KEYS:
            v = d.get(key)
            try:
                setattr(nd, key, list(v or []) if key in ("child_ids", "excluded_embed_metadata_keys",
                                                          "excluded_llm_metadata_keys") else v)
            except Exception:  # noqa: BLE001
                pass
    return nd


def _generation_of(name: str):
    """corpus.<gen>.bf16 -> gen, anything else -> None."""
    if name.startswith("corpus.") and name.endswith(".bf16"):
        try:
            return int(name[len("corpus."):-len(".bf16")])
        except ValueError:
            return None
    return None


def _next_generation(persist_dir: str) -> int:
    """One more than anything published or lying around (called under the directory's persist guard)."""
    gen = 0
    try:
        with open(os.path.join(persist_dir, "nodes.json")) as f:
            gen = int(json.load(f).get("generation", 0))
    except (OSError, ValueError):
        pass
    try:
        for name in os.listdir(persist_dir):
            g = _generation_of(name)
            if g is not None:
                gen = max(gen, g)
    except OSError:
        pass
    return gen + 1


_PERSIST_LOCKS: dict = {}
_PERSIST_LOCKS_GUARD = threading.Lock()


class _persist_guard:
    """Serialises persist() per directory: a process-wide lock per real path + flock on ``.persist.lock`` in it."""

    def __init__(self, persist_dir: str):
        self.path = os.path.realpath(persist_dir)
        with _PERSIST_LOCKS_GUARD:
            self.lock = _PERSIST_LOCKS.setdefault(self.path, threading.Lock())
        self.fh = None

    def __enter__(self):
        self.lock.acquire()
        try:
            import fcntl

            self.fh = open(os.path.join(self.path, ".persist.lock"), "a+")
            fcntl.flock(self.fh, fcntl.LOCK_EX)
        except (ImportError, OSError):          # no flock on this filesystem: the in-process lock still holds
            if self.fh is not None:
                self.fh.close()
                self.fh = None
        return self

    def __exit__(self, *exc):
        try:
            if self.fh is not None:
                import fcntl

                fcntl.flock(self.fh, fcntl.LOCK_UN)
                self.fh.close()
        finally:
            self.lock.release()
        return False


def _read_persisted(persist_dir: str, rows=None):
    """-> (nodes.json blob, int16 [n, dim] array of the bf16 rows).  ``rows``: a slice, or a function of the row count
    returning one -- only that row range is read (the file is memory-mapped)."""
    for attempt in range(3):
        with open(os.path.join(persist_dir, "nodes.json")) as f:
            blob = json.load(f)
        path = os.path.join(persist_dir, blob.get("corpus_file", "corpus.bf16"))
        n, dim = len(blob["leaf_ids"]), blob["dim"]
        try:
            if os.path.getsize(path) != n * dim * 2:
                raise ValueError(f"{path} and nodes.json disagree on the number of rows")
            raw = np.memmap(path, dtype=np.int16, mode="r", shape=(n, dim)) if n else np.zeros((0, dim), np.int16)
            break
        except FileNotFoundError:
            # a persist() published a newer generation and removed this one between our two opens: nodes.json has
            # moved on too -- read it again
            if attempt == 2:
                raise
    if callable(rows):
        rows = rows(n)
    return blob, (raw[rows] if rows is not None else raw)


class HipVectorRetriever:
    """``index.as_retriever(similarity_top_k=...)`` (rag_engine.py:639): query string ->
    ``List[NodeWithScore]`` sorted by score desc, at most ``similarity_top_k`` long.  Thread-safe:
    the reference calls it from up to 8 executor threads (rag_engine.py:420-424) and from
    ``run_in_executor`` request threads (api/routes/chat.py:367-374), one query per call.

    ``coalesce`` (default on): concurrent ``retrieve`` calls are merged by a ``coalesce.Coalescer`` into ONE
    query-embedding batch and ONE scan (up to ``max_batch`` queries share a pass over the corpus); a lone caller
    runs immediately as a batch of one.  Results are identical to serial calls: embeddings and scan results do not
    depend on the batch a query travels in (tests/test_coalesce_gpu.py)."""

    def __init__(self, index: HipVectorIndex, similarity_top_k: int = 10, coalesce: bool = True, max_batch: int = 64,
                 max_wait_s: float = 0.0):
        self.index = index
        self.similarity_top_k = similarity_top_k
        self._front = Coalescer(self._retrieve_batch, max_batch, max_wait_s) if coalesce else None
        self._stream: Optional[torch.cuda.Stream] = None

    def _gpu_stream(self):
        """The query embedding + scan of a batch run on the retriever's OWN high-priority stream (``TT_RETRIEVE_STREAM=0`` puts them
        back on the caller's).  On the stream the reranker uses, a retrieval batch queues behind whatever rerank batch is running
        (tens of ms), its callers come back late, and the GPU then idles while THEIR rerank batch is tokenised and packed: one
        ~8 ms gap per scan batch, 11 % of the time, in a kernel trace of the plugin-surface leg -- 95 -> 104 queries/s with the
        own stream.  Two streams mean scans run BESIDE encoder forwards; the results stay bit-identical to the one-stream ones
        (tests/test_concurrent_streams_gpu.py) now that no kernel holds the packed-f32 form that misbehaves beside another
        kernel's MFMAs (profiles/r03_pk_mfma_hazard.log; tests/test_lib_abi.py checks the disassembly).
        Several PROCESSES sharing one GPU should turn it off: their streams oversubscribe the hardware queues (bench.py, one-device mode)."""
        if os.environ.get("TT_RETRIEVE_STREAM", "1") == "0":
            return None
        if not _lib.isa_checked():
            # the library was not (or not successfully) disassembled by csrc/check_isa.py after its link: two-stream bit
            # reproducibility is not established for this binary -- stay on the caller's stream
            if not getattr(HipVectorRetriever, "_warned_isa", False):
                HipVectorRetriever._warned_isa = True
                logger.warning("libtt_hip.so carries no valid ISA-check stamp (csrc/check_isa.py): retrieval stays on the caller's stream")
            return None
        if self._stream is None:
            dev = self.index.device
            self._stream = torch.cuda.Stream(device=dev, priority=-1)
        return self._stream

    def retrieve(self, query) -> List[NodeWithScore]:
        qb = as_query_bundle(query)
        if self.index.num_live == 0:
            return []
        if self._front is not None:
            return self._front.submit(qb)
        return self._retrieve_batch([qb])[0]

    def _query_matrix(self, bundles) -> torch.Tensor:
        """One embedding per bundle, fp32 [B, D]: the precomputed ``QueryBundle.embedding`` or the mean over the
        bundle's ``embedding_strs`` (VectorIndexRetriever -> get_agg_embedding_from_queries); all
        strings of all bundles go through the encoder in one batch."""
        idx = self.index
        strs, owner = [], []
        for i, qb in enumerate(bundles):
            if getattr(qb, "embedding", None) is None:
                ss = list(qb.embedding_strs)
                strs += ss
                owner += [i] * len(ss)
        rows: List[Optional[torch.Tensor]] = [None] * len(bundles)
        if strs:
            em = idx.embed_model
            if em is None:
                raise ValueError("retriever needs an embed_model or a QueryBundle with an embedding")
            if hasattr(em, "query_embedding_device"):
                E = em.query_embedding_device(strs).to(idx.device, dtype=torch.float32)
            else:
                E = torch.tensor([em.get_agg_embedding_from_queries([s]) for s in strs], dtype=torch.float32,
                                 device=idx.device)
            if len(strs) == len(set(owner)):          # one string per bundle (the reference's case): no averaging
                for j, i in enumerate(owner):
                    rows[i] = E[j]
            else:
                for i in sorted(set(owner)):
                    sel = [j for j, o in enumerate(owner) if o == i]
                    rows[i] = E[sel].mean(dim=0)
        for i, qb in enumerate(bundles):
            if rows[i] is None:
                rows[i] = torch.tensor(qb.embedding, dtype=torch.float32, device=idx.device)
        return torch.stack(rows, 0)

    def _retrieve_batch(self, bundles) -> List[List[NodeWithScore]]:
        idx = self.index
        snap = idx.snapshot()
        mat, leaf_ids = snap
        k = min(self.similarity_top_k, max(idx.num_live, 0), mat.shape[0])
        if k < 1:
            return [[] for _ in bundles]
        stream = self._gpu_stream()
        if stream is None:
            scores, rows = idx.search_host(self._query_matrix(bundles), k, snapshot=snap)
            scores, rows = scores.tolist(), rows.tolist()
        else:
            written = getattr(idx, "_written", None)        # (read after the snapshot: covers every row the snapshot holds)
            with torch.cuda.stream(stream):
                if written is not None:
                    stream.wait_event(written)
                scores, rows = idx.search_host(self._query_matrix(bundles), k, snapshot=snap)     # waits for THIS stream only
                scores, rows = scores.tolist(), rows.tolist()
        return [self.nodes_from_hits(s, r, leaf_ids) for s, r in zip(scores, rows)]

    def nodes_from_hits(self, scores: Sequence[float], rows: Sequence[int], leaf_ids=None) -> List[NodeWithScore]:
        """(cosine, row) pairs of this retriever's index -> NodeWithScore list (padding rows < 0 skipped).
        ``leaf_ids``: the row -> id list of the snapshot the rows were found in (default: the index's current one);
        a row deleted since (id None / node gone from the docstore) is skipped."""
        idx = self.index
        ids = idx.leaf_ids if leaf_ids is None else leaf_ids
        out = []
        for s, r in zip(scores, rows):
            if r < 0 or r >= len(ids):
                continue
            src = idx.docstore.get(ids[r]) if ids[r] is not None else None
            if src is None:
                continue
            # a fresh node per hit with its own metadata dict: callers mutate it
            # (_source_index tagging, rag_engine.py:432-450) and may run concurrently
            node = TextNode(text=src.text, id_=src.id_, metadata=dict(src.metadata))
            _copy_node_attrs(src, node, _LINK_KEYS + _CONTENT_KEYS)
            out.append(NodeWithScore(node=node, score=float(idx.node_score(s))))
        return out

    _retrieve = retrieve


class HipIndexGroup:
    """Several module indexes packed into ONE matrix in HBM, searched with one pass
    (``tt_scan_topk_segmented``) instead of one search per module on a thread pool
    (reference: ``rag_engine.py:420-424``).

    Packing is zero-copy afterwards: every member index's matrix becomes a view of its row range in
    the group matrix.  A member that is mutated (``add`` / ``delete``) bumps its version and the group
    repacks on the next search."""

    def __init__(self, indexes: Sequence[HipVectorIndex]):
        if not indexes:
            raise ValueError("HipIndexGroup needs at least one index")
        if len({(ix.dim, ix.device) for ix in indexes}) != 1:
            raise ValueError("grouped indexes must share embedding width and device")
        self.indexes = list(indexes)
        self.dim, self.device = indexes[0].dim, indexes[0].device
        self._lock = threading.RLock()
        self._stamp = None
        self._mat = None
        self._leaf_ids: List[list] = []
        self.offsets: List[int] = []
        # fp8 shadows of the large modules (scan.ScanShadow, one per module, built on a lone caller's first search of this packing):
        # <= 4 queries then take one exact-prefilter pass per module instead of the dense segmented pass -- same bits, half the bytes
        self.fp8_shadow = os.environ.get("TT_SCAN_SHADOW", "1") != "0"
        self._seg_shadows: Dict[int, object] = {}

    def _pack(self) -> None:
        for ix in self.indexes:
            ix._lock.acquire()
        try:
            stamp = tuple((ix._version, ix.n) for ix in self.indexes)
            if stamp == self._stamp:
                return
            offs = [0]
            for ix in self.indexes:
                offs.append(offs[-1] + ix.n)
            mat = torch.empty((offs[-1], self.dim), dtype=torch.bfloat16, device=self.device)
            for ix, lo in zip(self.indexes, offs):
                mat[lo:lo + ix.n] = ix._mat[: ix.n]
                ix._mat = mat[lo:lo + ix.n]
                ix._mark_written()
            # the row -> id lists that belong to THIS packed matrix (a member's compaction replaces its list and
            # its matrix copy-on-write and bumps its version, so the pair below stays consistent until the repack)
            self._mat, self.offsets, self._stamp = mat, offs, stamp
            self._leaf_ids = [ix.leaf_ids for ix in self.indexes]
            self._seg_shadows = {}               # (they mirrored the previous packing)
        finally:
            for ix in self.indexes:
                ix._lock.release()

    def _snapshot_for(self, q: torch.Tensor, k: int):
        """Under the group lock: pack if a member changed, then (matrix, offsets, id lists, per-module shadows or None).  Shadows are
        used -- and built, on the first such call of a packing -- when the batch is a lone caller's (<= 4 queries) and at least one
        module is large enough for the prefilter to pay."""
        S = _scan.ScanShadow
        with self._lock:
            self._pack()
            mat, offs, ids = self._mat, list(self.offsets), list(self._leaf_ids)
            shadows = None
            big = [m for m, (lo, hi) in enumerate(zip(offs, offs[1:])) if hi - lo >= max(S.MIN_ROWS, 128 * k)]
            if self.fp8_shadow and 0 < q.shape[0] <= S.MAX_QUERIES and big:
                for m in big:
                    if m not in self._seg_shadows:
                        self._seg_shadows[m] = S(mat[offs[m]:offs[m + 1]])
                shadows = dict(self._seg_shadows)
        return mat, offs, ids, shadows

    @staticmethod
    def _scan_modules(mat, offs, q, k, shadows) -> torch.Tensor:
        """A lone caller over large modules: one pass per module -- through its shadow where it has one, the plain exact scan for the
        small ones -- issued back to back with NO host sync in between; every module's scores, rows and status word land in one
        int32 buffer [S, 2 Q k + 1] (scan_topk's packed layout).  Scores, module-local rows, ordering and padding are
        tt_scan_topk_segmented's: the same fragments and MFMA order score a row whichever pass reads it (tests/test_pipeline_gpu.py)."""
        nq, n_seg = q.shape[0], len(offs) - 1
        packed = torch.empty((n_seg, 2 * nq * k + 1), dtype=torch.int32, device=mat.device)
        for m, (lo, hi) in enumerate(zip(offs, offs[1:])):
            _scan.scan_topk(mat[lo:hi], q, k, _packed=packed[m], shadow=shadows.get(m))
        return packed

    @staticmethod
    def _unpack_modules(packed: torch.Tensor, mat, offs, q, k):
        """[S, 2 Q k + 1] (anywhere) -> (scores [Q, S, k] fp32, rows [Q, S, k] int32); a module whose candidate lists overflowed
        (status word != 0) is scanned again through the dense exact path, as scan_topk does."""
        nq, n_seg = q.shape[0], len(offs) - 1
        s = packed[:, : nq * k].view(torch.float32).view(n_seg, nq, k).permute(1, 0, 2).contiguous()
        r = packed[:, nq * k: 2 * nq * k].view(n_seg, nq, k).permute(1, 0, 2).contiguous()
        for m in torch.nonzero(packed[:, -1].cpu()).flatten().tolist():
            es, er = _scan.scan_topk(mat[offs[m]:offs[m + 1]], q, k, exact_dense=True)
            s[:, m], r[:, m] = es.to(s.device), er.to(r.device)
        return s, r

    def search(self, query_emb: torch.Tensor, k: int, return_snapshot: bool = False):
        """query_emb [Q, D] -> (cosine scores [Q, S, k] fp32, module-local rows [Q, S, k] int32); with
        ``return_snapshot`` also the per-module row -> id lists of the matrix that was scanned.  The scan runs
        outside the group lock on a snapshot (matrix, offsets, id lists) taken under it."""
        q = query_emb.to(self.device, dtype=torch.float32)
        q = (q / q.norm(dim=1, keepdim=True).clamp_min(1e-12)).to(torch.bfloat16).contiguous()
        mat, offs, ids, shadows = self._snapshot_for(q, k)
        if shadows is None:
            s, r = _scan.scan_topk_segmented(mat, q, k, offs)
        else:
            s, r = self._unpack_modules(self._scan_modules(mat, offs, q, k, shadows), mat, offs, q, k)
        return (s, r, ids) if return_snapshot else (s, r)

    def search_host(self, query_emb: torch.Tensor, k: int):
        """``search`` with the hits on the host (the retriever turns rows into nodes): -> (scores [Q, S, k], rows [Q, S, k], id lists),
        CPU tensors.  The per-module passes come back in ONE copy, which is the only host sync of the call."""
        q = query_emb.to(self.device, dtype=torch.float32)
        q = (q / q.norm(dim=1, keepdim=True).clamp_min(1e-12)).to(torch.bfloat16).contiguous()
        mat, offs, ids, shadows = self._snapshot_for(q, k)
        if shadows is None:
            s, r = _scan.scan_topk_segmented(mat, q, k, offs)
            return s.cpu(), r.cpu(), ids
        s, r = self._unpack_modules(self._scan_modules(mat, offs, q, k, shadows).cpu(), mat, offs, q, k)
        return s, r, ids
