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
import math
import os
import re
import threading
from datetime import datetime, timezone
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

from . import scan as _scan
from .schema import MetadataMode, NodeWithScore, QueryBundle, TextNode, as_query_bundle

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
        self.docstore: Dict[str, TextNode] = {}  # every node (leaves + parents), for auto-merging
        self.ref_docs: Dict[str, List[str]] = {}  # source document id -> ids of all its nodes (docstore ref_doc_info)
        self._lock = threading.RLock()
        self._version = 0                      # bumped by every mutation (HipIndexGroup repacks on change)
        self._dead = 0                         # tombstoned rows (NaN-filled, leaf id None), see delete()
        self._row_of: Optional[Dict[str, int]] = None   # leaf id -> row, built on the first delete

    # ---- build / mutate ------------------------------------------------------------------------
    def _reserve(self, n_new: int) -> None:
        need = self.n + n_new
        if need > self._mat.shape[0]:
            cap = max(need, 2 * self._mat.shape[0], 1024)
            grown = torch.empty((cap, self.dim), dtype=torch.bfloat16, device=self.device)
            grown[: self.n] = self._mat[: self.n]
            self._mat = grown

    def add(self, nodes: Sequence[TextNode], embeddings=None, show_progress: bool = False) -> List[str]:
        """Insert leaf nodes (embedding them with ``embed_model`` unless ``embeddings`` is given).
        Mirrors ``VectorStoreIndex(leaf_nodes, ...)`` / ``index.insert_nodes`` (text embedded with
        MetadataMode.EMBED, ``indexing/builder.py:437-442``, ``document_index.py:527``)."""
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
            for j, nd in enumerate(nodes):
                self.leaf_ids.append(nd.id_)
                self.docstore[nd.id_] = nd
                if self._row_of is not None:
                    self._row_of[nd.id_] = self.n + j
            self.n += len(nodes)
            self._version += 1
        return [nd.id_ for nd in nodes]

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
                for r in rows:
                    self.leaf_ids[r] = None
                self._dead += len(rows)
                self._version += 1
                if self._dead > max(1024, self.n // 4):
                    self._compact()
            for nid in drop:
                self.docstore.pop(nid, None)
        return len(rows)

    def _compact(self) -> None:
        with self._lock:
            if not self._dead:
                return
            keep = [i for i, nid in enumerate(self.leaf_ids) if nid is not None]
            idx = torch.tensor(keep, dtype=torch.long, device=self.device)
            self._mat[: len(keep)] = self._mat[: self.n].index_select(0, idx)
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
    def search(self, query_emb: torch.Tensor, k: int):
        """query_emb fp32/bf16 [Q, D] -> (scores [Q,k] fp32 cosine, rows [Q,k] int32)."""
        q = query_emb.to(self.device, dtype=torch.float32)
        q = (q / q.norm(dim=1, keepdim=True).clamp_min(1e-12)).to(torch.bfloat16).contiguous()
        with self._lock:
            mat = self._mat[: self.n]
            return _scan.scan_topk(mat, q, k)

    def node_score(self, cos: float) -> float:
        return math.exp(-(2.0 - 2.0 * cos)) if self.score_mode == "chroma" else cos

    def as_retriever(self, similarity_top_k: int = 10, **_kw) -> "HipVectorRetriever":
        return HipVectorRetriever(self, similarity_top_k)

    # ---- persistence ---------------------------------------------------------------------------------
    def persist(self, persist_dir: str, embedding_model: Optional[str] = None, chunk_sizes=None,
                chunking_strategy: Optional[str] = None, chunk_overlap: Optional[int] = None) -> None:
        os.makedirs(persist_dir, exist_ok=True)
        with self._lock:
            self._compact()                    # tombstones are not written
            self.matrix.cpu().view(torch.int16).numpy().tofile(os.path.join(persist_dir, "corpus.bf16"))
            nodes = {nid: {"text": nd.text, "metadata": nd.metadata, "parent_id": getattr(nd, "parent_id", None),
                           "child_ids": list(getattr(nd, "child_ids", []) or []),
                           "prev_id": getattr(nd, "prev_id", None), "next_id": getattr(nd, "next_id", None)}
                     for nid, nd in self.docstore.items()}
            with open(os.path.join(persist_dir, "nodes.json"), "w") as f:
                json.dump({"dim": self.dim, "leaf_ids": self.leaf_ids, "nodes": nodes, "ref_docs": self.ref_docs}, f)
        # the reference's index_metadata.json (indexing/metadata.py:103-146) + what this store adds
        model = embedding_model or getattr(self.embed_model, "model_name", None)
        meta = {"embedding_model": model, "embedding_model_id": sanitize_model_id(model) if model else None,
                "created_at": datetime.now(timezone.utc).isoformat(), "index_version": INDEX_VERSION,
                "chunk_sizes": list(chunk_sizes) if chunk_sizes is not None else None, "chunk_overlap": chunk_overlap,
                "chunking_strategy": chunking_strategy,
                "embedding_dim": self.dim, "num_vectors": self.n, "store": "tensor_truth_amd/corpus.bf16"}
        with open(os.path.join(persist_dir, INDEX_METADATA_FILENAME), "w") as f:
            json.dump(meta, f, indent=2)

    @classmethod
    def load(cls, persist_dir: str, device=None, embed_model=None, score_mode: str = "chroma") -> "HipVectorIndex":
        with open(os.path.join(persist_dir, "nodes.json")) as f:
            blob = json.load(f)
        idx = cls(blob["dim"], device, embed_model, score_mode)
        raw = np.fromfile(os.path.join(persist_dir, "corpus.bf16"), dtype=np.int16).reshape(-1, blob["dim"])
        if raw.shape[0] != len(blob["leaf_ids"]):
            raise ValueError("corpus.bf16 and nodes.json disagree on the number of rows")
        idx._mat = torch.from_numpy(raw).view(torch.bfloat16).to(idx.device).contiguous()
        idx.n = raw.shape[0]
        idx.leaf_ids = list(blob["leaf_ids"])
        idx.ref_docs = {k: list(v) for k, v in (blob.get("ref_docs") or {}).items()}
        for nid, d in blob["nodes"].items():
            nd = TextNode(text=d["text"], id_=nid, metadata=d["metadata"])
            for key in ("parent_id", "child_ids", "prev_id", "next_id"):
                try:
                    setattr(nd, key, d.get(key) if key != "child_ids" else list(d.get(key) or []))
                except Exception:  # noqa: BLE001 - LlamaIndex nodes keep links in .relationships
                    pass
            idx.docstore[nid] = nd
        return idx


class HipVectorRetriever:
    """``index.as_retriever(similarity_top_k=...)`` (rag_engine.py:639): query string ->
    ``List[NodeWithScore]`` sorted by score desc, at most ``similarity_top_k`` long.  Thread-safe:
    the reference calls it from up to 8 executor threads (rag_engine.py:420-424)."""

    def __init__(self, index: HipVectorIndex, similarity_top_k: int = 10):
        self.index = index
        self.similarity_top_k = similarity_top_k

    def retrieve(self, query) -> List[NodeWithScore]:
        qb = as_query_bundle(query)
        idx = self.index
        if idx.num_live == 0:
            return []
        if getattr(qb, "embedding", None) is not None:
            q = torch.tensor([qb.embedding], dtype=torch.float32)
        else:
            em = idx.embed_model
            if em is None:
                raise ValueError("retriever needs an embed_model or a QueryBundle with an embedding")
            strs = qb.embedding_strs
            if hasattr(em, "query_embedding_device"):
                q = em.query_embedding_device(strs).mean(dim=0, keepdim=True)
            else:
                q = torch.tensor([em.get_agg_embedding_from_queries(strs)], dtype=torch.float32)
        k = min(self.similarity_top_k, idx.num_live)
        scores, rows = idx.search(q, k)
        return self.nodes_from_hits(scores[0].cpu().tolist(), rows[0].cpu().tolist())

    def nodes_from_hits(self, scores: Sequence[float], rows: Sequence[int]) -> List[NodeWithScore]:
        """(cosine, row) pairs of this retriever's index -> NodeWithScore list (padding rows < 0 skipped)."""
        idx = self.index
        out = []
        for s, r in zip(scores, rows):
            if r < 0:
                continue
            src = idx.docstore[idx.leaf_ids[r]]
            # a fresh node per hit with its own metadata dict: callers mutate it
            # (_source_index tagging, rag_engine.py:432-450) and may run concurrently
            node = TextNode(text=src.text, id_=src.id_, metadata=dict(src.metadata))
            for key in ("parent_id", "child_ids", "prev_id", "next_id"):
                if hasattr(src, key):
                    try:
                        setattr(node, key, getattr(src, key))
                    except Exception:  # noqa: BLE001
                        pass
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
        self.offsets: List[int] = []

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
            self._mat, self.offsets, self._stamp = mat, offs, stamp
        finally:
            for ix in self.indexes:
                ix._lock.release()

    def search(self, query_emb: torch.Tensor, k: int):
        """query_emb [Q, D] -> (cosine scores [Q, S, k] fp32, module-local rows [Q, S, k] int32)."""
        q = query_emb.to(self.device, dtype=torch.float32)
        q = (q / q.norm(dim=1, keepdim=True).clamp_min(1e-12)).to(torch.bfloat16).contiguous()
        with self._lock:
            self._pack()
            return _scan.scan_topk_segmented(self._mat, q, k, self.offsets)
