"""Importer for indexes the reference has already built.

A reference index directory (``indexing/builder.py:424-444``, ``rag_engine.py:628-645``) holds
  * a Chroma ``PersistentClient`` store with ONE collection, ``"data"``: leaf-node ids, their fp32 embeddings,
    texts and flattened metadata (``ChromaVectorStore`` keeps the node JSON under ``_node_content``);
  * ``docstore.json`` written by ``storage_context.persist``: EVERY node of the hierarchy (leaves and parents) with
    its relationships -- what ``AutoMergingRetriever`` walks.
``import_reference_index`` turns that into a ``HipVectorIndex`` (row-major bf16 matrix in HBM + docstore side table).

``docstore.json`` is plain JSON and is parsed here without llama-index.  Reading the Chroma store needs the
``chromadb`` package on the importing host only (it is not a dependency of the search path): absent, the importer
says so -- or takes the collection's contents from the caller (``ids`` / ``embeddings``), e.g. exported elsewhere.
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .schema import TextNode

# llama_index.core.schema.NodeRelationship values
_REL_SOURCE, _REL_PREVIOUS, _REL_NEXT, _REL_PARENT, _REL_CHILD = "1", "2", "3", "4", "5"


def _rel_id(rel: Any) -> Optional[str]:
    if isinstance(rel, dict):
        return rel.get("node_id")
    return None


def node_from_llamaindex_dict(data: Dict[str, Any]) -> TextNode:
    """One ``__data__`` entry of ``docstore.json`` (a serialised ``TextNode``) -> the node type of this package."""
    rels = data.get("relationships") or {}
    children = rels.get(_REL_CHILD) or []
    if isinstance(children, dict):
        children = [children]
    node = TextNode(text=data.get("text", "") or "", id_=data.get("id_") or data.get("node_id"),
                    metadata=dict(data.get("metadata") or {}))
    try:
        node.excluded_embed_metadata_keys = list(data.get("excluded_embed_metadata_keys") or [])
        node.parent_id = _rel_id(rels.get(_REL_PARENT))
        node.child_ids = [c["node_id"] for c in children if isinstance(c, dict) and "node_id" in c]
        node.prev_id = _rel_id(rels.get(_REL_PREVIOUS))
        node.next_id = _rel_id(rels.get(_REL_NEXT))
    except Exception:  # noqa: BLE001 - llama-index's own TextNode keeps these inside .relationships
        pass
    return node


def load_llamaindex_docstore(path: str) -> Dict[str, TextNode]:
    """``docstore.json`` (SimpleDocumentStore.persist) -> {node id: node}, hierarchy links included."""
    with open(path) as f:
        blob = json.load(f)
    data = blob.get("docstore/data") or {}
    out: Dict[str, TextNode] = {}
    for nid, entry in data.items():
        payload = entry.get("__data__", entry)
        if isinstance(payload, str):
            payload = json.loads(payload)
        node = node_from_llamaindex_dict(payload)
        if not node.id_:
            node.id_ = nid
        out[node.id_] = node
    return out


def read_chroma_collection(persist_dir: str, collection: str = "data", page: int = 8192
                           ) -> Tuple[List[str], np.ndarray, List[Optional[str]], List[Dict[str, Any]]]:
    """(ids, embeddings [n, d] fp32, documents, metadatas) of a Chroma collection, read page by page."""
    try:
        import chromadb  # type: ignore
    except Exception as exc:  # noqa: BLE001
        raise ImportError("reading a Chroma store needs the 'chromadb' package on the importing host "
                          "(pip install chromadb), or pass ids= and embeddings= exported elsewhere") from exc
    col = chromadb.PersistentClient(path=persist_dir).get_collection(collection)
    n = col.count()
    ids: List[str] = []
    embs: List[np.ndarray] = []
    docs: List[Optional[str]] = []
    metas: List[Dict[str, Any]] = []
    for off in range(0, n, page):
        got = col.get(limit=page, offset=off, include=["embeddings", "documents", "metadatas"])
        ids += list(got["ids"])
        embs.append(np.asarray(got["embeddings"], dtype=np.float32))
        docs += list(got.get("documents") or [None] * len(got["ids"]))
        metas += [dict(m or {}) for m in (got.get("metadatas") or [{}] * len(got["ids"]))]
    emb = np.concatenate(embs, 0) if embs else np.zeros((0, 0), dtype=np.float32)
    return ids, emb, docs, metas


def _node_from_chroma(nid: str, doc: Optional[str], meta: Dict[str, Any]) -> TextNode:
    """A leaf that is missing from docstore.json: rebuild it from what ChromaVectorStore stored."""
    content = meta.get("_node_content")
    if content:
        try:
            node = node_from_llamaindex_dict(json.loads(content))
            if not node.text and doc:
                node.text = doc
            return node
        except Exception:  # noqa: BLE001
            pass
    clean = {k: v for k, v in meta.items() if not k.startswith("_")}
    return TextNode(text=doc or "", id_=nid, metadata=clean)


def import_reference_index(persist_dir: str, embed_model=None, device=None, score_mode: str = "chroma",
                           ids: Optional[Sequence[str]] = None, embeddings=None, documents: Optional[Sequence[str]] = None,
                           metadatas: Optional[Sequence[Dict[str, Any]]] = None, collection: str = "data"):
    """Build a ``HipVectorIndex`` from a reference index directory.

    Leaves = the Chroma collection's rows, in collection order; every node of ``docstore.json`` (parents included)
    goes to the docstore side table, so ``AutoMergingRetriever(index.as_retriever(k), index.docstore)`` behaves as it
    does on the reference index.  Embeddings are L2-normalised and rounded to bf16 on the way in (the reference's
    Chroma space is squared L2 over the same vectors: identical ranking)."""
    from .vector_index import HipVectorIndex

    if ids is None or embeddings is None:
        ids, embeddings, documents, metadatas = read_chroma_collection(persist_dir, collection)
    emb = np.asarray(embeddings, dtype=np.float32)
    if emb.ndim != 2 or emb.shape[0] != len(ids):
        raise ValueError(f"embeddings {emb.shape} do not match {len(ids)} ids")
    docstore_path = os.path.join(persist_dir, "docstore.json")
    nodes = load_llamaindex_docstore(docstore_path) if os.path.exists(docstore_path) else {}
    documents = list(documents) if documents is not None else [None] * len(ids)
    metadatas = list(metadatas) if metadatas is not None else [{}] * len(ids)
    leaves = []
    for nid, doc, meta in zip(ids, documents, metadatas):
        node = nodes.get(nid)
        if node is None:
            node = _node_from_chroma(nid, doc, dict(meta or {}))
            node.id_ = nid
        leaves.append(node)
    index = HipVectorIndex(emb.shape[1], device=device, embed_model=embed_model, score_mode=score_mode)
    index.add_to_docstore(nodes.values())
    index.add(leaves, embeddings=emb)
    return index
