"""Module index build: parse -> embed leaves on the GPU -> index + docstore -> persist.

The core of the reference's ``build_module`` (``src/tensortruth/indexing/builder.py:376-453``) behind its document
loading and metadata extraction (control plane, out of scope): same strategies (``ChunkingStrategy``), same defaults
(chunk sizes ``[2048, 512, 256]``, overlap 64, semantic buffer 1, breakpoint percentile 95), same
order of operations, same ``index_metadata.json``.  The embedding model is the HIP embedder; the semantic pass and the
leaf embedding both run through its pipelined batch path.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Set

from .node_parser import HierarchicalNodeParser, get_leaf_nodes
from .semantic import SemanticSplitter
from .vector_index import HipVectorIndex

DEFAULT_CHUNK_SIZES = [2048, 512, 256]      # builder.py:302-303
DEFAULT_CHUNK_OVERLAP = 64                  # indexing/builder.py DEFAULT_CHUNK_OVERLAP
CHUNKING_STRATEGIES = ("hierarchical", "semantic", "semantic_hierarchical")   # builder.py:48-66


def iter_parsed(documents: Sequence, embed_model, chunking_strategy: str = "hierarchical",
                chunk_sizes: Optional[Sequence[int]] = None, chunk_overlap: Optional[int] = None,
                semantic_buffer_size: int = 1, semantic_breakpoint_threshold: float = 95, node_parser=None,
                sub_batch: int = 2048, token_counter: str = "words"):
    """Strategy dispatch of builder.py:383-420, yielding the nodes in document order in pieces of about ``sub_batch`` inputs
    of the (host-only) hierarchical pass, so that a caller can hand each piece's leaves to the GPU while the next piece
    is being split.  ``node_parser``: any object with ``get_nodes_from_documents`` (e.g. llama-index's own
    HierarchicalNodeParser) to use instead of the host restatement."""
    if chunking_strategy not in CHUNKING_STRATEGIES:
        raise ValueError(f"'{chunking_strategy}' is not a valid ChunkingStrategy")
    documents = list(documents)
    sizes = list(chunk_sizes) if chunk_sizes is not None else list(DEFAULT_CHUNK_SIZES)
    overlap = DEFAULT_CHUNK_OVERLAP if chunk_overlap is None else chunk_overlap
    hier = node_parser or HierarchicalNodeParser.from_defaults(chunk_sizes=sizes, chunk_overlap=overlap,
                                                               tokenizer=_counter(token_counter, embed_model))
    step = max(1, sub_batch)

    def pieces(items):
        # The first pieces are SMALL and double up to ``sub_batch``: while the host splits a piece the GPU has only the
        # previous piece's leaves to embed, and before the first piece nothing at all -- with full-size pieces from the
        # start a kernel trace shows the GPU idle for the whole first piece of every window (0.4 s per 2048 inputs,
        # 20 % of a 512-document window); a 128-input first piece keeps that to a few tens of milliseconds.
        lo, size = 0, min(step, 128)
        while lo < len(items):
            yield items[lo:lo + size]
            lo += size
            size = min(step, size * 2)

    if chunking_strategy == "hierarchical":
        for piece in pieces(documents):
            yield hier.get_nodes_from_documents(piece)
        return
    sem = SemanticSplitter(embed_model, buffer_size=semantic_buffer_size,
                           breakpoint_percentile_threshold=semantic_breakpoint_threshold)
    semantic_nodes = sem.get_nodes_from_documents(documents)
    if chunking_strategy == "semantic":
        yield semantic_nodes
        return
    for piece in pieces(semantic_nodes):
        yield hier.get_nodes_from_documents(piece)


def _counter(token_counter: str, embed_model):
    """"words": word / punctuation count (node_parser.count_tokens); "embedder": sub-word tokens of the embedding model's own
    tokenizer (node_parser.tokenizer_counter) -- the stand-in for llama-index's tiktoken count on hosts without it."""
    if token_counter == "words":
        return None
    if token_counter == "embedder":
        from .node_parser import tokenizer_counter

        return tokenizer_counter(embed_model._tokenizer)
    raise ValueError(f"token_counter '{token_counter}': expected 'words' or 'embedder'")


def parse_documents(documents: Sequence, embed_model, chunking_strategy: str = "hierarchical",
                    chunk_sizes: Optional[Sequence[int]] = None, chunk_overlap: Optional[int] = None,
                    semantic_buffer_size: int = 1, semantic_breakpoint_threshold: float = 95, node_parser=None) -> List:
    """All nodes of ``documents`` (``iter_parsed`` in one list)."""
    out: List = []
    for nodes in iter_parsed(documents, embed_model, chunking_strategy, chunk_sizes, chunk_overlap, semantic_buffer_size,
                             semantic_breakpoint_threshold, node_parser):
        out.extend(nodes)
    return out


def build_index(documents: Sequence, embed_model, persist_dir: Optional[str] = None,
                chunking_strategy: str = "hierarchical", chunk_sizes: Optional[Sequence[int]] = None,
                chunk_overlap: Optional[int] = None, semantic_buffer_size: int = 1,
                semantic_breakpoint_threshold: float = 95, embedding_model: Optional[str] = None, node_parser=None,
                progress_callback: Optional[Callable[[str, int, int], None]] = None,
                window_docs: int = 8192, workers: Optional[int] = None, token_counter: str = "words",
                keep_leaf_token_ids: bool = False) -> HipVectorIndex:
    """-> the module's HipVectorIndex (persisted under ``persist_dir`` when given).

    ``workers`` (default: ``TT_INGEST_WORKERS``, else up to 8 of the host's cores; 0 = everything in this process): sentence
    splitting, hierarchical parsing and tokenization run in that many worker PROCESSES (``ingest_workers.py``) while this one
    only feeds the GPU -- round 3's single-process pipeline kept the GPU 0.75 busy from strings, its idle gaps the host-only
    stretches of one Python thread.  Same nodes, same rows, same order (node ids are uuid4 either way).  Used for the
    ``hierarchical`` and ``semantic_hierarchical`` strategies with the package's own parser and tokenizers.

    Documents are processed in windows of ``window_docs`` (parse -> docstore -> embed the leaves -> append the rows), so a
    100k-document build (BASELINE config 5) holds one window's sentence groups and strings at a time, not all of them.
    Every step is per document (percentile thresholds, hierarchy, metadata) and rows are appended in document order: the
    index is the one the reference's whole-corpus order of operations (builder.py:383-442) builds.  (A second thread
    parsing window i + 1 while this one embeds window i was measured SLOWER -- 250-255 vs 270 docs/s at 8000 documents:
    smaller windows mean shorter forward passes, and the overlap is already there without a thread, see the loop below.)
    ``window_docs <= 0``: one window.
    ``token_counter``: what the hierarchy's chunk sizes count -- "words" (default) or "embedder" (sub-word tokens of the embedding
    model's tokenizer, the offline stand-in for llama-index's tiktoken count: ``_counter``).
    ``keep_leaf_token_ids`` (worker-process builds): the index keeps every leaf's token ids as the embedder's tokenizer produced them
    (``HipVectorIndex.leaf_token_ids``, ~4 bytes per token of host memory), so that a reranker with the SAME tokenizer (bge-m3 and
    bge-reranker-v2-m3 share XLM-R's) never tokenises a retrieved leaf again: ``build_retrieval_service`` wires it up."""
    documents = list(documents)
    n_docs = len(documents)
    if chunking_strategy not in CHUNKING_STRATEGIES:
        raise ValueError(f"'{chunking_strategy}' is not a valid ChunkingStrategy")
    if progress_callback:
        progress_callback("parsing", 0, n_docs)
    index = HipVectorIndex(embed_model.config.hidden if hasattr(embed_model, "config") else len(embed_model.get_text_embedding("x")),
                           embed_model=embed_model)
    _counter(token_counter, embed_model)          # (validates the name)
    if _build_with_workers(index, documents, embed_model, chunking_strategy, chunk_sizes, chunk_overlap, semantic_buffer_size,
                           semantic_breakpoint_threshold, node_parser, workers, token_counter, keep_leaf_token_ids):
        n_docs = 0        # (done: skip the in-process loop below)
    step = max(n_docs, 1) if window_docs <= 0 else window_docs
    for lo in range(0, n_docs, step):
        # the leaf forward passes of one piece are only ENQUEUED by index.add (nothing below waits for the GPU except the
        # staging ring's back-pressure), so the GPU embeds piece i while this thread splits piece i + 1
        for nodes in iter_parsed(documents[lo:lo + step], embed_model, chunking_strategy, chunk_sizes, chunk_overlap,
                                 semantic_buffer_size, semantic_breakpoint_threshold, node_parser, token_counter=token_counter):
            index.add_to_docstore(nodes)                   # storage_context.docstore.add_documents(nodes), builder.py:430
            index.add(get_leaf_nodes(nodes), show_progress=True)   # VectorStoreIndex(leaf_nodes, ...), builder.py:437-442
        if progress_callback and lo + step < n_docs:
            progress_callback("embedding", index.n, index.n)
    if progress_callback:
        progress_callback("embedding", index.n, index.n)
    if persist_dir is not None:
        sizes = list(chunk_sizes) if chunk_sizes is not None else list(DEFAULT_CHUNK_SIZES)
        index.persist(persist_dir, embedding_model=embedding_model or getattr(embed_model, "model_name", None),
                      chunk_sizes=sizes, chunking_strategy=chunking_strategy,
                      chunk_overlap=DEFAULT_CHUNK_OVERLAP if chunk_overlap is None else chunk_overlap)
    return index


def _build_with_workers(index, documents, embed_model, chunking_strategy, chunk_sizes, chunk_overlap, buffer_size, percentile,
                        node_parser, workers, token_counter: str = "words", keep_leaf_token_ids: bool = False) -> bool:
    """The worker-process form of the build loop (``ingest_workers.IngestWorkers.run``).  -> False when it does not apply."""
    import torch

    from . import ingest_workers as iw
    from .semantic import adjacent_distances

    W = iw.default_workers() if workers is None else int(workers)
    min_docs = int(os.environ.get("TT_INGEST_WORKERS_MIN_DOCS", "64"))
    if (W <= 0 or node_parser is not None or chunking_strategy not in ("hierarchical", "semantic_hierarchical")
            or not hasattr(embed_model, "embed_token_batches") or len(documents) < min_docs):
        return False
    try:
        tk_spec = iw.tokenizer_spec(embed_model._tokenizer)
    except TypeError:
        return False
    spec = {"tokenizer": tk_spec, "max_length": embed_model.max_length, "text_instruction": getattr(embed_model, "text_instruction", "") or "",
            "buffer_size": buffer_size, "percentile": percentile,
            "chunk_sizes": list(chunk_sizes) if chunk_sizes is not None else list(DEFAULT_CHUNK_SIZES),
            "chunk_overlap": DEFAULT_CHUNK_OVERLAP if chunk_overlap is None else chunk_overlap,
            "token_counter": token_counter}
    def distances(emb):
        # the copy back is enqueued behind this chunk's own forward passes, with an event of its own: waiting for it waits for
        # this chunk only, not for what was enqueued after it
        d = adjacent_distances(emb)
        host = torch.empty(d.shape, dtype=d.dtype, pin_memory=True)
        host.copy_(d, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(d.device))

        def ready(block: bool = False) -> bool:
            if block:
                ev.synchronize()
                return True
            return ev.query()

        return host.numpy(), ready

    def on_nodes(nodes, leaf_pos, emb, tokens=None):
        index.add_to_docstore(nodes)                       # storage_context.docstore.add_documents(nodes), builder.py:430
        if emb is not None:
            tok = iw.unflatten(*tokens) if (keep_leaf_token_ids and tokens is not None) else None
            index.add([nodes[i] for i in leaf_pos], embeddings=emb, token_ids=tok)      # VectorStoreIndex(leaf_nodes, ...), builder.py:437-442

    # (a pool serves one build at a time: a second thread building with the same configuration gets a private pool)
    with iw.lease_workers(spec, W) as pool:
        pool.run(documents, chunking_strategy == "semantic_hierarchical", embed_model.embed_token_batches, distances, on_nodes,
                 chunk_docs=int(os.environ.get("TT_INGEST_CHUNK_DOCS", "48")), embed_flat=getattr(embed_model, "embed_flat", None),
                 leaf_tokens=keep_leaf_token_ids)
    return True


def build_index_sharded(documents: Sequence, embed_model, group=None, queries: str = "replicated", **build_kw):
    """BASELINE.json config 5 on N GPUs: every rank parses, splits and embeds documents ``rank, rank + world, ...`` on its own
    GPU (replica-parallel ingest, no collective: SURVEY.md section 8e) and the rank-local leaf rows become that rank's shard
    of ONE global index (``ShardedHipVectorIndex.from_local_index``: row counts + host side tables exchanged once, no matrix
    row moves).  ``documents`` is the same sequence on every rank.  -> the ``ShardedHipVectorIndex`` behind the Retriever
    surface (``as_retriever`` / ``AutoMergingRetriever`` as for a single-device index)."""
    import torch.distributed as dist

    from .sharded_index import ShardedHipVectorIndex

    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    if build_kw.get("persist_dir") is not None:
        raise ValueError("build_index_sharded: persist the shards per rank (index.persist) or persist a gathered index")
    local = build_index(list(documents)[rank::world], embed_model, **build_kw)
    return ShardedHipVectorIndex.from_local_index(local, group=group, queries=queries)


class HipDocumentIndex:
    """Incremental per-document index: the add / remove / inspect half of the reference's ``DocumentIndexBuilder``
    (``src/tensortruth/document_index.py:427-581``) on a ``HipVectorIndex`` -- PDF conversion, metadata extraction and
    settings hashing stay with the caller.  ``add_documents`` parses hierarchically, puts every node in the docstore and
    the leaves (embedded on the GPU) in the matrix, and records which nodes belong to which ``doc_id``;
    ``remove_document`` tombstones that document's leaves and drops its nodes; both persist when ``index_dir`` is set."""

    def __init__(self, embed_model, index_dir: Optional[str] = None, chunk_overlap: int = 20, node_parser_factory=None):
        self.embed_model = embed_model
        self.index_dir = index_dir
        self.chunk_overlap = chunk_overlap
        self._parser_factory = node_parser_factory or (
            lambda chunk_sizes: HierarchicalNodeParser.from_defaults(chunk_sizes=chunk_sizes, chunk_overlap=chunk_overlap))
        if index_dir is not None and os.path.exists(os.path.join(index_dir, "nodes.json")):
            self.index = HipVectorIndex.load(index_dir, embed_model=embed_model)
        else:
            dim = embed_model.config.hidden if hasattr(embed_model, "config") else len(embed_model.get_text_embedding("x"))
            self.index = HipVectorIndex(dim, embed_model=embed_model)
        self._chunk_sizes: Optional[List[int]] = None

    def index_exists(self) -> bool:          # document_index.py:135-139
        return self.index.num_live > 0

    def get_index_size(self) -> int:         # :427-441 (vectors in the collection)
        return self.index.num_live

    def get_document_count(self) -> int:     # :443-450
        return len(self.index.ref_docs)

    def get_indexed_doc_ids(self) -> Set[str]:   # :452-476
        return set(self.index.ref_docs)

    def _persist(self) -> None:
        if self.index_dir is not None:
            self.index.persist(self.index_dir, chunk_sizes=self._chunk_sizes, chunking_strategy="hierarchical",
                               chunk_overlap=self.chunk_overlap)

    def add_documents(self, documents: Sequence, doc_ids: Sequence[str], chunk_sizes: Sequence[int],
                      progress_callback: Optional[Callable] = None) -> None:
        """document_index.py:478-534.  A ``doc_id`` that is already indexed is replaced."""
        if len(documents) != len(doc_ids):
            raise ValueError("documents and doc_ids differ in length")
        cb = progress_callback or (lambda *a: None)
        cb("Embedding documents", 70, 100)
        parser = self._parser_factory(list(chunk_sizes))
        all_nodes, leaves = [], []
        for doc, doc_id in zip(documents, doc_ids):
            if doc_id in self.index.ref_docs:
                self.remove_document(doc_id, persist=False)
            nodes = parser.get_nodes_from_documents([doc])
            self.index.ref_docs[doc_id] = [n.id_ for n in nodes]
            all_nodes.extend(nodes)
            leaves.extend(get_leaf_nodes(nodes))
        self.index.add_to_docstore(all_nodes)
        self.index.add(leaves)
        self._chunk_sizes = list(chunk_sizes)
        self._persist()
        cb("Complete", 100, 100)

    def remove_document(self, doc_id: str, persist: bool = True) -> bool:
        """document_index.py:536-581: True if the document was indexed."""
        node_ids = self.index.ref_docs.pop(doc_id, None)
        if node_ids is None:
            return False
        self.index.delete(node_ids)          # leaves become tombstones, every node leaves the docstore
        if persist:
            self._persist()
        return True

    def as_retriever(self, similarity_top_k: int = 10):
        return self.index.as_retriever(similarity_top_k=similarity_top_k)
