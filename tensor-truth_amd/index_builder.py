"""Module index build: parse -> embed leaves on the GPU -> index + docstore -> persist.

The core of the reference's ``build_module`` (``src/tensortruth/indexing/builder.py:376-453``) behind its document
loading and metadata extraction (control plane, out of scope): same strategies (``ChunkingStrategy``), same defaults
(chunk sizes ``[2048, 512, 256]``, overlap 64, semantic buffer 1, breakpoint percentile 95), same
order of operations, same ``index_metadata.json``.  The embedding model is the HIP embedder; the semantic pass and the
leaf embedding both run through its pipelined batch path.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

from .node_parser import HierarchicalNodeParser, get_leaf_nodes
from .semantic import SemanticSplitter
from .vector_index import HipVectorIndex

DEFAULT_CHUNK_SIZES = [2048, 512, 256]      # builder.py:302-303
DEFAULT_CHUNK_OVERLAP = 64                  # indexing/builder.py DEFAULT_CHUNK_OVERLAP
CHUNKING_STRATEGIES = ("hierarchical", "semantic", "semantic_hierarchical")   # builder.py:48-66


def parse_documents(documents: Sequence, embed_model, chunking_strategy: str = "hierarchical",
                    chunk_sizes: Optional[Sequence[int]] = None, chunk_overlap: Optional[int] = None,
                    semantic_buffer_size: int = 1, semantic_breakpoint_threshold: float = 95, node_parser=None) -> List:
    """Strategy dispatch of builder.py:383-420.  ``node_parser``: any object with ``get_nodes_from_documents``
    (e.g. llama-index's own HierarchicalNodeParser) to use instead of the host restatement."""
    if chunking_strategy not in CHUNKING_STRATEGIES:
        raise ValueError(f"'{chunking_strategy}' is not a valid ChunkingStrategy")
    sizes = list(chunk_sizes) if chunk_sizes is not None else list(DEFAULT_CHUNK_SIZES)
    overlap = DEFAULT_CHUNK_OVERLAP if chunk_overlap is None else chunk_overlap
    hier = node_parser or HierarchicalNodeParser.from_defaults(chunk_sizes=sizes, chunk_overlap=overlap)
    if chunking_strategy == "hierarchical":
        return hier.get_nodes_from_documents(documents)
    sem = SemanticSplitter(embed_model, buffer_size=semantic_buffer_size,
                           breakpoint_percentile_threshold=semantic_breakpoint_threshold)
    semantic_nodes = sem.get_nodes_from_documents(documents)
    if chunking_strategy == "semantic":
        return semantic_nodes
    return hier.get_nodes_from_documents(semantic_nodes)


def build_index(documents: Sequence, embed_model, persist_dir: Optional[str] = None,
                chunking_strategy: str = "hierarchical", chunk_sizes: Optional[Sequence[int]] = None,
                chunk_overlap: Optional[int] = None, semantic_buffer_size: int = 1,
                semantic_breakpoint_threshold: float = 95, embedding_model: Optional[str] = None, node_parser=None,
                progress_callback: Optional[Callable[[str, int, int], None]] = None) -> HipVectorIndex:
    """-> the module's HipVectorIndex (persisted under ``persist_dir`` when given)."""
    if progress_callback:
        progress_callback("parsing", 0, len(documents))
    nodes = parse_documents(documents, embed_model, chunking_strategy, chunk_sizes, chunk_overlap,
                            semantic_buffer_size, semantic_breakpoint_threshold, node_parser)
    leaves = get_leaf_nodes(nodes)
    if progress_callback:
        progress_callback("embedding", 0, len(leaves))
    index = HipVectorIndex(embed_model.config.hidden if hasattr(embed_model, "config") else len(embed_model.get_text_embedding("x")),
                           embed_model=embed_model)
    index.add_to_docstore(nodes)                       # storage_context.docstore.add_documents(nodes), builder.py:430
    index.add(leaves, show_progress=True)              # VectorStoreIndex(leaf_nodes, ...), builder.py:437-442
    if progress_callback:
        progress_callback("embedding", len(leaves), len(leaves))
    if persist_dir is not None:
        sizes = list(chunk_sizes) if chunk_sizes is not None else list(DEFAULT_CHUNK_SIZES)
        index.persist(persist_dir, embedding_model=embedding_model or getattr(embed_model, "model_name", None),
                      chunk_sizes=sizes, chunking_strategy=chunking_strategy,
                      chunk_overlap=DEFAULT_CHUNK_OVERLAP if chunk_overlap is None else chunk_overlap)
    return index
