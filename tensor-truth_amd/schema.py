"""Node / query value types the plugin surface exchanges.

When LlamaIndex is importable its own classes are used (so the objects flow through the
reference's ``AutoMergingRetriever`` / chat engine unchanged); otherwise these minimal
stand-ins expose exactly the attributes the reference touches:
``NodeWithScore.score``, ``.node.metadata`` (mutable dict), ``.node.get_content()``,
``.node.id_``, ``QueryBundle.query_str``.
"""
from __future__ import annotations

import uuid
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

try:  # pragma: no cover - llama_index is absent in the build container
    from llama_index.core.schema import MetadataMode, NodeWithScore, QueryBundle, TextNode  # type: ignore

    HAVE_LLAMA_INDEX = True
except Exception:  # noqa: BLE001
    HAVE_LLAMA_INDEX = False

    class MetadataMode:
        ALL = "all"
        EMBED = "embed"
        LLM = "llm"
        NONE = "none"

    @dataclass
    class TextNode:
        text: str = ""
        id_: str = field(default_factory=lambda: str(uuid.uuid4()))
        metadata: Dict[str, Any] = field(default_factory=dict)
        excluded_embed_metadata_keys: List[str] = field(default_factory=list)
        excluded_llm_metadata_keys: List[str] = field(default_factory=list)
        embedding: Optional[List[float]] = None
        # hierarchy / sequence links (LlamaIndex NodeRelationship PARENT/CHILD/PREVIOUS/NEXT)
        parent_id: Optional[str] = None
        child_ids: List[str] = field(default_factory=list)
        prev_id: Optional[str] = None
        next_id: Optional[str] = None

        @property
        def node_id(self) -> str:
            return self.id_

        def get_content(self, metadata_mode: str = MetadataMode.NONE) -> str:
            if metadata_mode == MetadataMode.NONE or not self.metadata:
                return self.text
            keys = [k for k in self.metadata if not k.startswith("_")]
            if metadata_mode == MetadataMode.EMBED:
                keys = [k for k in keys if k not in self.excluded_embed_metadata_keys]
            elif metadata_mode == MetadataMode.LLM:
                keys = [k for k in keys if k not in self.excluded_llm_metadata_keys]
            meta = "\n".join(f"{k}: {self.metadata[k]}" for k in keys)
            return f"{meta}\n\n{self.text}" if meta else self.text

    @dataclass
    class NodeWithScore:
        node: TextNode
        score: Optional[float] = None

        @property
        def text(self) -> str:
            return self.node.text

        @property
        def metadata(self) -> Dict[str, Any]:
            return self.node.metadata

        @property
        def id_(self) -> str:
            return self.node.id_

        @property
        def node_id(self) -> str:
            return self.node.id_

        def get_content(self, metadata_mode: str = MetadataMode.NONE) -> str:
            return self.node.get_content(metadata_mode)

    @dataclass
    class QueryBundle:
        query_str: str
        embedding: Optional[List[float]] = None
        custom_embedding_strs: Optional[List[str]] = None

        @property
        def embedding_strs(self) -> List[str]:
            return self.custom_embedding_strs or [self.query_str]


def as_query_bundle(q) -> "QueryBundle":
    return q if hasattr(q, "query_str") else QueryBundle(query_str=str(q))
