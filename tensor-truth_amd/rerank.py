"""Cross-encoder rerank postprocessor with the ``SentenceTransformerRerank`` surface.

Stands in for the object built at ``src/tensortruth/services/model_manager.py:333-337`` and
called at ``services/rag_service.py:343-346,617-620`` (keyword ``query_bundle``) and
``utils/web_search.py:155,222`` (positional).  Behaviour restated from llama-index /
sentence-transformers: pairs ``(query_str, node.get_content(EMBED))``,
truncation to 512, sigmoid scores, ``node.score = float(score)``, sort desc, ``[:top_n]``;
empty input -> ``[]``; missing query -> ``ValueError``.
"""
from __future__ import annotations

import logging
from typing import Any, Dict, List, Optional, Sequence

import torch

logger = logging.getLogger(__name__)

from . import weights as _weights
from .coalesce import Coalescer
from . import precision as _precision
from .encoder import pack_tokens
from .schema import MetadataMode, NodeWithScore
from .tokenization import load_tokenizer


class HipSentenceTransformerRerank:
    def __init__(self, model: str = "BAAI/bge-reranker-v2-m3", top_n: int = 2, device: Optional[str] = None,
                 keep_retrieval_score: bool = False, model_kwargs: Optional[Dict[str, Any]] = None,
                 max_length: int = 512, batch_pairs: int = 1024, coalesce: bool = True, max_coalesced_calls: int = 64,
                 coalesce_wait_s: float = 0.0, **_ignored):
        dev = torch.device("cuda" if device in (None, "cuda") else device)
        if dev.type != "cuda":
            raise RuntimeError(f"device '{device}': tensor_truth_amd runs on HIP devices only (no CPU path)")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.model_name = model
        self.top_n = top_n
        self.device = dev
        self.keep_retrieval_score = keep_retrieval_score
        self.max_length = max_length
        self.batch_pairs = batch_pairs
        cfg, state, mdir = _weights.resolve(model, model_kwargs, dev, want_head=True)
        if not cfg.num_labels:
            raise ValueError(f"'{model}' has no classification head (not a cross-encoder)")
        self.config = cfg
        # CrossEncoder.predict's activation: sigmoid for the BGE rerankers, raw logits for checkpoints that say so (ms-marco)
        self.activation = _weights.head_activation(model, mdir, model_kwargs)
        # BERT cross-encoders mark the passage segment with token type 1 (XLM-R has a single type)
        self._use_types = cfg.arch == "bert" and cfg.type_vocab > 1
        # pairs are truncated by the tokenizer (longest-first, specials kept): never beyond what the model has positions for
        self.max_length = min(max_length, cfg.max_seq_len)
        # precision.resolve(): model_kwargs, ModelManager.precision, TT_PRECISION -- "reference" gives the unchanged
        # reference call (no dtype: fp32, model_manager.py:333-337) its fp32 semantics, and is the default; bf16 / fp16 / fp8 when named
        # (`.model` is what the reference's memory accounting reads)
        self.model, self._encoder, self.precision = _precision.build_encoder(cfg, state, dev, model_kwargs, f"reranker {model}")
        self._tokenizer = (model_kwargs or {}).get("tokenizer") or load_tokenizer(mdir, cfg.arch, cfg.vocab_size)
        # concurrent predict() / postprocess_nodes() calls (one per request thread in the reference,
        # rag_service.py:343-346,617-620) share ONE tokenizer call and ONE encoder batch; scores do not depend on the
        # batch a pair travels in (tests/test_configs_gpu.py), so callers see exactly their serial results
        # two-phase: the host side of a batch (tokenise + pack) runs while the previous batch is on the GPU
        self.stats = {"pairs": 0, "tokens": 0}
        self._token_source = None
        from .tokenization import HFTokenizer

        if coalesce and isinstance(self._tokenizer, HFTokenizer):
            # the worker processes of the coalesced batches' pair tokenisation start NOW, in the background: not inside the first
            # request that brings 96 pairs (seconds of interpreter start-up + tokenizer parsing)
            try:
                from . import ingest_workers as iw

                iw.warm_pair_pool(self._tokenizer, self.max_length)
            except Exception as exc:  # noqa: BLE001
                logger.warning("reranker: pair tokenizer pool not started (%s)", exc)
        # depth 3 (round 5): with a real sub-word tokenizer the host turn-around of a batch's callers (retrieve + tokenise + pack: ~100 ms
        # for 8 callers) is longer than ONE batch on the GPU, so with two unfinished batches the GPU waited for the third
        # (profiles/r05_surface_busy.log: 97.6 -> 102.3 q/s from 32 threads); TT_COALESCE_DEPTH overrides
        self._front = (Coalescer(self._prepare_many, max_coalesced_calls, coalesce_wait_s, execute=self._enqueue_many,
                                 finish=self._collect_many, depth=3) if coalesce else None)

    # ---- token-id level ---------------------------------------------------------------------------
    def _pack(self, pair_ids: Sequence[Sequence[int]], type_ids: Optional[Sequence[Sequence[int]]] = None):
        if not self._use_types:
            type_ids = None
        return [pack_tokens(pair_ids[lo:lo + self.batch_pairs], self.config,
                            None if type_ids is None else type_ids[lo:lo + self.batch_pairs], self.max_length)
                for lo in range(0, len(pair_ids), self.batch_pairs)]

    def _score_packed(self, batches) -> torch.Tensor:
        if self.activation == "identity":
            outs = [self._encoder.rerank_packed(b, want_logits=True)[1] for b in batches]
        else:
            outs = [self._encoder.rerank_packed(b) for b in batches]
        return torch.cat(outs) if outs else torch.empty(0, device=self.device)

    def score_token_pairs(self, pair_ids: Sequence[Sequence[int]],
                          type_ids: Optional[Sequence[Sequence[int]]] = None) -> torch.Tensor:
        """Relevance (sigmoid, or the raw logit for ``activation == "identity"``) of already tokenised
        ``<s> q </s></s> p </s>`` / ``[CLS] q [SEP] p [SEP]`` sequences, fp32 [n] (device).  ``type_ids``: BERT segment ids."""
        if self._use_types and type_ids is None:
            raise ValueError("a BERT cross-encoder needs the pairs' token type ids (0 for [CLS] q [SEP], 1 for p [SEP])")
        return self._score_packed(self._pack(pair_ids, type_ids))

    def _tokenize_pairs(self, pairs: Sequence[Sequence[str]]):
        """-> (ids per pair, token type ids per pair)."""
        tk = self._tokenizer
        enc = None
        from .tokenization import HFTokenizer, PreTokenized, assemble_pairs

        pre = [i for i, pr in enumerate(pairs) if isinstance(pr[1], PreTokenized)]
        if pre:
            # passages whose ids the index kept at ingest: only their queries are tokenised (once per distinct string)
            done = assemble_pairs(tk, [pairs[i] for i in pre], self.max_length)
            rest = [i for i, pr in enumerate(pairs) if not isinstance(pr[1], PreTokenized)]
            ids: List = [None] * len(pairs)
            types: List = [None] * len(pairs)
            for i, a in zip(pre, done):
                ids[i] = a
            if rest:
                r_ids, r_types = self._tokenize_pairs([pairs[i] for i in rest])
                for i, a, t in zip(rest, r_ids, r_types):
                    ids[i], types[i] = a, t
            st = self.stats
            st["pairs"] += len(pre)
            st["tokens"] += sum(len(a) for a in done)
            st["pretokenized"] = st.get("pretokenized", 0) + len(pre)
            return ids, types

        if isinstance(tk, HFTokenizer) and len(pairs) >= 96:
            # a coalesced batch's pairs go to single-threaded worker processes (ingest_workers.PairTokenizerPool): same ids, a
            # third of the time, and not under this process's GIL; a lone caller's 50 pairs stay here (3 ms)
            # Nothing about the pool can fail (or stall) a request: a pool that is still starting, busy, dead, hung past its deadline
            # or raising hands back None and the pairs are tokenised right here -- identical ids either way.
            try:
                from . import ingest_workers as iw

                pool = iw.get_pair_pool(tk)
                got = pool.encode(list(pairs), self.max_length) if pool is not None else None
            except Exception as exc:  # noqa: BLE001
                logger.warning("reranker: pair tokenizer pool unavailable (%s: %s); tokenising in process", type(exc).__name__, exc)
                got = None
            if got is not None:
                enc = list(zip(*got))
        if enc is not None:
            pass
        elif hasattr(tk, "encode_pair_batch"):
            enc = tk.encode_pair_batch(list(pairs), self.max_length)
        else:
            enc = [tk.encode_pair(q, p, self.max_length) for q, p in pairs]
        ids = [e[0] for e in enc]
        st = self.stats                                   # running totals (bench.py reports the pair lengths a leg really ran)
        st["pairs"] += len(ids)
        st["tokens"] += sum(len(x) for x in ids)
        return ids, [e[1] for e in enc]

    def _predict_flat(self, pairs: Sequence[Sequence[str]]) -> List[float]:
        return self.score_token_pairs(*self._tokenize_pairs(pairs)).cpu().tolist()

    def _prepare_many(self, calls: List[Sequence[Sequence[str]]]):
        """Host phase of a coalesced batch: the pair lists of several concurrent callers, tokenised and packed."""
        flat = [p for c in calls for p in c]
        return [len(c) for c in calls], self._pack(*self._tokenize_pairs(flat)) if flat else []

    def _enqueue_many(self, prepared):
        """Device phase: the forward over the packed pairs is ENQUEUED (asynchronous), followed by the copy of its
        scores into a pinned host buffer and an event of THIS batch; nothing waits here."""
        sizes, batches = prepared
        if not batches:
            return sizes, None, None
        dev_scores = self._score_packed(batches)
        with torch.cuda.device(self.device):
            host = torch.empty(dev_scores.shape, dtype=dev_scores.dtype, pin_memory=True)
            host.copy_(dev_scores, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
        return sizes, host, ev

    def _collect_many(self, pending) -> List[List[float]]:
        """Wait for THIS batch's event only (the next batch's forward is already enqueued behind it: a stream-wide
        synchronising copy would make these callers wait for that one too) and hand each caller its own scores."""
        sizes, host, ev = pending
        if ev is not None:
            ev.synchronize()
        scores = host.tolist() if host is not None else []
        out, lo = [], 0
        for n in sizes:
            out.append(scores[lo:lo + n])
            lo += n
        return out

    def predict(self, pairs: Sequence[Sequence[str]]) -> List[float]:
        """CrossEncoder.predict: [(query, passage), ...] -> scores (sigmoid, or logits where the checkpoint says Identity)."""
        pairs = list(pairs)
        if not pairs:
            return []
        if self._front is not None:
            return self._front.submit(pairs)
        return self._predict_flat(pairs)

    # ---- passages tokenised once, at ingest -----------------------------------------------------------------
    def accepts_token_source(self, signature: str, instruction: str = "") -> bool:
        """Would ids made by the tokenizer ``signature`` (with ``instruction`` prepended to the text) be THIS model's ids?"""
        from .tokenization import tokenizer_signature

        return not self._use_types and not instruction and signature == tokenizer_signature(self._tokenizer)

    def attach_token_source(self, source, signature: str, instruction: str = "") -> bool:
        """``source(node_id) -> int32 body ids (no specials) or None``: the ids of a node's EMBED-mode content as the index's
        embedder tokenised them at ingest (``HipVectorIndex.leaf_token_ids``).  Accepted only when that tokenizer IS this model's
        (same signature: bge-m3 and bge-reranker-v2-m3 share XLM-R's vocabulary), no text instruction was prepended, and the model has one
        token type -- then ``postprocess_nodes`` hands known passages to the batch as ids and only queries are tokenised per call.
        Scores are bit-identical to the string path (tests/test_config5_gpu.py).  -> whether it was accepted."""
        ok = self.accepts_token_source(signature, instruction)
        self._token_source = source if ok else None
        if not ok:
            logger.info("reranker: token source refused (another tokenizer, a text instruction, or a model with segment ids)")
        return ok

    def detach_token_source(self) -> None:
        self._token_source = None

    # ---- postprocessor surface ------------------------------------------------------------------------
    def postprocess_nodes(self, nodes: List[NodeWithScore], query_bundle=None, query_str: Optional[str] = None, token_source=None):
        """``token_source`` (keyword, optional): a per-CALL source of stored passage ids -- what ``RerankerWithTokenSource`` passes for
        its service -- instead of the one attached to this (shared, ModelManager-cached) instance."""
        if query_bundle is None and query_str is None:
            raise ValueError("Missing query bundle in extra info.")
        q = query_str if query_bundle is None else query_bundle.query_str
        if len(nodes) == 0:
            return []
        src = token_source if token_source is not None else self._token_source
        if src is None:
            passages = [n.node.get_content(metadata_mode=MetadataMode.EMBED) for n in nodes]
        else:
            from .tokenization import PreTokenized

            passages = []
            for n in nodes:
                ids = src(n.node.id_)
                passages.append(PreTokenized(ids) if ids is not None else n.node.get_content(metadata_mode=MetadataMode.EMBED))
        scores = self.predict([(q, t) for t in passages])
        for n, s in zip(nodes, scores):
            if self.keep_retrieval_score:
                n.node.metadata["retrieval_score"] = n.score
            n.score = float(s)
        return sorted(nodes, key=lambda x: -x.score if x.score else 0)[: self.top_n]

    _postprocess_nodes = postprocess_nodes

    # ---- core/ranking.py Reranker protocol (reference: core/ranking.py:16-32) ---------------------------
    def rerank(self, query: str, documents: Sequence[str], top_n: Optional[int] = None):
        scores = self.predict([(query, d) for d in documents])
        order = sorted(range(len(documents)), key=lambda i: -scores[i])
        if top_n is not None:
            order = order[:top_n]
        return [{"index": i, "relevance_score": float(scores[i])} for i in order]


class RerankerWithTokenSource:
    """One retrieval service's view of a shared reranker: the same model, tokenizer, coalescing front and statistics -- plus THIS
    service's source of stored passage ids, handed over per call.  ``ModelManager.get_reranker`` returns one cached instance per
    (model, device, top_n) (services/model_manager.py:143-186); attaching a service's token source to it would let the last service
    built replace or detach every earlier service's (its lookups would then go through the other service's indexes)."""

    def __init__(self, reranker: HipSentenceTransformerRerank, token_source):
        self._reranker = reranker
        self.token_source = token_source

    def postprocess_nodes(self, nodes, query_bundle=None, query_str: Optional[str] = None):
        return self._reranker.postprocess_nodes(nodes, query_bundle, query_str, token_source=self.token_source)

    _postprocess_nodes = postprocess_nodes

    def __getattr__(self, name):          # model, top_n, predict, rerank, stats ...: the shared instance's
        return getattr(self._reranker, name)
