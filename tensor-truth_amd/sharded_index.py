"""Row-sharded vector index behind the Retriever plugin surface (BASELINE.json config 4).

north_star: "The corpus matrix is row-sharded across the 8 GPUs of one node with per-shard partial top-k merged via
RCCL all-gather over xGMI ... keeping the repo's Retriever surface".  The reference builds its retriever at
``src/tensortruth/rag_engine.py:626-645`` (``index.as_retriever(similarity_top_k)`` wrapped in an
``AutoMergingRetriever``); ``ShardedHipVectorIndex.as_retriever`` is the drop-in for that object when the corpus does
not belong on one device (or when one process per GPU serves it):

* every rank keeps rows ``[lo_r, hi_r)`` of the matrix in ITS GPU's HBM (``sharded.shard_bounds``) and the complete host
  side tables (leaf id per global row, docstore) -- those are small next to the matrix and are what turns a global row
  back into a node on whichever rank the caller runs;
* ``retrieve(query)``: embed locally -> (``queries="partitioned"``: all-gather the query embeddings, every shard scans
  the whole gathered batch) -> local exact top-k with GLOBAL row ids -> ONE all-gather of the packed partials ->
  ``tt_topk_merge`` -> nodes.  ``queries="replicated"`` (default) is the serving arrangement in which the front end
  hands every rank the same request: no query gather, identical results on all ranks.
* ``logical_shards > 1`` cuts the local rows once more and merges the pieces through the same merge kernel: one GPU
  exercises the whole protocol (tests), and a single process can hold several shards.

Collectives are issued under a lock in call order.  ``queries="replicated"``: SPMD callers make their ``retrieve`` calls
in the same order on every rank (one collective round per call).  ``queries="partitioned"`` behind ``as_retriever`` on
several ranks: a lock-step TICK front (``_TickFront``) -- every rank serves its own request threads, the ranks' rounds are
kept aligned by one tiny all-gather per tick, concurrent callers of a rank share its rounds, and query embedding and the
rerank above it are partitioned over the GPUs (throughput scales with the ranks; ``replicated`` buys capacity only).
"""
from __future__ import annotations

import math
import os
import threading
from typing import Callable, Dict, List, Optional, Sequence, Tuple  # noqa: F401

import torch
import torch.distributed as dist

from . import _lib
from . import sharded as _sh
from .schema import NodeWithScore, TextNode
from .vector_index import (HipVectorIndex, HipVectorRetriever, _read_persisted, _node_from_dict)


def _world(group) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


class ShardRoundError(RuntimeError):
    """A rank's local scan failed inside a collective round.  The failing rank still takes part in the round's all-gather
    (with a poisoned partial list), so EVERY rank raises this for the round's queries and nobody is left waiting in a collective."""


_POISON = -2          # row id marking a rank's partial list as "my scan failed" (-1 is ordinary padding)


class ShardedHipVectorIndex:
    def __init__(self, dim: int, local_rows: torch.Tensor, row_lo: int, n_total: int, leaf_ids: Sequence[Optional[str]],
                 docstore: Dict[str, TextNode], embed_model=None, score_mode: str = "chroma", group=None,
                 logical_shards: int = 1, queries: str = "replicated", ragged_queries: bool = True,
                 scan_fn: Optional[Callable] = None, merge_fn: Optional[Callable] = None):
        if queries not in ("replicated", "partitioned"):
            raise ValueError("queries must be 'replicated' or 'partitioned'")
        if local_rows.dtype != torch.bfloat16 or local_rows.dim() != 2 or local_rows.shape[1] != dim:
            raise ValueError("local_rows must be a [n_local, dim] bfloat16 matrix")
        if leaf_ids is not None and len(leaf_ids) != n_total:
            raise ValueError(f"leaf_ids has {len(leaf_ids)} entries for {n_total} global rows")
        if scan_fn is None and not local_rows.is_cuda:
            raise RuntimeError("ShardedHipVectorIndex needs its shard on a HIP device; tensor_truth_amd has no CPU path")
        self.dim, self.device = dim, local_rows.device
        self.row_lo, self.n_total = int(row_lo), int(n_total)
        # any sequence works (a list; a lazy row -> id mapping for corpora whose ids are derived from the row);
        # None = search-only index (``search`` works, ``as_retriever`` has no nodes to return)
        self.leaf_ids = leaf_ids if leaf_ids is None or not isinstance(leaf_ids, (list, tuple)) else list(leaf_ids)
        self.docstore = docstore
        self.embed_model = embed_model
        self.score_mode = score_mode
        self.group = group
        self.queries = queries
        self.ragged_queries = ragged_queries      # partitioned mode: ranks may bring different query counts (one extra tiny all-gather)
        # product path: the HIP scan / merge kernels.  (The gloo protocol test injects CPU stand-ins: the exchange and
        # the row bookkeeping are what it checks.)
        if scan_fn is None or merge_fn is None:
            from . import scan as _scan

            scan_fn = scan_fn or self._scan_shard
            merge_fn = merge_fn or _scan.topk_merge
        self._scan, self._merge = scan_fn, merge_fn
        # fp8 shadow of every local shard (scan.ScanShadow): an exact prefilter for batches of <= 4 queries -- a lone caller's scan
        # reads half the bytes, same bits out.  Built on first use (one pass over the shard, + 50 % of its HBM); TT_SCAN_SHADOW=0 turns it off.
        self.fp8_shadow = os.environ.get("TT_SCAN_SHADOW", "1") != "0"
        self._shadows: Dict[int, object] = {}
        self._shadow_lock = threading.Lock()
        self._collective_lock = threading.Lock()
        self._tick_fronts = 0                      # live _TickFront threads (they own this index's collectives)
        self._written = None                       # (HipVectorRetriever._gpu_stream: event behind the last write to the rows)
        if local_rows.is_cuda:
            self._written = torch.cuda.Event()
            self._written.record(torch.cuda.current_stream(local_rows.device))
        n_local = local_rows.shape[0]
        pieces = max(1, min(int(logical_shards), max(n_local, 1)))
        self._shards: List[Tuple[torch.Tensor, int]] = []
        for p in range(pieces):
            lo, hi = _sh.shard_bounds(n_local, pieces, p)
            self._shards.append((local_rows[lo:hi], self.row_lo + lo))

    # ---- construction ------------------------------------------------------------------------------------
    @classmethod
    def from_index(cls, index: HipVectorIndex, group=None, **kw) -> "ShardedHipVectorIndex":
        """Shard an index every rank holds in full (tests; small corpora): this rank keeps its row range."""
        world, rank = _world(group)
        index._compact()
        mat, leaf_ids = index.snapshot()
        lo, hi = _sh.shard_bounds(mat.shape[0], world, rank)
        return cls(index.dim, mat[lo:hi].contiguous(), lo, mat.shape[0], leaf_ids, index.docstore,
                   embed_model=kw.pop("embed_model", index.embed_model), score_mode=kw.pop("score_mode", index.score_mode),
                   group=group, **kw)

    @classmethod
    def from_local(cls, dim: int, local_rows: torch.Tensor, local_leaf_ids: Sequence[str], local_docstore: Dict[str, TextNode],
                   group=None, **kw) -> "ShardedHipVectorIndex":
        """Every rank has ingested ITS OWN share of the documents (replica-parallel ingest, SURVEY.md section 8e: "chunks
        partitioned, no collective") and holds the rows of its own leaves: stitch the shards into one global index WITHOUT
        moving a matrix row.  Rank r's rows become global rows [sum(n_0..n_{r-1}), + n_r) (one all-gather of the counts);
        the host side tables -- leaf id per row, and every node's text / metadata / links, which is what turns a hit on
        another rank's rows into a node here -- are exchanged once with ``all_gather_object`` (text only: a few hundred
        bytes per node against 2 KiB per matrix row)."""
        world, rank = _world(group)
        local_leaf_ids = list(local_leaf_ids)
        if local_rows.shape[0] != len(local_leaf_ids):
            raise ValueError(f"{local_rows.shape[0]} local rows but {len(local_leaf_ids)} local leaf ids")
        if world == 1:
            return cls(dim, local_rows, 0, len(local_leaf_ids), local_leaf_ids, dict(local_docstore), group=group, **kw)
        from .vector_index import _node_to_dict

        mine = (local_leaf_ids, {nid: _node_to_dict(nd) for nid, nd in local_docstore.items()})
        parts: List = [None] * world
        dist.all_gather_object(parts, mine, group=group)
        counts = [len(p[0]) for p in parts]
        leaf_ids: List[Optional[str]] = []
        docstore: Dict[str, TextNode] = {}
        for r, (ids, nodes) in enumerate(parts):
            leaf_ids.extend(ids)
            if r == rank:
                docstore.update(local_docstore)           # keep this rank's own node objects as they are
            else:
                for nid, d in nodes.items():
                    docstore.setdefault(nid, _node_from_dict(nid, d))
        if len(set(i for i in leaf_ids if i is not None)) != sum(1 for i in leaf_ids if i is not None):
            raise ValueError("the ranks' leaf ids collide: every document must be ingested by exactly one rank")
        return cls(dim, local_rows, sum(counts[:rank]), sum(counts), leaf_ids, docstore, group=group, **kw)

    @classmethod
    def from_local_index(cls, index: HipVectorIndex, group=None, **kw) -> "ShardedHipVectorIndex":
        """``from_local`` for a rank-local ``HipVectorIndex`` (e.g. ``build_index`` over this rank's documents)."""
        index._compact()
        mat, leaf_ids = index.snapshot()
        return cls.from_local(index.dim, mat.contiguous(), leaf_ids, index.docstore, group=group,
                              embed_model=kw.pop("embed_model", index.embed_model),
                              score_mode=kw.pop("score_mode", index.score_mode), **kw)

    @classmethod
    def load(cls, persist_dir: str, device=None, embed_model=None, score_mode: str = "chroma", group=None,
             **kw) -> "ShardedHipVectorIndex":
        """``HipVectorIndex.persist``'s directory: every rank reads nodes.json and ONLY its rows of the matrix
        (memory-mapped), so a 10M x 1024 corpus is never materialised whole on a host or a device."""
        world, rank = _world(group)
        blob, raw = _read_persisted(persist_dir, rows=lambda n_rows: slice(*_sh.shard_bounds(n_rows, world, rank)))
        n = len(blob["leaf_ids"])
        lo, hi = _sh.shard_bounds(n, world, rank)
        dev = torch.device("cuda" if device in (None, "cuda") else device)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        import numpy as np

        rows = torch.from_numpy(np.array(raw, copy=True)).view(torch.bfloat16).to(dev).contiguous()
        docstore = {nid: _node_from_dict(nid, d) for nid, d in blob["nodes"].items()}
        return cls(blob["dim"], rows, lo, n, blob["leaf_ids"], docstore, embed_model=embed_model, score_mode=score_mode,
                   group=group, **kw)

    # ---- search ----------------------------------------------------------------------------------------------
    @property
    def num_live(self) -> int:
        return self.n_total

    def node_score(self, cos: float) -> float:
        return math.exp(-(2.0 - 2.0 * cos)) if self.score_mode == "chroma" else cos

    def _scan_shard(self, rows: torch.Tensor, q16: torch.Tensor, k: int, base: int):
        from . import scan as _scan

        shadow = None
        if self.fp8_shadow and 0 < q16.shape[0] <= _scan.ScanShadow.MAX_QUERIES and rows.shape[0] >= _scan.ScanShadow.MIN_ROWS:
            key = rows.data_ptr()
            shadow = self._shadows.get(key)
            if shadow is None:
                with self._shadow_lock:
                    shadow = self._shadows.get(key)
                    if shadow is None:
                        shadow = self._shadows[key] = _scan.ScanShadow(rows)
        return _scan.scan_topk(rows, q16, k, idx_base=base, shadow=shadow)

    def _local_topk(self, q16: torch.Tensor, k: int):
        """Partial top-k of this rank's rows (global ids): one scan per logical shard + a local merge."""
        parts = [self._scan(rows, q16, k, base) for rows, base in self._shards]
        if len(parts) == 1:
            return parts[0]
        s = torch.cat([p[0] for p in parts], dim=1)
        i = torch.cat([p[1] for p in parts], dim=1)
        return self._merge(s, i, k)

    def _collective_topk(self, q16: torch.Tensor, k: int):
        """Local scan of the (identical on every rank) batch -> ONE all-gather of the packed partial lists -> merge.  A local
        scan that raises is carried through the collective as a poisoned list: all ranks then raise ``ShardRoundError``
        together (the failing rank with its own exception as the cause) and the next round starts aligned."""
        world, rank = _world(self.group)
        err = None
        try:
            s, i = self._local_topk(q16, k)
        except Exception as exc:  # noqa: BLE001 - reported to every rank below
            err = exc
            s = torch.zeros((q16.shape[0], k), dtype=torch.float32, device=q16.device)
            i = torch.full((q16.shape[0], k), _POISON, dtype=torch.int32, device=q16.device)
        all_s, all_i = _sh.gather_partials(s, i, self.group)
        bad = (all_i[:1, ::k] == _POISON).cpu().tolist()[0] if all_i.shape[0] else [False] * world   # first slot of every rank's block
        if err is not None or any(bad):
            failed = [r for r, b in enumerate(bad) if b] or [rank]
            raise ShardRoundError(f"shard scan failed on rank(s) {failed}: the round's queries have no complete result") from err
        if all_s.shape[1] == k:
            return all_s, all_i
        return self._merge(all_s, all_i, k)

    def _unit_bf16(self, query_emb: torch.Tensor) -> torch.Tensor:
        q = query_emb.to(self.device, dtype=torch.float32)
        return (q / q.norm(dim=1, keepdim=True).clamp_min(1e-12)).to(torch.bfloat16).contiguous()

    def search(self, query_emb: torch.Tensor, k: int):
        """query_emb [Q, D] (this rank's queries) -> (cosine scores [Q, k] fp32, GLOBAL rows [Q, k] int32) for them."""
        q16 = self._unit_bf16(query_emb)
        world, rank = _world(self.group)
        if world == 1:
            return self._local_topk(q16, k)
        if self._tick_fronts:
            raise RuntimeError("this index is being served by a lock-step tick front (as_retriever at world > 1, "
                               "queries='partitioned'): a direct search() would interleave its collectives with the front's; "
                               "go through retrieve(), or close() the retriever first")
        with self._collective_lock:
            nq = q16.shape[0]
            if self.queries == "partitioned" and not self.ragged_queries:
                return self._round_partitioned(q16, k, nq)
            if self.queries == "partitioned":
                # ranks may bring different numbers of queries: agree on the largest count (one tiny all-gather)
                counts = torch.zeros(world, dtype=torch.int64, device=q16.device)
                dist.all_gather_into_tensor(counts, torch.tensor([nq], dtype=torch.int64, device=q16.device), group=self.group)
                return self._round_partitioned(q16, k, int(counts.max().item()))
            return self._collective_topk(q16, k)

    def _round_partitioned(self, q16: torch.Tensor, k: int, nmax: int):
        """One collective round in which rank r brings q16[r] ([nq_r, D] unit bf16, nq_r <= nmax, nmax agreed beforehand):
        pad to nmax with zero rows (they score 0 everywhere and are dropped again), all-gather the queries, every shard
        scans the whole gathered batch, ONE all-gather of the packed partial top-k, merge, keep this rank's slice."""
        world, rank = _world(self.group)
        nq = q16.shape[0]
        if nmax == 0:
            return (torch.empty((0, k), dtype=torch.float32, device=q16.device),
                    torch.empty((0, k), dtype=torch.int32, device=q16.device))
        if nq < nmax:
            q16 = torch.cat([q16, torch.zeros((nmax - nq, q16.shape[1]), dtype=q16.dtype, device=q16.device)], 0)
        all_q = _sh.gather_queries(q16, self.group)
        s, i = self._collective_topk(all_q, k)
        return s[rank * nmax: rank * nmax + nq], i[rank * nmax: rank * nmax + nq]

    def as_retriever(self, similarity_top_k: int = 10, coalesce: bool = True, max_batch: int = 64,
                     max_wait_s: float = 0.0, **_kw) -> "ShardedHipVectorRetriever":
        """One process (world 1): the leader/follower coalescer of ``HipVectorRetriever``.  Several ranks with
        ``queries="partitioned"``: a TICK front (``_TickFront``) -- every rank queues its own callers and all ranks run
        collective rounds in lock step, so coalescing stays on and each rank embeds (and, above this retriever, reranks)
        only ITS OWN callers' queries: N GPUs serve N times the queries, not N copies of the same ones.
        ``queries="replicated"`` keeps one collective round per call (the front end hands every rank the same request)."""
        if self.leaf_ids is None or self.docstore is None:
            raise ValueError("this ShardedHipVectorIndex was built without node tables (search-only)")
        world, _ = _world(self.group)
        tick = coalesce and world > 1 and self.queries == "partitioned"
        return ShardedHipVectorRetriever(self, similarity_top_k, coalesce=coalesce and world == 1, max_batch=max_batch,
                                         max_wait_s=max_wait_s, tick=tick)

    # what HipVectorRetriever's shared code reads
    def snapshot(self):
        return None, self.leaf_ids


class _TickSlot:
    __slots__ = ("bundle", "event", "result", "error")

    def __init__(self, bundle):
        self.bundle, self.event, self.result, self.error = bundle, threading.Event(), None, None


class _TickFront:
    """Lock-step serving front for a row-sharded index with one process per GPU ("pairs can be scattered
    to all 8 GPUs regardless of which shard found them").

    Every rank runs ONE tick thread.  A tick: take up to ``max_batch`` queued callers; embed THEIR queries on this GPU;
    all-gather (query count, closing flag) -- the only collective of an idle tick; if any rank brought queries: all-gather
    the padded query blocks, every shard scans the gathered batch, one all-gather of the packed partial top-k, merge,
    and each rank turns ITS slice into nodes for its own callers.  Ranks therefore always issue the same collectives in
    the same order whatever their callers do, concurrent callers of a rank share its rounds (coalescing stays on), and
    embedding -- and the reranker above -- only ever see a rank's own queries.  A query whose embedding fails is answered
    with its exception before the round; it never reaches a collective.  A shard scan that fails INSIDE a round is carried through
    the round's all-gather as a poisoned list: the round's callers on every rank get ``ShardRoundError`` and the next tick starts
    aligned (a rank that simply stopped would leave the others in a collective for the process group's timeout).  ``close()`` raises this rank's closing flag;
    the thread keeps serving the other ranks' rounds (they need this shard) until every rank has raised its own."""

    def __init__(self, retriever: "ShardedHipVectorRetriever", max_batch: int, idle_sleep_s: float = 2e-4, idle_sleep_max_s: float = 5e-3):
        self.retriever, self.index = retriever, retriever.index
        self.max_batch = max(1, int(max_batch))
        # Idle ticks back off: an idle front used to run a collective + a host sync every 200 us for the life of the retriever
        # (~5 k collectives/s on every GPU of an idle server).  The pause doubles per idle tick up to idle_sleep_max_s and
        # snaps back on the first tick that carries a query; the count of idle ticks is a function of what every rank saw in
        # the same collectives, so all ranks pause alike.  A local submit() cuts this rank's own pause short (its tick then
        # waits in the collective for the slowest sleeper: at most idle_sleep_max_s of added latency for the first query after
        # an idle spell).
        self.idle_sleep_s, self.idle_sleep_max_s = idle_sleep_s, max(idle_sleep_s, idle_sleep_max_s)
        self._wake = threading.Event()
        self._lock = threading.Lock()
        self._queue: List[_TickSlot] = []
        self._closing = False
        self._dead: Optional[BaseException] = None
        self.ticks = self.rounds = self.items = 0
        self.index._tick_fronts += 1
        self._thread = threading.Thread(target=self._loop, name="tt-tick-front", daemon=True)
        self._thread.start()

    def submit(self, bundle):
        slot = _TickSlot(bundle)
        with self._lock:
            if self._closing or self._dead is not None:
                raise RuntimeError("the sharded retriever's serving front is closed") from self._dead
            self._queue.append(slot)
        self._wake.set()
        slot.event.wait()
        if slot.error is not None:
            raise slot.error
        return slot.result

    def close(self, timeout: Optional[float] = None) -> None:
        with self._lock:
            self._closing = True
        self._wake.set()
        self._thread.join(timeout)

    def _embed_own(self, batch: List[_TickSlot]):
        """-> (slots that go into the round, their unit bf16 embeddings [n, D]); failures are answered here."""
        r = self.retriever
        try:
            return batch, self.index._unit_bf16(r._query_matrix([s.bundle for s in batch]))
        except Exception as first:  # noqa: BLE001 - isolate the offender(s): each query alone
            if len(batch) == 1:
                batch[0].error = first
                batch[0].event.set()
                return [], None
        good, rows = [], []
        for s in batch:
            try:
                rows.append(self.index._unit_bf16(r._query_matrix([s.bundle])))
                good.append(s)
            except Exception as exc:  # noqa: BLE001
                s.error = exc
                s.event.set()
        return good, (torch.cat(rows, 0) if rows else None)

    def _loop(self) -> None:
        idx, r = self.index, self.retriever
        world, rank = _world(idx.group)
        dev = idx.device
        batch: List[_TickSlot] = []
        idle_ticks = 0
        import contextlib

        ctx = contextlib.nullcontext()
        if dev.type == "cuda":
            torch.cuda.set_device(dev)         # (the current device is per thread)
            if os.environ.get("TT_RETRIEVE_STREAM", "1") != "0" and _lib.isa_checked():       # see HipVectorRetriever._gpu_stream
                # the rounds' GPU work (query embedding, the collectives' device side, the shard scan, the merge) on the
                # front's own high-priority stream: it must not queue behind the rerank batches of this rank's callers
                # (HipVectorRetriever._gpu_stream: one idle gap per scan batch otherwise)
                stream = torch.cuda.Stream(device=dev, priority=-1)
                if idx._written is not None:
                    stream.wait_event(idx._written)
                ctx = torch.cuda.stream(stream)
        try:
            with ctx:
                while True:
                    with self._lock:
                        batch = self._queue[: self.max_batch]
                        del self._queue[: len(batch)]
                        closing = self._closing and not self._queue
                    q16, round_error = None, None
                    if batch:
                        batch, q16 = self._embed_own(batch)
                    nq = len(batch)
                    flags = torch.zeros((world, 2), dtype=torch.int64, device=dev)
                    with idx._collective_lock:
                        dist.all_gather_into_tensor(flags.view(-1), torch.tensor([nq, 1 if (closing and nq == 0) else 0],
                                                                                 dtype=torch.int64, device=dev), group=idx.group)
                        flags_h = flags.cpu()
                        nmax = int(flags_h[:, 0].max().item())
                        self.ticks += 1
                        if nmax == 0:
                            if bool(flags_h[:, 1].all().item()):
                                return                                   # every rank is closing and idle: all leave at this tick
                            hits = None
                        else:
                            k = min(r.similarity_top_k, idx.n_total)
                            if q16 is None:
                                q16 = torch.zeros((0, idx.dim), dtype=torch.bfloat16, device=dev)
                            try:
                                hits = idx._round_partitioned(q16, k, nmax) if k >= 1 else None
                            except ShardRoundError as exc:      # raised by EVERY rank in this round: the protocol stays aligned
                                round_error, hits = exc, None
                            self.rounds += 1
                    if nmax == 0:
                        idle_ticks += 1
                        self._wake.wait(min(self.idle_sleep_max_s, self.idle_sleep_s * (1 << min(idle_ticks - 1, 10))))
                        self._wake.clear()
                        continue
                    idle_ticks = 0
                    self._wake.clear()
                    if batch and round_error is not None:           # this round's callers fail, the front lives on
                        for s_ in batch:
                            s_.error = round_error
                            s_.event.set()
                        batch = []
                    if batch:
                        if hits is None:
                            results = [[] for _ in batch]
                        else:
                            scores, rows = hits[0].cpu().tolist(), hits[1].cpu().tolist()
                            results = [r.nodes_from_hits(s_, r_, idx.leaf_ids) for s_, r_ in zip(scores, rows)]
                        self.items += len(batch)
                        for s_, res in zip(batch, results):
                            s_.result = res
                            s_.event.set()
                        batch = []
        except BaseException as exc:  # noqa: BLE001 - a failed collective: nobody may wait forever
            with self._lock:
                self._dead = exc
                self._closing = True
                pending = batch + self._queue
                self._queue = []
            for s_ in pending:
                if not s_.event.is_set():
                    s_.error = RuntimeError(f"sharded serving front stopped: {exc!r}")
                    s_.event.set()
        finally:
            idx._tick_fronts -= 1


class ShardedHipVectorRetriever(HipVectorRetriever):
    """``retrieve(query)`` over the sharded index: same surface, same node construction as ``HipVectorRetriever``."""

    def __init__(self, index, similarity_top_k: int = 10, coalesce: bool = True, max_batch: int = 64, max_wait_s: float = 0.0,
                 tick: bool = False):
        super().__init__(index, similarity_top_k, coalesce=coalesce, max_batch=max_batch, max_wait_s=max_wait_s)
        self._tick = _TickFront(self, max_batch) if tick else None

    def retrieve(self, query) -> List[NodeWithScore]:
        if self._tick is not None:
            from .schema import as_query_bundle

            return self._tick.submit(as_query_bundle(query))
        return super().retrieve(query)

    _retrieve = retrieve

    def close(self, timeout: Optional[float] = None) -> None:
        """Leave the lock-step serving front (multi-rank ``queries="partitioned"`` only).  SPMD: EVERY rank must call it -- the
        front's thread keeps answering the other ranks' rounds (they need this shard) until all ranks have raised their closing
        flag, and while it lives it owns the process group's collectives (``index.search()`` from another thread raises)."""
        if self._tick is not None:
            self._tick.close(timeout)

    def _retrieve_batch(self, bundles) -> List[List[NodeWithScore]]:
        idx = self.index
        k = min(self.similarity_top_k, idx.n_total)
        if k < 1:
            return [[] for _ in bundles]
        world, _ = _world(idx.group)
        stream = self._gpu_stream() if world == 1 and idx.device.type == "cuda" else None    # (collective rounds stay where they are)
        if stream is None:
            scores, rows = idx.search(self._query_matrix(bundles), k)
            scores, rows = scores.cpu().tolist(), rows.cpu().tolist()
        else:
            with torch.cuda.stream(stream):
                if idx._written is not None:
                    stream.wait_event(idx._written)
                scores, rows = idx.search(self._query_matrix(bundles), k)
                scores, rows = scores.cpu().tolist(), rows.cpu().tolist()
        return [self.nodes_from_hits(s, r, idx.leaf_ids) for s, r in zip(scores, rows)]
