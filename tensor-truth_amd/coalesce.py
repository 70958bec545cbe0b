"""Cross-caller batching for the plugin surface.

The reference calls ``retriever.retrieve(query)`` and ``reranker.postprocess_nodes(nodes, query_bundle)`` ONE query at a
time, from executor threads (``src/tensortruth/rag_engine.py:418-424``, ``api/routes/chat.py:367-374``,
``services/orchestrator_tool_wrappers.py:238-247``).  On an MI355X a single query leaves the chip nearly idle (one
34-token embedding, one scan tile, 50 rerank pairs), so concurrent callers are merged here into one embed batch, one
scan tile and one rerank batch -- without changing the surface and without a background thread.

``Coalescer`` is leader/follower batching with NATURAL windows: the first caller to arrive while nothing is running
becomes the leader and runs a batch of whatever is queued (at least its own item); callers that arrive while a batch
is on the GPU queue up, and when the batch completes the leader hands leadership to the oldest waiter, which runs the
next batch with everything that accumulated meanwhile.  A lone caller therefore pays no added latency (batch of one,
immediately), and under load the batch size grows to the arrival rate times the batch time.  ``max_wait_s`` > 0 adds
a fixed collection window in front of every batch for callers that prefer throughput.

Failures stay with the caller that caused them: the reference isolates errors per call (an index whose retrieve()
raises is skipped, ``rag_engine.py:453-455``; a rerank that raises falls back to the un-reranked nodes of THAT request,
``services/rag_service.py:347-350``), so when a coalesced batch raises, its members are re-run one by one and only the
offender sees an exception -- the other callers get exactly their serial results.
"""
from __future__ import annotations

import os
import threading
import time
from typing import Any, Callable, List, Optional, Sequence


class _Slot:
    __slots__ = ("item", "event", "done", "result", "error")

    def __init__(self, item):
        self.item = item
        self.event = threading.Event()
        self.done = False
        self.result = None
        self.error = None


class Coalescer:
    """``run_batch(items) -> results`` (same length, same order) is called by ONE thread at a time.

    Pipelined form: ``Coalescer(prepare, execute=..., finish=...)`` -- ``prepare(items) -> prepared`` is the host side
    of a batch (tokenising, packing), ``execute(prepared) -> pending`` ENQUEUES its device side (asynchronous launches
    on the stream), ``finish(pending) -> results`` waits for it and splits the results.  Leadership is handed on as soon
    as a batch is prepared; ``execute`` calls run one at a time in batch order (so batches reach the stream in order,
    back to back -- the GPU never waits for the host between them), ``finish`` runs outside every lock.  A new batch
    is taken off the queue only while fewer than ``depth`` (2) batches are unfinished: one on the GPU, one prepared
    right behind it -- so under load a batch collects everything that arrives during a whole batch time instead of
    whatever arrived during the few milliseconds of a prepare.

    A batch of several callers that raises (in any phase) is re-run member by member, so an exception reaches only the
    caller whose item causes it.  Whatever happens to a leader -- including a ``BaseException`` outside the guarded
    calls -- its batch is answered, its place in the execute order is released and leadership is handed on."""

    FOLLOWER_POLL_S = 0.5      # a waiting caller re-checks at this interval that somebody still leads

    def __init__(self, run_batch: Callable[[List[Any]], Any], max_batch: int = 256, max_wait_s: float = 0.0,
                 execute: Optional[Callable[[Any], Any]] = None, finish: Optional[Callable[[Any], Sequence[Any]]] = None,
                 depth: int = 2):
        if max_batch < 1:
            raise ValueError("max_batch must be >= 1")
        self._run = run_batch
        self._execute = execute
        self._exec_lock = threading.Lock()
        self._exec_turn = 0        # batches execute in the order they were taken off the queue
        self._exec_skipped = set() # tickets whose holder left without executing (released out of order)
        self._exec_cv = threading.Condition(self._exec_lock)
        self._taken = 0
        self._finished = 0
        self._finish = finish
        self._retry_lock = threading.Lock()
        self.depth = max(1, int(os.environ.get("TT_COALESCE_DEPTH", depth)))     # (TT_COALESCE_DEPTH: A/B switch of the pipeline depth)
        self.max_batch = max_batch
        self.max_wait_s = max_wait_s
        self._lock = threading.Lock()
        self._queue: List[_Slot] = []
        self._running = False
        self.batches = 0          # statistics: batches run / items served / batches re-run member by member after a failure
        self.items = 0
        self.isolated = 0

    def submit(self, item):
        slot = _Slot(item)
        with self._lock:
            self._queue.append(slot)
            lead = not self._running
            if lead:
                self._running = True
        while not lead:
            if slot.event.wait(self.FOLLOWER_POLL_S):
                if slot.done:
                    break
                slot.event.clear()      # woken as the new leader: the slot is still queued
                lead = True
            else:
                # nobody woke us: if leadership was lost (it cannot be, short of a killed thread) the queue head takes it
                with self._lock:
                    if not slot.done and not self._running and self._queue and self._queue[0] is slot:
                        self._running = True
                        lead = True
        if lead and not slot.done:
            self._lead(slot)
        if slot.error is not None:
            raise slot.error
        return slot.result

    # ---- the leader's turn ---------------------------------------------------------------------------------
    def _chain(self, items):
        """All phases for `items`, in this thread (the member-by-member re-run of a failed batch)."""
        out = self._run(items)
        if self._execute is not None:
            out = self._execute(out)
            if self._finish is not None:
                out = self._finish(out)
        return out

    def _isolate(self, batch, failure) -> None:
        """A batch of several callers failed: each member alone, so only the offender raises."""
        if len(batch) == 1 or not isinstance(failure, Exception):
            self._deliver(batch, None, failure)
            return
        with self._retry_lock:
            outcomes = []
            for s in batch:
                try:
                    r = list(self._chain([s.item]))
                    if len(r) != 1:
                        raise RuntimeError(f"coalesced batch returned {len(r)} results for 1 item")
                    outcomes.append((r[0], None))
                except Exception as exc:  # noqa: BLE001 - this member's own failure
                    outcomes.append((None, exc))
        with self._lock:
            self.batches += 1
            self.isolated += 1
            self.items += len(batch)
            for s, (r, e) in zip(batch, outcomes):
                s.result, s.error, s.done = r, e, True

    def _release_turn(self, ticket, skipped: bool) -> None:
        with self._exec_cv:
            if skipped and self._exec_turn != ticket:
                self._exec_skipped.add(ticket)
            else:
                self._exec_turn += 1
            while self._exec_turn in self._exec_skipped:
                self._exec_skipped.discard(self._exec_turn)
                self._exec_turn += 1
            self._exec_cv.notify_all()

    def _lead(self, slot) -> None:
        batch, ticket = None, None
        handed = turn_released = finished = delivered = False
        try:
            if self.max_wait_s > 0:
                deadline = time.perf_counter() + self.max_wait_s
                while time.perf_counter() < deadline:
                    with self._lock:
                        if len(self._queue) >= self.max_batch:
                            break
                    time.sleep(min(2e-4, self.max_wait_s))
            if self._execute is not None:
                with self._exec_cv:         # pipeline depth: wait until fewer than `depth` batches are unfinished
                    while self._taken - self._finished >= self.depth:
                        self._exec_cv.wait()
            with self._lock:
                batch = self._queue[: self.max_batch]
                del self._queue[: len(batch)]
                ticket = self._taken
                self._taken += 1
            # the queue is FIFO and a leader is either the first arrival or the promoted queue head, so its own slot is
            # always the first item of the batch it takes
            assert batch and batch[0] is slot
            prepared, failure = None, None
            try:
                prepared = self._run([s.item for s in batch])
            except BaseException as exc:  # noqa: BLE001
                failure = exc
            if self._execute is None:
                if failure is None:
                    self._deliver(batch, prepared, None)
                else:
                    self._isolate(batch, failure)
                delivered = True
            self._hand_over()               # two-phase: BEFORE executing -- the next batch is prepared meanwhile
            handed = True
            if self._execute is not None:
                with self._exec_cv:         # device phases run one at a time, in the order the batches were taken
                    while self._exec_turn != ticket:
                        self._exec_cv.wait()
                try:
                    if failure is None:
                        prepared = self._execute(prepared)
                except BaseException as exc:  # noqa: BLE001
                    failure = exc
                finally:
                    self._release_turn(ticket, skipped=False)
                    turn_released = True
                try:
                    if failure is None and self._finish is not None:
                        prepared = self._finish(prepared)
                except BaseException as exc:  # noqa: BLE001
                    failure = exc
                finally:
                    with self._exec_cv:
                        self._finished += 1
                        self._exec_cv.notify_all()
                    finished = True
                if failure is None:
                    self._deliver(batch, prepared, None)
                else:
                    self._isolate(batch, failure)
                delivered = True
        except BaseException as exc:  # noqa: BLE001 - anything outside the guarded calls (interrupts, a failed wait)
            if batch is None:
                with self._lock:            # died before taking a batch: answer at least its own slot
                    if slot in self._queue:
                        self._queue.remove(slot)
                slot.error, slot.done = exc, True
            elif not delivered:
                self._deliver(batch, None, exc)
                delivered = True
            raise
        finally:
            if self._execute is not None and ticket is not None:
                if not turn_released:
                    self._release_turn(ticket, skipped=True)
                if not finished:
                    with self._exec_cv:
                        self._finished += 1
                        self._exec_cv.notify_all()
            if not handed:
                self._hand_over()
            for s in (batch or [])[1:]:
                s.event.set()

    def _hand_over(self) -> None:
        """The oldest waiter leads the next batch (a leader never serves others forever); nobody waiting: idle."""
        with self._lock:
            nxt = self._queue[0] if self._queue else None
            if nxt is None:
                self._running = False
        if nxt is not None:
            nxt.event.set()

    def _deliver(self, batch, results, failure) -> None:
        if failure is None:
            try:
                results = list(results)
                if len(results) != len(batch):
                    raise RuntimeError(f"coalesced batch returned {len(results)} results for {len(batch)} items")
            except BaseException as exc:  # noqa: BLE001
                failure = exc
        with self._lock:
            self.batches += 1
            self.items += len(batch)
            for i, s in enumerate(batch):
                if failure is None:
                    s.result = results[i]
                else:
                    s.error = failure
                s.done = True
