"""Host side of the ingest in WORKER PROCESSES: sentence splitting, hierarchical parsing and tokenization leave the
process that feeds the GPU.

The composed ingest (``index_builder.build_index``: the reference's ``build_module``, ``indexing/builder.py:376-453``)
is host-bound from strings: a kernel trace of round 3's single-process pipeline shows the GPU 0.75 busy, its 45 idle gaps
the stretches where ONE Python thread splits sentences, walks the chunk hierarchy and tokenizes (a second parsing thread
lost to the GIL, ``profiles/r03_ingest_busy.log``).  Here that work runs in ``W`` spawned processes; the feeding process
only moves arrays:

    worker, phase "split"   texts of a document chunk -> sentences, sentence groups, their token ids            (flat int32 + lengths)
    feeder                  group tokens -> embeddings -> adjacent distances (GPU, enqueued) -> host, one event per chunk
    worker, phase "cut"     distances -> per-document percentile cuts -> semantic chunks -> hierarchical nodes,
                            leaf token ids                                                                      (nodes + flat int32)
    feeder                  docstore, leaf tokens -> embeddings (GPU, enqueued) -> index rows

A chunk's two phases run on the SAME worker (its sentences stay there: only distances travel back), chunks are dealt
round-robin and consumed in document order, so the index is the one the in-process path builds (node ids apart: uuid4
either way) -- embeddings do not depend on the batch they travel in.  A worker is a FRESH interpreter (``python -c``, pickled
messages over its stdin / stdout): it imports numpy and this package's host modules only -- no torch, never the GPU -- it does
not re-import the parent's main module (as ``multiprocessing``'s spawn would), and it may be started after the parent has
touched the GPU (nothing is forked).  Workers are kept for the life of the process and reused by later builds.
"""
from __future__ import annotations

import atexit
import contextlib
import logging
import os
import pickle
import select
import struct
import subprocess
import sys
import threading
import time
from collections import deque
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

logger = logging.getLogger(__name__)


class _PipeConn:
    """Length-prefixed pickles over a pair of raw file descriptors: send / recv / poll."""

    def __init__(self, rfd: int, wfd: int, duplex_safe: bool = False):
        self.rfd, self.wfd, self.buf = rfd, wfd, bytearray()
        # duplex_safe (the FEEDER's end): a send never blocks on a full pipe without draining the other direction -- the worker
        # at the far end may itself be blocked writing a large answer (800 KB of token ids against a 64 KB pipe) that nobody reads
        # while the feeder sits in write(): both would wait forever.  The write end is non-blocking and the send loop reads
        # whatever arrives in between.
        self.duplex_safe = duplex_safe
        self.out, self.out_pos = bytearray(), 0          # (feeder's end) bytes posted but not yet written: post() / pump()
        if duplex_safe:
            os.set_blocking(wfd, False)

    # ---- non-blocking sends (round 5).  The feeder hands a worker its NEXT work unit while it is still busy with the current one; the
    # unit (hundreds of KB of document text) is larger than the pipe, so a blocking send parked the feeder until that worker came
    # back to read -- 44 of 78 seconds of a 6000-document build at the reference's chunk geometry, the GPU idle meanwhile
    # (profiles/r05_ingest_ref_geometry_profile_6000_before.log).  post() queues the bytes, pump() writes what the pipe takes now.
    def post(self, obj) -> None:
        data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
        if self.out_pos and self.out_pos == len(self.out):
            self.out, self.out_pos = bytearray(), 0
        self.out += struct.pack("<Q", len(data))
        self.out += data
        self.pump()

    def pending(self) -> bool:
        return self.out_pos < len(self.out)

    def pump(self) -> bool:
        """-> True when nothing is left to write."""
        while self.out_pos < len(self.out):
            try:
                n = os.write(self.wfd, memoryview(self.out)[self.out_pos:self.out_pos + (1 << 16)])
            except BlockingIOError:
                return False
            self.out_pos += n
        if self.out_pos:
            self.out, self.out_pos = bytearray(), 0
        return True

    def send(self, obj) -> None:
        data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
        view = memoryview(struct.pack("<Q", len(data)) + data)
        while len(view):
            if not self.duplex_safe:
                view = view[os.write(self.wfd, view):]
                continue
            readable, writable, _ = select.select([self.rfd], [self.wfd], [])
            if readable:
                chunk = os.read(self.rfd, 1 << 20)
                if not chunk:
                    raise EOFError("ingest worker closed its pipe")
                self.buf += chunk
            if writable:
                try:
                    view = view[os.write(self.wfd, view[: 1 << 16]):]
                except BlockingIOError:
                    pass

    def _fill(self, block: bool) -> bool:
        """Read what is there (block: wait for at least one byte).  -> False at end of file."""
        if not block and not select.select([self.rfd], [], [], 0)[0]:
            return True
        chunk = os.read(self.rfd, 1 << 20)
        if not chunk:
            return False
        self.buf += chunk
        return True

    def _complete(self) -> bool:
        return len(self.buf) >= 8 and len(self.buf) >= 8 + struct.unpack_from("<Q", self.buf)[0]

    def poll(self) -> bool:
        while not self._complete():
            before = len(self.buf)
            if not self._fill(False) or len(self.buf) == before:
                break
        return self._complete()

    def recv(self):
        while not self._complete():
            if self.pending():                       # posted bytes first: the reply may be the answer to them
                readable, writable, _ = select.select([self.rfd], [self.wfd], [])
                if writable:
                    self.pump()
                if not readable:
                    continue
            if not self._fill(True):
                raise EOFError("ingest worker closed its pipe")
        n = struct.unpack_from("<Q", self.buf)[0]
        obj = pickle.loads(bytes(self.buf[8:8 + n]))
        del self.buf[:8 + n]
        return obj


def tokenizer_spec(tk):
    """What a worker needs to rebuild the feeder's tokenizer."""
    from .tokenization import HFTokenizer, HashTokenizer

    if isinstance(tk, HashTokenizer):
        return ("hash", tk.arch, tk.vocab_size)
    if isinstance(tk, HFTokenizer):
        return ("hf", tk._json, tk.arch)
    raise TypeError(f"ingest workers cannot rebuild a {type(tk).__name__}: pass workers=0 to stay in process")


def _tokenizer_from_spec(spec):
    from .tokenization import HFTokenizer, HashTokenizer

    if spec[0] == "hash":
        return HashTokenizer(spec[1], spec[2])
    return HFTokenizer(None, spec[2], json_str=spec[1])


def _flatten(seqs: Sequence[Sequence[int]]):
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int32, count=len(seqs))
    flat = np.empty(int(lens.sum()), dtype=np.int32)
    pos = 0
    for s, n in zip(seqs, lens):
        flat[pos:pos + n] = s
        pos += n
    return flat, lens


def unflatten(flat: np.ndarray, lens: np.ndarray) -> List[np.ndarray]:
    ends = np.cumsum(lens)
    return [flat[e - n:e] for e, n in zip(ends, lens)]


def _doc_record(doc):
    return (doc.get_content() if hasattr(doc, "get_content") else str(doc), dict(getattr(doc, "metadata", {}) or {}),
            list(getattr(doc, "excluded_embed_metadata_keys", None) or []), list(getattr(doc, "excluded_llm_metadata_keys", None) or []))


class _Host:
    """The host-only steps, shared by the workers and by the in-process reference run of the tests."""

    def __init__(self, spec: Dict):
        from .node_parser import HierarchicalNodeParser

        self.tk = _tokenizer_from_spec(spec["tokenizer"])
        self.state: Dict[int, tuple] = {}
        if spec.get("role") == "pairs":            # a rerank front's pair tokeniser (PairTokenizerPool): the tokenizer and nothing else
            return
        self.max_length = spec["max_length"]
        self.prefix = spec["text_instruction"] or ""
        self.buffer_size = spec["buffer_size"]
        self.percentile = spec["percentile"]
        counter = None
        if spec.get("token_counter", "words") == "embedder":       # chunk sizes in the embedder's sub-word tokens (index_builder._counter)
            from .node_parser import tokenizer_counter

            counter = tokenizer_counter(self.tk)
        self.hier = HierarchicalNodeParser.from_defaults(chunk_sizes=spec["chunk_sizes"], chunk_overlap=spec["chunk_overlap"],
                                                         tokenizer=counter)

    def pairs(self, pairs: List[tuple], max_length: int):
        """(query, passage) strings -> (flat ids, lengths, flat token-type ids or None): CrossEncoder's tokenizer call on a slice."""
        enc = self.tk.encode_pair_batch(pairs, max_length)
        flat, lens = _flatten([e[0] for e in enc])
        types = None
        if enc and enc[0][1] is not None:
            types, _ = _flatten([e[1] for e in enc])
        return flat, lens, types

    def _tokens(self, texts: List[str]):
        full = [self.prefix + t for t in texts] if self.prefix else texts
        seqs = self.tk.encode_batch(full, self.max_length) if hasattr(self.tk, "encode_batch") else [self.tk.encode(t, self.max_length) for t in full]
        return _flatten(seqs)

    def split(self, chunk_id: int, docs: List[tuple]):
        """-> (spans [(first group, sentences) or (-1, 0) per document], flat group tokens, lengths)"""
        from .semantic_host import split_sentences

        b = self.buffer_size
        sents = [split_sentences(d[0]) for d in docs]
        spans, groups = [], []
        for ss in sents:
            if len(ss) > 1:
                spans.append((len(groups), len(ss)))
                groups.extend("".join(ss[max(0, i - b): i + b + 1]) for i in range(len(ss)))
            else:
                spans.append((-1, 0))
        self.state[chunk_id] = (docs, sents, spans)
        flat, lens = self._tokens(groups) if groups else (np.zeros(0, np.int32), np.zeros(0, np.int32))
        return np.asarray(spans, dtype=np.int64).reshape(-1, 2), flat, lens

    def cut(self, chunk_id: int, dist: np.ndarray):
        """distances of the chunk's concatenated groups -> (all hierarchy nodes, leaf positions, flat leaf tokens, lengths)"""
        from .schema import TextNode
        from .semantic_host import breakpoints_from_distances, join_chunks

        docs, sents, spans = self.state.pop(chunk_id)
        sem_nodes = []
        for (text, meta, ex_embed, ex_llm), ss, (lo, n) in zip(docs, sents, spans):
            if lo < 0:
                chunks = [text] if text.strip() else []
            else:
                chunks = join_chunks(ss, breakpoints_from_distances(dist[lo:lo + n - 1], self.percentile))
            for c in chunks:
                nd = TextNode(text=c, metadata=dict(meta))
                for key, val in (("excluded_embed_metadata_keys", ex_embed), ("excluded_llm_metadata_keys", ex_llm)):
                    if val:
                        try:
                            setattr(nd, key, list(val))
                        except Exception:  # noqa: BLE001
                            pass
                sem_nodes.append(nd)
        return self._hierarchy(sem_nodes)

    def parse(self, docs: List[tuple]):
        """hierarchical strategy: documents -> nodes (no semantic pass)"""
        from .schema import TextNode

        srcs = []
        for text, meta, ex_embed, ex_llm in docs:
            nd = TextNode(text=text, metadata=dict(meta))
            for key, val in (("excluded_embed_metadata_keys", ex_embed), ("excluded_llm_metadata_keys", ex_llm)):
                if val:
                    try:
                        setattr(nd, key, list(val))
                    except Exception:  # noqa: BLE001
                        pass
            srcs.append(nd)
        return self._hierarchy(srcs)

    def _hierarchy(self, sources):
        from .schema import MetadataMode

        nodes = self.hier.get_nodes_from_documents(sources)
        leaf_pos = np.asarray([i for i, n in enumerate(nodes) if not getattr(n, "child_ids", None)], dtype=np.int64)
        texts = [nodes[i].get_content(metadata_mode=MetadataMode.EMBED) for i in leaf_pos]
        flat, lens = self._tokens(texts) if texts else (np.zeros(0, np.int32), np.zeros(0, np.int32))
        return nodes, leaf_pos, flat, lens


def _stdio_worker():
    """Entry point of a worker process: messages on stdin, answers on stdout (anything the host code prints goes to stderr)."""
    conn = _PipeConn(os.dup(0), os.dup(1))
    os.dup2(2, 1)                       # stray prints must not corrupt the message stream
    _worker_main(conn, conn.recv())


def _worker_main(conn, spec):
    host = _Host(spec)
    conn.send(("ready",))
    while True:
        msg = conn.recv()
        op = msg[0]
        if op == "stop":
            break
        try:
            if op == "split":
                conn.send(("split", msg[1]) + tuple(host.split(msg[1], msg[2])))
            elif op == "cut":
                conn.send(("cut", msg[1]) + tuple(host.cut(msg[1], msg[2])))
            elif op == "parse":
                conn.send(("cut", msg[1]) + tuple(host.parse(msg[2])))
            elif op == "pairs":
                conn.send(("pairs", msg[1]) + tuple(host.pairs(msg[2], msg[3])))
        except Exception as exc:  # noqa: BLE001 - reported to the feeder, which raises
            import traceback

            conn.send(("error", msg[1], f"{exc!r}\n{traceback.format_exc()}"))


def default_workers() -> int:
    env = os.environ.get("TT_INGEST_WORKERS")
    if env is not None and env != "":
        return max(0, int(env))
    try:
        cpus = len(os.sched_getaffinity(0))
    except AttributeError:
        cpus = os.cpu_count() or 1
    # up to 8 by default on small hosts; a large host (the 256-thread GPU box) gives a sixteenth of its cores, at most 24: with a real
    # sub-word tokenizer at the reference's chunk geometry one worker prepares ~6-10 documents/s and the GPU consumes ~30
    return max(0, min(max(8, min(24, cpus // 16)), cpus - 2))


_POOLS: Dict[bytes, "IngestWorkers"] = {}
_POOLS_LOCK = threading.Lock()


def get_workers(spec: Dict, workers: int) -> "IngestWorkers":
    """The process's worker pool for this host configuration (started on first use, reused by later builds, closed at exit).
    A pool serves ONE build at a time (``run`` holds its lock): concurrent builds go through ``lease_workers``."""
    key = pickle.dumps((sorted(spec.items(), key=lambda kv: kv[0]), workers))
    with _POOLS_LOCK:
        pool = _POOLS.get(key)
        if pool is None or not pool.alive():
            pool = _POOLS[key] = IngestWorkers(spec, workers)
        return pool


@contextlib.contextmanager
def lease_workers(spec: Dict, workers: int):
    """A pool nobody else is using: the shared one of ``get_workers`` if it is idle, otherwise a PRIVATE pool started for
    this build and closed after it.  Two threads calling ``build_index`` with the same configuration used to share the pipes and the
    ``buffers[w][(op, chunk_id)]`` keys -- both builds number their chunks from 0, so replies of one were consumed by the other,
    and an ``abort`` of one killed the workers under the other."""
    pool = get_workers(spec, workers)
    private = not pool._busy.acquire(blocking=False)
    if private:
        pool = IngestWorkers(spec, workers)
        pool._busy.acquire()
    try:
        yield pool
    finally:
        pool._busy.release()
        if private:
            pool.close()


@atexit.register
def _close_pools() -> None:
    with _POOLS_LOCK:
        pools = list(_POOLS.values())
        _POOLS.clear()
    for pool in pools:
        pool.close()


class IngestWorkers:
    """W host workers behind pipes.  ``run`` drives one build's chunks through them in document order."""

    def __init__(self, spec: Dict, workers: int, worker_threads: Optional[int] = None):
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))          # where the tensor_truth_amd import shim lives
        code = f"import sys; sys.path.insert(0, {root!r}); from tensor_truth_amd.ingest_workers import _stdio_worker; _stdio_worker()"
        self.conns, self.procs = [], []
        # The Rust tokenizer starts one thread per host core in EVERY process that uses it (rayon's default): eight workers + the
        # feeder on a 256-core host are ~2300 threads contending for the same cores, and a worker's reply took 2-3x its single-thread
        # time (profiles/r05_ingest_ref_geometry_profile.log: the feeder blocked on worker replies for half of the build).  The
        # workers are the parallelism here: each gets a small pool of its own (TT_INGEST_WORKER_THREADS, default 2).
        env = dict(os.environ)
        env["RAYON_NUM_THREADS"] = str(worker_threads) if worker_threads else os.environ.get("TT_INGEST_WORKER_THREADS", "2")
        env.setdefault("TOKENIZERS_PARALLELISM", "true")
        for _ in range(workers):
            p = subprocess.Popen([sys.executable, "-c", code], stdin=subprocess.PIPE, stdout=subprocess.PIPE, bufsize=0, env=env)
            self.procs.append(p)
            self.conns.append(_PipeConn(p.stdout.fileno(), p.stdin.fileno(), duplex_safe=True))
        self.buffers: List[Dict] = [dict() for _ in range(workers)]
        self._busy = threading.RLock()              # held for the whole of run(): pipes and reply buffers belong to one build
        for c in self.conns:
            c.send(spec)
        for c in self.conns:                      # (interpreter start + imports: a few tenths of a second, paid once per process)
            msg = c.recv()
            if msg[0] != "ready":
                raise RuntimeError(f"ingest worker failed to start: {msg}")

    def alive(self) -> bool:
        return bool(self.procs) and all(p.poll() is None for p in self.procs)

    def close(self) -> None:
        for c, p in zip(self.conns, self.procs):
            try:
                c.send(("stop",))
                p.stdin.close()
            except Exception:  # noqa: BLE001
                pass
        for p in self.procs:
            try:
                p.wait(timeout=5)
            except Exception:  # noqa: BLE001
                p.kill()
        self.conns, self.procs = [], []

    def abort(self) -> None:
        """Kill the workers without a handshake (a build failed half-way: chunks in flight, half-read frames)."""
        for p in self.procs:
            try:
                p.kill()
            except Exception:  # noqa: BLE001
                pass
        for p in self.procs:
            try:
                p.wait(timeout=5)
            except Exception:  # noqa: BLE001
                pass
        self.conns, self.procs = [], []
        self.buffers = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _store(self, w: int, msg) -> None:
        if msg[0] == "error":
            raise RuntimeError(f"ingest worker {w} failed on chunk {msg[1]}: {msg[2]}")
        self.buffers[w][(msg[0], msg[1])] = msg[2:]

    def pump_all(self) -> None:
        for c in self.conns:
            if c.pending():
                c.pump()

    def poll(self, w: int, key) -> Optional[tuple]:
        while key not in self.buffers[w] and self.conns[w].poll():
            self._store(w, self.conns[w].recv())
        return self.buffers[w].pop(key, None)

    def wait(self, w: int, key) -> tuple:
        """Blocks until worker ``w`` has answered ``key`` -- serving every OTHER pipe meanwhile: posted work units keep flowing to
        the workers that can take them and their replies are stored as they arrive (nobody idles because the feeder waits for one)."""
        while key not in self.buffers[w]:
            self.serve(None)
        return self.buffers[w].pop(key)

    def serve(self, timeout: Optional[float]) -> None:
        """One select() over every pipe: write posted bytes where a pipe takes them, store the replies that have arrived."""
        rfds = {c.rfd: i for i, c in enumerate(self.conns)}
        wfds = {c.wfd: i for i, c in enumerate(self.conns) if c.pending()}
        readable, writable, _ = select.select(list(rfds), list(wfds), [], timeout)
        for fd in writable:
            self.conns[wfds[fd]].pump()
        for fd in readable:
            i = rfds[fd]
            c = self.conns[i]
            if not c._fill(False):
                raise EOFError(f"ingest worker {i} closed its pipe")
            while c._complete():
                self._store(i, c.recv())

    def run(self, documents: Sequence, semantic: bool, embed_tokens: Callable, distances: Callable, on_nodes: Callable,
            chunk_docs: int = 48, inflight_per_worker: int = 3, embed_flat: Optional[Callable] = None,
            leaf_tokens: bool = False) -> None:
        """``embed_tokens(list of int32 arrays) -> embeddings`` (enqueues GPU work), ``distances(embeddings) -> (host array,
        ready())`` (adjacent distances copied back asynchronously; ``ready(block)`` tells / waits), ``on_nodes(nodes,
        leaf positions, leaf embeddings)`` (docstore + index rows), called in document order.
        A build that raises -- in a worker, in a callback, on a dead pipe -- leaves chunks in flight and replies unread: the pool
        is killed (``abort``) and the error re-raised; ``get_workers`` starts a fresh pool for the next build."""
        with self._busy:
            try:
                if embed_flat is None:       # (flat int32 array + lengths, as the workers send them: no per-sequence Python in the feeder)
                    embed_flat = lambda flat, lens: embed_tokens(unflatten(flat, lens))      # noqa: E731
                if leaf_tokens:              # on_nodes(nodes, leaf positions, leaf embeddings, (flat leaf token ids, lengths))
                    cb = on_nodes
                else:
                    cb = lambda nodes, pos, emb, _tok: on_nodes(nodes, pos, emb)              # noqa: E731
                self._run(documents, semantic, embed_flat, distances, cb, chunk_docs, inflight_per_worker)
            except BaseException:
                self.abort()
                raise

    def _run(self, documents, semantic, embed_flat, distances, on_nodes, chunk_docs, inflight_per_worker) -> None:
        W = len(self.conns)
        docs = [_doc_record(d) for d in documents]
        # small first chunks put the GPU to work early; then chunk_docs -- counted in documents of ~7000 characters (the ~1.1 k-word
        # documents chunk_docs was tuned on): a chunk of 48 documents of 6 k words each would be six times the work unit, and a
        # 256-document build only eight chunks for sixteen workers
        bounds, size, unit = [0], max(4, chunk_docs // 8), 7000
        while bounds[-1] < len(docs):
            hi, chars = bounds[-1], 0
            while hi < len(docs) and (hi == bounds[-1] or (hi - bounds[-1] < size and chars + len(docs[hi][0]) <= size * unit)):
                chars += len(docs[hi][0])
                hi += 1
            bounds.append(hi)
            size = min(chunk_docs, size * 2)
        chunks = list(zip(bounds[:-1], bounds[1:]))
        n = len(chunks)
        sent = n_split = n_cut_sent = done = 0
        dist_q: deque = deque()                       # (chunk, host array, ready)
        max_ahead = W * inflight_per_worker
        while done < n:
            progressed = False
            self.pump_all()
            while sent < n and sent - done < max_ahead:                                   # a) hand out chunks
                lo, hi = chunks[sent]
                self.conns[sent % W].post(("split" if semantic else "parse", sent, docs[lo:hi]))
                sent += 1
                progressed = True
            if semantic:
                while n_split < sent:                                                     # b) groups -> GPU
                    r = self.poll(n_split % W, ("split", n_split))
                    if r is None:
                        break
                    spans, flat, lens = r
                    if len(lens):
                        host, ready = distances(embed_flat(flat, lens))
                    else:
                        host, ready = np.zeros(0, np.float32), (lambda block=False: True)
                    dist_q.append((n_split, host, ready))
                    n_split += 1
                    progressed = True
                while dist_q and dist_q[0][2](False):                                     # c) distances -> worker
                    c, host, _ = dist_q.popleft()
                    self.conns[c % W].post(("cut", c, np.array(host, copy=True)))
                    n_cut_sent += 1
                    progressed = True
            else:
                n_split = n_cut_sent = sent
            while done < n_cut_sent:                                                      # d) nodes + leaf tokens -> GPU, index
                r = self.poll(done % W, ("cut", done))
                if r is None:
                    break
                nodes, leaf_pos, flat, lens = r
                on_nodes(nodes, leaf_pos, embed_flat(flat, lens) if len(lens) else None, (flat, lens))
                done += 1
                progressed = True
            if progressed:
                continue
            # nothing could move: wait for the most urgent thing
            if done < n_cut_sent:
                self.buffers[done % W][("cut", done)] = self.wait(done % W, ("cut", done))
            elif dist_q:
                while not dist_q[0][2](False):          # the GPU owes us distances: keep the pipes moving meanwhile
                    self.serve(0.0005)
            elif n_split < sent:
                self.buffers[n_split % W][("split", n_split)] = self.wait(n_split % W, ("split", n_split))


class PairTokenizerPool:
    """Pair tokenisation of a coalesced rerank batch in worker PROCESSES.

    With a real sub-word tokenizer (XLM-R's 250 002-piece Unigram model) the 400 pairs x ~292 tokens of a coalesced batch cost
    ~25 ms in the request process -- the Rust library's thread pool tops out at ~5x on this job whatever its size (16 or 256
    threads: profiles/r05_tokenizer_threads.log), its result lists are built under the GIL, and a retrieval batch's query
    tokenisation queues behind it in the same pool -- which the plugin-surface leg showed as GPU idle gaps (0.84 busy, 83 q/s
    against 107 with one hashed id per word).  Here the pairs are dealt in slices to W single-threaded workers (the ingest workers'
    process + pipe machinery, ``role: pairs``): 133 ms of tokenisation / W in parallel, the parent only pickles strings and
    concatenates int32 arrays.  Results are the in-process tokenizer's, id for id (tests/test_host_logic.py)."""

    MIN_PAIRS = 96

    def __init__(self, tokenizer, workers: int):
        self.pool = IngestWorkers({"role": "pairs", "tokenizer": tokenizer_spec(tokenizer)}, workers, worker_threads=1)
        self.lock = threading.Lock()
        self._seq = 0

    def alive(self) -> bool:
        return self.pool.alive()

    def close(self) -> None:
        self.pool.close()

    DEADLINE_S = 30.0      # a batch of pairs is tens of milliseconds of work: a pool that has not answered by then is hung

    def encode(self, pairs: Sequence, max_length: int, deadline_s: Optional[float] = None):
        """-> (list of int32 arrays, list of type-id arrays or None per pair), in order; None when the caller should tokenise in
        process instead -- the pool is busy (another thread's batch is in it), a worker died or raised (EOFError on its pipe, its own
        error), or the pool did not answer within ``deadline_s``: the in-process tokenizer returns the same ids, so a user's rerank
        request never fails (or hangs) because of the pool.  A failed pool is killed; ``get_pair_pool`` starts a fresh one."""
        if not self.lock.acquire(blocking=False):
            return None
        try:
            pool, W = self.pool, len(self.pool.conns)
            if W == 0:
                return None
            limit = time.monotonic() + (self.DEADLINE_S if deadline_s is None else deadline_s)
            n = len(pairs)
            per = (n + W - 1) // W
            jobs = []
            for w in range(W):
                lo, hi = w * per, min(n, (w + 1) * per)
                if lo >= hi:
                    break
                self._seq += 1
                pool.conns[w].send(("pairs", self._seq, [tuple(p) for p in pairs[lo:hi]], max_length))
                jobs.append((w, self._seq))
            ids, types = [], []
            for w, key in jobs:
                k = ("pairs", key)
                while k not in pool.buffers[w]:
                    left = limit - time.monotonic()
                    if left <= 0:
                        raise TimeoutError(f"pair tokenizer pool: no answer from worker {w} within the deadline")
                    pool.serve(min(left, 1.0))
                flat, lens, tflat = pool.buffers[w].pop(k)
                ids.extend(unflatten(flat, lens))
                types.extend(unflatten(tflat, lens) if tflat is not None else [None] * len(lens))
            return ids, types
        except Exception as exc:  # noqa: BLE001 - a dead / hung / failing worker: the caller falls back, the pool is replaced
            logger.warning("pair tokenizer pool failed (%s: %s); tokenising in process", type(exc).__name__, exc)
            self.pool.abort()
            return None
        except BaseException:
            self.pool.abort()
            raise
        finally:
            self.lock.release()


_PAIR_POOLS: Dict[bytes, PairTokenizerPool] = {}


def pair_workers_default() -> int:
    env = os.environ.get("TT_PAIR_WORKERS")
    if env is not None and env != "":
        return max(0, int(env))
    try:
        cpus = len(os.sched_getaffinity(0))
    except AttributeError:
        cpus = os.cpu_count() or 1
    return 0 if cpus < 16 else min(16, cpus // 8)      # small hosts tokenise in process


_PAIR_POOLS_STARTING: set = set()


def _start_pair_pool(tokenizer, key: bytes, W: int, max_length: int) -> Optional[PairTokenizerPool]:
    """Spawn, warm and publish the pool for ``key`` (the caller has put ``key`` into _PAIR_POOLS_STARTING); runs OUTSIDE the registry lock."""
    pool = None
    try:
        pool = PairTokenizerPool(tokenizer, W)
        # one job through every worker BEFORE the pool is published, at the reranker's own max_length: a worker builds its truncating
        # pair tokenizer (a second parse of the 17 MB tokenizer.json, keyed on the length) on its first pair -- ~0.5 s, measured as a
        # 488-580 ms prepare phase of the first coalesced batch that reached a fresh pool (tools/probes/threads_variance.py).  Paid
        # here, in the starting thread; request threads tokenise in process until the pool is in the registry.
        if pool.encode([("warm up", "a short passage for the pair tokenizer to see once")] * (2 * W), max_length, deadline_s=120.0) is None:
            pool.close()
            pool = None
    except Exception as exc:  # noqa: BLE001 - no workers: every caller tokenises in process
        logger.warning("pair tokenizer pool could not be started (%s: %s)", type(exc).__name__, exc)
        pool = None
    finally:
        with _POOLS_LOCK:
            _PAIR_POOLS_STARTING.discard(key)
            if pool is not None:
                _PAIR_POOLS[key] = pool
    return pool


def get_pair_pool(tokenizer, wait: bool = False, max_length: int = 512) -> Optional[PairTokenizerPool]:
    """The process's pair-tokenisation pool for this tokenizer; None: disabled, a tokenizer the workers cannot rebuild, the pool
    could not be started -- or (``wait=False``, what a request thread passes) it is not up yet: up to 16 interpreters each parsing a
    17 MB tokenizer.json take seconds, which no request waits for -- it tokenises in process meanwhile, and if nobody is starting the
    pool (the first one died, or no reranker warmed it) a background thread is set to.  ``wait=True`` (``warm_pair_pool``, the bench's
    steady-state legs) starts it in the calling thread or waits for whoever does."""
    W = pair_workers_default()
    if W <= 0:
        return None
    try:
        key = pickle.dumps((tokenizer_spec(tokenizer), W))
    except TypeError:
        return None
    with _POOLS_LOCK:
        pool = _PAIR_POOLS.get(key)
        if pool is not None and pool.alive():
            return pool
        mine = key not in _PAIR_POOLS_STARTING
        if mine:
            _PAIR_POOLS_STARTING.add(key)
    if not wait:
        if mine:
            threading.Thread(target=_start_pair_pool, args=(tokenizer, key, W, max_length), name="tt-pair-pool-start", daemon=True).start()
        return None
    if not mine:           # another thread is starting it
        while True:
            time.sleep(0.02)
            with _POOLS_LOCK:
                if key not in _PAIR_POOLS_STARTING:
                    pool = _PAIR_POOLS.get(key)
                    return pool if pool is not None and pool.alive() else None
    return _start_pair_pool(tokenizer, key, W, max_length)


def warm_pair_pool(tokenizer, max_length: int = 512) -> None:
    """Start the pair pool in a background thread (called when a reranker is constructed, with its max_length), so the first coalesced
    batch finds it started AND warm."""
    def _warm():
        try:       # the request process's own truncating pair tokenizer first (a lone caller's first rerank would build it: ~0.5 s)
            if hasattr(tokenizer, "encode_pair_batch"):
                tokenizer.encode_pair_batch([("warm up", "a short passage for the pair tokenizer to see once")], max_length)
        except Exception as exc:  # noqa: BLE001 - a warm-up: the first request builds it instead
            logger.warning("pair tokenizer warm-up failed (%s: %s)", type(exc).__name__, exc)
        if pair_workers_default() > 0:
            get_pair_pool(tokenizer, True, max_length)

    threading.Thread(target=_warm, name="tt-pair-pool-start", daemon=True).start()


@atexit.register
def _close_pair_pools() -> None:
    with _POOLS_LOCK:
        pools = list(_PAIR_POOLS.values())
        _PAIR_POOLS.clear()
    for pool in pools:
        pool.close()
