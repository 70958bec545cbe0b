"""Exact similarity scan + top-k on MI355X (host side of include/tt_hip.h's scan API).

Stands in for the vector search the reference reaches through
``index.as_retriever(similarity_top_k=...)`` (``src/tensortruth/rag_engine.py:639``)
-> ``ChromaVectorStore.query``: given L2-normalised query embeddings it returns,
per query, the ``k`` corpus rows with the largest dot product, ordered by
(score desc, row index asc).  The corpus is a row-major bf16 matrix resident in
HBM (a shard of it under multi-GPU row sharding, see ``sharded.py``).
"""
from __future__ import annotations

import threading
from typing import Optional, Tuple

import torch

from . import _lib


def _require_cuda(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} must live on a HIP device (got {t.device}); tensor_truth_amd has no CPU path"
        )


class _Workspace(threading.local):
    """Per-thread scratch buffers, so concurrent retrieve() calls from the
    reference's executor threads (rag_engine.py:420-424) never share scratch."""

    def __init__(self):
        self.bufs = {}

    def get(self, device: torch.device, nbytes: int) -> torch.Tensor:
        key = (device.type, device.index)
        buf = self.bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
            self.bufs[key] = buf
        return buf


_ws = _Workspace()


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class ScanShadow:
    """The fp8 shadow of a corpus matrix (``tt_scan_shadow_build``, csrc/shadow.hip): one e4m3 byte per element plus two floats
    per row, an EXACT prefilter for the scan of a lone caller -- ``scan_topk(..., shadow=)`` then reads half the bytes and returns
    bit-identical scores and indices.  ``rows`` = how many rows of the matrix the block currently mirrors (``extend`` adds the rows
    appended since; a row rewritten in place -- a tombstone -- needs ``rebuild``)."""

    MIN_ROWS = 1 << 20           # below this a bf16 pass is a few hundred microseconds: nothing to win
    MAX_QUERIES = 4

    def __init__(self, corpus: torch.Tensor, cap_rows: Optional[int] = None):
        _require_cuda(corpus, "corpus")
        if corpus.dtype != torch.bfloat16 or corpus.dim() != 2 or not corpus.is_contiguous():
            raise TypeError("the shadow mirrors a contiguous [N, D] torch.bfloat16 matrix")
        self.dim = int(corpus.shape[1])
        self.cap_rows = int(cap_rows if cap_rows is not None else corpus.shape[0])
        lib = _lib.load_library()
        nbytes = lib.tt_scan_shadow_bytes(self.cap_rows, self.dim)
        if nbytes == 0:
            raise ValueError(f"no shadow for {self.cap_rows} rows x {self.dim}")
        self._raw = torch.empty(nbytes + 256, dtype=torch.uint8, device=corpus.device)
        self.ptr = (self._raw.data_ptr() + 255) // 256 * 256
        self.rows = 0
        self.base_ptr = corpus.data_ptr()
        self.ready = torch.cuda.Event()       # recorded behind every build: a search on ANOTHER stream waits for it (scan_topk)
        self.extend(corpus, int(corpus.shape[0]))

    def extend(self, corpus: torch.Tensor, n_rows: int) -> None:
        """Mirror rows [self.rows, n_rows) of ``corpus`` (rows appended since the last call)."""
        if n_rows > self.cap_rows or corpus.shape[1] != self.dim:
            raise ValueError("the shadow's capacity / width does not fit this matrix")
        if n_rows <= self.rows:
            return
        lib = _lib.load_library()
        with torch.cuda.device(corpus.device):
            rc = lib.tt_scan_shadow_build(corpus.data_ptr(), self.dim, self.rows, n_rows, self.ptr, self.cap_rows, _stream_ptr(corpus.device))
            _lib.check(rc, "tt_scan_shadow_build")
            self.ready.record(torch.cuda.current_stream(corpus.device))
        self.rows = n_rows

    def rebuild(self, corpus: torch.Tensor, n_rows: int) -> None:
        self.rows = 0
        self.extend(corpus, n_rows)

    def serves(self, n_rows: int, n_queries: int, k: int) -> bool:
        return (self.rows >= n_rows >= max(self.MIN_ROWS, 128 * k) and 0 < n_queries <= self.MAX_QUERIES)


def scan_topk(
    corpus: torch.Tensor,
    queries: torch.Tensor,
    k: int,
    idx_base: int = 0,
    exact_dense: bool = False,
    check_overflow: bool = True,
    return_flag: bool = False,
    _packed: Optional[torch.Tensor] = None,
    shadow: Optional[ScanShadow] = None,
):
    """corpus [N,D] bf16, queries [Q,D] bf16 (same HIP device) ->
    (scores [Q,k] fp32, idx [Q,k] int32), idx = idx_base + row, padding (-inf, -1).

    ``exact_dense`` forces the dense-score path (N*Q*4 bytes of scratch).
    ``check_overflow`` reads the device status word (one sync) and transparently
    re-runs through the dense path if a candidate list overflowed.
    ``return_flag`` (tests / diagnostics): no fallback, returns ``(scores, idx, overflowed)``.
    """
    lib = _lib.load_library()
    _require_cuda(corpus, "corpus")
    _require_cuda(queries, "queries")
    if corpus.dtype != torch.bfloat16 or queries.dtype != torch.bfloat16:
        raise TypeError("corpus and queries must be torch.bfloat16")
    if corpus.dim() != 2 or queries.dim() != 2 or corpus.shape[1] != queries.shape[1]:
        raise ValueError(f"shape mismatch: corpus {tuple(corpus.shape)} queries {tuple(queries.shape)}")
    if corpus.device != queries.device:
        raise ValueError("corpus and queries must be on the same device")
    if not corpus.is_contiguous() or not queries.is_contiguous():
        raise ValueError("corpus and queries must be contiguous row-major")
    n, d = corpus.shape
    q = queries.shape[0]
    dev = corpus.device
    if _packed is not None:      # scan_topk_host: scores, indices and the status word in ONE buffer (one copy back, one sync)
        out_s = _packed[: q * k].view(torch.float32).view(q, k)
        out_i = _packed[q * k: 2 * q * k].view(q, k)
    else:
        out_s = torch.empty((q, k), dtype=torch.float32, device=dev)
        out_i = torch.empty((q, k), dtype=torch.int32, device=dev)
    if q == 0:
        return out_s, out_i
    with torch.cuda.device(dev):
        st = _stream_ptr(dev)
        if exact_dense:
            need = lib.tt_scan_exact_workspace_bytes(n, d, q, k)
            ws = _ws.get(dev, need)
            rc = lib.tt_scan_topk_exact(corpus.data_ptr(), n, d, queries.data_ptr(), q, k, idx_base,
                                        out_s.data_ptr(), out_i.data_ptr(), ws.data_ptr(), ws.numel(), st)
            _lib.check(rc, "tt_scan_topk_exact")
            return out_s, out_i
        use_shadow = shadow is not None and shadow.dim == d and shadow.serves(n, q, k)
        need = lib.tt_scan_shadow_workspace_bytes(n, d, q, k) if use_shadow else lib.tt_scan_workspace_bytes(n, d, q, k)
        ws = _ws.get(dev, need + 256)
        base = (ws.data_ptr() + 255) // 256 * 256
        if _packed is not None:
            flag = _packed[2 * q * k:]
            flag.zero_()
        else:
            flag = torch.zeros(1, dtype=torch.int32, device=dev)
        if use_shadow:
            torch.cuda.current_stream(dev).wait_event(shadow.ready)     # (built / extended on whatever stream its first user ran on)
            # a lone caller's scan: one pass over the fp8 shadow, the survivors re-scored from the bf16 rows -- the same bits out
            rc = lib.tt_scan_topk_shadow(corpus.data_ptr(), shadow.ptr, shadow.cap_rows, n, d, queries.data_ptr(), q, k, idx_base,
                                         out_s.data_ptr(), out_i.data_ptr(), base, ws.numel() - (base - ws.data_ptr()),
                                         flag.data_ptr(), st)
            _lib.check(rc, "tt_scan_topk_shadow")
        else:
            rc = lib.tt_scan_topk(corpus.data_ptr(), n, d, queries.data_ptr(), q, k, idx_base,
                                  out_s.data_ptr(), out_i.data_ptr(), base, ws.numel() - (base - ws.data_ptr()),
                                  flag.data_ptr(), st)
            _lib.check(rc, "tt_scan_topk")
        if _packed is not None:
            return out_s, out_i
        if return_flag:
            return out_s, out_i, int(flag.item())
        if check_overflow and int(flag.item()) != 0:
            return scan_topk(corpus, queries, k, idx_base, exact_dense=True)
    return out_s, out_i


def scan_topk_host(corpus: torch.Tensor, queries: torch.Tensor, k: int, idx_base: int = 0, shadow: Optional[ScanShadow] = None):
    """``scan_topk`` for callers that need the hits ON THE HOST (the retriever turns rows into nodes): scores, indices and the
    overflow status word live in one device buffer and come back in ONE copy -- the wait for the hits is the only host sync of a
    scan batch (``scan_topk`` itself reads the flag with its own ``.item()`` before the caller's ``.cpu()``: two syncs, 0.04 ms of
    a 0.74 ms shard batch).  A flagged overflow re-runs the dense exact path, as ``scan_topk`` does.
    -> (scores [Q,k] fp32, idx [Q,k] int32), CPU tensors."""
    q = queries.shape[0]
    if q == 0:
        return torch.empty((0, k), dtype=torch.float32), torch.empty((0, k), dtype=torch.int32)
    packed = torch.empty(2 * q * k + 1, dtype=torch.int32, device=corpus.device)
    scan_topk(corpus, queries, k, idx_base, _packed=packed, shadow=shadow)
    host = packed.cpu()                                   # the one sync (current stream only)
    if int(host[-1]) != 0:
        s, i = scan_topk(corpus, queries, k, idx_base, exact_dense=True)
        return s.cpu(), i.cpu()
    return host[: q * k].view(torch.float32).view(q, k), host[q * k: 2 * q * k].view(q, k)


def scan_topk_segmented(corpus: torch.Tensor, queries: torch.Tensor, k: int, seg_offsets) -> Tuple[torch.Tensor, torch.Tensor]:
    """Several index modules in one matrix: rows ``[seg_offsets[s], seg_offsets[s+1])`` are module ``s``.
    One pass over the matrix, then an exact top-k per (query, module) ->
    (scores [Q,S,k] fp32, rows [Q,S,k] int32, module-local, padding (-inf, -1)).
    Stands in for the reference's per-module thread fan-out (``rag_engine.py:420-424``)."""
    import ctypes
    lib = _lib.load_library()
    _require_cuda(corpus, "corpus")
    _require_cuda(queries, "queries")
    if corpus.dtype != torch.bfloat16 or queries.dtype != torch.bfloat16:
        raise TypeError("corpus and queries must be torch.bfloat16")
    if corpus.dim() != 2 or queries.dim() != 2 or corpus.shape[1] != queries.shape[1]:
        raise ValueError(f"shape mismatch: corpus {tuple(corpus.shape)} queries {tuple(queries.shape)}")
    if corpus.device != queries.device:
        raise ValueError("corpus and queries must be on the same device")
    if not corpus.is_contiguous() or not queries.is_contiguous():
        raise ValueError("corpus and queries must be contiguous row-major")
    offs = [int(o) for o in seg_offsets]
    n_seg = len(offs) - 1
    if n_seg < 1:
        raise ValueError("seg_offsets needs at least two entries")
    n, d = corpus.shape
    q = queries.shape[0]
    dev = corpus.device
    out_s = torch.empty((q, n_seg, k), dtype=torch.float32, device=dev)
    out_i = torch.empty((q, n_seg, k), dtype=torch.int32, device=dev)
    if q == 0:
        return out_s, out_i
    c_offs = (ctypes.c_int64 * (n_seg + 1))(*offs)
    with torch.cuda.device(dev):
        need = lib.tt_scan_segmented_workspace_bytes(max(offs[-1] - offs[0], 0), d, q, k)
        ws = _ws.get(dev, need + 256)
        base = (ws.data_ptr() + 255) // 256 * 256
        rc = lib.tt_scan_topk_segmented(corpus.data_ptr(), n, d, queries.data_ptr(), q, k, c_offs, n_seg,
                                        out_s.data_ptr(), out_i.data_ptr(), base,
                                        ws.numel() - (base - ws.data_ptr()), _stream_ptr(dev))
        _lib.check(rc, "tt_scan_topk_segmented")
    return out_s, out_i


def topk_merge(scores: torch.Tensor, idx: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Merge candidate lists [Q, M] (fp32 scores, int32 global indices, idx<0 =
    padding) into the top-k by (score desc, idx asc)."""
    lib = _lib.load_library()
    _require_cuda(scores, "scores")
    _require_cuda(idx, "idx")
    if scores.dtype != torch.float32 or idx.dtype != torch.int32:
        raise TypeError("scores must be float32 and idx int32")
    if scores.shape != idx.shape or scores.dim() != 2:
        raise ValueError("scores/idx must both be [Q, M]")
    scores = scores.contiguous()
    idx = idx.contiguous()
    q, m = scores.shape
    dev = scores.device
    out_s = torch.empty((q, k), dtype=torch.float32, device=dev)
    out_i = torch.empty((q, k), dtype=torch.int32, device=dev)
    if q == 0:
        return out_s, out_i
    with torch.cuda.device(dev):
        rc = lib.tt_topk_merge(scores.data_ptr(), idx.data_ptr(), q, m, k, out_s.data_ptr(), out_i.data_ptr(),
                               _stream_ptr(dev))
        _lib.check(rc, "tt_topk_merge")
    return out_s, out_i


def similarity_from_cosine(cos: torch.Tensor) -> torch.Tensor:
    """Reference score mapping for NodeWithScore.score: ChromaVectorStore returns
    exp(-squared_L2) and squared_L2 = 2 - 2 cos for unit vectors."""
    return torch.exp(-(2.0 - 2.0 * cos))
