"""CPU oracle: exact similarity scan + stable top-k, shard merge.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Restates the vector search behind ``VectorIndexRetriever`` (built at
``src/tensortruth/rag_engine.py:639,674``; collection created at
``rag_engine.py:628-630`` / ``indexing/builder.py:424-426``).  The reference
runs chromadb's HNSW (approximate, squared-L2, fp32) and maps
``similarity = exp(-distance)``.  For unit vectors ``distance = 2 - 2 cos`` so
the rank order equals the cosine/dot order; BASELINE.json asks for the *exact*
brute-force scan over a bf16 row-major corpus, which is what is restated here:

    S = Q . C^T   (inputs rounded to bf16, fp32 products, fp32 accumulation)
    top-K per query ordered by (score desc, row index asc)

``chroma_similarity`` gives the reference's score mapping for callers that want
LlamaIndex-compatible ``NodeWithScore.score`` values.
"""
from __future__ import annotations

import numpy as np
import torch


def to_bf16_f32(x: torch.Tensor) -> torch.Tensor:
    """Round to bf16 (RNE) and widen back to fp32."""
    return x.to(torch.bfloat16).to(torch.float32)


def scan_scores(corpus_bf16: torch.Tensor, queries_bf16: torch.Tensor) -> torch.Tensor:
    """[N,D] bf16, [Q,D] bf16 -> [Q,N] fp32 (fp32 accumulate)."""
    return queries_bf16.to(torch.float32) @ corpus_bf16.to(torch.float32).T


def stable_topk(scores: torch.Tensor, k: int):
    """Top-k per row ordered by (score desc, index asc).  scores [Q,N] fp32.

    Returns (vals [Q,k] fp32, idx [Q,k] int64, gap [Q] fp32) where ``gap`` is
    the smallest difference between adjacent scores among the first k+1 sorted
    scores -- the tie-free margin used by the parity tests.
    """
    Q, N = scores.shape
    k_eff = min(k, N)
    # stable sort on the negated scores keeps index-ascending order inside ties
    order = torch.sort(-scores, dim=1, stable=True).indices
    idx = order[:, :k_eff]
    vals = torch.gather(scores, 1, idx)
    kk = min(k_eff + 1, N)
    top = torch.gather(scores, 1, order[:, :kk])
    if kk >= 2:
        gap = (top[:, :-1] - top[:, 1:]).min(dim=1).values
    else:
        gap = torch.full((Q,), float("inf"))
    if k_eff < k:  # pad like the HIP path: -inf score, index -1
        pad_v = torch.full((Q, k - k_eff), float("-inf"))
        pad_i = torch.full((Q, k - k_eff), -1, dtype=torch.int64)
        vals = torch.cat([vals, pad_v], 1)
        idx = torch.cat([idx, pad_i], 1)
    return vals, idx, gap


def scan_topk(corpus_bf16, queries_bf16, k: int, chunk: int = 262144):
    """Exact scan in row chunks (bounded memory).  Returns (vals, idx, gap)."""
    N = corpus_bf16.shape[0]
    Q = queries_bf16.shape[0]
    if N == 0:
        return (torch.full((Q, k), float("-inf")), torch.full((Q, k), -1, dtype=torch.int64),
                torch.full((Q,), float("inf")))
    if N <= chunk:
        return stable_topk(scan_scores(corpus_bf16, queries_bf16), k)
    cand_v, cand_i = [], []
    for lo in range(0, N, chunk):
        s = scan_scores(corpus_bf16[lo:lo + chunk], queries_bf16)
        v, i, _ = stable_topk(s, min(k + 1, s.shape[1]))
        cand_v.append(v)
        cand_i.append(i + lo)
    v = torch.cat(cand_v, 1)
    i = torch.cat(cand_i, 1)
    return merge_topk(v, i, k, want_gap=True)


def scan_topk_segmented(corpus_bf16, queries_bf16, k: int, seg_offsets):
    """One exact search per index module, as the reference runs them (one retriever per module on a
    thread pool, ``rag_engine.py:420-424``): module ``s`` = rows ``[seg_offsets[s], seg_offsets[s+1])``.
    Returns (vals [Q,S,k], module-local idx [Q,S,k], gap [Q,S])."""
    vs, ixs, gaps = [], [], []
    for s in range(len(seg_offsets) - 1):
        v, i, g = scan_topk(corpus_bf16[seg_offsets[s]:seg_offsets[s + 1]], queries_bf16, k)
        vs.append(v)
        ixs.append(i)
        gaps.append(g)
    return torch.stack(vs, 1), torch.stack(ixs, 1), torch.stack(gaps, 1)


def merge_topk(vals: torch.Tensor, idx: torch.Tensor, k: int, want_gap: bool = False):
    """Merge candidate lists [Q,M] by (score desc, global index asc) -> top-k.

    This is also the oracle for the multi-GPU shard merge: each
    rank contributes K candidates with *global* row ids; padding entries carry
    idx -1 / score -inf and sort last.
    """
    Q, M = vals.shape
    v = vals.numpy().astype(np.float64)
    i = idx.numpy().astype(np.int64)
    i_key = np.where(i < 0, np.iinfo(np.int64).max, i)
    order = np.lexsort((i_key, -v), axis=1)  # primary -v, secondary index
    kk = min(k + 1, M)
    top = order[:, :kk]
    tv = np.take_along_axis(vals.numpy(), top, axis=1)
    ti = np.take_along_axis(i, top, axis=1)
    out_v = torch.from_numpy(tv[:, :k].copy())
    out_i = torch.from_numpy(ti[:, :k].copy())
    if out_v.shape[1] < k:
        pad = k - out_v.shape[1]
        out_v = torch.cat([out_v, torch.full((Q, pad), float("-inf"))], 1)
        out_i = torch.cat([out_i, torch.full((Q, pad), -1, dtype=torch.int64)], 1)
    if not want_gap:
        return out_v, out_i
    if kk >= 2:
        with np.errstate(invalid="ignore"):
            d = tv[:, :-1] - tv[:, 1:]
        d = np.where(np.isnan(d), np.inf, d)
        gap = torch.from_numpy(d.min(axis=1).astype(np.float32))
    else:
        gap = torch.full((Q,), float("inf"))
    return out_v, out_i, gap


def chroma_similarity(cos: torch.Tensor) -> torch.Tensor:
    """Reference score mapping: ChromaVectorStore returns exp(-squared_L2) and
    for unit vectors squared_L2 = 2 - 2 cos  [UPSTREAM-K A8/A9]."""
    return torch.exp(-(2.0 - 2.0 * cos))


def synth_corpus(n: int, d: int, seed: int = 1234) -> torch.Tensor:
    """randn -> L2 normalise (fp32) -> bf16, row-major."""
    g = torch.Generator().manual_seed(seed)
    c = torch.randn(n, d, generator=g)
    c = c / c.norm(dim=1, keepdim=True)
    return c.to(torch.bfloat16).contiguous()


def synth_queries(corpus_bf16: torch.Tensor, q: int, seed: int = 4321):
    """Half planted neighbours normalise(c_j + 0.5 u) (u unit-norm random, cos ~0.89
    to row j), half pure random.  Returns (queries bf16 [q,D], planted row or -1)."""
    n, d = corpus_bf16.shape
    g = torch.Generator().manual_seed(seed)
    out = torch.empty(q, d)
    planted = torch.full((q,), -1, dtype=torch.int64)
    for i in range(q):
        u = torch.randn(d, generator=g)
        u = u / u.norm()
        if i % 2 == 0 and n > 0:
            j = int(torch.randint(0, n, (1,), generator=g))
            v = corpus_bf16[j].to(torch.float32) + 0.5 * u
            planted[i] = j
        else:
            v = u
        out[i] = v / v.norm()
    return out.to(torch.bfloat16).contiguous(), planted
