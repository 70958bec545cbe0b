"""Order / top-n agreement checks that cannot silently not run ("rerank top-10 set/ordering
identical where oracle score gaps > tolerance").  ``bound`` is the stated score tolerance of the mode under test:
two candidates whose ORACLE scores differ by more than ``bound`` are separable and must come out in the oracle's
order; closer pairs may legitimately swap.  Every function returns how many separable pairs / items it checked so
the caller can assert that the subset was not empty and report it."""
import numpy as np


def assert_order_on_separable(want, got, bound, what=""):
    """All pairs (i, j) with want[i] - want[j] > bound must satisfy got[i] > got[j].  -> number of such pairs."""
    w = np.asarray(want, dtype=np.float64).reshape(-1)
    g = np.asarray(got, dtype=np.float64).reshape(-1)
    assert w.shape == g.shape
    sep = (w[:, None] - w[None, :]) > bound
    bad = sep & ~(g[:, None] > g[None, :])
    if bad.any():
        i, j = np.argwhere(bad)[0]
        raise AssertionError(f"{what}: candidates {i} / {j} are separable in the oracle ({w[i]:.5f} vs {w[j]:.5f}, "
                             f"gap > {bound}) but come out as {g[i]:.5f} vs {g[j]:.5f} ({int(bad.sum())} such pairs)")
    return int(sep.sum())


def assert_topn_on_separable(want, got, n, bound, what=""):
    """Top-n membership wherever the oracle decides it by more than ``bound``: every candidate scoring more than
    ``bound`` above the oracle's (n+1)-th best must be in got's top-n, every candidate more than ``bound`` below the
    oracle's n-th best must not be.  When the n-th / (n+1)-th gap itself exceeds ``bound`` this IS set equality.
    -> (must-in count, must-out count, exact-set-required flag)."""
    w = np.asarray(want, dtype=np.float64).reshape(-1)
    g = np.asarray(got, dtype=np.float64).reshape(-1)
    n = min(n, len(w))
    order_w = np.argsort(-w, kind="stable")
    top_g = set(np.argsort(-g, kind="stable")[:n].tolist())
    nth = w[order_w[n - 1]]
    nxt = w[order_w[n]] if n < len(w) else -np.inf
    must_in = [int(i) for i in range(len(w)) if w[i] > nxt + bound]
    must_out = [int(i) for i in range(len(w)) if w[i] < nth - bound]
    missing = [i for i in must_in if i not in top_g]
    intruders = [i for i in must_out if i in top_g]
    assert not missing and not intruders, (f"{what}: top-{n} differs where the oracle is decisive (bound {bound}): "
                                           f"missing {missing}, intruders {intruders}")
    exact = bool(nth - nxt > bound)
    if exact:
        assert top_g == set(order_w[:n].tolist()), f"{what}: top-{n} set differs although the cut gap exceeds {bound}"
    return len(must_in), len(must_out), exact


def kendall_tau(a, b):
    """Kendall rank correlation (tau-a) of two score vectors."""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    sa = np.sign(a[:, None] - a[None, :])
    sb = np.sign(b[:, None] - b[None, :])
    n = len(a)
    return float((sa * sb).sum() / (n * (n - 1)))


def topn_overlap(want, got, n):
    w = set(np.argsort(-np.asarray(want), kind="stable")[:n].tolist())
    g = set(np.argsort(-np.asarray(got), kind="stable")[:n].tolist())
    return len(w & g) / float(n)


def weights_checksum(W) -> str:
    """sha256 over the names and the first 4 KiB of every tensor of a weight dict: ties a golden fixture to its weights."""
    import hashlib

    h = hashlib.sha256()
    for k in sorted(W):
        h.update(k.encode())
        h.update(W[k].contiguous().numpy().tobytes()[:4096])
    return h.hexdigest()
