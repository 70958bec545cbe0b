"""Seeded random sweep of the scan through the C ABI against the CPU oracle: shapes drawn around the planner's
thresholds (dense / sampled-threshold / tiled 65+-query paths: csrc/scan_api.hip make_plan), with NaN tombstones
(vector_index.delete) and duplicated rows mixed in.  Bar as in test_scan_gpu.py: indices bit-exact on tie-free queries,
scores within 1e-3 relative, never a tombstoned row."""
import numpy as np
import pytest
import torch

from oracle import scan as osc
from test_scan_gpu import _check

pytestmark = pytest.mark.gpu


def _draw(rng):
    regime = rng.integers(0, 4)
    d = int(rng.choice([128, 256, 384, 512, 640, 768, 896, 1024]))
    if regime == 0:      # dense path
        n, q, k = int(rng.integers(1, 6000)), int(rng.integers(1, 200)), int(rng.integers(1, 130))
    elif regime == 1:    # sampled threshold, streaming filter (<= 64 queries per tile, any number of tiles)
        n, q, k = int(rng.integers(65_537, 160_000)), int(rng.integers(1, 300)), int(rng.integers(1, 130))
    elif regime == 2:    # tiled MFMA filter: 65+ queries, >= 262144 rows, rows >= 2048 k
        k = int(rng.integers(1, 129))
        n, q = int(rng.integers(max(262_144, 2048 * k), 340_000)), int(rng.integers(65, 600))
    else:                # just below the tiled path's thresholds (stays on the streaming kernel)
        k = int(rng.integers(100, 160))
        n, q = int(rng.integers(200_000, 2048 * k if 2048 * k > 200_001 else 262_143)), int(rng.integers(60, 70))
        n = min(n, 320_000)
    return n, d, q, k


CASES = [_draw(np.random.default_rng(1000 + s)) for s in range(24)]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_random_shapes_with_tombstones_and_duplicates(dev, built_lib, case):
    from tensor_truth_amd import scan as tscan

    n, d, q, k = CASES[case]
    rng = np.random.default_rng(50_000 + case)
    corpus = osc.synth_corpus(n, d, seed=case + 11)
    queries, planted = osc.synth_queries(corpus, q, seed=case + 500)
    # duplicates: a handful of rows copied elsewhere (exact score ties -> the oracle's gap is 0 there, order by index)
    if n > 64:
        src = torch.from_numpy(rng.integers(0, n, size=8))
        dst = torch.from_numpy(rng.integers(0, n, size=8))
        corpus[dst] = corpus[src]
    # tombstones: ~1 % of the rows (at least one where there is more than one row), planted hits among them
    n_dead = min(n - 1, max(1, n // 100)) if n > 1 else 0
    dead = torch.from_numpy(rng.choice(n, size=n_dead, replace=False)) if n_dead else torch.zeros(0, dtype=torch.long)
    hit = planted[planted >= 0][:4]
    dead = torch.unique(torch.cat([dead, hit.long()]))[: max(n - 1, 0)]
    live = torch.ones(n, dtype=torch.bool)
    live[dead] = False
    # oracle on the live rows only, indices mapped back (stable order preserved: the map is monotone)
    live_idx = torch.nonzero(live).squeeze(1)
    want_s, want_i, gap = osc.scan_topk(corpus[live_idx], queries, k)
    want_i = torch.where(want_i >= 0, live_idx[want_i.clamp_min(0)], want_i)
    c = corpus.to(dev)
    if len(dead):
        c[dead.to(dev)] = float("nan")
    s, i = tscan.scan_topk(c, queries.to(dev), k)
    torch.cuda.synchronize()
    assert not torch.isin(i.cpu().long(), dead).any(), "a tombstoned row was returned"
    n_tf = _check(s, i, want_s, want_i, gap)
    # the sweep must have teeth: on a random corpus a good share of the queries is tie-free (fewer at large k: each of
    # the k adjacent gaps of a query has to clear 1e-6)
    assert q < 8 or n < 64 or n_tf >= q // 8, (n_tf, q)
