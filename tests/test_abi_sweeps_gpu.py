"""Seeded random sweeps of the smaller C-ABI entry points against the CPU oracle: the shard merge (`tt_topk_merge`,
SURVEY.md 8e), the one-pass multi-module scan (`tt_scan_topk_segmented`, row a8) and the semantic splitter's adjacent
distances (`tt_adjacent_cosine`, row f4).  Integer / index results bit-exact, fp32 values as in the fixed-shape tests."""
import numpy as np
import pytest
import torch

from oracle import scan as osc
from test_scan_gpu import _check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(12))
def test_topk_merge_random_lists(dev, built_lib, seed):
    """Random (queries, lists, k): padding entries (-inf / -1) anywhere, exact score ties across lists (tie-break by global
    index), fewer live candidates than k, duplicate scores inside a list."""
    from tensor_truth_amd import scan as tscan

    rng = np.random.default_rng(900 + seed)
    q, lists, k = int(rng.integers(1, 300)), int(rng.integers(1, 17)), int(rng.integers(1, 130))
    per = int(rng.integers(1, k + 1)) if seed % 3 == 0 else k       # lists shorter than k now and then
    m = lists * per
    g = torch.Generator().manual_seed(seed)
    # scores from a small set of values -> many exact ties; global indices unique per query
    vals = (torch.randint(0, 40, (q, m), generator=g).float() / 8.0) if seed % 2 else torch.randn(q, m, generator=g)
    # unique global indices per query, in random order, spread over the int32 range
    idx = torch.stack([torch.randperm(m, generator=g) * 1000 + torch.randint(0, 1000, (m,), generator=g) for _ in range(q)])
    idx = (idx * (2_000_000 // max(m, 1))).to(torch.int32)
    pad = torch.rand(q, m, generator=g) < 0.15
    vals[pad] = float("-inf")
    idx[pad] = -1
    want_v, want_i = osc.merge_topk(vals, idx.to(torch.int64), k)
    got_v, got_i = tscan.topk_merge(vals.to(dev), idx.to(dev), k)
    torch.cuda.synchronize()
    assert torch.equal(got_i.cpu().to(torch.int64), want_i), (q, lists, per, k)
    assert torch.equal(got_v.cpu(), want_v)


@pytest.mark.parametrize("seed", range(10))
def test_segmented_scan_random_modules(dev, built_lib, seed):
    """Random module tables: 1..16 modules of 0..90 000 rows (empty ones, ones shorter than k, ones longer than a selection
    piece), rows before the first and after the last module that belong to nobody."""
    from tensor_truth_amd import scan as tscan

    rng = np.random.default_rng(300 + seed)
    n_seg = int(rng.integers(1, 17))
    sizes = [int(rng.choice([0, int(rng.integers(1, 64)), int(rng.integers(64, 5000)), int(rng.integers(5000, 90_000))],
                            p=[0.15, 0.2, 0.4, 0.25])) for _ in range(n_seg)]
    lead = int(rng.integers(0, 50))
    offsets = [lead]
    for s in sizes:
        offsets.append(offsets[-1] + s)
    d = int(rng.choice([128, 384, 640, 1024]))
    k, nq = int(rng.integers(1, 101)), int(rng.integers(1, 9))
    n = offsets[-1] + int(rng.integers(0, 300))
    if n == 0:
        n = 8
    corpus = osc.synth_corpus(n, d, seed=seed + 40)
    queries, _ = osc.synth_queries(corpus, nq, seed=seed + 41)
    want_s, want_i, gap = osc.scan_topk_segmented(corpus, queries, k, offsets)
    s, i = tscan.scan_topk_segmented(corpus.to(dev), queries.to(dev), k, offsets)
    torch.cuda.synchronize()
    assert s.shape == (nq, n_seg, k) and i.shape == (nq, n_seg, k)
    _check(s.reshape(nq * n_seg, k), i.reshape(nq * n_seg, k), want_s.reshape(nq * n_seg, k),
           want_i.reshape(nq * n_seg, k), gap.reshape(-1))
    for m, size in enumerate(sizes):
        if size == 0:
            assert (i[:, m] == -1).all() and torch.isinf(s[:, m]).all()
        else:
            assert int(i[:, m].max()) < size, "module-local index out of its module"


@pytest.mark.parametrize("n,h", [(2, 128), (3, 384), (17, 1024), (1000, 1024), (4097, 384), (33, 768), (20_000, 1024)])
def test_adjacent_cosine_random(dev, built_lib, n, h):
    from tensor_truth_amd.semantic import adjacent_distances

    g = torch.Generator().manual_seed(n + h)
    e = torch.randn(n, h, generator=g)
    e[n // 2] = e[n // 2 - 1] * 3.0 if n > 2 else e[n // 2]         # a parallel pair: distance 0 whatever the norms
    if n > 5:
        e[3] = -e[4]                                                  # an antiparallel pair: distance 2
    a, b = e[:-1].double(), e[1:].double()
    want = (1.0 - (a * b).sum(1) / (a.norm(dim=1) * b.norm(dim=1))).float()
    got = adjacent_distances(e.to(dev)).cpu()
    assert got.shape == want.shape
    assert (got - want).abs().max().item() < 2e-6


@pytest.mark.parametrize("m", [1, 2, 63, 64, 65, 127, 128, 129, 1000, 1024, 1025, 4096, 5000, 16385, 70001])
@pytest.mark.parametrize("k", [1, 10, 50, 64])
def test_wave_register_select_list_lengths(dev, built_lib, m, k):
    """The k <= 64 selection kernel (select.hip select_wave_kernel: per-wave bitonic sort in registers, DPP / v_permlane swaps,
    chunk skipping, one LDS merge) on list lengths around every chunk (64) and wave-count (128, 1024) boundary, with NaN scores,
    -0.0 / +0.0, exact ties (index ascending), padding entries and ascending input (no chunk can be skipped): indices and scores
    bit-exact against the oracle's stable merge."""
    from tensor_truth_amd import scan as tscan

    g = torch.Generator().manual_seed(31 * m + k)
    q = 7
    vals = torch.randn(q, m, generator=g)
    vals[0] = torch.sort(vals[0]).values                               # ascending: every chunk beats the running list
    vals[1] = torch.sort(vals[1], descending=True).values             # descending: only the first chunk(s) matter
    if m >= 4:
        vals[2, :: 3] = 0.25                                           # exact ties
        vals[3, 0], vals[3, m // 2] = 0.0, -0.0                        # the two zeros compare equal: index decides
        vals[3, 1::2] = -1.0
        vals[4, ::5] = float("nan")                                    # never selected
    idx = torch.stack([torch.randperm(m, generator=g) for _ in range(q)]).to(torch.int32) * 7 + 3
    pad = torch.rand(q, m, generator=g) < 0.1
    pad[5] = True                                                      # a query with no live candidate at all
    vals[pad] = float("-inf")
    idx[pad] = -1
    live = vals.clone()
    live[torch.isnan(live)] = float("-inf")                            # the oracle's view of "never selected"
    li = idx.clone()
    li[torch.isnan(vals)] = -1
    want_v, want_i = osc.merge_topk(live, li.to(torch.int64), k)
    got_v, got_i = tscan.topk_merge(vals.to(dev), idx.to(dev), k)
    torch.cuda.synchronize()
    assert torch.equal(got_i.cpu().to(torch.int64), want_i), (m, k)
    gv, wv = got_v.cpu(), want_v
    assert torch.equal(gv == 0, wv == 0) and torch.equal(torch.where(gv == 0, torch.zeros_like(gv), gv), torch.where(wv == 0, torch.zeros_like(wv), wv))


def test_wave_select_equals_lds_network_in_a_child_process(dev, built_lib, tmp_path, diag_lib_env):
    """A/B: TT_SELECT_WAVE=0 (the LDS network of rounds 1-3, still the kernel for k > 64) returns the same bits as the
    wave-register kernel on a sampled-threshold scan (sample select with the fused threshold outputs + final select over private
    and shared candidate lists) and on a tiled 256-query batch."""
    import os
    import subprocess
    import sys

    from tensor_truth_amd import scan as tscan

    def run():
        out = []
        for n, nq, k in ((200_000, 16, 50), (300_007, 256, 50), (300_007, 100, 7)):
            corpus = osc.synth_corpus(n, 1024, seed=5)
            queries, _ = osc.synth_queries(corpus, nq, seed=6)
            s, i = tscan.scan_topk(corpus.to(dev), queries.to(dev), k)
            out += [s.cpu(), i.cpu()]
        return out

    mine = run()
    if os.environ.get("TT_SELECT_AB_CHILD"):
        torch.save(mine, os.environ["TT_SELECT_AB_CHILD"])
        return
    other = tmp_path / "lds.pt"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(diag_lib_env, TT_SELECT_WAVE="0", TT_SELECT_AB_CHILD=str(other))
    subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__) + "::test_wave_select_equals_lds_network_in_a_child_process"],
                   check=True, env=env, cwd=root, timeout=900, capture_output=True)
    theirs = torch.load(str(other))
    assert len(mine) == len(theirs) and all(torch.equal(a, b) for a, b in zip(mine, theirs))


def test_contraction_sample_equals_streaming_sample_in_a_child_process(dev, built_lib, tmp_path, diag_lib_env):
    """A/B: the tiled scan's threshold sample on the contraction kernel (round 4) against the streaming sample kernel
    (TT_SCAN_GEMM_SAMPLE=0).  The two samples look at different rows, so the thresholds differ -- the RESULT may not: top-k scores
    and indices bit-identical, on a full 256-query batch, a ragged one (zero-padded copy), one with NaN tombstones in the sampled
    tiles, and a shard whose rows are not a multiple of 256."""
    import os
    import subprocess
    import sys

    from tensor_truth_amd import scan as tscan

    def run():
        out = []
        for n, nq, k, holes in ((300_032, 256, 50, False), (300_007, 100, 7, False), (524_288, 256, 50, True), (1_250_001, 130, 50, False)):
            corpus = osc.synth_corpus(n, 1024, seed=15)
            queries, _ = osc.synth_queries(corpus, nq, seed=16)
            if holes:      # tombstone whole sampled tiles and scattered rows (after the queries were planted on live rows)
                corpus[:512] = float("nan")
                corpus[torch.arange(1000, n, 997)] = float("nan")
            s, i = tscan.scan_topk(corpus.to(dev), queries.to(dev), k)
            out += [s.cpu(), i.cpu()]
        return out

    mine = run()
    if os.environ.get("TT_SAMPLE_AB_CHILD"):
        torch.save(mine, os.environ["TT_SAMPLE_AB_CHILD"])
        return
    other = tmp_path / "streaming_sample.pt"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(diag_lib_env, TT_SCAN_GEMM_SAMPLE="0", TT_SAMPLE_AB_CHILD=str(other))
    subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu",
                    os.path.abspath(__file__) + "::test_contraction_sample_equals_streaming_sample_in_a_child_process"],
                   check=True, env=env, cwd=root, timeout=900, capture_output=True)
    theirs = torch.load(str(other))
    assert len(mine) == len(theirs) and all(torch.equal(a, b) for a, b in zip(mine, theirs))
    for s, i in zip(mine[0::2], mine[1::2]):
        assert torch.isfinite(s).all() and (i >= 0).all()
