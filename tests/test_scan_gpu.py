"""GPU parity: HIP scan/top-k (through the C ABI) against the CPU oracle.

Bar: top-k indices bit-exact wherever the oracle's adjacent-score
gap exceeds 1e-6 (tie-free), fp32 scores within 1e-3 relative.
"""
import os

import numpy as np
import pytest
import torch

from oracle import scan as osc

pytestmark = pytest.mark.gpu

REL_TOL = 1e-3   # north_star: fp scores within 1e-3 relative
TIE_GAP = 1e-6   # oracle adjacent-score gap above which the order is unambiguous


def _check(got_s, got_i, want_s, want_i, gap, k_valid=None):
    got_s, got_i = got_s.cpu(), got_i.cpu().to(torch.int64)
    tie_free = gap > TIE_GAP
    assert torch.equal(got_i[tie_free], want_i[tie_free]), "indices differ on tie-free queries"
    # queries with near-ties: same set up to the ambiguous positions, scores still match
    finite = torch.isfinite(want_s)
    assert torch.equal(torch.isfinite(got_s), finite)
    rel = ((got_s - want_s)[finite].abs() / want_s[finite].abs().clamp_min(1e-6))
    assert rel.numel() == 0 or rel.max().item() < REL_TOL
    assert (got_i[~finite] == -1).all()
    return int(tie_free.sum())


def _run(tscan, dev, corpus, queries, k, **kw):
    s, i = tscan.scan_topk(corpus.to(dev), queries.to(dev), k, **kw)
    torch.cuda.synchronize()
    return s, i


def test_golden_scan_fixture(dev, built_lib, golden_dir):
    from tensor_truth_amd import scan as tscan

    z = np.load(os.path.join(golden_dir, "scan_4096x1024_k50.npz"))
    corpus = osc.synth_corpus(4096, 1024, seed=1234)
    queries, planted = osc.synth_queries(corpus, 16, seed=4321)
    s, i = _run(tscan, dev, corpus, queries, 50)
    n_tf = _check(s, i, torch.from_numpy(z["scores"]), torch.from_numpy(z["idx"]).to(torch.int64),
                  torch.from_numpy(z["gap"]))
    assert n_tf >= 12
    for q in range(16):
        if planted[q] >= 0:
            assert int(i[q, 0]) == int(planted[q])


def test_config_c1_toy_corpus(dev, built_lib):
    """BASELINE config 1: 1k x 384 (bge-small shape), top-10."""
    from tensor_truth_amd import scan as tscan

    corpus = osc.synth_corpus(1000, 384, seed=1234)
    queries, _ = osc.synth_queries(corpus, 16, seed=4321)
    want = osc.scan_topk(corpus, queries, 10)
    s, i = _run(tscan, dev, corpus, queries, 10)
    _check(s, i, *want)


@pytest.mark.parametrize("n,d,q,k", [
    (0, 128, 3, 5), (1, 128, 1, 4), (31, 256, 2, 8), (33, 512, 5, 64), (257, 768, 63, 50),
    (1000, 1024, 65, 10), (4097, 1024, 130, 50), (300, 128, 7, 1000), (65536, 384, 4, 50),
    (2000, 640, 9, 20), (3000, 896, 70, 33),      # every multiple of 128 the header promises (include/tt_hip.h:52)
])
def test_ragged_shapes_dense_path(dev, built_lib, n, d, q, k):
    from tensor_truth_amd import scan as tscan

    corpus = osc.synth_corpus(n, d, seed=n + d)
    queries, _ = osc.synth_queries(corpus, q, seed=q + k)
    want = osc.scan_topk(corpus, queries, k)
    s, i = _run(tscan, dev, corpus, queries, k)
    _check(s, i, *want)


@pytest.mark.parametrize("n,d,q,k", [
    (65537, 1024, 3, 50), (100_003, 1024, 64, 50), (131_072 + 17, 512, 70, 10), (200_000, 384, 33, 100),
    (150_001, 1024, 256, 50),     # the gathered query batch of an 8-GPU bench step: four query tiles
    (60_000, 1024, 1024, 50),     # SURVEY.md 8d's largest scan-only batch: sixteen query tiles (above the MFMA ridge)
    # 65+ queries over >= 262144 rows: the filter pass is the 256-query-wide tiled MFMA contraction (ONE pass over the
    # shard per 256 queries, csrc/gemm.hip TT_EPI_SCAN) + the streaming kernel for the < 256 tail rows
    (300_007, 1024, 256, 50),     # 8-GPU bench step shape; 231 tail rows
    (262_144, 512, 65, 10),       # smallest shard / batch on that path, no tail
    (280_000, 128, 300, 20),      # two 256-query blocks (the second one mostly padding), narrow rows
    (270_001, 384, 100, 128),     # bge-small width, k above a sort group
    (90_000, 640, 20, 50), (263_000, 896, 80, 10),   # the widths between the common model sizes, both filter kernels
    # tiny k on the tiled path: the threshold IS a sampled row's score (k = 1: the best sampled group), computed by the
    # streaming kernel while the filter recomputes it on another MFMA shape -- the defining row must still pass
    (262_144, 1024, 65, 1), (262_144, 1024, 80, 2), (262_144, 1024, 256, 3), (262_144, 1024, 130, 4),
])
def test_sampled_threshold_path(dev, built_lib, n, d, q, k):
    """Shards above 65536 rows: sample -> threshold -> filtered main pass -> select."""
    from tensor_truth_amd import scan as tscan

    corpus = osc.synth_corpus(n, d, seed=n % 1000)
    queries, planted = osc.synth_queries(corpus, q, seed=q)
    want = osc.scan_topk(corpus, queries, k)
    s, i = _run(tscan, dev, corpus, queries, k, idx_base=7_000_000)
    i = i - 7_000_000 * (i >= 0).to(i.dtype)
    _check(s, i, *want)
    # forced dense path gives the same answer
    s2, i2 = _run(tscan, dev, corpus, queries, k, exact_dense=True)
    _check(s2, i2, *want)


def test_tiled_path_returns_a_sampled_best_row_at_k1(dev, built_lib):
    """k = 1 on the 65+-query path with the global best row of every query INSIDE the threshold sample (262144 rows: every
    second 32-row group is sampled): thr is that row's own score from the sample kernel, and the tiled filter's
    recomputation of it must not fall below thr -- no query may come back empty, and no status flag."""
    from tensor_truth_amd import scan as tscan

    n, d, q = 262_144, 1024, 96
    corpus = osc.synth_corpus(n, d, seed=5)
    g = torch.Generator().manual_seed(6)
    rows = (torch.randperm(n // 64, generator=g)[:q] * 64 + torch.randint(0, 32, (q,), generator=g))   # even groups: sampled
    noise = torch.nn.functional.normalize(torch.randn(q, d, generator=g), dim=1)
    queries = torch.nn.functional.normalize(corpus[rows].float() + 0.3 * noise, dim=1).to(torch.bfloat16)
    want = osc.scan_topk(corpus, queries, 1)
    assert torch.equal(want[1][:, 0], rows)
    s, i, flag = tscan.scan_topk(corpus.to(dev), queries.to(dev), 1, return_flag=True)
    torch.cuda.synchronize()
    assert not flag and (i.cpu() >= 0).all()
    _check(s, i, *want)


def test_tiled_filter_pass_skips_tombstones_and_reports_overflow(dev, built_lib):
    """The 65+-query path with deleted rows (NaN tombstones, vector_index.delete) and with a clustered corpus that
    overflows the shared candidate list: NaN rows never rank, and an overflow raises the status flag (the wrapper then
    re-runs the dense exact path) instead of returning a truncated answer."""
    from tensor_truth_amd import scan as tscan

    corpus = osc.synth_corpus(262_144 + 300, 256, seed=77)
    queries, planted = osc.synth_queries(corpus, 96, seed=78)
    dead = planted[planted >= 0][:20]
    want_corpus = corpus.clone()
    want_corpus[dead] = 0                                   # a zero row scores 0: never in a top-10 of planted / random hits
    want = osc.scan_topk(want_corpus, queries, 10)
    c = corpus.to(dev)
    c[dead.to(dev)] = float("nan")
    s, i, overflowed = tscan.scan_topk(c, queries.to(dev), 10, return_flag=True)
    torch.cuda.synchronize()
    assert not overflowed and not torch.isin(i.cpu().long(), dead).any() and torch.isfinite(s).all()
    _check(s, i, *want)
    # every row identical to the query: nothing separates the sample from the rest -> every row passes the filter
    one = torch.nn.functional.normalize(torch.randn(1, 256, generator=torch.Generator().manual_seed(1)), dim=1).to(torch.bfloat16)
    flat = one.repeat(300_000, 1).to(dev)
    qs = one.repeat(70, 1).to(dev)
    s2, i2, overflowed = tscan.scan_topk(flat, qs, 10, return_flag=True)
    assert overflowed
    s3, i3 = tscan.scan_topk(flat, qs, 10)                  # default wrapper: dense fallback, exact (ties -> lowest rows)
    assert torch.equal(i3.cpu(), torch.arange(10, dtype=torch.int32).repeat(70, 1))


def test_both_corpus_load_modes(dev, built_lib, tmp_path, diag_lib_env):
    """The streaming kernel's two corpus load paths: LDS-transposed full lines (the product) against the oracle, and direct
    fragment-shaped loads (TT_SCAN_MODE=0, round 1's first form: a switch of the diagnostic library, run in a child process)
    against the product, bit for bit."""
    import subprocess
    import sys

    from tensor_truth_amd import scan as tscan

    corpus = osc.synth_corpus(90_000, 1024, seed=5)
    queries, _ = osc.synth_queries(corpus, 20, seed=6)
    s, i = _run(tscan, dev, corpus, queries, 50)
    if os.environ.get("TT_SCAN_MODE_AB_CHILD"):
        torch.save((s.cpu(), i.cpu()), os.environ["TT_SCAN_MODE_AB_CHILD"])
        return
    _check(s, i, *osc.scan_topk(corpus, queries, 50))
    other = tmp_path / "mode0.pt"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(diag_lib_env, TT_SCAN_MODE="0", TT_SCAN_MODE_AB_CHILD=str(other))
    subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__) + "::test_both_corpus_load_modes"],
                   check=True, env=env, cwd=root, timeout=900, capture_output=True)
    s0, i0 = torch.load(str(other))
    assert torch.equal(i0, i.cpu()) and torch.allclose(s0, s.cpu(), rtol=1e-6, atol=1e-7)


def test_topic_ordered_corpus_keeps_the_filter_path(dev, built_lib):
    """Corpora are ingested document by document, so neighbouring rows share a topic.  The threshold sample is spread
    evenly over the shard (every (N / n0)-th 32-row group), not its first rows: a query about a topic that only occurs
    late in the matrix still gets a tight threshold -- exact results AND no candidate-list overflow (a sample of the
    first rows would see none of that topic, let the whole cluster through the filter and fall back to the dense path)."""
    from tensor_truth_amd import scan as tscan

    g = torch.Generator().manual_seed(31)
    n_topics, per_topic, d = 60, 5000, 256
    centers = torch.nn.functional.normalize(torch.randn(n_topics, d, generator=g), dim=1)
    rows = centers.repeat_interleave(per_topic, 0) + 0.35 * torch.randn(n_topics * per_topic, d, generator=g) / d ** 0.5 * 4
    corpus = torch.nn.functional.normalize(rows, dim=1).to(torch.bfloat16)          # 300k rows, sorted by topic
    topics = torch.tensor([59, 58, 40, 31, 7, 0, 22, 50])
    queries = torch.nn.functional.normalize(centers[topics] + 0.05 * torch.randn(8, d, generator=g), dim=1).to(torch.bfloat16)
    want_s, want_i, gap = osc.scan_topk(corpus, queries, 50)
    assert all(int(want_i[q, 0]) // per_topic == int(topics[q]) for q in range(8))    # hits lie in the query's topic
    s, i, overflowed = tscan.scan_topk(corpus.to(dev), queries.to(dev), 50, return_flag=True)
    torch.cuda.synchronize()
    assert not overflowed
    _check(s, i, want_s, want_i, gap)


def test_duplicate_rows_tie_break_by_index(dev, built_lib):
    from tensor_truth_amd import scan as tscan

    base = osc.synth_corpus(40_000, 1024, seed=9)
    corpus = torch.cat([base, base], 0).contiguous()
    queries, _ = osc.synth_queries(base, 8, seed=10)
    s, i = _run(tscan, dev, corpus, queries, 50)
    s, i = s.cpu(), i.cpu()
    # every hit appears as the pair (r, r + 40000), equal scores, lower index first
    assert torch.equal(i[:, 0::2] + 40_000, i[:, 1::2])
    assert torch.equal(s[:, 0::2], s[:, 1::2])
    want_s, want_i, _ = osc.scan_topk(base, queries, 25)
    assert torch.equal(i[:, 0::2].to(torch.int64), want_i) or (want_s[:, :-1] - want_s[:, 1:]).min() <= TIE_GAP


def test_candidate_overflow_falls_back_exact(dev, built_lib):
    """Adversarial layout: exactly the sampled row groups (every (N/32 // 1024)-th 32-row group, scan_api.hip) hold
    unrelated rows, every other row beats the sample's k-th best -> the candidate lists overflow, the raw call says so."""
    from tensor_truth_amd import scan as tscan

    d = 256
    g = torch.Generator().manual_seed(3)
    qv = torch.randn(d, generator=g)
    qv = qv / qv.norm()
    n = 130_000
    close = qv + 0.3 * torch.randn(n, d, generator=g) / (d ** 0.5)
    close = close / close.norm(dim=1, keepdim=True)
    rnd = torch.randn(n, d, generator=g)
    rnd = rnd / rnd.norm(dim=1, keepdim=True)
    stride = (n // 32) // 1024                       # 32768 sample rows = 1024 groups
    grp = torch.arange(n) // 32
    sampled = (grp % stride == 0) & (grp < 1024 * stride)
    rows = torch.where(sampled[:, None], rnd, close)
    corpus = rows.to(torch.bfloat16).contiguous()
    queries = qv.view(1, d).to(torch.bfloat16).contiguous()
    want = osc.scan_topk(corpus, queries, 50)
    c_dev, q_dev = corpus.to(dev), queries.to(dev)
    # the raw call reports the overflow ...
    from tensor_truth_amd import _lib
    lib = _lib.load_library()
    need = lib.tt_scan_workspace_bytes(corpus.shape[0], d, 1, 50)
    ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
    base = (ws.data_ptr() + 255) // 256 * 256
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    out_s = torch.empty(1, 50, dtype=torch.float32, device=dev)
    out_i = torch.empty(1, 50, dtype=torch.int32, device=dev)
    rc = lib.tt_scan_topk(c_dev.data_ptr(), corpus.shape[0], d, q_dev.data_ptr(), 1, 50, 0, out_s.data_ptr(),
                          out_i.data_ptr(), base, need, flag.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert int(flag.item()) == 1
    # ... and the Python wrapper falls back to the dense path and stays exact
    s, i = tscan.scan_topk(c_dev, q_dev, 50)
    _check(s, i, *want)
    # ... and so does the retriever's one-copy form, which reads the status word from the same copy as the hits
    hs, hi = tscan.scan_topk_host(c_dev, q_dev, 50)
    assert torch.equal(hs, s.cpu()) and torch.equal(hi, i.cpu())


def test_scan_topk_host_one_copy_equals_scan_topk(dev, built_lib):
    """The retriever's form (scan.scan_topk_host): scores, indices and the status word come back in ONE copy -- same bits as
    scan_topk on the streaming and on the tiled path (the overflow fallback: test_candidate_overflow_falls_back_exact)."""
    from tensor_truth_amd import scan as tscan

    corpus = osc.synth_corpus(300_007, 1024, seed=21)
    for nq in (1, 20, 130):
        queries, _ = osc.synth_queries(corpus, nq, seed=22)
        c, q = corpus.to(dev), queries.to(dev)
        s, i = tscan.scan_topk(c, q, 50)
        hs, hi = tscan.scan_topk_host(c, q, 50, idx_base=7)
        assert not hs.is_cuda and torch.equal(hs, s.cpu()) and torch.equal(hi, i.cpu() + 7)

def test_full_size_config_c2_against_oracle(dev, built_lib):
    """BASELINE config 2 scan half: 1M x 1024 bf16, top-50, Q=64, vs the CPU oracle."""
    from tensor_truth_amd import scan as tscan

    torch.set_num_threads(os.cpu_count() or 8)
    corpus = osc.synth_corpus(1_000_000, 1024, seed=1234)
    queries, planted = osc.synth_queries(corpus, 64, seed=4321)
    want_s, want_i, gap = osc.scan_topk(corpus, queries, 50)
    s, i = _run(tscan, dev, corpus, queries, 50)
    n_tf = _check(s, i, want_s, want_i, gap)
    assert n_tf >= 32, f"only {n_tf}/64 tie-free queries"
    for q in range(64):
        if planted[q] >= 0:
            assert int(i[q, 0]) == int(planted[q])
    # size-independent properties on the device result
    s_c = s.cpu()
    assert (s_c[:, :-1] >= s_c[:, 1:]).all(), "scores not sorted"
    c_dev = corpus.to(dev)
    re = (queries.to(dev).float().unsqueeze(1) * c_dev[i.long().clamp_min(0)].float()).sum(-1)
    assert torch.allclose(re.cpu(), s_c, rtol=REL_TOL, atol=1e-5), "returned scores are not the rows' dot products"


def test_topk_merge_matches_oracle(dev, built_lib):
    from tensor_truth_amd import scan as tscan

    g = torch.Generator().manual_seed(0)
    q, lists, k = 37, 8, 50
    vals = torch.randn(q, lists * k, generator=g)
    idx = torch.stack([torch.randperm(10_000_000, generator=g)[: lists * k] for _ in range(q)]).to(torch.int32)
    # padding entries and exact ties
    vals[:, 5] = float("-inf")
    idx[:, 5] = -1
    vals[:, 10] = vals[:, 11]
    want_v, want_i = osc.merge_topk(vals, idx.to(torch.int64), k)
    got_v, got_i = tscan.topk_merge(vals.to(dev), idx.to(dev), k)
    assert torch.equal(got_i.cpu().to(torch.int64), want_i)
    assert torch.equal(got_v.cpu(), want_v)
    # fewer candidates than k -> padded
    got_v, got_i = tscan.topk_merge(vals[:, :7].contiguous().to(dev), idx[:, :7].contiguous().to(dev), 10)
    assert (got_i.cpu()[:, 6:] == -1).all() and torch.isinf(got_v.cpu()[:, 6:]).all()


def test_abi_argument_errors(dev, built_lib):
    from tensor_truth_amd import _lib, scan as tscan

    c = torch.zeros(64, 100, dtype=torch.bfloat16, device=dev)  # dim not multiple of 128
    with pytest.raises(_lib.TTError, match="dim"):
        tscan.scan_topk(c, c[:2].contiguous(), 5)
    c = torch.zeros(64, 128, dtype=torch.bfloat16, device=dev)
    with pytest.raises(_lib.TTError, match="k="):
        tscan.scan_topk(c, c[:2].contiguous(), 5000)
    with pytest.raises(TypeError):
        tscan.scan_topk(c.float(), c[:2].float(), 5)


def test_full_size_config_c4_shard_layouts_vs_torch(dev, built_lib):
    """BASELINE config 4 at full size: 10M x 1024 bf16 (20.5 GB), K=50, a 32-query batch, as ONE shard and as the eight
    row shards of an 8-GPU node merged with tt_topk_merge -- against an independent full-size checker (torch matmul of
    the same bf16 data with fp32 accumulation, running top-k over 1M-row pieces; the CPU oracle would need minutes).
    Properties checked at this size: planted neighbours are found first, scores are sorted, indices are unique, and
    the single-shard and sharded answers are identical."""
    import bench
    from tensor_truth_amd import scan as tscan
    from tensor_truth_amd.sharded import shard_bounds

    n, d, q, k = 10_000_000, 1024, 32, 50
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 40 * 2 ** 30:
        pytest.skip("needs ~40 GB of free HBM")
    corpus = bench.synth_corpus_shard(n, d, 1234, dev)
    g = torch.Generator(device=dev).manual_seed(4321)
    planted = torch.randint(0, n, (q // 2,), generator=g, device=dev)
    u = torch.randn(q // 2, d, generator=g, device=dev)
    u = u / u.norm(dim=1, keepdim=True)
    qa = corpus[planted].float() + 0.5 * u                     # cos ~ 0.89 to its row: the known top-1
    qb = torch.randn(q - q // 2, d, generator=g, device=dev)
    queries = torch.cat([qa, qb])
    queries = (queries / queries.norm(dim=1, keepdim=True)).to(torch.bfloat16)

    s, i = tscan.scan_topk(corpus, queries, k)
    assert (i[: q // 2, 0].long() == planted).all()
    assert (s[:, :-1] >= s[:, 1:]).all() and (i >= 0).all() and (i < n).all()
    assert all(len(set(row.tolist())) == k for row in i.cpu())

    # independent checker: running exact top-k over 1M-row pieces
    best_s = torch.full((q, k), -float("inf"), device=dev)
    best_i = torch.zeros((q, k), dtype=torch.int64, device=dev)
    qf = queries.float()
    for lo in range(0, n, 1_000_000):
        sc = qf @ corpus[lo:lo + 1_000_000].float().T          # bf16 values, fp32 accumulate
        ps, pi = torch.topk(sc, k, dim=1)
        cat_s, cat_i = torch.cat([best_s, ps], 1), torch.cat([best_i, pi + lo], 1)
        order = torch.argsort(cat_s, dim=1, descending=True, stable=True)[:, :k]
        best_s, best_i = torch.gather(cat_s, 1, order), torch.gather(cat_i, 1, order)
    assert torch.allclose(s, best_s, rtol=1e-3, atol=1e-6)
    gap = (best_s[:, :-1] - best_s[:, 1:]).min(dim=1).values
    tie_free = gap > 1e-6
    assert tie_free.float().mean().item() >= 0.5
    assert torch.equal(i[tie_free].long(), best_i[tie_free])

    # the same search as eight row shards + merge (what 8 ranks compute, minus the all-gather)
    parts = []
    for r in range(8):
        lo, hi = shard_bounds(n, 8, r)
        parts.append(tscan.scan_topk(corpus[lo:hi], queries, k, idx_base=lo))
    ms, mi = tscan.topk_merge(torch.cat([p[0] for p in parts], 1), torch.cat([p[1] for p in parts], 1), k)
    assert torch.equal(ms, s) and torch.equal(mi[tie_free], i[tie_free])


def _planted_queries(corpus, q, seed, dev):
    """Half of the queries are a corpus row + 0.5 x a unit direction (cos ~ 0.89: the known top-1), half pure noise."""
    n, d = corpus.shape
    g = torch.Generator(device=dev).manual_seed(seed)
    planted = torch.randint(0, n, (q // 2,), generator=g, device=dev)
    u = torch.randn(q // 2, d, generator=g, device=dev)
    u = u / u.norm(dim=1, keepdim=True)
    qa = corpus[planted].float() + 0.5 * u
    qb = torch.randn(q - q // 2, d, generator=g, device=dev)
    queries = torch.cat([qa, qb])
    return (queries / queries.norm(dim=1, keepdim=True)).to(torch.bfloat16), planted


def _running_topk_checker(corpus, queries, k, piece=500_000):
    """Independent full-size checker: torch matmul of the same bf16 data with fp32 accumulation, exact running top-k over pieces."""
    dev = corpus.device
    q, n = queries.shape[0], corpus.shape[0]
    best_s = torch.full((q, k), -float("inf"), device=dev)
    best_i = torch.zeros((q, k), dtype=torch.int64, device=dev)
    qf = queries.float()
    for lo in range(0, n, piece):
        sc = qf @ corpus[lo:lo + piece].float().T
        ps, pi = torch.topk(sc, min(k, sc.shape[1]), dim=1)
        cat_s, cat_i = torch.cat([best_s, ps], 1), torch.cat([best_i, pi + lo], 1)
        order = torch.argsort(cat_s, dim=1, descending=True, stable=True)[:, :k]
        best_s, best_i = torch.gather(cat_s, 1, order), torch.gather(cat_i, 1, order)
        del sc
    return best_s, best_i


@pytest.mark.parametrize("n", [1_250_000, 10_000_000])
def test_tiled_scan_256_queries_at_bench_sizes_vs_torch(dev, built_lib, n):
    """The path an 8-GPU step takes, at its own sizes: 256 gathered queries over one GPU's 1.25 M x 1024 shard
    and over the whole 10 M x 1024 corpus THROUGH tt_scan_topk's tiled MFMA filter pass (65+ queries; the threshold sample shrinks
    with the shard, scan_api.hip) -- indices bit-exact against the fp32 running top-k checker on every tie-free query, status flag
    clear, planted neighbours first; at 10 M rows also eight row shards of 256 queries each + tt_topk_merge = the single pass."""
    from tensor_truth_amd import scan as tscan
    from tensor_truth_amd.sharded import shard_bounds
    import bench

    d, q, k = 1024, 256, 50
    free, _ = torch.cuda.mem_get_info(dev)
    if free < (40 if n > 2_000_000 else 12) * 2 ** 30:
        pytest.skip("not enough free HBM")
    corpus = bench.synth_corpus_shard(n, d, 1234, dev)
    queries, planted = _planted_queries(corpus, q, 4321 + n % 97, dev)
    s, i, overflowed = tscan.scan_topk(corpus, queries, k, return_flag=True)
    assert not overflowed, "the tiled pass flagged a candidate-list overflow on the bench's own data"
    assert (i[: q // 2, 0].long() == planted).all()
    assert (s[:, :-1] >= s[:, 1:]).all() and (i >= 0).all() and (i < n).all()
    # the checker keeps k + 1 rows: a query is tie-free when every adjacent gap INCLUDING the one between rank k and rank k + 1 exceeds
    # 1e-6 (with 256 queries a near-tie at the cut -- two fp32 evaluations of the same dot product differ by ~1e-7 -- is likely in a run)
    best_s, best_i = _running_topk_checker(corpus, queries, k + 1)
    gap = (best_s[:, :-1] - best_s[:, 1:]).min(dim=1).values
    best_s, best_i = best_s[:, :k], best_i[:, :k]
    assert torch.allclose(s, best_s, rtol=REL_TOL, atol=1e-6)
    tie_free = gap > 1e-6
    assert tie_free.float().mean().item() >= 0.5, f"tie-free fraction {tie_free.float().mean().item():.2f}"
    assert torch.equal(i[tie_free].long(), best_i[tie_free])
    # every returned index appears once, and the streaming kernel (<= 64 queries per pass) gives the same bits for the same queries
    assert all(len(set(row.tolist())) == k for row in i[::16].cpu())
    s64, i64 = tscan.scan_topk(corpus, queries[:64].contiguous(), k)
    assert torch.equal(i64[tie_free[:64]], i[:64][tie_free[:64]])
    assert torch.allclose(s64, s[:64], rtol=1e-5, atol=1e-6)
    if n == 10_000_000:
        parts = []
        for r in range(8):
            lo, hi = shard_bounds(n, 8, r)
            ps, pi, ov = tscan.scan_topk(corpus[lo:hi], queries, k, idx_base=lo, return_flag=True)
            assert not ov
            parts.append((ps, pi))
        ms, mi = tscan.topk_merge(torch.cat([p[0] for p in parts], 1), torch.cat([p[1] for p in parts], 1), k)
        assert torch.equal(ms, s) and torch.equal(mi[tie_free], i[tie_free])


@pytest.mark.parametrize("offsets,k,nq", [
    ([0, 1000, 1000, 1037, 5000, 12288], 10, 3),     # ragged modules, an empty one, one shorter than 64 rows
    ([5, 9, 300, 4096], 50, 1),                      # rows before/after the modules are ignored; 4 rows < k
    ([0, 70000], 20, 2),                             # one module longer than a selection piece (65536 rows)
    ([3, 140000, 140100, 206000], 50, 2),            # long + short + long: pieces merged per module
])
def test_segmented_scan_matches_per_module_oracle(dev, built_lib, offsets, k, nq):
    """Several index modules in one matrix: per-(query, module) top-k from one pass
    equals the oracle's one-search-per-module, and is bit-identical to separate tt_scan_topk calls."""
    from tensor_truth_amd import scan as tscan

    n = max(offsets[-1] + 17, 4200)
    corpus = osc.synth_corpus(n, 1024, seed=99)
    queries, _ = osc.synth_queries(corpus, nq, seed=7)
    want_s, want_i, gap = osc.scan_topk_segmented(corpus, queries, k, offsets)
    cd, qd = corpus.to(dev), queries.to(dev)
    s, i = tscan.scan_topk_segmented(cd, qd, k, offsets)
    torch.cuda.synchronize()
    n_seg = len(offsets) - 1
    assert s.shape == (nq, n_seg, k) and i.shape == (nq, n_seg, k)
    _check(s.reshape(nq * n_seg, k), i.reshape(nq * n_seg, k), want_s.reshape(nq * n_seg, k),
           want_i.reshape(nq * n_seg, k), gap.reshape(-1))
    for m in range(n_seg):
        lo, hi = offsets[m], offsets[m + 1]
        if hi == lo:
            assert (i[:, m] == -1).all() and torch.isinf(s[:, m]).all()
            continue
        s1, i1 = tscan.scan_topk(cd[lo:hi].contiguous(), qd, k, exact_dense=True)
        assert torch.equal(s1, s[:, m]) and torch.equal(i1, i[:, m])


def test_segmented_scan_argument_errors(dev, built_lib):
    from tensor_truth_amd import scan as tscan

    corpus = osc.synth_corpus(256, 128, seed=1).to(dev)
    q = corpus[:2].contiguous()
    with pytest.raises(RuntimeError, match="non-decreasing"):
        tscan.scan_topk_segmented(corpus, q, 5, [0, 100, 50])
    with pytest.raises(RuntimeError, match="n_rows"):
        tscan.scan_topk_segmented(corpus, q, 5, [0, 300])
    with pytest.raises(RuntimeError, match="n_segments"):
        tscan.scan_topk_segmented(corpus, q, 5, list(range(0, 67)))


def test_fp8_shadow_prefilter_is_bit_identical_to_the_bf16_scan(dev, built_lib):
    """Round 6: ``tt_scan_topk_shadow`` -- one pass over an e4m3 shadow of the corpus lists the rows whose rigorous upper bound reaches
    the exact threshold, those rows are re-scored from the bf16 matrix with the streaming kernel's own arithmetic -- returns the SAME
    scores and indices as ``tt_scan_topk``, bit for bit: random rows, planted neighbours, exact duplicates (ties broken by row),
    NaN tombstones, rows that are not unit-norm, an index base, several widths, a shadow built in two pieces; a corpus of identical rows
    overflows the survivor lists and comes back through the flagged fallback, still exact.  Indices also against the CPU oracle."""
    from tensor_truth_amd import scan as tscan

    g = torch.Generator(device=dev).manual_seed(99)

    def corpus_of(n, d):
        x = torch.randn((n, d), generator=g, device=dev)
        return (x / x.norm(dim=1, keepdim=True)).to(torch.bfloat16)

    n, d, k = 1_300_000, 1024, 50
    c = corpus_of(n, d)
    q = corpus_of(3, d).float()
    q[1] = c[777_777].float() + 0.5 * q[1]                       # planted neighbour
    q = (q / q.norm(dim=1, keepdim=True)).to(torch.bfloat16)
    c[5] = c[1_200_000]                                           # exact duplicates: equal scores, the lower row first
    c[900_001] = c[123]
    c[40_000:40_064] = float("nan")                               # tombstones
    c[1000:1100] *= 3.0                                           # not unit-norm: the bounds are per row, not assumed
    c[2000:2100] *= 0.01
    sh = tscan.ScanShadow(c[: n // 2].contiguous(), cap_rows=n)   # built in two pieces (rows appended later)
    sh.extend(c, n)
    assert sh.rows == n and sh.serves(n, 3, k) and not sh.serves(n, 5, k)
    for nq in (1, 3):
        want_s, want_i, flag0 = tscan.scan_topk(c, q[:nq], k, idx_base=7, return_flag=True)
        got_s, got_i, flag1 = tscan.scan_topk(c, q[:nq], k, idx_base=7, return_flag=True, shadow=sh)
        torch.cuda.synchronize()
        assert not flag0 and not flag1, (flag0, flag1)
        assert torch.equal(got_i, want_i) and torch.equal(got_s.view(torch.int32), want_s.view(torch.int32))
    assert int(got_i[1, 0]) == 777_777 + 7
    hs, hi = tscan.scan_topk_host(c, q[:1], k, shadow=sh)
    assert torch.equal(hi, want_i[:1].cpu() - 7) and torch.equal(hs, want_s[:1].cpu())
    # against the CPU oracle (tie-free queries: bit-exact indices)
    clean = c.clone()
    clean[40_000:40_064] = 0
    o_s, o_i, gap = osc.scan_topk(clean.cpu(), q.cpu(), k)
    tf = gap > 1e-6
    assert tf.any() and torch.equal((got_i.cpu().to(torch.int64) - 7)[tf], o_i[tf])
    del c, clean, sh
    # other widths and k
    for d2, k2 in ((384, 10), (128, 100), (768, 1)):
        c2 = corpus_of(1_100_000, d2)
        q2 = corpus_of(2, d2)
        sh2 = tscan.ScanShadow(c2)
        a = tscan.scan_topk(c2, q2, k2, return_flag=True)
        b = tscan.scan_topk(c2, q2, k2, return_flag=True, shadow=sh2)
        torch.cuda.synchronize()
        assert not b[2] and torch.equal(a[1], b[1]) and torch.equal(a[0].view(torch.int32), b[0].view(torch.int32)), (d2, k2)
        del c2, sh2
    # every row identical: every row survives the prefilter, the lists overflow, the flag sends the call to the exact fallback
    row = corpus_of(1, 256)
    c3 = row.repeat(1_050_000, 1).contiguous()
    sh3 = tscan.ScanShadow(c3)
    s3, i3, flagged = tscan.scan_topk(c3, row, 20, return_flag=True, shadow=sh3)
    assert flagged
    s3, i3 = tscan.scan_topk(c3, row, 20, shadow=sh3)
    assert i3[0].tolist() == list(range(20))
