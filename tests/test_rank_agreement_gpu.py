"""GPU: the SURVEY.md section 8d rerank gate AT DEPTH -- "top-10 set / ordering identical where oracle score gaps >
tolerance" for the call the reference makes at ``services/rag_service.py:617-627``: 50 candidates per query through the
24-layer bge-reranker-v2-m3 shape, ~292 tokens per (query, chunk) pair, HIP bf16 (and the fp8 mode) against the plain
fp32 CPU oracle on the same token ids.

The oracle forward is the expensive part (4 queries x 50 pairs x 292 tokens x 24 layers of fp32 on host cores), so it
runs once per module and every mode is checked against it.  Weights: HF-style random init (no network for checkpoints);
vocabulary and position tables cut to what the inputs use -- every layer has the named model's full shape.
"""
import numpy as np
import pytest
import torch

from oracle import encoder as oe
from rank_checks import assert_order_on_separable, assert_topn_on_separable, kendall_tau, topn_overlap

pytestmark = pytest.mark.gpu

N_QUERIES, N_PAIRS, PAIR_TOKENS, QUERY_TOKENS, TOP_N = 4, 50, 292, 32, 10
SHAPE = dict(arch="xlmr", vocab_size=8192, hidden=1024, layers=24, heads=16, ffn=4096, max_pos=514, type_vocab=1,
             pad_id=1, ln_eps=1e-5, num_labels=1)
WEIGHT_SEED = 17
GOLDEN_NAME = "rank_oracle_24L_4x50x292.npz"   # tests/golden/make_rank_golden.py
STRESS_EMB_GOLDEN_NAME = "embed_oracle_stress_24L_16x292.npz"     # ... --stress-embeddings: normalised first-row hidden states, same weights
STRESS_EMB_PAIRS = 16
STRESS_GOLDEN_NAME = "rank_oracle_stress_24L_4x50x292.npz"   # ... --stress: the stress fixture (tests/stress_weights.py: the builder's hostile construction, not a trained model)
BF16_BOUND = 2e-2      # stated absolute bound of the bf16 mode on a sigmoid score after 24 layers
FP16_BOUND = 3e-3      # ... of the fp16 mode (same rate, three more mantissa bits per rounding point; measured 2.4e-3)
FP8_BOUND = 0.2        # stated bound of the fp8 throughput mode; its QUALITY gate is rank agreement, below


def _pairs():
    """<s> q </s></s> chunk </s>: every query shares its 32 tokens across its 50 pairs, as the postprocessor builds them."""
    rng = np.random.default_rng(20261003)
    out = np.empty((N_QUERIES, N_PAIRS, PAIR_TOKENS), dtype=np.int64)
    for q in range(N_QUERIES):
        qtok = rng.integers(4, SHAPE["vocab_size"], size=QUERY_TOKENS)
        out[q, :, 0] = 0
        out[q, :, 1:1 + QUERY_TOKENS] = qtok
        out[q, :, 1 + QUERY_TOKENS:3 + QUERY_TOKENS] = 2
        out[q, :, 3 + QUERY_TOKENS:-1] = rng.integers(4, SHAPE["vocab_size"], size=(N_PAIRS, PAIR_TOKENS - QUERY_TOKENS - 4))
        out[q, :, -1] = 2
    return out


@pytest.fixture(scope="module")
def oracle_scores():
    """fp32 oracle sigmoid scores [4, 50] + the weights + the token ids (computed once)."""
    ocfg = oe.EncoderConfig(**SHAPE)
    W = oe.synth_weights(ocfg, seed=WEIGHT_SEED)
    pairs = _pairs()
    import hashlib
    import os

    # the oracle's scores for exactly these weights and token ids are a committed fixture (minutes of host time to
    # recompute: tests/golden/make_rank_golden.py); anything else -- missing file, other inputs -- recomputes them here
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", GOLDEN_NAME)
    if os.path.exists(path) and os.environ.get("TT_RECOMPUTE_RANK_ORACLE") != "1":
        from rank_checks import weights_checksum

        z = np.load(path)
        if str(z["pairs_sha256"]) == hashlib.sha256(pairs.tobytes()).hexdigest() and str(z["weights_sha256"]) == weights_checksum(W):
            return ocfg, W, pairs, torch.from_numpy(z["scores"].astype(np.float32))

    before = torch.get_num_threads()
    try:
        torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 128)))      # large matmuls: use the host's cores
    except Exception:  # noqa: BLE001
        pass
    want = torch.empty(N_QUERIES, N_PAIRS)
    with torch.no_grad():
        for q in range(N_QUERIES):
            ids = torch.from_numpy(pairs[q])
            want[q] = oe.rerank_scores(ids, torch.ones_like(ids), W, ocfg)     # plain fp32 math, fp32 weights
    torch.set_num_threads(before)
    return ocfg, W, pairs, want


def _product_scores(dev, W, pairs, gemm_dtype):
    from tensor_truth_amd.encoder import Encoder, EncoderConfig, EncoderWeights, pack_token_matrix

    cfg = EncoderConfig(**SHAPE)
    enc = Encoder(EncoderWeights(cfg, W, dev, dtype=torch.float16 if gemm_dtype == "fp16" else torch.bfloat16))
    flat = pairs.reshape(-1, PAIR_TOKENS).astype(np.int32)
    batch = pack_token_matrix(flat, cfg)
    if gemm_dtype == "fp8":
        enc.calibrate_fp8(pack_token_matrix(flat[:64], cfg))
        enc.w.set_gemm_dtype("fp8")
    got = enc.rerank_packed(batch).cpu().view(N_QUERIES, N_PAIRS)
    torch.cuda.synchronize()
    return got


def test_bf16_top10_of_50_at_full_depth(dev, built_lib, oracle_scores):
    ocfg, W, pairs, want = oracle_scores
    got = _product_scores(dev, W, pairs, "bf16")
    err = (got - want).abs().max().item()
    assert err <= BF16_BOUND, f"bf16 score error after 24 layers: {err}"
    n_sep = n_in = n_out = n_exact = 0
    for q in range(N_QUERIES):
        n_sep += assert_order_on_separable(want[q].numpy(), got[q].numpy(), 2 * BF16_BOUND, f"query {q}")
        a, b, exact = assert_topn_on_separable(want[q].numpy(), got[q].numpy(), TOP_N, 2 * BF16_BOUND, f"query {q}")
        n_in, n_out, n_exact = n_in + a, n_out + b, n_exact + int(exact)
    total = N_QUERIES * N_PAIRS * (N_PAIRS - 1) // 2
    # the gate must have teeth: a random head spreads 50 scores over ~0.3, so a good share of all pairs is separable
    assert n_sep >= total // 10, (n_sep, total)
    assert n_in + n_out >= N_QUERIES * 10
    taus = [kendall_tau(want[q].numpy(), got[q].numpy()) for q in range(N_QUERIES)]
    over = [topn_overlap(want[q].numpy(), got[q].numpy(), TOP_N) for q in range(N_QUERIES)]
    print(f"bf16 @24L, {N_QUERIES}x{N_PAIRS} pairs x {PAIR_TOKENS} tok: max |err| {err:.4f}; {n_sep}/{total} pairs separable "
          f"at gap > {2 * BF16_BOUND} all ordered as the oracle; top-{TOP_N}: {n_in} decisive members in, {n_out} decisive "
          f"non-members out, exact set required for {n_exact}/{N_QUERIES} queries; Kendall tau {min(taus):.3f}..{max(taus):.3f}; "
          f"top-{TOP_N} overlap {min(over):.2f}..{max(over):.2f}")
    # informational floor (measured on this random-init model: tau 0.85-0.90 -- 50 scores spread over ~0.3 with bf16
    # noise of ~1e-2 on each; the GATE is the separable-pair / decisive-member assertions above)
    assert min(taus) >= 0.8 and min(over) >= 0.7
    # Would a head that spreads the candidates over the whole sigmoid range give the top-10 check more teeth?
    # Scaling classifier.out_proj by a power of two a and shifting its bias maps every logit l to a (l - c) EXACTLY in both the
    # oracle and the product (the head's last dot product is fp32 on both sides), so the scores such a model would produce follow
    # from the measured ones: s' = sigmoid(a (logit(s) - c)).  Error and spread grow together -- the cut between rank 10 and 11
    # stays inside twice the error on every query -- which is why the exact-top-10 assertion lives in the reference-precision test.
    lw, lg = torch.logit(want.double()), torch.logit(got.double())
    c = lw.median()
    for a in (4.0, 8.0):
        w2, g2 = torch.sigmoid(a * (lw - c)), torch.sigmoid(a * (lg - c))
        e2 = (g2 - w2).abs().max().item()
        n_exact2 = 0
        for q in range(N_QUERIES):
            assert_order_on_separable(w2[q].numpy(), g2[q].numpy(), 2 * e2, f"scaled head x{a:g}, query {q}")
            n_exact2 += int(assert_topn_on_separable(w2[q].numpy(), g2[q].numpy(), TOP_N, 2 * e2, f"scaled head x{a:g}, query {q}")[2])
        print(f"  head scaled x{a:g}: oracle scores span {w2.min().item():.2f}..{w2.max().item():.2f}, bf16 max |err| {e2:.3f}; "
              f"exact top-{TOP_N} set decidable (cut gap > 2 x err) on {n_exact2}/{N_QUERIES} queries")


def test_fp16_top10_of_50_at_full_depth(dev, built_lib, oracle_scores):
    """The fp16 mode (`precision="fp16"`): the bf16 mode's rate, an order of magnitude closer to the oracle -- so the gate that is
    thin for bf16 (few separable pairs, top-10 membership rarely decidable) has teeth here."""
    ocfg, W, pairs, want = oracle_scores
    got = _product_scores(dev, W, pairs, "fp16")
    err = (got - want).abs().max().item()
    assert err <= FP16_BOUND, f"fp16 score error after 24 layers: {err}"
    n_sep = n_in = n_out = n_exact = 0
    for q in range(N_QUERIES):
        n_sep += assert_order_on_separable(want[q].numpy(), got[q].numpy(), 2 * FP16_BOUND, f"query {q}")
        a, b, exact = assert_topn_on_separable(want[q].numpy(), got[q].numpy(), TOP_N, 2 * FP16_BOUND, f"query {q}")
        n_in, n_out, n_exact = n_in + a, n_out + b, n_exact + int(exact)
    total = N_QUERIES * N_PAIRS * (N_PAIRS - 1) // 2
    taus = [kendall_tau(want[q].numpy(), got[q].numpy()) for q in range(N_QUERIES)]
    over = [topn_overlap(want[q].numpy(), got[q].numpy(), TOP_N) for q in range(N_QUERIES)]
    print(f"fp16 @24L, {N_QUERIES}x{N_PAIRS} pairs x {PAIR_TOKENS} tok: max |err| {err:.5f}; {n_sep}/{total} pairs separable "
          f"at gap > {2 * FP16_BOUND} all ordered as the oracle; top-{TOP_N}: {n_in} decisive members in, {n_out} decisive "
          f"non-members out, exact set required for {n_exact}/{N_QUERIES} queries; Kendall tau {min(taus):.3f}..{max(taus):.3f}; "
          f"top-{TOP_N} overlap {min(over):.2f}..{max(over):.2f}")
    assert n_sep >= total * 6 // 10, (n_sep, total)                 # most pairs are separable at this bound (bf16: ~a quarter)
    assert n_in + n_out >= N_QUERIES * 35                            # and most of the 50 candidates are decisive for the top-10
    assert min(taus) >= 0.97 and min(over) >= 0.9


def test_fp8_rank_quality_at_full_depth(dev, built_lib, oracle_scores):
    """The fp8 mode's stated metric (BASELINE config 5): rank agreement with the fp32 oracle, not a loose score bound."""
    ocfg, W, pairs, want = oracle_scores
    got = _product_scores(dev, W, pairs, "fp8")
    err = (got - want).abs().max().item()
    assert err <= FP8_BOUND, f"fp8 score error after 24 layers: {err}"
    taus = [kendall_tau(want[q].numpy(), got[q].numpy()) for q in range(N_QUERIES)]
    over = [topn_overlap(want[q].numpy(), got[q].numpy(), TOP_N) for q in range(N_QUERIES)]
    # candidates the oracle separates by more than twice the MEASURED fp8 error must still be ordered
    n_sep = sum(assert_order_on_separable(want[q].numpy(), got[q].numpy(), 2 * err, f"fp8 query {q}") for q in range(N_QUERIES))
    print(f"fp8 @24L: max |err| {err:.4f}; Kendall tau {min(taus):.3f}..{max(taus):.3f} (mean {np.mean(taus):.3f}); "
          f"top-{TOP_N} overlap {min(over):.2f}..{max(over):.2f} (mean {np.mean(over):.2f}); {n_sep} pairs separable at 2x err")
    # Measured on this random-init model (scores of 50 candidates within ~0.3 of each other, e4m3 noise up to ~0.1 on
    # each after 96 fp8 GEMMs): Kendall tau 0.44-0.54, top-10 overlap 0.6-0.7 -- the fp8 mode trades this much ranking
    # fidelity for 1.27x reranker throughput when candidates are this close; the floors below catch regressions.
    assert np.mean(taus) >= 0.4 and np.mean(over) >= 0.5, (taus, over)


def test_fp32_scores_within_1e3_relative_at_full_depth(dev, built_lib, oracle_scores):
    """The reference-precision path (model_kwargs torch_dtype=float32, csrc/f32_path.hip) at full depth: north_star's
    "fp scores within 1e-3 relative" for all 200 pairs, and the oracle's own ranking reproduced."""
    from tensor_truth_amd.encoder import EncoderConfig, pack_token_matrix
    from tensor_truth_amd.encoder_f32 import EncoderF32, EncoderWeightsF32

    ocfg, W, pairs, want = oracle_scores
    cfg = EncoderConfig(**SHAPE)
    enc = EncoderF32(EncoderWeightsF32(cfg, W, dev))
    import time
    got = torch.empty_like(want)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for q in range(N_QUERIES):                                   # one query's 50 pairs per call: the interactive case
        got[q] = enc.rerank_packed(pack_token_matrix(pairs[q].astype(np.int32), cfg)).cpu()
    dt = (time.perf_counter() - t0) / N_QUERIES
    rel = ((got - want).abs() / want.abs()).max().item()
    assert rel <= 1e-3, f"fp32 path: relative score error {rel}"
    n_sep = 0
    for q in range(N_QUERIES):
        n_sep += assert_order_on_separable(want[q].numpy(), got[q].numpy(), 1e-4, f"fp32 query {q}")
        assert_topn_on_separable(want[q].numpy(), got[q].numpy(), TOP_N, 1e-4, f"fp32 query {q}")
    taus = [kendall_tau(want[q].numpy(), got[q].numpy()) for q in range(N_QUERIES)]
    print(f"fp32 @24L: max relative score error {rel:.2e}; {n_sep} pairs separable at 1e-4 all ordered as the oracle; "
          f"Kendall tau min {min(taus):.4f}; {dt * 1e3:.0f} ms per query of {N_PAIRS} pairs x {PAIR_TOKENS} tok")
    assert min(taus) >= 0.995


def test_bf16x3_scores_within_1e3_relative_at_full_depth(dev, built_lib, oracle_scores):
    """The FAST reference-precision mode (TT_PRECISION=reference / torch_dtype=float32 on a 1024-wide model: split-bf16 on
    the bf16 matrix cores, csrc/x3_path.hip) at full depth: north_star's "fp scores within 1e-3 relative" for all 200
    pairs, Kendall tau = 1.000 against the oracle, top-10 identical -- and a whole query (50 pairs x 292 tokens x 24
    layers) in about a fifth of the fp32-MFMA path's time."""
    import time

    from tensor_truth_amd.encoder import EncoderConfig, pack_token_matrix
    from tensor_truth_amd.encoder_x3 import EncoderWeightsX3, EncoderX3

    ocfg, W, pairs, want = oracle_scores
    cfg = EncoderConfig(**SHAPE)
    enc = EncoderX3(EncoderWeightsX3(cfg, W, dev))
    got = torch.empty_like(want)
    enc.rerank_packed(pack_token_matrix(pairs[0].astype(np.int32), cfg))          # warm-up (workspace, LDS attributes)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for q in range(N_QUERIES):                                   # one query's 50 pairs per call: the interactive case
        got[q] = enc.rerank_packed(pack_token_matrix(pairs[q].astype(np.int32), cfg)).cpu()
    dt = (time.perf_counter() - t0) / N_QUERIES
    rel = ((got - want).abs() / want.abs()).max().item()
    assert rel <= 1e-3, f"bf16x3 path: relative score error {rel}"
    n_sep = 0
    for q in range(N_QUERIES):
        n_sep += assert_order_on_separable(want[q].numpy(), got[q].numpy(), 2e-4, f"bf16x3 query {q}")
        assert_topn_on_separable(want[q].numpy(), got[q].numpy(), TOP_N, 2e-4, f"bf16x3 query {q}")
    taus = [kendall_tau(want[q].numpy(), got[q].numpy()) for q in range(N_QUERIES)]
    over = [topn_overlap(want[q].numpy(), got[q].numpy(), TOP_N) for q in range(N_QUERIES)]
    # all four queries in one batch: the throughput form
    flat = pairs.reshape(-1, PAIR_TOKENS).astype(np.int32)
    enc.rerank_packed(pack_token_matrix(flat, cfg))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got_b = enc.rerank_packed(pack_token_matrix(flat, cfg)).cpu().view(N_QUERIES, N_PAIRS)
    dt_b = (time.perf_counter() - t0) / N_QUERIES
    assert ((got_b - want).abs() / want.abs()).max().item() <= 1e-3
    print(f"bf16x3 @24L: max relative score error {rel:.2e}; {n_sep} pairs separable at 2e-4 all ordered as the oracle; "
          f"Kendall tau min {min(taus):.4f}; top-{TOP_N} overlap min {min(over):.2f}; {dt * 1e3:.1f} ms per query of {N_PAIRS} "
          f"pairs x {PAIR_TOKENS} tok alone ({1.0 / dt:.1f} q/s), {dt_b * 1e3:.1f} ms per query in a batch of {N_QUERIES} ({1.0 / dt_b:.1f} q/s)")
    assert min(taus) >= 0.999 and min(over) == 1.0


def test_f16c_scores_within_1e3_relative_at_full_depth(dev, built_lib, oracle_scores):
    """The DEFAULT mode of the unchanged reference calls since round 4 (precision "reference" on a 1024-wide model: fp16 main
    products + block-scaled e4m3 correction terms, two matrix-time units -- csrc/f16c_path.hip) at full depth: north_star's
    "fp scores within 1e-3 relative" for all 200 pairs, Kendall tau >= 0.999 against the oracle, top-10 identical -- the
    gate VERDICT r03 item 1 names -- at 1.4x the split-bf16 path's rate (the CPU emulation that preceded the kernels,
    tools/probes/f16c_emulation.py, predicted 8.3e-5)."""
    import time

    from tensor_truth_amd.encoder import EncoderConfig, pack_token_matrix
    from tensor_truth_amd.encoder_f16c import EncoderF16C, EncoderWeightsF16C

    ocfg, W, pairs, want = oracle_scores
    cfg = EncoderConfig(**SHAPE)
    enc = EncoderF16C(EncoderWeightsF16C(cfg, W, dev))
    got = torch.empty_like(want)
    enc.rerank_packed(pack_token_matrix(pairs[0].astype(np.int32), cfg))          # warm-up (workspace, LDS attributes)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for q in range(N_QUERIES):                                   # one query's 50 pairs per call: the interactive case
        got[q] = enc.rerank_packed(pack_token_matrix(pairs[q].astype(np.int32), cfg)).cpu()
    dt = (time.perf_counter() - t0) / N_QUERIES
    rel = ((got - want).abs() / want.abs()).max().item()
    assert rel <= 1e-3, f"f16c path: relative score error {rel}"
    n_sep = 0
    for q in range(N_QUERIES):
        n_sep += assert_order_on_separable(want[q].numpy(), got[q].numpy(), 4e-4, f"f16c query {q}")
        assert_topn_on_separable(want[q].numpy(), got[q].numpy(), TOP_N, 4e-4, f"f16c query {q}")
    taus = [kendall_tau(want[q].numpy(), got[q].numpy()) for q in range(N_QUERIES)]
    over = [topn_overlap(want[q].numpy(), got[q].numpy(), TOP_N) for q in range(N_QUERIES)]
    flat = pairs.reshape(-1, PAIR_TOKENS).astype(np.int32)
    enc.rerank_packed(pack_token_matrix(flat, cfg))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got_b = enc.rerank_packed(pack_token_matrix(flat, cfg)).cpu().view(N_QUERIES, N_PAIRS)
    dt_b = (time.perf_counter() - t0) / N_QUERIES
    assert ((got_b - want).abs() / want.abs()).max().item() <= 1e-3
    assert torch.equal(got_b, got)                               # a pair's score does not depend on what shares its batch
    print(f"f16c @24L: max relative score error {rel:.2e}; {n_sep} pairs separable at 4e-4 all ordered as the oracle; "
          f"Kendall tau min {min(taus):.4f}; top-{TOP_N} overlap min {min(over):.2f}; {dt * 1e3:.1f} ms per query of {N_PAIRS} "
          f"pairs x {PAIR_TOKENS} tok alone ({1.0 / dt:.1f} q/s), {dt_b * 1e3:.1f} ms per query in a batch of {N_QUERIES} ({1.0 / dt_b:.1f} q/s)")
    assert min(taus) >= 0.999 and min(over) == 1.0


# ---- the same gate on the STRESS FIXTURE (tests/stress_weights.py: hostile construction, not a trained model) ----------------------
# Six hidden dimensions carry massive activations (20-60x the ordinary features) through every layer, half the heads attend with
# an entropy of 1.5 bits (logits of tens), and the calibrated head spreads a query's candidates over ~6 logits.  There is no
# network for real checkpoints: this is the offline stand-in for them.  Bounds below are MEASURED on this fixture and asserted
# with a margin; DESIGN.md section 2 carries them as the "stress" column of the tolerance table.
STRESS_REFERENCE_REL = 1e-3    # north_star's bar: the DEFAULT implementation of the reference precision (f16x3) holds it here too
STRESS_REL = {"f16x3": 1e-3,   # measured 1.7e-4 (CPU emulation: profiles/r04_f16c_emulation_stress.log)
              "bf16x3": 1.5e-3,  # measured 1.0017e-3: two bf16 planes carry 16 bits -- AT the bar when logits reach ~100
              "f16c": 1.2e-2}    # measured 7.2e-3 (on scores of ~0.005: 2e-3 absolute): the fast variant's stated bound
STRESS_FP16_BOUND = 3e-2       # absolute, fp16 mode (measured 2.0e-2)
STRESS_BF16_BOUND = 2e-1       # absolute, bf16 mode (measured 1.4e-1)


@pytest.fixture(scope="module")
def stress_oracle_scores():
    import hashlib
    import os

    import stress_weights
    from rank_checks import weights_checksum

    ocfg = oe.EncoderConfig(**SHAPE)
    pairs = _pairs()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", STRESS_GOLDEN_NAME)
    assert os.path.exists(path), f"{STRESS_GOLDEN_NAME} missing: python tests/golden/make_rank_golden.py --stress"
    z = np.load(path)
    W = stress_weights.with_head(stress_weights.apply(oe.synth_weights(ocfg, seed=WEIGHT_SEED), ocfg, qk_scales=z["qk_scales"]),
                                 z["head_w"], z["head_b"])
    assert str(z["pairs_sha256"]) == hashlib.sha256(pairs.tobytes()).hexdigest() and str(z["weights_sha256"]) == weights_checksum(W), \
        "the stress fixture belongs to other weights / token ids: regenerate it"
    return ocfg, W, pairs, torch.from_numpy(z["scores"].astype(np.float32))


def _mode_scores(dev, W, pairs, mode):
    from tensor_truth_amd.encoder import Encoder, EncoderConfig, EncoderWeights, pack_token_matrix

    cfg = EncoderConfig(**SHAPE)
    flat = pairs.reshape(-1, PAIR_TOKENS).astype(np.int32)
    if mode == "f16c":
        from tensor_truth_amd.encoder_f16c import EncoderF16C, EncoderWeightsF16C

        enc = EncoderF16C(EncoderWeightsF16C(cfg, W, dev))
    elif mode in ("bf16x3", "f16x3"):
        from tensor_truth_amd.encoder_x3 import EncoderWeightsX3, EncoderX3

        enc = EncoderX3(EncoderWeightsX3(cfg, W, dev, dtype=torch.float16 if mode == "f16x3" else torch.bfloat16))
    else:
        enc = Encoder(EncoderWeights(cfg, W, dev, dtype=torch.float16 if mode == "fp16" else torch.bfloat16))
        if mode == "fp8":
            enc.calibrate_fp8(pack_token_matrix(flat[:64], cfg))
            enc.w.set_gemm_dtype("fp8")
    got = enc.rerank_packed(pack_token_matrix(flat, cfg)).cpu().view(N_QUERIES, N_PAIRS)
    torch.cuda.synchronize()
    return got


def _report(name, want, got):
    err = (got - want).abs()
    rel = (err / want.abs()).max().item()
    taus = [kendall_tau(want[q].numpy(), got[q].numpy()) for q in range(N_QUERIES)]
    over = [topn_overlap(want[q].numpy(), got[q].numpy(), TOP_N) for q in range(N_QUERIES)]
    print(f"stress @24L {name}: scores {want.min().item():.3f}..{want.max().item():.3f}; max |err| {err.max().item():.2e}, max relative {rel:.2e}; "
          f"Kendall tau {min(taus):.4f}..{max(taus):.4f}; top-{TOP_N} overlap {min(over):.2f}..{max(over):.2f}; finite {bool(torch.isfinite(got).all())}")
    return err.max().item(), rel, min(taus), min(over)


@pytest.mark.parametrize("mode", ["f16x3", "f16c", "bf16x3"])
def test_stress_weights_reference_precision_implementations(dev, built_lib, stress_oracle_scores, mode):
    """The implementations of the reference mode on the stress weights.  f16x3 -- the default -- is still inside north_star's
    1e-3 relative with the oracle's ranking reproduced; bf16x3 sits at the bar (16-bit operands against logits of ~100) and the
    two-unit f16c path is outside it on the smallest scores (its e4m3 correction terms and single-fp16 P.V leave ~2e-3
    absolute; ranking intact): each is held to its MEASURED, stated bound, which is why f16x3 is the default."""
    ocfg, W, pairs, want = stress_oracle_scores
    got = _mode_scores(dev, W, pairs, mode)
    err, rel, tau, over = _report(mode, want, got)
    assert torch.isfinite(got).all()
    assert rel <= STRESS_REL[mode], f"{mode} on stress weights: relative score error {rel}"
    if mode == "f16x3":
        assert rel <= STRESS_REFERENCE_REL
    assert tau >= 0.995 and over >= 0.9
    for q in range(N_QUERIES):
        assert_order_on_separable(want[q].numpy(), got[q].numpy(), 2 * err + 1e-6, f"{mode} stress query {q}")


@pytest.mark.parametrize("mode", ["fp16", "bf16", "fp8"])
def test_stress_weights_16bit_modes_stay_finite_and_bounded(dev, built_lib, stress_oracle_scores, mode):
    """The named 16-bit modes (and the fp8 throughput mode) on the stress weights: finite -- fp16 saturates instead of overflowing
    -- and inside their MEASURED bounds, which are what DESIGN.md states for the stress fixture."""
    ocfg, W, pairs, want = stress_oracle_scores
    got = _mode_scores(dev, W, pairs, mode)
    err, rel, tau, over = _report(mode, want, got)
    assert torch.isfinite(got).all()
    if mode == "fp16":
        assert err <= STRESS_FP16_BOUND
    elif mode == "bf16":
        assert err <= STRESS_BF16_BOUND
    else:
        assert tau >= 0.2            # (a throughput mode: its gate is that it still ranks better than chance, stated as measured)


def test_f16x3_scores_within_1e3_relative_at_full_depth(dev, built_lib, oracle_scores):
    """The DEFAULT implementation of the reference precision (round 4): split-fp16 planes, three fp16 MFMA products per product --
    north_star's "fp scores within 1e-3 relative" for all 200 pairs, Kendall tau = 1.000, top-10 identical (and, unlike the
    other implementations, on the stress fixture too: test_stress_weights_reference_precision_implementations)."""
    import time

    from tensor_truth_amd.encoder import EncoderConfig, pack_token_matrix
    from tensor_truth_amd.encoder_x3 import EncoderWeightsX3, EncoderX3

    ocfg, W, pairs, want = oracle_scores
    cfg = EncoderConfig(**SHAPE)
    enc = EncoderX3(EncoderWeightsX3(cfg, W, dev, dtype=torch.float16))
    flat = pairs.reshape(-1, PAIR_TOKENS).astype(np.int32)
    enc.rerank_packed(pack_token_matrix(flat, cfg))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = enc.rerank_packed(pack_token_matrix(flat, cfg)).cpu().view(N_QUERIES, N_PAIRS)
    dt_b = (time.perf_counter() - t0) / N_QUERIES
    rel = ((got - want).abs() / want.abs()).max().item()
    assert rel <= 1e-3, f"f16x3 path: relative score error {rel}"
    n_sep = 0
    for q in range(N_QUERIES):
        n_sep += assert_order_on_separable(want[q].numpy(), got[q].numpy(), 2e-4, f"f16x3 query {q}")
        assert_topn_on_separable(want[q].numpy(), got[q].numpy(), TOP_N, 2e-4, f"f16x3 query {q}")
    taus = [kendall_tau(want[q].numpy(), got[q].numpy()) for q in range(N_QUERIES)]
    over = [topn_overlap(want[q].numpy(), got[q].numpy(), TOP_N) for q in range(N_QUERIES)]
    print(f"f16x3 @24L: max relative score error {rel:.2e}; {n_sep} pairs separable at 2e-4 all ordered as the oracle; Kendall tau min "
          f"{min(taus):.4f}; top-{TOP_N} overlap min {min(over):.2f}; {dt_b * 1e3:.1f} ms per query in a batch of {N_QUERIES} ({1.0 / dt_b:.1f} q/s)")
    assert min(taus) >= 0.999 and min(over) == 1.0


# ---- the bi-encoder side of the stress fixture: embeddings (normalised first-row hidden states) under the same hostile weights ----
STRESS_EMB_BOUNDS = {            # mode -> (min cosine, max |component error|); measured values in the test's report line
    # measured: f16x3 1.000000000 / 1.9e-6, bf16x3 0.999999998 / 8.1e-6, f16c 0.99999979 / 9.6e-5, fp16 0.99996853 / 8.5e-4,
    # bf16 0.99911500 / 5.5e-3 (largest component 0.55) -- bf16 holds SURVEY's 0.999 on these weights too, narrowly
    "f16x3": (0.99999999, 1e-5), "bf16x3": (0.9999999, 4e-5), "f16c": (0.999999, 4e-4), "fp16": (0.9999, 4e-3), "bf16": (0.999, 2.5e-2)}


@pytest.mark.parametrize("mode", ["f16x3", "bf16x3", "f16c", "fp16", "bf16"])
def test_stress_weights_embeddings(dev, built_lib, stress_oracle_scores, mode):
    """SURVEY section 8d's embedding gate (cosine >= 0.999) on the stress weights, per precision: 16 pairs' first-row hidden states,
    L2-normalised, against the fp64-accumulated oracle (tests/golden/make_rank_golden.py --stress-embeddings).  The outlier
    dimensions dominate these vectors as they do in trained encoders, so the cosine is a weak gate by itself -- the component-wise
    error is asserted too."""
    import hashlib
    import os

    from tensor_truth_amd.encoder import Encoder, EncoderConfig, EncoderWeights, pack_token_matrix

    ocfg, W, pairs, _ = stress_oracle_scores
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", STRESS_EMB_GOLDEN_NAME))
    ids = np.concatenate([pairs[q][:STRESS_EMB_PAIRS // N_QUERIES] for q in range(N_QUERIES)]).astype(np.int64)
    assert str(z["ids_sha256"]) == hashlib.sha256(ids.tobytes()).hexdigest(), "the embedding fixture belongs to other token ids"
    want = torch.from_numpy(z["embeddings"])
    cfg = EncoderConfig(**SHAPE)
    if mode == "f16c":
        from tensor_truth_amd.encoder_f16c import EncoderF16C, EncoderWeightsF16C
        enc = EncoderF16C(EncoderWeightsF16C(cfg, W, dev))
    elif mode in ("bf16x3", "f16x3"):
        from tensor_truth_amd.encoder_x3 import EncoderWeightsX3, EncoderX3
        enc = EncoderX3(EncoderWeightsX3(cfg, W, dev, dtype=torch.float16 if mode == "f16x3" else torch.bfloat16))
    else:
        enc = Encoder(EncoderWeights(cfg, W, dev, dtype=torch.float16 if mode == "fp16" else torch.bfloat16))
    got = enc.embed_packed(pack_token_matrix(ids.astype(np.int32), cfg))[0].cpu()
    torch.cuda.synchronize()
    cos = torch.nn.functional.cosine_similarity(got.double(), want.double(), dim=1).min().item()
    err = (got - want).abs().max().item()
    print(f"stress @24L embeddings {mode}: min cosine {cos:.9f}, max |component error| {err:.2e} (largest component {want.abs().max().item():.2f})")
    assert torch.isfinite(got).all()
    lo, hi = STRESS_EMB_BOUNDS[mode]
    assert cos >= lo and err <= hi, (mode, cos, err)
