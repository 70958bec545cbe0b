"""CPU: the oracle against the committed golden fixtures (made by
tests/golden/make_golden.py from transformers + the reference's own modules)."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from oracle import encoder as oe
from oracle import scan as osc


def _cfg(z):
    return oe.EncoderConfig(**json.loads(str(z["cfg"])))


def _wsum(W):
    return float(sum(float(v.double().abs().sum()) for v in W.values()))


@pytest.mark.parametrize("name", ["xlmr_encoder.npz", "bert_encoder.npz"])
def test_encoder_matches_transformers_golden(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    cfg = _cfg(z)
    W = oe.synth_weights(cfg, seed=int(z["seed"]))
    assert abs(_wsum(W) - float(z["wsum"])) < 1e-6 * float(z["wsum"]), "seeded weights changed"
    ids, mask = torch.from_numpy(z["ids"]), torch.from_numpy(z["mask"])
    type_ids = torch.from_numpy(z["type_ids"]) if "type_ids" in z.files else None
    hid = oe.encoder_forward(ids, mask, W, cfg, type_ids=type_ids)
    want = torch.from_numpy(z["hidden"])
    m = mask.bool()
    assert (hid - want)[m].abs().max().item() < 2e-4
    emb = oe.cls_pool_normalize(hid)
    assert (emb - torch.from_numpy(z["emb"])).abs().max().item() < 1e-5
    assert torch.allclose(emb.norm(dim=1), torch.ones(emb.shape[0]), atol=1e-6)


def test_rerank_head_matches_transformers_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "xenc_head.npz"))
    cfg = _cfg(z)
    W = oe.synth_weights(cfg, seed=int(z["seed"]))
    ids, mask = torch.from_numpy(z["ids"]), torch.from_numpy(z["mask"])
    logits = oe.rerank_logits(ids, mask, W, cfg)
    assert (logits - torch.from_numpy(z["logits"])).abs().max().item() < 2e-4
    scores = oe.rerank_scores(ids, mask, W, cfg)
    assert (scores - torch.from_numpy(z["scores"])).abs().max().item() < 1e-4
    assert ((scores > 0) & (scores < 1)).all()


def test_bf16_emulation_close_to_fp32(golden_dir):
    z = np.load(os.path.join(golden_dir, "xlmr_encoder.npz"))
    cfg = _cfg(z)
    W = {k: v.to(torch.bfloat16) for k, v in oe.synth_weights(cfg, seed=int(z["seed"])).items()}
    ids, mask = torch.from_numpy(z["ids"]), torch.from_numpy(z["mask"])
    e32 = oe.embed(ids, mask, W, cfg, emulate_bf16=False)
    e16 = oe.embed(ids, mask, W, cfg, emulate_bf16=True)
    cos = (e32 * e16).sum(dim=1)
    assert cos.min().item() > 0.999


def test_scan_golden_and_c_restatement(golden_dir):
    z = np.load(os.path.join(golden_dir, "scan_4096x1024_k50.npz"))
    corpus = osc.synth_corpus(int(z["n"]), int(z["d"]), seed=int(z["corpus_seed"]))
    queries, planted = osc.synth_queries(corpus, 16, seed=int(z["query_seed"]))
    assert abs(float(corpus.float().double().sum()) - float(z["corpus_sum"])) < 1e-6
    assert abs(float(queries.float().double().sum()) - float(z["query_sum"])) < 1e-6
    vals, idx, gap = osc.scan_topk(corpus, queries, int(z["k"]))
    tie_free = torch.from_numpy(z["gap"]) > 1e-6
    assert torch.equal(idx[tie_free].to(torch.int32), torch.from_numpy(z["idx"])[tie_free])
    assert np.allclose(vals.numpy(), z["scores"], rtol=1e-5, atol=1e-6)
    assert np.array_equal(planted.numpy(), z["planted"])
    # planted neighbours are found at rank 0
    for q in range(16):
        if planted[q] >= 0:
            assert idx[q, 0] == planted[q]
    # chunked path == single-shot path
    v2, i2, _ = osc.scan_topk(corpus, queries, int(z["k"]), chunk=1000)
    assert torch.equal(i2[tie_free], idx[tie_free])
    # independent plain-C restatement
    so = os.path.join(os.path.dirname(golden_dir), "..", "oracle", "liboracle_scan.so")
    if not os.path.exists(so):
        import subprocess

        subprocess.run(["make", "-C", os.path.dirname(so)], check=True)
    lib = ctypes.CDLL(so)
    cs = np.empty((16, 50), np.float32)
    ci = np.empty((16, 50), np.int32)
    rc = lib.tt_oracle_scan_topk(corpus.view(torch.int16).numpy().ctypes.data_as(ctypes.c_void_p),
                                 ctypes.c_int64(4096), 1024,
                                 queries.view(torch.int16).numpy().ctypes.data_as(ctypes.c_void_p), 16, 50,
                                 cs.ctypes.data_as(ctypes.c_void_p), ci.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    assert np.array_equal(ci[tie_free.numpy()], z["idx"][tie_free.numpy()])


def test_scan_oracle_edge_cases():
    corpus = osc.synth_corpus(7, 128, seed=1)
    queries, _ = osc.synth_queries(corpus, 3, seed=2)
    v, i, _ = osc.scan_topk(corpus, queries, 10)
    assert v.shape == (3, 10) and (i[:, 7:] == -1).all() and torch.isinf(v[:, 7:]).all()
    # duplicate rows: ties resolved by ascending index
    dup = torch.cat([corpus, corpus], 0)
    v, i, gap = osc.scan_topk(dup, queries, 4)
    assert (i[:, 0] + 7 == i[:, 1]).all() and (gap == 0).all()
    # empty corpus
    v, i, _ = osc.scan_topk(corpus[:0], queries, 5)
    assert (i == -1).all()
    # merge of shard-local lists == global
    big = osc.synth_corpus(3000, 128, seed=3)
    q, _ = osc.synth_queries(big, 4, seed=4)
    gv, gi, _ = osc.scan_topk(big, q, 20)
    parts_v, parts_i = [], []
    for r in range(3):
        v, i, _ = osc.scan_topk(big[r * 1000:(r + 1) * 1000], q, 20)
        parts_v.append(v)
        parts_i.append(i + r * 1000)
    mv, mi = osc.merge_topk(torch.cat(parts_v, 1), torch.cat(parts_i, 1), 20)
    assert torch.equal(mi, gi) and torch.allclose(mv, gv)


def test_full_depth_rank_fixture_matches_the_oracle_on_sampled_pairs(golden_dir):
    """tests/golden/rank_oracle_24L_4x50x292.npz (what the GPU suite's full-depth rerank gate compares the product with) is
    what oracle/encoder.py computes for the same seeded weights and token ids: two of its 200 pairs are re-derived here
    (a pair costs seconds on a few cores, all 200 cost minutes -- which is why they are a fixture)."""
    import hashlib

    import test_rank_agreement_gpu as t
    from rank_checks import weights_checksum

    z = np.load(os.path.join(golden_dir, t.GOLDEN_NAME))
    ocfg = oe.EncoderConfig(**t.SHAPE)
    W = oe.synth_weights(ocfg, seed=t.WEIGHT_SEED)
    pairs = t._pairs()
    assert str(z["pairs_sha256"]) == hashlib.sha256(pairs.tobytes()).hexdigest()
    assert str(z["weights_sha256"]) == weights_checksum(W)
    assert z["scores"].shape == (t.N_QUERIES, t.N_PAIRS)
    for q, j in ((0, 0), (3, 41)):
        ids = torch.from_numpy(pairs[q, j:j + 1])
        with torch.no_grad():
            got = oe.rerank_scores(ids, torch.ones_like(ids), W, ocfg)
        assert abs(float(got[0]) - float(z["scores"][q, j])) < 2e-5, (q, j, float(got[0]), float(z["scores"][q, j]))


def test_stress_fixtures_match_the_oracle_on_sampled_pairs(golden_dir):
    """The two stress fixtures (scores of 200 pairs; normalised first-row hidden states of 16) are what oracle/encoder.py computes
    on the weights rebuilt from the recipe (tests/stress_weights.py + the factors and head stored in the scores fixture): one
    pair's score and one pair's embedding are re-derived here."""
    import hashlib

    import stress_weights
    import test_rank_agreement_gpu as t
    from rank_checks import weights_checksum

    z = np.load(os.path.join(golden_dir, t.STRESS_GOLDEN_NAME))
    ze = np.load(os.path.join(golden_dir, t.STRESS_EMB_GOLDEN_NAME))
    ocfg = oe.EncoderConfig(**t.SHAPE)
    W = stress_weights.with_head(stress_weights.apply(oe.synth_weights(ocfg, seed=t.WEIGHT_SEED), ocfg, qk_scales=z["qk_scales"]),
                                 z["head_w"], z["head_b"])
    pairs = t._pairs()
    assert str(z["pairs_sha256"]) == hashlib.sha256(pairs.tobytes()).hexdigest()
    assert str(z["weights_sha256"]) == weights_checksum(W) == str(ze["weights_sha256"])
    ids16 = np.concatenate([pairs[q][:t.STRESS_EMB_PAIRS // t.N_QUERIES] for q in range(t.N_QUERIES)]).astype(np.int64)
    assert str(ze["ids_sha256"]) == hashlib.sha256(ids16.tobytes()).hexdigest() and ze["embeddings"].shape == (t.STRESS_EMB_PAIRS, ocfg.hidden)
    assert np.allclose(np.linalg.norm(ze["embeddings"], axis=1), 1.0, atol=1e-5)
    ids = torch.from_numpy(pairs[2, 7:8])
    with torch.no_grad():
        got = oe.rerank_scores(ids, torch.ones_like(ids), W, ocfg)
        emb = oe.embed(torch.from_numpy(ids16[5:6]), torch.ones(1, ids16.shape[1], dtype=torch.int64), W, ocfg)
    assert abs(float(got[0]) - float(z["scores"][2, 7])) < 1e-4 * max(1.0, 1.0), (float(got[0]), float(z["scores"][2, 7]))
    assert np.abs(emb[0].numpy() - ze["embeddings"][5]).max() < 2e-5
