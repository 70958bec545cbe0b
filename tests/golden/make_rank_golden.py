"""Generates tests/golden/rank_oracle_24L_4x50x292.npz: the fp32 CPU oracle's sigmoid scores for the full-depth rerank gate
(tests/test_rank_agreement_gpu.py) -- 4 queries x 50 pairs x 292 tokens through the 24-layer bge-reranker-v2-m3 shape with
the seeded HF-style random weights of ``oracle.encoder.synth_weights``.  The oracle forward costs minutes of host time;
the GPU suite reads this fixture instead of recomputing it (and a CPU test re-derives one pair to keep it honest).

    python tests/golden/make_rank_golden.py
    python tests/golden/make_rank_golden.py --stress     # -> rank_oracle_stress_24L_4x50x292.npz (tests/stress_weights.py)

--stress: the same token ids through the stress fixture's hostile weights (outlier LayerNorm gains, peaked attention on
half the heads) and a head CALIBRATED on these pairs: the first principal direction of the pre-head features tanh(dense(h)),
scaled so that the 200 logits span +-3 (scores 0.05 .. 0.95).  The fixture carries the head (4 KiB), the fp32 oracle's
scores, and what the generator measured about the stress it applies (attention entropy per head class, largest activation).
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))                  # tests/
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))  # repo root

from oracle import encoder as oe  # noqa: E402
import test_rank_agreement_gpu as t  # noqa: E402
from rank_checks import weights_checksum  # noqa: E402


def _attention_probs(x, W, ocfg, layer):
    import math

    B, L, H = x.shape
    nh, dh = ocfg.heads, ocfg.head_dim
    p = f"encoder.layer.{layer}."
    q = (x @ W[p + "attention.self.query.weight"].T + W[p + "attention.self.query.bias"]).view(B, L, nh, dh).transpose(1, 2)
    k = (x @ W[p + "attention.self.key.weight"].T + W[p + "attention.self.key.bias"]).view(B, L, nh, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    return s, torch.softmax(s, dim=-1)


def _entropy_bits(pr):
    return -(pr * torch.log2(pr.clamp_min(1e-30))).sum(-1).mean(dim=(0, 2))           # per head


def stress_forward(ids, W, ocfg, stats, calibrate=None):
    """oracle.encoder.encoder_forward (no padding: full-length pairs) that also records attention entropies and the largest
    activations -- the same arithmetic, restated so the statistics can be read off.  ``calibrate`` (a list to fill): before a
    layer's attention, find the factor on its even heads' logits that brings their mean entropy to the target (bisection on
    the actual softmax), apply it to W in place and go on with the scaled weights."""
    import stress_weights

    f = lambda n: W[n]  # noqa: E731
    B, L = ids.shape
    H, nh, dh = ocfg.hidden, ocfg.heads, ocfg.head_dim
    mask = torch.ones_like(ids)
    pos = oe.position_ids(mask, ocfg)
    x = f("embeddings.word_embeddings.weight")[ids] + f("embeddings.position_embeddings.weight")[pos] + \
        f("embeddings.token_type_embeddings.weight")[torch.zeros_like(ids)]
    x = oe.layer_norm(x, f("embeddings.LayerNorm.weight"), f("embeddings.LayerNorm.bias"), ocfg.ln_eps)
    for i in range(ocfg.layers):
        p = f"encoder.layer.{i}."
        if calibrate is not None:
            s0, _ = _attention_probs(x, W, ocfg, i)
            lo, hi = 1.0, 4096.0
            for _ in range(18):
                mid = (lo * hi) ** 0.5
                ent = float(_entropy_bits(torch.softmax(s0[:, 0::2] * mid, dim=-1)).mean())
                lo, hi = (mid, hi) if ent > stress_weights.ENTROPY_TARGET_BITS else (lo, mid)
            alpha = (lo * hi) ** 0.5
            stress_weights.scale_qk(W, ocfg, i, alpha)
            calibrate.append(alpha)
        s, pr = _attention_probs(x, W, ocfg, i)
        v = (x @ f(p + "attention.self.value.weight").T + f(p + "attention.self.value.bias")).view(B, L, nh, dh).transpose(1, 2)
        ent = _entropy_bits(pr)
        stats.setdefault("entropy_even_heads_by_layer", {}).setdefault(i, []).append(round(float(ent[0::2].mean()), 3))
        stats.setdefault("entropy_odd_heads_by_layer", {}).setdefault(i, []).append(round(float(ent[1::2].mean()), 3))
        stats["max_logit"] = max(stats.get("max_logit", 0.0), float(s.abs().max()))
        ctx = (pr @ v).transpose(1, 2).reshape(B, L, H)
        a = ctx @ f(p + "attention.output.dense.weight").T + f(p + "attention.output.dense.bias")
        x = oe.layer_norm(a + x, f(p + "attention.output.LayerNorm.weight"), f(p + "attention.output.LayerNorm.bias"), ocfg.ln_eps)
        h = oe.gelu_erf(x @ f(p + "intermediate.dense.weight").T + f(p + "intermediate.dense.bias"))
        stats["max_ffn_act"] = max(stats.get("max_ffn_act", 0.0), float(h.abs().max()))
        o = h @ f(p + "output.dense.weight").T + f(p + "output.dense.bias")
        stats["max_pre_ln"] = max(stats.get("max_pre_ln", 0.0), float((o + x).abs().max()))
        x = oe.layer_norm(o + x, f(p + "output.LayerNorm.weight"), f(p + "output.LayerNorm.bias"), ocfg.ln_eps)
        stats["max_ln_out"] = max(stats.get("max_ln_out", 0.0), float(x.abs().max()))
        stats.setdefault("median_abs_ln_out_by_layer", {})[i] = round(float(x.abs().median()), 4)
    return x[:, 0, :]


def main_stress():
    import stress_weights

    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 64)))
    ocfg = oe.EncoderConfig(**t.SHAPE)
    W = stress_weights.apply(oe.synth_weights(ocfg, seed=t.WEIGHT_SEED), ocfg)
    pairs = t._pairs()
    stats = {}
    with torch.no_grad():
        # 1. attention sharpening, calibrated layer by layer on 8 pairs (two of every query)
        qk_scales = []
        cal_ids = torch.from_numpy(np.concatenate([pairs[q][:2] for q in range(t.N_QUERIES)]))
        stress_forward(cal_ids, W, ocfg, {}, calibrate=qk_scales)              # scales W in place
        print("calibrated logit factors of the even heads:", [round(a, 1) for a in qk_scales], flush=True)
        # 2. the oracle forward of all pairs with those weights
        cls = torch.cat([stress_forward(torch.from_numpy(pairs[q]), W, ocfg, stats) for q in range(t.N_QUERIES)])
        feat = torch.tanh(cls @ W["classifier.dense.weight"].T + W["classifier.dense.bias"])            # [200][H]
        # 3. the head: the direction along which the candidates of a query differ most (every query weighted alike), scaled so
        #    that a query's candidates span ~6 logits between their 5th and 95th percentile
        per_q = [feat[q * t.N_PAIRS:(q + 1) * t.N_PAIRS] for q in range(t.N_QUERIES)]
        centred = torch.cat([(fq - fq.mean(0, keepdim=True)) / (fq - fq.mean(0, keepdim=True)).norm() for fq in per_q])
        _, _, vh = torch.linalg.svd(centred.double(), full_matrices=False)
        w = vh[0].float()
        spans = [float(torch.quantile(fq @ w, 0.95) - torch.quantile(fq @ w, 0.05)) for fq in per_q]
        w = w * (6.0 / float(np.median(spans)))
        b = -float((feat @ w).median())
        head_w, head_b = w.reshape(1, -1), torch.tensor([b])
        Wh = stress_weights.with_head(W, head_w, head_b)
        want = torch.sigmoid(feat @ Wh["classifier.out_proj.weight"].T + Wh["classifier.out_proj.bias"])[:, 0].view(t.N_QUERIES, t.N_PAIRS)
        # the restated forward IS the oracle's: one query re-scored through oracle.encoder on weights rebuilt from the recipe
        W2 = stress_weights.with_head(stress_weights.apply(oe.synth_weights(ocfg, seed=t.WEIGHT_SEED), ocfg, qk_scales=qk_scales), head_w, head_b)
        ids0 = torch.from_numpy(pairs[0])
        chk = oe.rerank_scores(ids0, torch.ones_like(ids0), W2, ocfg)
        assert (chk - want[0]).abs().max().item() < 1e-4, (chk - want[0]).abs().max().item()
        Wh = W2
    for q in range(t.N_QUERIES):
        print(f"query {q}: scores {want[q].min().item():.4f} .. {want[q].max().item():.4f}", flush=True)
    stats = {k: ({kk: (round(float(np.mean(vv)), 3) if isinstance(vv, list) else vv) for kk, vv in v.items()} if isinstance(v, dict) else v)
             for k, v in stats.items()}
    print("stress statistics:", stats)
    np.savez(os.path.join(HERE, t.STRESS_GOLDEN_NAME), scores=want.numpy().astype(np.float32), head_w=head_w.numpy(), head_b=head_b.numpy(),
             qk_scales=np.asarray(qk_scales, dtype=np.float64), pairs_sha256=hashlib.sha256(pairs.tobytes()).hexdigest(),
             weights_sha256=weights_checksum(Wh), stats=np.array(repr(stats)), torch_version=torch.__version__)
    print("wrote", t.STRESS_GOLDEN_NAME)


def main_stress_embeddings():
    """The bi-encoder side of the stress fixture: L2-normalised first-row (CLS) hidden states of 16 pairs under the COMMITTED stress
    weights (factors and head read from the scores fixture, nothing re-calibrated) -> tests/golden/<STRESS_EMB_GOLDEN_NAME>."""
    import stress_weights

    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 64)))
    ocfg = oe.EncoderConfig(**t.SHAPE)
    z = np.load(os.path.join(HERE, t.STRESS_GOLDEN_NAME))
    W = stress_weights.with_head(stress_weights.apply(oe.synth_weights(ocfg, seed=t.WEIGHT_SEED), ocfg, qk_scales=z["qk_scales"]),
                                 z["head_w"], z["head_b"])
    assert str(z["weights_sha256"]) == weights_checksum(W)
    pairs = t._pairs()
    ids = torch.from_numpy(np.concatenate([pairs[q][:t.STRESS_EMB_PAIRS // t.N_QUERIES] for q in range(t.N_QUERIES)]))
    with torch.no_grad():
        cls = stress_forward(ids, W, ocfg, {})
        emb = torch.nn.functional.normalize(cls.double(), dim=1).float()
        # the restated forward IS the oracle's
        chk = oe.embed(ids[:2], torch.ones_like(ids[:2]), W, ocfg)
        assert (chk - emb[:2]).abs().max().item() < 2e-5, (chk - emb[:2]).abs().max().item()
    np.savez(os.path.join(HERE, t.STRESS_EMB_GOLDEN_NAME), embeddings=emb.numpy().astype(np.float32),
             ids_sha256=hashlib.sha256(ids.numpy().tobytes()).hexdigest(), weights_sha256=weights_checksum(W), torch_version=torch.__version__)
    print("wrote", t.STRESS_EMB_GOLDEN_NAME, tuple(emb.shape), "component range", float(emb.min()), float(emb.max()))


def main():
    if "--stress-embeddings" in sys.argv:
        return main_stress_embeddings()
    if "--stress" in sys.argv:
        return main_stress()
    ocfg = oe.EncoderConfig(**t.SHAPE)
    W = oe.synth_weights(ocfg, seed=t.WEIGHT_SEED)
    pairs = t._pairs()
    want = torch.empty(t.N_QUERIES, t.N_PAIRS)
    with torch.no_grad():
        for q in range(t.N_QUERIES):
            ids = torch.from_numpy(pairs[q])
            want[q] = oe.rerank_scores(ids, torch.ones_like(ids), W, ocfg)
            print(f"query {q}: scores {want[q].min().item():.4f} .. {want[q].max().item():.4f}", flush=True)
    np.savez(os.path.join(HERE, t.GOLDEN_NAME), scores=want.numpy().astype(np.float32),
             pairs_sha256=hashlib.sha256(pairs.tobytes()).hexdigest(), weights_sha256=weights_checksum(W),
             torch_version=torch.__version__)
    print("wrote", t.GOLDEN_NAME)


if __name__ == "__main__":
    main()
