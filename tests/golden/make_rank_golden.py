"""Generates tests/golden/rank_oracle_24L_4x50x292.npz: the fp32 CPU oracle's sigmoid scores for the full-depth rerank gate
(tests/test_rank_agreement_gpu.py) -- 4 queries x 50 pairs x 292 tokens through the 24-layer bge-reranker-v2-m3 shape with
the seeded HF-style random weights of ``oracle.encoder.synth_weights``.  The oracle forward costs minutes of host time;
the GPU suite reads this fixture instead of recomputing it (and a CPU test re-derives one pair to keep it honest).

    python tests/golden/make_rank_golden.py
    python tests/golden/make_rank_golden.py --stress     # -> rank_oracle_stress_24L_4x50x292.npz (tests/stress_weights.py)

--stress: the same token ids through weights with trained-model statistics (outlier LayerNorm gains, peaked attention on
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


def stress_forward(ids, W, ocfg, stats):
    """oracle.encoder.encoder_forward (no padding: full-length pairs) that also records attention entropies and the largest
    activations -- the same arithmetic, restated so the statistics can be read off."""
    import math

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
        q = (x @ f(p + "attention.self.query.weight").T + f(p + "attention.self.query.bias")).view(B, L, nh, dh).transpose(1, 2)
        k = (x @ f(p + "attention.self.key.weight").T + f(p + "attention.self.key.bias")).view(B, L, nh, dh).transpose(1, 2)
        v = (x @ f(p + "attention.self.value.weight").T + f(p + "attention.self.value.bias")).view(B, L, nh, dh).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
        pr = torch.softmax(s, dim=-1)
        if i in (0, ocfg.layers // 2, ocfg.layers - 1):
            ent = -(pr * torch.log2(pr.clamp_min(1e-30))).sum(-1).mean(dim=(0, 2))         # bits, per head
            stats.setdefault("entropy_even_heads", []).append(float(ent[0::2].mean()))
            stats.setdefault("entropy_odd_heads", []).append(float(ent[1::2].mean()))
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
    return x[:, 0, :]


def main_stress():
    import stress_weights

    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 64)))
    ocfg = oe.EncoderConfig(**t.SHAPE)
    W = stress_weights.apply(oe.synth_weights(ocfg, seed=t.WEIGHT_SEED), ocfg)
    pairs = t._pairs()
    stats = {}
    with torch.no_grad():
        cls = torch.cat([stress_forward(torch.from_numpy(pairs[q]), W, ocfg, stats) for q in range(t.N_QUERIES)])
        feat = torch.tanh(cls @ W["classifier.dense.weight"].T + W["classifier.dense.bias"])            # [200][H]
        # the head: the direction along which the candidates of a query differ most, scaled so that the logits span +-3
        centred = torch.cat([feat[q * t.N_PAIRS:(q + 1) * t.N_PAIRS] - feat[q * t.N_PAIRS:(q + 1) * t.N_PAIRS].mean(0, keepdim=True)
                             for q in range(t.N_QUERIES)])
        _, _, vh = torch.linalg.svd(centred.double(), full_matrices=False)
        w = vh[0].float()
        proj = feat @ w
        w = w * (6.0 / float(proj.max() - proj.min()))
        b = -float((feat @ w).median())
        head_w, head_b = w.reshape(1, -1), torch.tensor([b])
        Wh = stress_weights.with_head(W, head_w, head_b)
        want = torch.sigmoid(feat @ Wh["classifier.out_proj.weight"].T + Wh["classifier.out_proj.bias"])[:, 0].view(t.N_QUERIES, t.N_PAIRS)
        # the restated forward IS the oracle's: one query re-scored through oracle.encoder
        ids0 = torch.from_numpy(pairs[0])
        chk = oe.rerank_scores(ids0, torch.ones_like(ids0), Wh, ocfg)
        assert (chk - want[0]).abs().max().item() < 1e-5, (chk - want[0]).abs().max().item()
    for q in range(t.N_QUERIES):
        print(f"query {q}: scores {want[q].min().item():.4f} .. {want[q].max().item():.4f}", flush=True)
    print("stress statistics:", stats)
    np.savez(os.path.join(HERE, t.STRESS_GOLDEN_NAME), scores=want.numpy().astype(np.float32), head_w=head_w.numpy(), head_b=head_b.numpy(),
             pairs_sha256=hashlib.sha256(pairs.tobytes()).hexdigest(), weights_sha256=weights_checksum(Wh),
             stats=np.array(repr(stats)), torch_version=torch.__version__)
    print("wrote", t.STRESS_GOLDEN_NAME)


def main():
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
