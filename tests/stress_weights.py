"""Seeded weights with TRAINED-MODEL STATISTICS for the precision gates.

``oracle.encoder.synth_weights`` is HF's init (normal(0, 0.02)): no outlier features, near-uniform attention, and a
classification head that squeezes a query's 50 candidates into ~0.3 of the sigmoid range -- the easy case for every
reduced-precision mode.  Trained XLM-R encoders differ in exactly the ways that hurt low precision:

  * OUTLIER FEATURES: a handful of hidden dimensions whose LayerNorm gain is 20-60x the others' (massive activations that
    ride the residual stream through every layer);
  * PEAKED ATTENTION: heads whose softmax concentrates on a few keys (entropy < 2 bits of the ~8.2 bits 292 keys allow)
    -- logits of tens, where an operand rounding of 2^-9 moves a probability by percents;
  * a HEAD THAT DECIDES: scores spread over (0.05, 0.95), not bunched around one value.

``apply`` builds the first two from the seeded init (deterministic, no data); the head is CALIBRATED on the fixture's own
pairs by the golden generator (``tests/golden/make_rank_golden.py --stress``: first principal direction of the pre-head
features, scaled so the logits span +-3) and travels inside the fixture -- ``with_head`` installs it.
There is no network for real checkpoints; this is the offline stand-in for them.
"""
from __future__ import annotations

import torch

OUTLIER_DIMS = (7, 133, 402, 588, 771, 1009)          # six of the 1024 hidden dimensions
OUTLIER_GAINS = (60.0, 45.0, 30.0, 25.0, 20.0, 40.0)  # LayerNorm gamma multipliers
QK_SCALE = 3.5                                          # on W_q, b_q, W_k, b_k of every EVEN head: logits x 12


def apply(W: dict, cfg, seed: int = 0) -> dict:
    """-> a new weight dict (fp32) with outlier LayerNorm gains and peaked-attention heads; the classifier is untouched."""
    out = {k: v.clone().to(torch.float32) for k, v in W.items()}
    H, dh = cfg.hidden, cfg.hidden // cfg.heads
    dims = [d for d in OUTLIER_DIMS if d < H]
    ln_names = ["embeddings.LayerNorm.weight"]
    for i in range(cfg.layers):
        ln_names += [f"encoder.layer.{i}.attention.output.LayerNorm.weight", f"encoder.layer.{i}.output.LayerNorm.weight"]
    for name in ln_names:
        for d, gain in zip(dims, OUTLIER_GAINS):
            out[name][d] *= gain
    even = torch.zeros(H, dtype=torch.bool)
    for h in range(0, cfg.heads, 2):
        even[h * dh:(h + 1) * dh] = True
    for i in range(cfg.layers):
        p = f"encoder.layer.{i}.attention.self."
        for nm in ("query", "key"):
            out[p + nm + ".weight"][even] *= QK_SCALE
            out[p + nm + ".bias"][even] *= QK_SCALE
        # the outlier dimensions' columns of the projections are damped, as in trained models (the consumers of a massive
        # activation carry small weights for it): otherwise six dimensions would BE the layer's output
        for nm in ("attention.self.query", "attention.self.key", "attention.self.value", "intermediate.dense"):
            w = out[f"encoder.layer.{i}.{nm}.weight"]
            for d, gain in zip(dims, OUTLIER_GAINS):
                w[:, d] *= 4.0 / gain
    return out


def with_head(W: dict, head_w, head_b) -> dict:
    out = dict(W)
    out["classifier.out_proj.weight"] = torch.as_tensor(head_w, dtype=torch.float32).reshape(1, -1).clone()
    out["classifier.out_proj.bias"] = torch.as_tensor(head_b, dtype=torch.float32).reshape(1).clone()
    return out
