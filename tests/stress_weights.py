"""The STRESS FIXTURE of the precision gates: seeded weights made hostile the way trained checkpoints are hostile.
This is the builder's own construction -- outlier dimensions, peaked attention, a head that spreads the scores -- not a trained model
and not fitted to one; no network, no real checkpoint exists here.

``oracle.encoder.synth_weights`` is HF's init (normal(0, 0.02)): no outlier features, near-uniform attention, and a
classification head that squeezes a query's 50 candidates into ~0.3 of the sigmoid range -- the easy case for every
reduced-precision mode.  Trained XLM-R encoders differ in exactly the ways that hurt low precision:

  * OUTLIER FEATURES: a handful of hidden dimensions carrying massive activations, 20-60x the others', through every layer.
    Here they are the LayerNorm's OFFSET in six dimensions (gain 0, beta = +-20 ... +-60), i.e. near-constant across tokens, as
    the massive activations of trained transformers are.  (A 20-60x LayerNorm GAIN on a dimension -- the first construction
    tried -- does not survive a post-LN residual stream: gamma_d > 1 on a dimension that feeds the next LayerNorm is amplified
    at every layer until the six dimensions own the normalisation; at depth the ordinary dimensions had shrunk to 1e-3, the
    outliers stood at 2300, and sharpening the attention took logit factors of 4096.  Offsets alone do the same more slowly:
    they keep every pre-LN variance at ~10, so the ordinary dimensions lose a factor 3 per LayerNorm.  Trained models hold
    the balance with their ORDINARY gains; so does this construction: the ordinary dimensions' gamma is multiplied by
    c = sqrt(1 + sum(beta_d^2) / H) ~ 3.15, which makes "ordinary features of unit scale, six offsets of 20-60" the fixed
    point of the residual stream at every depth.)
  * PEAKED ATTENTION: heads whose softmax concentrates on a few keys (entropy < 2 bits of the ~8.2 bits 292 keys allow)
    -- logits of tens, where an operand rounding of 2^-9 moves a probability by percents;
  * a HEAD THAT DECIDES: scores spread over (0.05, 0.95), not bunched around one value.

``apply`` builds the outlier gains from the seeded init (deterministic, no data).  The attention sharpening and the head are
CALIBRATED on the fixture's own pairs by the golden generator (``tests/golden/make_rank_golden.py --stress``): per layer, the
factor on the even heads' query / key projections that brings their mean attention entropy to 1.5 bits (a fixed factor
sharpens the first layer only: the outlier dimensions take over the normalisation with depth and the ordinary dimensions'
logits fade); the head = the direction along which the candidates of a query differ most, scaled so that a query's
candidates span ~6 logits.  Both travel inside the fixture (24 + 1025 floats): ``apply(..., qk_scales=)``, ``with_head``.
There is no network for real checkpoints; this is the offline stand-in for them.
"""
from __future__ import annotations

import torch

OUTLIER_DIMS = (7, 133, 402, 588, 771, 1009)          # six of the 1024 hidden dimensions
OUTLIER_GAINS = (60.0, -45.0, 30.0, -25.0, 20.0, -40.0)  # the LayerNorm output in those dimensions (gamma 0, beta = this)
ENTROPY_TARGET_BITS = 1.5                               # mean attention entropy of the even heads after calibration


def scale_qk(W: dict, cfg, layer: int, alpha: float) -> None:
    """In place: the logits of every EVEN head of ``layer`` x alpha (sqrt(alpha) on W_q, b_q and on W_k, b_k)."""
    H, dh = cfg.hidden, cfg.hidden // cfg.heads
    even = torch.zeros(H, dtype=torch.bool)
    for h in range(0, cfg.heads, 2):
        even[h * dh:(h + 1) * dh] = True
    p = f"encoder.layer.{layer}.attention.self."
    r = float(alpha) ** 0.5
    for nm in ("query", "key"):
        W[p + nm + ".weight"][even] *= r
        W[p + nm + ".bias"][even] *= r


def apply(W: dict, cfg, seed: int = 0, qk_scales=None) -> dict:
    """-> a new weight dict (fp32) with outlier LayerNorm gains (and, with ``qk_scales`` [layers], the calibrated attention
    sharpening of the even heads); the classifier is untouched."""
    out = {k: v.clone().to(torch.float32) for k, v in W.items()}
    H, dh = cfg.hidden, cfg.hidden // cfg.heads
    dims = [d for d in OUTLIER_DIMS if d < H]
    ln_names = ["embeddings.LayerNorm.weight"]
    for i in range(cfg.layers):
        ln_names += [f"encoder.layer.{i}.attention.output.LayerNorm.weight", f"encoder.layer.{i}.output.LayerNorm.weight"]
    c = (1.0 + sum(g * g for _, g in zip(dims, OUTLIER_GAINS)) / H) ** 0.5
    for name in ln_names:
        out[name] *= c
        for d, gain in zip(dims, OUTLIER_GAINS):
            out[name][d] = 0.0
            out[name.replace(".weight", ".bias")][d] = gain
    for i in range(cfg.layers):
        # the outlier dimensions' columns of the projections are damped, as in trained models (the consumers of a massive
        # activation carry small weights for it): otherwise six dimensions would BE the layer's output
        for nm in ("attention.self.query", "attention.self.key", "attention.self.value", "intermediate.dense"):
            w = out[f"encoder.layer.{i}.{nm}.weight"]
            for d, gain in zip(dims, OUTLIER_GAINS):
                w[:, d] *= 4.0 / abs(gain)
    if qk_scales is not None:
        for i in range(cfg.layers):
            scale_qk(out, cfg, i, float(qk_scales[i]))
    return out


def with_head(W: dict, head_w, head_b) -> dict:
    out = dict(W)
    out["classifier.out_proj.weight"] = torch.as_tensor(head_w, dtype=torch.float32).reshape(1, -1).clone()
    out["classifier.out_proj.bias"] = torch.as_tensor(head_b, dtype=torch.float32).reshape(1).clone()
    return out
