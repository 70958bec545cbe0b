"""Round 6: LayerNorm FOLDED into the neighbouring GEMMs -- CPU emulation of the bf16 path's rounding points
with and without it, before anything is built.

Today (post-LN block, bf16 mode; oracle.encoder.encoder_forward(emulate_bf16=True)):
    y  = bf16(GEMM(.) + b + x)              residual epilogue
    x' = bf16(LayerNorm(y))                 row kernel: reads y, writes x'      <- the pass folding removes
    consumers: GEMM(x', bf16(W)) ; the next residual epilogue adds x'
Folded:
    y  = bf16(GEMM(.) + b + LN_prev(y_prev))   the residual epilogue rebuilds LayerNorm(y_prev) in fp32 from raw y_prev + (mu, rstd)
    (mu, rstd) of y from the epilogue's own bf16-rounded values (per-row partials per column tile, combined by the consumer)
    consumers: acc = GEMM(y, bf16(gamma * W));  out = rstd * acc - rstd * mu * colsum(bf16(gamma * W)) + (b + W beta)
i.e. ONE rounding fewer per LayerNorm (x' is never rounded to bf16), the gain folded into the weights before they are rounded,
the offset folded into the bias in fp32.  The embedding LayerNorm and the last layer's output LayerNorm stay real.

Usage: python tools/probes/ln_folding_emulation.py [--layers N] [--stress] [--small]
"""
import argparse
import math
import os
import sys
import time

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
import torch

from oracle import encoder as oe


def bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def forward(ids, mask, W, cfg, fold, wcache):
    f = lambda name: W[name].to(torch.float32)  # noqa: E731
    ids = ids.to(torch.int64)
    B, L = ids.shape
    H, nh, dh = cfg.hidden, cfg.heads, cfg.head_dim
    pos = oe.position_ids(mask, cfg)
    x = f("embeddings.word_embeddings.weight")[ids] + f("embeddings.position_embeddings.weight")[pos] + \
        f("embeddings.token_type_embeddings.weight")[torch.zeros_like(ids)]
    x = bf16(oe.layer_norm(x, f("embeddings.LayerNorm.weight"), f("embeddings.LayerNorm.bias"), cfg.ln_eps))
    neg = torch.zeros(B, 1, 1, L, dtype=torch.float32)
    neg.masked_fill_(mask.to(torch.bool).logical_not().view(B, 1, 1, L), float("-inf"))
    scale = 1.0 / math.sqrt(dh)

    # `state`: either a materialised activation ("x", tensor) or a pending LayerNorm ("ln", raw y (bf16 values), mu, rstd, gamma, beta)
    state = ("x", x)

    def ln_value(st):              # what the residual epilogue adds: the activation itself, or LayerNorm(y) rebuilt in fp32
        if st[0] == "x":
            return st[1]
        _, y, mu, rstd, g, b = st
        return (y - mu) * rstd * g + b

    def lin(st, wn, bn):
        if st[0] == "x":
            key = ("w", wn)
            if key not in wcache:
                wcache[key] = bf16(f(wn))
            return st[1] @ wcache[key].T + f(bn)
        _, y, mu, rstd, g, b = st
        key = ("wf", wn, id(g))
        if key not in wcache:
            wf = bf16(f(wn) * g[None, :])                       # gamma folded into the weight BEFORE the bf16 rounding
            wcache[key] = (wf, wf.sum(dim=1), f(bn) + bf16(f(wn)) @ b)      # colsum of the rounded folded weight; beta . W^T + bias (fp32)
        wf, cs, b2 = wcache[key]
        acc = y @ wf.T
        return rstd * acc - (rstd * mu) * cs + b2

    def pending(y, gname, bname):
        """y: the bf16-rounded pre-LayerNorm sum.  Statistics from those rounded values, fp32 (what the epilogue has in registers)."""
        mu = y.mean(dim=-1, keepdim=True)
        var = ((y - mu) ** 2).mean(dim=-1, keepdim=True)
        return ("ln", y, mu, 1.0 / torch.sqrt(var + cfg.ln_eps), f(gname), f(bname))

    for i in range(cfg.layers):
        p = f"encoder.layer.{i}."
        q = bf16(lin(state, p + "attention.self.query.weight", p + "attention.self.query.bias"))
        k = bf16(lin(state, p + "attention.self.key.weight", p + "attention.self.key.bias"))
        v = bf16(lin(state, p + "attention.self.value.weight", p + "attention.self.value.bias"))
        q = q.view(B, L, nh, dh).transpose(1, 2)
        k = k.view(B, L, nh, dh).transpose(1, 2)
        v = v.view(B, L, nh, dh).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) * scale + neg
        m = s.max(dim=-1, keepdim=True).values
        e = torch.exp(s - m)
        denom = e.sum(dim=-1, keepdim=True)
        ctx = (bf16(e) @ v) / denom
        ctx = bf16(ctx.transpose(1, 2).reshape(B, L, H))
        a = ctx @ wcache.setdefault(("w", p + "o"), bf16(f(p + "attention.output.dense.weight"))).T + f(p + "attention.output.dense.bias")
        y = bf16(a + ln_value(state))
        if fold:
            state = pending(y, p + "attention.output.LayerNorm.weight", p + "attention.output.LayerNorm.bias")
        else:
            state = ("x", bf16(oe.layer_norm(y, f(p + "attention.output.LayerNorm.weight"), f(p + "attention.output.LayerNorm.bias"), cfg.ln_eps)))
        h = bf16(oe.gelu_erf(lin(state, p + "intermediate.dense.weight", p + "intermediate.dense.bias")))
        o = h @ wcache.setdefault(("w", p + "d"), bf16(f(p + "output.dense.weight"))).T + f(p + "output.dense.bias")
        y = bf16(o + ln_value(state))
        if fold and i + 1 < cfg.layers:
            state = pending(y, p + "output.LayerNorm.weight", p + "output.LayerNorm.bias")
        else:
            state = ("x", bf16(oe.layer_norm(y, f(p + "output.LayerNorm.weight"), f(p + "output.LayerNorm.bias"), cfg.ln_eps)))
    h = state[1][:, 0, :]
    t = torch.tanh(h @ bf16(f("classifier.dense.weight")).T + f("classifier.dense.bias"))
    return torch.sigmoid((t @ bf16(f("classifier.out_proj.weight")).T + f("classifier.out_proj.bias"))[:, 0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--stress", action="store_true")
    args = ap.parse_args()
    torch.set_num_threads(8)
    from rank_checks import kendall_tau, topn_overlap
    import test_rank_agreement_gpu as t

    shape = dict(t.SHAPE)
    shape["layers"] = args.layers
    cfg = oe.EncoderConfig(**shape)
    W = oe.synth_weights(cfg, seed=t.WEIGHT_SEED)
    pairs = t._pairs()
    nq = 1 if args.small else t.N_QUERIES
    groups = [(torch.from_numpy(pairs[q]), torch.ones_like(torch.from_numpy(pairs[q]))) for q in range(nq)]
    ref = None
    if args.layers == 24 and not args.stress:
        ref = torch.from_numpy(np.load(os.path.join("tests", "golden", t.GOLDEN_NAME))["scores"])[:nq]
    if args.stress:
        import stress_weights

        z = np.load(os.path.join("tests", "golden", t.STRESS_GOLDEN_NAME))
        W = stress_weights.with_head(stress_weights.apply(W, cfg, qk_scales=z["qk_scales"]), z["head_w"], z["head_b"])
        if args.layers == 24:
            ref = torch.from_numpy(z["scores"].astype(np.float32))[:nq]
    with torch.no_grad():
        if ref is None:
            ref = torch.stack([oe.rerank_scores(i, m, W, cfg) for i, m in groups])
        print(f"fixture: {'stress' if args.stress else 'standard'}, {args.layers} layers, {nq} x {groups[0][0].shape[0]} pairs x {groups[0][0].shape[1]} tokens; "
              f"fp32 oracle scores {ref.min():.4f} .. {ref.max():.4f}", flush=True)
        for name, fold in (("bf16 (today)", False), ("bf16 + LN folded", True)):
            t0 = time.time()
            wcache = {}
            got = torch.stack([forward(i, m, W, cfg, fold, wcache) for i, m in groups])
            err = (got - ref).abs()
            rel = err / ref.abs().clamp_min(1e-6)
            taus = [kendall_tau(ref[g].numpy(), got[g].numpy()) for g in range(len(groups))]
            ov = [topn_overlap(ref[g].numpy(), got[g].numpy(), 10) for g in range(len(groups))]
            print(f"{name:>18}: |err| mean {err.mean():.2e} max {err.max():.2e}  relative mean {rel.mean():.2e} max {rel.max():.2e}  "
                  f"tau {min(taus):.4f}..{max(taus):.4f}  top-10 overlap {min(ov):.1f}..{max(ov):.1f}  ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
