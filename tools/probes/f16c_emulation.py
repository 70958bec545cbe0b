"""Can the `reference` (fp32-semantics) mode run on TWO matrix-time units instead of split-bf16's three?  

Scheme "f16c" (fp16 + corrections).  A GEMM operand x is carried as
    hi  = fp16(x)                                   (11-bit significand)
    x8  = e4m3(x * 2^-s)                            block scale s = floor(log2(absmax of 32 consecutive K elements)) - 7
    lo8 = e4m3((x - hi) * 2^-(s - 11))              |x - hi| <= 2^(E-11): the same block exponent, shifted
and a product as
    a . w  ~=  a_hi . w_hi   (v_mfma_f32_16x16x32_f16, fp32 accumulate: 1 unit)
             + a8 . wlo8     (v_mfma_scale_f32_16x16x128_f8f6f4, block scales: 1/2 unit)
             + alo8 . w8     (1/2 unit)
The two cross terms are 2^-12 of the result and only need e4m3's 2^-4: the dropped lo.lo term is 2^-24.  Attention runs
on single fp16 products (Q, K, V, P rounded to fp16; fp32 scores, softmax, accumulators).  LayerNorm, GELU (exact erf),
the residual stream and the head stay fp32 -- as in the split-bf16 path.

CPU emulation with the oracle, full depth, against the committed fp32 fixture of tests/test_rank_agreement_gpu.py
(4 queries x 50 pairs x 292 tokens) or a smaller seeded case.  Variants are switched on the command line so that the
contribution of each approximation can be read off.

Usage: python tools/probes/f16c_emulation.py [--small] [--variants a,b,...] [--stress]
"""
import argparse
import math
import os
import re
import sys
import time

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
import torch

from oracle import encoder as oe

E4M3_MAX = 448.0


def f16(x):
    return x.clamp(-65504.0, 65504.0).to(torch.float16).to(torch.float32)


def bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def block_exp(x, block=32):
    """floor(log2(absmax)) of every block of `block` consecutive elements of the last axis (as fp32, -126 for a zero block)."""
    shp = x.shape
    a = x.reshape(*shp[:-1], shp[-1] // block, block).abs().amax(-1, keepdim=True)
    e = torch.floor(torch.log2(a.clamp_min(2.0 ** -126)))
    return e


def q8(x, e_blk, shift, block=32):
    """e4m3(x * 2^-(e_blk - 7 + shift)) back in fp32 at its true magnitude (saturating)."""
    shp = x.shape
    xb = x.reshape(*shp[:-1], shp[-1] // block, block)
    s = torch.exp2(e_blk - 7.0 + shift)
    q = (xb / s).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).to(torch.float32)
    return (q * s).reshape(shp)


class Split:
    """The three planes of an operand."""

    def __init__(self, x, hi_fn=f16, lo_shift=-11.0, block=32):
        self.hi = hi_fn(x)
        lo = x - self.hi
        e = block_exp(x, block)
        self.x8 = q8(x, e, 0.0, block)
        self.lo8 = q8(lo, e, lo_shift, block)
        self.lo = lo


OUTLIER_PERM = {}


def outlier_perm(K):
    """Round 5: a static permutation of a K = hidden axis that moves the stress fixture's six outlier dimensions
    (tests/stress_weights.py OUTLIER_DIMS) into ONE 32-wide block (block 0) -- an exact re-indexing of the hidden dimension."""
    if K not in OUTLIER_PERM:
        import stress_weights
        dims = [d for d in stress_weights.OUTLIER_DIMS if d < K]
        rest = [d for d in range(K) if d not in dims]
        OUTLIER_PERM[K] = torch.tensor(dims + rest)
    return OUTLIER_PERM[K]


def lin_f16c(x, w_split, b, variant, perm=False, side=False, wperm=None):
    """perm: both operands' K axis re-indexed so the outlier dimensions share block 0 (w_split is then the split of the permuted
    weight); side: block 0's two cross terms on exact fp16 lo planes (a <= 32-column fp16 side product) instead of e4m3."""
    if perm or side:
        x = x[..., outlier_perm(x.shape[-1])]
    xs = Split(x)
    if side:
        y = xs.hi @ w_split.hi.T
        y = y + xs.x8[..., 32:] @ w_split.lo8[:, 32:].T + xs.lo8[..., 32:] @ w_split.x8[:, 32:].T
        y = y + xs.hi[..., :32] @ f16(w_split.lo[:, :32]).T + f16(xs.lo[..., :32]) @ w_split.hi[:, :32].T
        return y + b
    y = xs.hi @ w_split.hi.T
    if variant == "f16_only":
        return y + b
    if variant == "f16_exact_corr":        # what the corrections would give with exact cross terms (= fp16 x3)
        return y + xs.hi @ w_split.lo.T + xs.lo @ w_split.hi.T + b
    # cross terms on e4m3 operands
    y = y + xs.x8 @ w_split.lo8.T + xs.lo8 @ w_split.x8.T
    return y + b


def forward(ids, mask, W, cfg, variant, wcache):
    """oracle.encoder.encoder_forward with the projections (and attention) of the scheme under test."""
    f = lambda name: W[name].to(torch.float32)  # noqa: E731

    do_perm = "+perm" in variant or "+side" in variant
    do_side = "+side" in variant

    def wsplit(name):
        if name not in wcache:
            w = f(name)
            if do_perm and w.shape[1] == cfg.hidden:
                w = w[:, outlier_perm(cfg.hidden)]
            wcache[name] = Split(w)
        return wcache[name]

    # "mix=<families>": the listed projection families (qk, v, o, up, down, joined by '.') run f16c, the others f16x3
    mix = None
    mm = re.search(r"mix=([a-z.]*)", variant)
    if mm:
        mix = set(x for x in mm.group(1).split(".") if x)

    def family(wn):
        if ".query." in wn or ".key." in wn:
            return "qk"
        if ".value." in wn:
            return "v"
        if "attention.output.dense" in wn:
            return "o"
        if "intermediate.dense" in wn:
            return "up"
        return "down"

    if variant == "fp32":
        lin = lambda x, wn, bn: x @ f(wn).T + f(bn)  # noqa: E731
    elif variant == "bf16x3":
        def lin(x, wn, bn):
            if wn not in wcache:
                w = f(wn); wh = bf16(w); wcache[wn] = (wh, bf16(w - wh))
            wh, wl = wcache[wn]
            xh = bf16(x); xl = bf16(x - xh)
            return xh @ wh.T + xh @ wl.T + xl @ wh.T + f(bn)
    elif variant.startswith("f16x3"):
        # fp16 hi + fp16 lo planes, three fp16 products per product (split-bf16's scheme on fp16 planes: 22 significand bits)
        def lin(x, wn, bn):
            if wn not in wcache:
                w = f(wn); wh = f16(w); wcache[wn] = (wh, f16(w - wh))
            wh, wl = wcache[wn]
            xh = f16(x); xl = f16(x - xh)
            return xh @ wh.T + xh @ wl.T + xl @ wh.T + f(bn)
    else:
        def lin_h3(x, wn, bn):
            if ("h3", wn) not in wcache:
                w = f(wn); wh = f16(w); wcache[("h3", wn)] = (wh, f16(w - wh))
            wh, wl = wcache[("h3", wn)]
            xh = f16(x); xl = f16(x - xh)
            return xh @ wh.T + xh @ wl.T + xl @ wh.T + f(bn)

        def lin(x, wn, bn):
            # "+qkh3": the query / key projections on three fp16 products (their outputs ARE the logits' operands), the rest f16c
            if "+qkh3" in variant and (".query." in wn or ".key." in wn):
                return lin_h3(x, wn, bn)
            if mix is not None and family(wn) not in mix:
                return lin_h3(x, wn, bn)
            hidden_k = W[wn].shape[1] == cfg.hidden
            return lin_f16c(x, wsplit(wn), f(bn), "f16c" if mix is not None else variant.split("+")[0],
                            perm=do_perm and hidden_k, side=do_side and hidden_k)
    att16 = variant not in ("fp32", "bf16x3") and "+att32" not in variant
    ra = f16 if att16 else (lambda t: t)
    if variant.startswith("f16x3") or "+att3" in variant:
        ra = lambda t: f16(t) + f16(t - f16(t))  # noqa: E731   (hi + lo planes everywhere in the attention)
    # "+qk32": Q and K at full precision in the score product (what hi + lo planes for Q / K would give), V and P still fp16
    rqk = (lambda t: t) if "+qk32" in variant else ra
    # "+qkb16": Q and K as TWO bf16 planes (the split-bf16 kernel's score product: 16 significand bits)
    if "+qkb16" in variant:
        rqk = lambda t: bf16(t) + bf16(t - bf16(t))  # noqa: E731
    # "+respl": the residual branch reads hi + lo8 of the LayerNorm output's c-planes instead of an fp32 copy (saves 8 of the
    # ~40 bytes per element and layer the row kernels and residual epilogues move): what does it cost?
    if "+respl" in variant:
        def rs(t):
            s = Split(t)
            return s.hi + s.lo8
    else:
        rs = lambda t: t  # noqa: E731

    ids = ids.to(torch.int64)
    B, L = ids.shape
    H, nh, dh = cfg.hidden, cfg.heads, cfg.head_dim
    pos = oe.position_ids(mask, cfg)
    x = f("embeddings.word_embeddings.weight")[ids] + f("embeddings.position_embeddings.weight")[pos] + \
        f("embeddings.token_type_embeddings.weight")[torch.zeros_like(ids)]
    x = oe.layer_norm(x, f("embeddings.LayerNorm.weight"), f("embeddings.LayerNorm.bias"), cfg.ln_eps)
    neg = torch.zeros(B, 1, 1, L, dtype=torch.float32)
    xr = rs(x)
    neg.masked_fill_(mask.to(torch.bool).logical_not().view(B, 1, 1, L), float("-inf"))
    scale = 1.0 / math.sqrt(dh)
    for i in range(cfg.layers):
        p = f"encoder.layer.{i}."
        q = rqk(lin(x, p + "attention.self.query.weight", p + "attention.self.query.bias"))
        k = rqk(lin(x, p + "attention.self.key.weight", p + "attention.self.key.bias"))
        v = ra(lin(x, p + "attention.self.value.weight", p + "attention.self.value.bias"))
        q = q.view(B, L, nh, dh).transpose(1, 2)
        k = k.view(B, L, nh, dh).transpose(1, 2)
        v = v.view(B, L, nh, dh).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) * scale + neg
        m = s.max(dim=-1, keepdim=True).values
        e = torch.exp(s - m)
        denom = e.sum(dim=-1, keepdim=True)
        ctx = (ra(e) @ v) / denom
        ctx = ctx.transpose(1, 2).reshape(B, L, H)
        a = lin(ctx, p + "attention.output.dense.weight", p + "attention.output.dense.bias")
        x = oe.layer_norm(a + xr, f(p + "attention.output.LayerNorm.weight"), f(p + "attention.output.LayerNorm.bias"), cfg.ln_eps)
        xr = rs(x)
        h = oe.gelu_erf(lin(x, p + "intermediate.dense.weight", p + "intermediate.dense.bias"))
        o = lin(h, p + "output.dense.weight", p + "output.dense.bias")
        x = oe.layer_norm(o + xr, f(p + "output.LayerNorm.weight"), f(p + "output.LayerNorm.bias"), cfg.ln_eps)
        xr = rs(x)
    h = x[:, 0, :]
    t = torch.tanh(h @ f("classifier.dense.weight").T + f("classifier.dense.bias"))
    return torch.sigmoid((t @ f("classifier.out_proj.weight").T + f("classifier.out_proj.bias"))[:, 0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--small", action="store_true", help="100 pairs x 128 tokens, seed 11 (the fp16_emulation.py case) instead of the fixture")
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--variants", default="f16_only,f16c,f16c+att32,f16_exact_corr,bf16x3")
    ap.add_argument("--stress", action="store_true", help="outlier-feature / peaked-attention weights (tests/stress_weights.py)")
    args = ap.parse_args()
    torch.set_num_threads(8)
    from rank_checks import kendall_tau, topn_overlap

    if args.small:
        from tensor_truth_amd.encoder import BGE_RERANKER_V2_M3 as cfgp

        kw = dict(cfgp.__dict__); kw["layers"] = args.layers
        cfg = oe.EncoderConfig(**kw)
        W = oe.synth_weights(cfg, seed=11)
        ids, mask = oe.synth_tokens(100, 128, cfg, seed=31)
        groups = [(ids[g:g + 50], mask[g:g + 50]) for g in (0, 50)]
        ref = None
    else:
        import test_rank_agreement_gpu as t

        shape = dict(t.SHAPE); shape["layers"] = args.layers
        cfg = oe.EncoderConfig(**shape)
        W = oe.synth_weights(cfg, seed=t.WEIGHT_SEED)
        pairs = t._pairs()
        groups = [(torch.from_numpy(pairs[q]), torch.ones_like(torch.from_numpy(pairs[q]))) for q in range(t.N_QUERIES)]
        ref = None
        if args.layers == 24 and not args.stress:
            ref = torch.from_numpy(np.load(os.path.join("tests", "golden", t.GOLDEN_NAME))["scores"])
    if args.stress:
        import stress_weights
        import test_rank_agreement_gpu as t

        z = np.load(os.path.join("tests", "golden", t.STRESS_GOLDEN_NAME))
        W = stress_weights.with_head(stress_weights.apply(W, cfg, qk_scales=z["qk_scales"]), z["head_w"], z["head_b"])
        if args.layers == 24 and not args.small:
            ref = torch.from_numpy(z["scores"].astype(np.float32))
    with torch.no_grad():
        if ref is None:
            t0 = time.time()
            ref = torch.stack([forward(i, m, W, cfg, "fp32", {}) for i, m in groups])
            print(f"fp32 oracle: {time.time() - t0:.0f} s; scores {ref.min():.4f} .. {ref.max():.4f}, spread (std) {ref.std():.4f}", flush=True)
        for variant in args.variants.split(","):
            t0 = time.time()
            wcache = {}
            got = torch.stack([forward(i, m, W, cfg, variant, wcache) for i, m in groups])
            err = (got - ref).abs()
            rel = (err / ref.abs().clamp_min(1e-6))
            taus = [kendall_tau(ref[g].numpy(), got[g].numpy()) for g in range(len(groups))]
            ov = [topn_overlap(ref[g].numpy(), got[g].numpy(), 10) for g in range(len(groups))]
            print(f"{variant:>16}: |err| mean {err.mean():.2e} max {err.max():.2e}  relative mean {rel.mean():.2e} max {rel.max():.2e}  "
                  f"tau {min(taus):.4f}..{max(taus):.4f}  top-10 overlap {min(ov):.1f}..{max(ov):.1f}  ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
