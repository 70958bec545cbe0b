"""Would an fp16 compute mode (FlagEmbedding's own default for these models, `use_fp16=True`; same MFMA rate as bf16) close the
bf16 mode's score error?  CPU emulation: the oracle forward with the HIP path's six rounding points per layer (and the weights)
rounded to bf16, then to fp16, against fp32 -- full depth, reranker shape, synthetic weights.  Also prints the largest activation
magnitude seen at the rounding points (fp16 overflows at 65504; synthetic weights have no outlier features, real checkpoints may).

Usage: python tools/probes/fp16_emulation.py [layers] [pairs] [tokens]"""
import sys

sys.path.insert(0, ".")
import torch

from oracle import encoder as oe


def kendall(a, b):
    n = len(a)
    s = 0
    for i in range(n):
        s += (torch.sign(a[i] - a[i + 1:]) * torch.sign(b[i] - b[i + 1:])).sum().item()
    return s / (n * (n - 1) / 2)


def main():
    layers = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    tokens = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    from tensor_truth_amd.encoder import BGE_RERANKER_V2_M3 as cfgp

    kw = dict(cfgp.__dict__); kw["layers"] = layers
    cfg = oe.EncoderConfig(**kw)
    W = oe.synth_weights(cfg, seed=11)
    ids, mask = oe.synth_tokens(pairs, tokens, cfg, seed=31)
    torch.set_num_threads(8)
    peak = {"v": 0.0}
    orig = oe._rnd
    with torch.no_grad():
        ref = oe.rerank_scores(ids, mask, W, cfg)
        print(f"{layers} layers, {pairs} pairs x {tokens} tokens, synthetic weights (seed 11); score spread (std) {ref.std():.4f}")
        for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
            def rnd(x, on, dt=dt):
                if on:
                    peak["v"] = max(peak["v"], float(x.abs().max()))
                    return x.to(dt).to(torch.float32)
                return x
            oe._rnd = rnd
            Wr = {k: (v.to(dt).to(torch.float32) if v.dim() == 2 else v) for k, v in W.items()}      # matrices in the mode's dtype
            peak["v"] = 0.0
            s = oe.rerank_scores(ids, mask, Wr, cfg, emulate_bf16=True)
            oe._rnd = orig
            taus, overlaps = [], []
            for g in range(0, pairs - 49, 50):
                a, b = ref[g:g + 50], s[g:g + 50]
                taus.append(kendall(a, b))
                overlaps.append(len(set(a.topk(10).indices.tolist()) & set(b.topk(10).indices.tolist())) / 10)
            rel = ((s - ref).abs() / ref.abs().clamp_min(1e-6)).max()
            print(f"{name}: score error mean {(s - ref).abs().mean():.2e}  max {(s - ref).abs().max():.2e}  max relative {rel:.2e}  "
                  f"tau {sum(taus) / len(taus):.3f}  top-10 overlap {sum(overlaps) / len(overlaps):.2f}  largest |value| at a rounding point {peak['v']:.1f}")


if __name__ == "__main__":
    main()
