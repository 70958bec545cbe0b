"""Would MX (per-32-element, power-of-two) block scales earn the fp8 mode its rank fidelity?  

The fp8 mode already scales finely: activations per TOKEN, weights per OUTPUT CHANNEL (dynamic absmax -> 448), only the FFN
intermediate uses a calibrated per-layer scale.  e4m3 is a floating-point format: above its subnormal range an element's
relative rounding error is 2^-4 whatever the scale, so finer scales only help the elements that a coarse scale pushes
into the subnormals (below 2^-6 of 448 / absmax) or to zero.  This probe measures that on the path's own operands, on the CPU,
with the emulation the parity tests pin the HIP fp8 path against (oracle/encoder.py linear_fp8):

  * share of GEMM-operand elements that land in e4m3's subnormal range / flush to zero under the per-row scale;
  * relative error of each projection's output (vs the fp32 product) with per-row scales and with OCP-MX blocks of 32
    (shared e8m0 exponent = floor(log2(absmax_block)) - 8, saturating e4m3 elements), same operands;
  * the reranker's scores through the whole stack both ways: mean abs error, Kendall tau, top-10 overlap vs fp32.

Usage: python tools/probes/fp8_mx_emulation.py [layers] [pairs]     (CPU only; minutes at the default 24 layers x 100 pairs)"""
import sys

sys.path.insert(0, ".")
import torch

from oracle import encoder as oe

E4M3_MAX = 448.0


def q_rows(t):
    q, s = oe.quantize_rows_e4m3(t)
    return q * s


def q_mx(t, block=32):
    """OCP microscaling: blocks of 32 along the reduction axis share one power-of-two scale."""
    shp = t.shape
    x = t.reshape(-1, shp[-1] // block, block)
    amax = x.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    scale = torch.exp2(torch.floor(torch.log2(amax)) - 8.0)
    q = (x / scale).clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).to(torch.float32)
    return (q * scale).reshape(shp)


def subnormal_share(t):
    """Elements whose scaled magnitude is below e4m3's smallest normal (2^-6) under the per-row scale, and those that flush to 0."""
    amax = t.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    y = (t * (E4M3_MAX / amax)).abs()
    nz = t != 0
    return ((y < 2.0 ** -6) & nz).float().mean().item(), ((y < 2.0 ** -10) & nz).float().mean().item()


class Stats:
    def __init__(self):
        self.rows = {}

    def add(self, name, x, w, b):
        ref = x @ w.T
        e_row = ((q_rows(x) @ q_rows(w).T) - ref).norm() / ref.norm()
        e_mx = ((q_mx(x) @ q_mx(w).T) - ref).norm() / ref.norm()
        sub, zero = subnormal_share(x)
        r = self.rows.setdefault(name, [0, 0.0, 0.0, 0.0, 0.0])
        r[0] += 1; r[1] += e_row.item(); r[2] += e_mx.item(); r[3] += sub; r[4] += zero


def forward(ids, mask, W, cfg, mode, stats=None):
    """oracle.encoder.encoder_forward's layer loop with the projection operands quantised by ``mode`` (None: fp32;
    'row': the HIP fp8 path's scales; 'mx': blocks of 32).  Attention, LayerNorm, GELU, residuals in fp32 -- only the
    fp8 operand rounding is under test."""
    names = ["q,k,v", "attn out", "ffn up", "ffn down"]
    quant = {None: lambda t: t, "row": q_rows, "mx": q_mx}[mode]

    def lin(name, x, w, b):
        if stats is not None:
            stats.add(name, x.reshape(-1, x.shape[-1]), w, b)
        return quant(x) @ quant(w).T + b

    B, L = ids.shape
    pos = oe.position_ids(mask, cfg)
    x = W["embeddings.word_embeddings.weight"][ids] + W["embeddings.position_embeddings.weight"][pos] + \
        W["embeddings.token_type_embeddings.weight"][torch.zeros_like(ids)]
    x = oe.layer_norm(x, W["embeddings.LayerNorm.weight"], W["embeddings.LayerNorm.bias"], cfg.ln_eps)
    neg = (1.0 - mask[:, None, None, :].float()) * -1e30
    H, Dh = cfg.heads, cfg.hidden // cfg.heads
    for layer in range(cfg.layers):
        p = f"encoder.layer.{layer}."
        wq = torch.cat([W[p + f"attention.self.{n}.weight"] for n in ("query", "key", "value")], 0)
        bq = torch.cat([W[p + f"attention.self.{n}.bias"] for n in ("query", "key", "value")], 0)
        qkv = lin(names[0], x, wq, bq).view(B, L, 3, H, Dh)
        q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
        a = torch.softmax(q @ k.transpose(-1, -2) / Dh ** 0.5 + neg, -1) @ v
        a = a.transpose(1, 2).reshape(B, L, cfg.hidden)
        x = oe.layer_norm(lin(names[1], a, W[p + "attention.output.dense.weight"], W[p + "attention.output.dense.bias"]) + x,
                          W[p + "attention.output.LayerNorm.weight"], W[p + "attention.output.LayerNorm.bias"], cfg.ln_eps)
        h = oe.gelu_erf(lin(names[2], x, W[p + "intermediate.dense.weight"], W[p + "intermediate.dense.bias"]))
        x = oe.layer_norm(lin(names[3], h, W[p + "output.dense.weight"], W[p + "output.dense.bias"]) + x,
                          W[p + "output.LayerNorm.weight"], W[p + "output.LayerNorm.bias"], cfg.ln_eps)
    c = x[:, 0, :]
    if "classifier.dense.weight" in W:
        c = torch.tanh(c @ W["classifier.dense.weight"].T + W["classifier.dense.bias"])
        return torch.sigmoid((c @ W["classifier.out_proj.weight"].T + W["classifier.out_proj.bias"])[:, 0])
    return torch.sigmoid((c @ W["classifier.weight"].T + W["classifier.bias"])[:, 0])


def kendall(a, b):
    n = len(a)
    s = 0
    for i in range(n):
        s += (torch.sign(a[i] - a[i + 1:]) * torch.sign(b[i] - b[i + 1:])).sum().item()
    return s / (n * (n - 1) / 2)


def main():
    layers = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    from tensor_truth_amd.encoder import BGE_RERANKER_V2_M3 as cfgp

    kw = dict(cfgp.__dict__); kw["layers"] = layers
    cfg = oe.EncoderConfig(**kw)
    W = oe.synth_weights(cfg, seed=11)
    ids, mask = oe.synth_tokens(pairs, 128, cfg, seed=31)
    torch.set_num_threads(8)
    with torch.no_grad():
        st = Stats()
        ref = forward(ids, mask, W, cfg, None, st)
        print(f"fp8 operand emulation, {layers} layers, {pairs} pairs x 128 tokens, synthetic weights (seed 11); CPU fp32 arithmetic")
        print("projection   rel. output error per-row scales | MX-32 scales | operand elements subnormal / flushed under the per-row scale")
        for name, r in st.rows.items():
            n = r[0]
            print(f"  {name:9s}  {r[1] / n:.4f} | {r[2] / n:.4f} | {100 * r[3] / n:.2f} % / {100 * r[4] / n:.3f} %")
        for mode in ("row", "mx"):
            s = forward(ids, mask, W, cfg, mode)
            taus, overlaps = [], []
            for g in range(0, pairs - 49, 50):
                a, b = ref[g:g + 50], s[g:g + 50]
                taus.append(kendall(a, b))
                overlaps.append(len(set(a.topk(10).indices.tolist()) & set(b.topk(10).indices.tolist())) / 10)
            print(f"scores, {mode:3s} scales: mean abs err {(s - ref).abs().mean():.4f}  max {(s - ref).abs().max():.4f}  "
                  f"tau {sum(taus) / len(taus):.3f}  top-10 overlap {sum(overlaps) / len(overlaps):.2f}   (score spread {ref.std():.3f})")


if __name__ == "__main__":
    main()
