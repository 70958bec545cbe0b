"""Would keeping the residual stream in fp32 (GEMM operands still 16-bit) bring a 16-bit mode inside 1e-3?  CPU emulation on the
oracle arithmetic: 24 layers, 100 pairs x 128 tokens, synthetic weights; the two stream roundings (pre-LayerNorm sums, LayerNorm
outputs on the residual branch) on / off, weight matrices rounded / fp32.  Usage: python tools/probes/fp32_stream_emulation.py"""
import sys, math, time
sys.path.insert(0, ".")
import torch
from oracle import encoder as oe
from tensor_truth_amd.encoder import BGE_RERANKER_V2_M3 as cfgp

def fwd_stream32(ids, mask, W, cfg, dt, round_stream=False):
    r = lambda t: t.to(dt).to(torch.float32)
    rs = r if round_stream else (lambda t: t)
    f = lambda n: W[n].to(torch.float32)
    ids = ids.to(torch.int64); B, L = ids.shape; H, nh, dh = cfg.hidden, cfg.heads, cfg.head_dim
    pos = oe.position_ids(mask, cfg)
    x = f("embeddings.word_embeddings.weight")[ids] + f("embeddings.position_embeddings.weight")[pos] + f("embeddings.token_type_embeddings.weight")[torch.zeros_like(ids)]
    x = rs(oe.layer_norm(x, f("embeddings.LayerNorm.weight"), f("embeddings.LayerNorm.bias"), cfg.ln_eps))
    neg = torch.zeros(B, 1, 1, L); neg.masked_fill_(mask.to(torch.bool).logical_not().view(B, 1, 1, L), float("-inf"))
    scale = 1.0 / math.sqrt(dh)
    for i in range(cfg.layers):
        p = f"encoder.layer.{i}."
        xin = r(x)
        q = r(xin @ f(p + "attention.self.query.weight").T + f(p + "attention.self.query.bias")).view(B, L, nh, dh).transpose(1, 2)
        k = r(xin @ f(p + "attention.self.key.weight").T + f(p + "attention.self.key.bias")).view(B, L, nh, dh).transpose(1, 2)
        v = r(xin @ f(p + "attention.self.value.weight").T + f(p + "attention.self.value.bias")).view(B, L, nh, dh).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) * scale + neg
        m = s.max(dim=-1, keepdim=True).values; e = torch.exp(s - m); denom = e.sum(dim=-1, keepdim=True)
        ctx = r(((r(e) @ v) / denom).transpose(1, 2).reshape(B, L, H))
        a = ctx @ f(p + "attention.output.dense.weight").T + f(p + "attention.output.dense.bias")
        x = rs(oe.layer_norm(rs(a + x), f(p + "attention.output.LayerNorm.weight"), f(p + "attention.output.LayerNorm.bias"), cfg.ln_eps))
        h = r(oe.gelu_erf(r(x) @ f(p + "intermediate.dense.weight").T + f(p + "intermediate.dense.bias")))
        o = h @ f(p + "output.dense.weight").T + f(p + "output.dense.bias")
        x = rs(oe.layer_norm(rs(o + x), f(p + "output.LayerNorm.weight"), f(p + "output.LayerNorm.bias"), cfg.ln_eps))
    c = x[:, 0, :]
    t = torch.tanh(c @ f("classifier.dense.weight").T + f("classifier.dense.bias"))
    return torch.sigmoid((t @ f("classifier.out_proj.weight").T + f("classifier.out_proj.bias"))[:, 0])

layers, pairs, tokens = 24, 100, 128
kw = dict(cfgp.__dict__); kw["layers"] = layers
cfg = oe.EncoderConfig(**kw)
W = oe.synth_weights(cfg, seed=11)
ids, mask = oe.synth_tokens(pairs, tokens, cfg, seed=31)
torch.set_num_threads(8)
with torch.no_grad():
    t0 = time.time(); ref = oe.rerank_scores(ids, mask, W, cfg); print("ref", time.time() - t0, flush=True)
    for name, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
        Wr = {k: (v.to(dt).to(torch.float32) if v.dim() == 2 else v) for k, v in W.items()}
        for rs in (True, False):
            s = fwd_stream32(ids, mask, Wr, cfg, dt, round_stream=rs)
            d = (s - ref).abs()
            print(f"{name} stream {'16-bit' if rs else 'fp32  '}: mean {d.mean():.2e} max {d.max():.2e} max rel {(d / ref.abs()).max():.2e}", flush=True)
        s = fwd_stream32(ids, mask, W, cfg, dt, round_stream=False)
        d = (s - ref).abs()
        print(f"{name} stream fp32, fp32 weights (activation roundings only): mean {d.mean():.2e} max {d.max():.2e}", flush=True)
