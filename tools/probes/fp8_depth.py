"""Where does the fp8 mode's error at full depth come from?  24-layer reranker shape, synthetic weights, vs fp32 oracle."""
import sys
sys.path.insert(0, ".")
import torch
from oracle import encoder as oe
from tensor_truth_amd.encoder import BGE_RERANKER_V2_M3, Encoder, EncoderWeights, pack_tokens
dev = torch.device("cuda", 0)
cfg = BGE_RERANKER_V2_M3; ocfg = oe.EncoderConfig(**cfg.__dict__)
W = oe.synth_weights(ocfg, seed=11)
g = torch.Generator().manual_seed(5)
lens = [96, 33, 64, 17, 80, 50, 120, 45, 70, 28, 101, 60]
seqs = [[0] + torch.randint(4, cfg.vocab_size, (n - 2,), generator=g).tolist() + [2] for n in lens]
L = max(lens); ids = torch.full((len(seqs), L), cfg.pad_id, dtype=torch.int64); mask = torch.zeros(len(seqs), L, dtype=torch.int64)
for b, s in enumerate(seqs): ids[b, :len(s)] = torch.tensor(s); mask[b, :len(s)] = 1
torch.set_num_threads(32)
with torch.no_grad():
    ref_l = oe.rerank_logits(ids, mask, W, ocfg)
ref = torch.sigmoid(ref_l)
enc = Encoder(EncoderWeights(cfg, W, dev))
def report(name):
    s, l = enc.rerank(seqs, want_logits=True)
    print(f"{name:34s} score err max {(s.cpu()-ref).abs().max().item():.4f} mean {(s.cpu()-ref).abs().mean().item():.4f}  logit err max {(l.cpu()-ref_l).abs().max().item():.3f}  (ref logits span {ref_l.min().item():.2f}..{ref_l.max().item():.2f})", flush=True)
report("bf16")
enc.w.ffn_act_scales = None; enc.w.set_gemm_dtype("fp8"); report("fp8: qkv + o + up (down bf16)")
for margin in (1.0, 2.0, 4.0):
    enc.w.set_gemm_dtype("bf16"); enc.calibrate_fp8(pack_tokens(seqs, cfg, None, 512), margin=margin); enc.w.set_gemm_dtype("fp8")
    report(f"fp8: all four, FFN margin {margin}")
