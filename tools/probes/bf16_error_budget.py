"""Where the bf16 mode's score error comes from: the fp32-grade split-bf16 forward (csrc/x3_path.hip) with
ONE of the bf16 path's rounding points switched back on at a time (TT_X3_ROUND_MASK / EncoderWeightsX3(round_weights=True)),
at full depth -- 4 queries x 50 pairs x 292 tokens x 24 layers, the inputs and fp32-oracle scores of
tests/test_rank_agreement_gpu.py (committed fixture) -- reporting each setting's sigmoid-score error and rank agreement."""
import os as _os

_os.environ.setdefault("TT_LIB_NAME", "libtt_hip_diag.so")   # the switches this probe sweeps exist in the diagnostic library only (csrc: make DIAG=1)
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import test_rank_agreement_gpu as T
from oracle import encoder as oe
from rank_checks import kendall_tau, topn_overlap
from tensor_truth_amd.encoder import Encoder, EncoderConfig, EncoderWeights, pack_token_matrix
from tensor_truth_amd.encoder_x3 import EncoderWeightsX3, EncoderX3

dev = torch.device("cuda", 0)
torch.set_num_threads(32)
ocfg = oe.EncoderConfig(**T.SHAPE)
W = oe.synth_weights(ocfg, seed=T.WEIGHT_SEED)
pairs = T._pairs()
z = np.load(os.path.join(ROOT, "tests", "golden", T.GOLDEN_NAME))
want = torch.from_numpy(z["scores"].astype(np.float32))
cfg = EncoderConfig(**T.SHAPE)
batch = pack_token_matrix(pairs.reshape(-1, T.PAIR_TOKENS).astype(np.int32), cfg)


def report(name, got):
    err = (got - want).abs()
    taus = [kendall_tau(want[q].numpy(), got[q].numpy()) for q in range(T.N_QUERIES)]
    over = [topn_overlap(want[q].numpy(), got[q].numpy(), T.TOP_N) for q in range(T.N_QUERIES)]
    print(f"{name:58s} max |err| {err.max().item():.2e}  mean |err| {err.mean().item():.2e}  tau {np.mean(taus):.3f}  top-10 overlap {np.mean(over):.2f}", flush=True)


BITS = [(2, "Q / K / V projection outputs"), (4, "softmax probabilities P"), (8, "attention context"),
        (16, "pre-LayerNorm sums (GEMM + residual outputs)"), (32, "LayerNorm outputs (+ embedding LN, residual branch)"),
        (64, "FFN intermediate (GELU output)")]
enc = EncoderX3(EncoderWeightsX3(cfg, W, dev))
os.environ["TT_X3_ROUND_MASK"] = "0"
report("bf16x3, nothing rounded (the reference mode)", enc.rerank_packed(batch).cpu().view(T.N_QUERIES, T.N_PAIRS))
for bit, what in BITS:
    os.environ["TT_X3_ROUND_MASK"] = str(bit)
    report(f"+ bf16 rounding of: {what}", enc.rerank_packed(batch).cpu().view(T.N_QUERIES, T.N_PAIRS))
os.environ["TT_X3_ROUND_MASK"] = str(sum(b for b, _ in BITS))
report("+ all six activation roundings", enc.rerank_packed(batch).cpu().view(T.N_QUERIES, T.N_PAIRS))
for drop, what in BITS:
    os.environ["TT_X3_ROUND_MASK"] = str(sum(b for b, _ in BITS) - drop)
    report(f"  all six EXCEPT: {what}", enc.rerank_packed(batch).cpu().view(T.N_QUERIES, T.N_PAIRS))
del enc
enc = EncoderX3(EncoderWeightsX3(cfg, W, dev, round_weights=True))
os.environ["TT_X3_ROUND_MASK"] = "0"
report("+ bf16 WEIGHTS only (matrices, tables, head)", enc.rerank_packed(batch).cpu().view(T.N_QUERIES, T.N_PAIRS))
os.environ["TT_X3_ROUND_MASK"] = str(sum(b for b, _ in BITS))
report("+ bf16 weights + all six activation roundings", enc.rerank_packed(batch).cpu().view(T.N_QUERIES, T.N_PAIRS))
os.environ["TT_X3_ROUND_MASK"] = str(sum(b for b, _ in BITS) - 16)
report("+ bf16 weights + all activation roundings except pre-LN sums", enc.rerank_packed(batch).cpu().view(T.N_QUERIES, T.N_PAIRS))
os.environ["TT_X3_ROUND_MASK"] = "0"
del enc
bf = Encoder(EncoderWeights(cfg, W, dev))
report("the bf16 path itself (tt_encoder_forward_cls)", bf.rerank_packed(batch).cpu().view(T.N_QUERIES, T.N_PAIRS))
