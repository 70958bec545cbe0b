"""Which projections make the fp8 reranker lose rank fidelity?  (diagnostic; `TT_FP8_MASK` / `TT_FP8_SKIP_FIRST` /
`TT_FP8_SKIP_LAST` in csrc/encoder_api.hip)

8 queries x 50 candidate pairs x 292 tokens through the bge-reranker-v2-m3-shaped model (synthetic weights, the bench's
seed): scores of the fp32 reference-precision path vs the fp8 mode with a subset of the layer projections in e4m3 and the
rest in bf16 -> max |score error|, Kendall tau, top-10 overlap, and the time of one 1600-pair rerank batch.
"""
import os as _os

_os.environ.setdefault("TT_LIB_NAME", "libtt_hip_diag.so")   # the switches this probe sweeps exist in the diagnostic library only (csrc: make DIAG=1)
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from tensor_truth_amd.encoder import (BGE_RERANKER_V2_M3, Encoder, EncoderWeights, pack_token_matrix,  # noqa: E402
                                      synthetic_state_device)
from tensor_truth_amd.encoder_f32 import EncoderF32, EncoderWeightsF32  # noqa: E402

K, NQ, TOPN, TOK = 50, 8, 10, 292


def agreement(a, b):
    sa, sb = np.sign(a[:, None] - a[None, :]), np.sign(b[:, None] - b[None, :])
    tau = float((sa * sb).sum() / (K * (K - 1)))
    over = len(set(np.argsort(-a)[:TOPN].tolist()) & set(np.argsort(-b)[:TOPN].tolist())) / TOPN
    return tau, over


def main():
    dev = torch.device("cuda", 0)
    cfg = BGE_RERANKER_V2_M3
    rr = Encoder(EncoderWeights(cfg, synthetic_state_device(cfg, dev, seed=2), dev))
    rng = np.random.default_rng(5)
    # pairs as the bench builds them: <s> q </s></s> chunk </s>, the query shared by the 50 pairs of a query
    q_tok = rng.integers(4, cfg.vocab_size, size=(NQ, 32), dtype=np.int32)
    ids = np.empty((NQ * K, TOK), dtype=np.int32)
    ids[:, 0] = 0
    ids[:, 1:33] = np.repeat(q_tok, K, axis=0)
    ids[:, 33:35] = 2
    ids[:, 35:-1] = rng.integers(4, cfg.vocab_size, size=(NQ * K, TOK - 36), dtype=np.int32)
    ids[:, -1] = 2
    enc32 = EncoderF32(EncoderWeightsF32(cfg, rr.w.state_dict(), dev))
    s32 = torch.cat([enc32.rerank_packed(pack_token_matrix(ids[q * K:(q + 1) * K], cfg)) for q in range(NQ)]).cpu().view(NQ, K).numpy().astype(np.float64)
    del enc32
    batch = pack_token_matrix(ids, cfg)
    rr.calibrate_fp8(batch)                                      # static e4m3 scale of the FFN intermediate, per layer
    big = pack_token_matrix(np.tile(ids, (4, 1)), cfg)          # 1600 pairs: the bench's rerank batch
    cases = [("bf16", None, 0, 0)]
    for mask, name in [(0xF, "all four"), (0xE, "all but QKV"), (0xD, "all but o-proj"), (0x3, "QKV + o-proj"), (0xC, "FFN pair"),
                       (0x1, "QKV only"), (0x2, "o-proj only"), (0x4, "FFN-up only (bf16 intermediate)")]:
        cases.append((f"fp8 {name}", mask, 0, 0))
    for first, last in [(2, 2), (4, 4), (0, 4), (4, 0), (8, 8)]:
        cases.append((f"fp8 all four, first {first} / last {last} layers bf16", 0xF, first, last))
    print(f"{NQ} queries x {K} pairs x {TOK} tok, 24 layers; reference = fp32 path on the same weights")
    print(f"{'mode':58s} {'max|err|':>9s} {'tau':>7s} {'top10':>6s} {'ms/1600 pairs':>14s}")
    for name, mask, first, last in cases:
        if mask is None:
            rr.w.set_gemm_dtype("bf16")
        else:
            rr.w.set_gemm_dtype("fp8")
            os.environ["TT_FP8_MASK"] = hex(mask)
            os.environ["TT_FP8_SKIP_FIRST"], os.environ["TT_FP8_SKIP_LAST"] = str(first), str(last)
        s = rr.rerank_packed(batch).cpu().view(NQ, K).numpy().astype(np.float64)
        ag = [agreement(s32[q], s[q]) for q in range(NQ)]
        rr.rerank_packed(big)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            rr.rerank_packed(big)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        print(f"{name:58s} {np.abs(s - s32).max():9.4f} {np.mean([a[0] for a in ag]):7.3f} {np.mean([a[1] for a in ag]):6.2f} {ms:14.1f}")


if __name__ == "__main__":
    main()
