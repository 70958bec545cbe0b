"""Interactive case (BASELINE config 3: one query, 1M x 1024 corpus, top-50, rerank 50 -> 10): wall-clock latency of
each stage with a device sync after it, median of 20.  Usage: python tools/probes/single_query_latency.py [rows] [queries] [reference]"""
import statistics
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench as B  # noqa: E402
from tensor_truth_amd.encoder import (BGE_M3, BGE_RERANKER_V2_M3, Encoder, EncoderWeights, pack_token_matrix,  # noqa: E402
                                      synthetic_state_device)
from tensor_truth_amd import scan as tscan  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
Bq = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
corpus = B.synth_corpus_shard(N, 1024, 1234, dev)
if len(sys.argv) > 3 and sys.argv[3] == "reference":      # both encoders in the reference precision (split-bf16, encoder_x3)
    from tensor_truth_amd.encoder_x3 import EncoderWeightsX3, EncoderX3

    emb = EncoderX3(EncoderWeightsX3(BGE_M3, synthetic_state_device(BGE_M3, dev, seed=1, dtype=torch.float32), dev))
    rr = EncoderX3(EncoderWeightsX3(BGE_RERANKER_V2_M3, synthetic_state_device(BGE_RERANKER_V2_M3, dev, seed=2, dtype=torch.float32), dev))
else:
    emb = Encoder(EncoderWeights(BGE_M3, synthetic_state_device(BGE_M3, dev, seed=1), dev))
    rr = Encoder(EncoderWeights(BGE_RERANKER_V2_M3, synthetic_state_device(BGE_RERANKER_V2_M3, dev, seed=2), dev))
rng = np.random.default_rng(777)
vocab = BGE_M3.vocab_size
K, QL, CL = 50, 32, 256
acc = {}


def seg(name, t0):
    torch.cuda.synchronize()
    t = time.perf_counter()
    acc.setdefault(name, []).append((t - t0) * 1e3)
    return t


for it in range(24):
    if it == 4:
        acc.clear()
    q_tok = rng.integers(4, vocab, size=(Bq, QL), dtype=np.int32)
    torch.cuda.synchronize()
    t = time.perf_counter()
    q_ids = np.empty((Bq, QL + 2), dtype=np.int32)
    q_ids[:, 0], q_ids[:, 1:-1], q_ids[:, -1] = 0, q_tok, 2
    _, q16 = emb.embed_packed(pack_token_matrix(q_ids, BGE_M3))
    t = seg("embed query", t)
    s, i = tscan.scan_topk(corpus, q16, K)
    t = seg("scan top-50", t)
    mine = i.cpu().numpy()
    ptok = B.passage_tokens(np.maximum(mine.reshape(-1), 0), CL, vocab)
    pair = np.empty((Bq * K, QL + CL + 4), dtype=np.int32)
    pair[:, 0] = 0
    pair[:, 1:1 + QL] = np.repeat(q_tok, K, axis=0)
    pair[:, 1 + QL:3 + QL] = 2
    pair[:, 3 + QL:-1] = ptok
    pair[:, -1] = 2
    t = seg("candidates to host + pair build", t)
    sc = rr.rerank_packed(pack_token_matrix(pair, BGE_RERANKER_V2_M3)).view(Bq, K)
    t = seg("rerank 50 pairs", t)
    ts, tj = torch.topk(sc, 10, dim=1)
    out = (ts.cpu(), torch.gather(i.long(), 1, tj).cpu())
    t = seg("top-10 to host", t)
tot = 0.0
for k, v in acc.items():
    m = statistics.median(v)
    tot += m
    print(f"{k:34s} {m:8.3f} ms (min {min(v):.3f})")
print(f"{'total':34s} {tot:8.3f} ms  -> {Bq / tot * 1e3:.1f} queries/s at {Bq} query per step, {N} rows")
