"""Where the non-kernel time of a bench step goes: wall-clock segments with a device sync after each."""
import sys, time, os
sys.path.insert(0, ".")
import numpy as np, torch
import bench as B
from tensor_truth_amd import _lib
from tensor_truth_amd.encoder import BGE_M3, BGE_RERANKER_V2_M3, Encoder, EncoderWeights, pack_token_matrix, synthetic_state_device
from tensor_truth_amd.sharded import ShardedCorpus, gather_queries
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
N = 10_000_000
corpus = ShardedCorpus(B.synth_corpus_shard(N, 1024, 1234, dev), 0, N)
emb = Encoder(EncoderWeights(BGE_M3, synthetic_state_device(BGE_M3, dev, seed=1), dev))
rr = Encoder(EncoderWeights(BGE_RERANKER_V2_M3, synthetic_state_device(BGE_RERANKER_V2_M3, dev, seed=2), dev))
rng = np.random.default_rng(777); vocab = BGE_M3.vocab_size
Bq, K, QL, CL = 16, 50, 32, 256
def sync(): torch.cuda.synchronize()
acc = {}
def seg(name, t0):
    sync(); t = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t - t0); return t
for it in range(4):
    if it == 1: acc.clear()
    q_tok = rng.integers(4, vocab, size=(Bq, QL), dtype=np.int32)
    sync(); t = time.perf_counter()
    q_ids = np.empty((Bq, QL + 2), dtype=np.int32); q_ids[:, 0], q_ids[:, 1:-1], q_ids[:, -1] = 0, q_tok, 2
    batch = pack_token_matrix(q_ids, BGE_M3); t = seg("pack_q", t)
    _, q16 = emb.embed_packed(batch); t = seg("embed_q", t)
    s, i = corpus.search(gather_queries(q16), K); t = seg("search", t)
    mine = i.cpu().numpy(); t = seg("idx_to_host", t)
    ptok = B.passage_tokens(np.maximum(mine.reshape(-1), 0), CL, vocab); t = seg("passage_tokens", t)
    pair = np.empty((Bq * K, QL + CL + 4), dtype=np.int32)
    pair[:, 0] = 0; pair[:, 1:1 + QL] = np.repeat(q_tok, K, axis=0); pair[:, 1 + QL:3 + QL] = 2; pair[:, 3 + QL:-1] = ptok; pair[:, -1] = 2
    t = seg("pair_build", t)
    rb = pack_token_matrix(pair, BGE_RERANKER_V2_M3); t = seg("pack_pairs", t)
    sc = rr.rerank_packed(rb).view(Bq, K); t = seg("rerank", t)
    ts, tj = torch.topk(sc, 10, dim=1); rows = torch.gather(i.long(), 1, tj); a, b = ts.cpu(), rows.cpu(); t = seg("topk_out", t)
for k, v in acc.items(): print(f"{k:16s} {v / 3 * 1e3:8.3f} ms")
print("total", sum(acc.values()) / 3 * 1e3)
