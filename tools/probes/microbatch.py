"""Does processing the rerank batch in smaller pieces (activations closer to the 256 MB MALL) pay?"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import bench as B
from tensor_truth_amd.encoder import BGE_RERANKER_V2_M3, Encoder, EncoderWeights, pack_token_matrix, synthetic_state_device
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
cfg = BGE_RERANKER_V2_M3
rr = Encoder(EncoderWeights(cfg, synthetic_state_device(cfg, dev, seed=2), dev))
rng = np.random.default_rng(1)
pairs = rng.integers(4, cfg.vocab_size, size=(800, 292), dtype=np.int32); pairs[:, 0] = 0; pairs[:, -1] = 2
for nsplit in (1, 2, 4, 8, 1, 2, 4):
    chunks = np.array_split(pairs, nsplit)
    batches = [pack_token_matrix(c, cfg) for c in chunks]
    for b in batches: rr.rerank_packed(b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        for b in batches: rr.rerank_packed(b)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"split {nsplit}: rows/call {batches[0].n_rows}  {dt*1e3:.2f} ms per 800 pairs")
