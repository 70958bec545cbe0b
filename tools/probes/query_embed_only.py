"""One query's embedding (34 tokens, bge-m3 shape, bf16), 60 times with a device sync after each: the lone caller's dependent launch
chain.  Run under `rocprofv3 --kernel-trace` (tools/gpu_query_embed_trace.sh) to split its latency into kernel time and gaps."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from tensor_truth_amd.encoder import BGE_M3, Encoder, EncoderWeights, pack_token_matrix, synthetic_state_device  # noqa: E402

dev = torch.device("cuda", 0)
emb = Encoder(EncoderWeights(BGE_M3, synthetic_state_device(BGE_M3, dev, seed=1), dev))
rng = np.random.default_rng(7)
ts = []
for it in range(64):
    q = rng.integers(4, BGE_M3.vocab_size, size=(1, 34), dtype=np.int32)
    q[:, 0], q[:, -1] = 0, 2
    pk = pack_token_matrix(q, BGE_M3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    emb.embed_packed(pk)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(f"query embedding, wall incl. sync: median {sorted(ts[4:])[30]:.3f} ms, min {min(ts[4:]):.3f} ms")
