"""How launch-bound is ONE query's embedding (B = 1, ~34 tokens, 24 layers of weight-streaming GEMMs)?
Wall time per call against the sum of its kernels' durations (run under `rocprofv3 --kernel-trace --stats` for the latter)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from tensor_truth_amd.encoder import BGE_M3, Encoder, EncoderWeights, pack_token_matrix, synthetic_state_device  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
enc = Encoder(EncoderWeights(BGE_M3, synthetic_state_device(BGE_M3, dev, seed=1), dev))
rng = np.random.default_rng(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200


def one():
    ids = np.empty((1, 34), dtype=np.int32)
    ids[:, 0], ids[:, 1:-1], ids[:, -1] = 0, rng.integers(4, BGE_M3.vocab_size, size=(1, 32)), 2
    return enc.embed_packed(pack_token_matrix(ids, BGE_M3))


for _ in range(10):
    one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    one()
    torch.cuda.synchronize()           # a lone caller waits for its embedding
dt = (time.perf_counter() - t0) / n
print(f"one query's embedding, synchronous calls: {dt * 1e3:.3f} ms wall per call ({n} calls)")
t0 = time.perf_counter()
for _ in range(n):
    one()
torch.cuda.synchronize()
dt2 = (time.perf_counter() - t0) / n
print(f"back to back (no sync between calls): {dt2 * 1e3:.3f} ms per call")
