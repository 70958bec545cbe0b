"""Generates tests/golden/rank_oracle_24L_4x50x292.npz: the fp32 CPU oracle's sigmoid scores for the full-depth rerank gate
(tests/test_rank_agreement_gpu.py) -- 4 queries x 50 pairs x 292 tokens through the 24-layer bge-reranker-v2-m3 shape with
the seeded HF-style random weights of ``oracle.encoder.synth_weights``.  The oracle forward costs minutes of host time;
the GPU suite reads this fixture instead of recomputing it (and a CPU test re-derives one pair to keep it honest).

    python tests/golden/make_rank_golden.py
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))                  # tests/
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))  # repo root

from oracle import encoder as oe  # noqa: E402
import test_rank_agreement_gpu as t  # noqa: E402
from rank_checks import weights_checksum  # noqa: E402


def main():
    ocfg = oe.EncoderConfig(**t.SHAPE)
    W = oe.synth_weights(ocfg, seed=t.WEIGHT_SEED)
    pairs = t._pairs()
    want = torch.empty(t.N_QUERIES, t.N_PAIRS)
    with torch.no_grad():
        for q in range(t.N_QUERIES):
            ids = torch.from_numpy(pairs[q])
            want[q] = oe.rerank_scores(ids, torch.ones_like(ids), W, ocfg)
            print(f"query {q}: scores {want[q].min().item():.4f} .. {want[q].max().item():.4f}", flush=True)
    np.savez(os.path.join(HERE, t.GOLDEN_NAME), scores=want.numpy().astype(np.float32),
             pairs_sha256=hashlib.sha256(pairs.tobytes()).hexdigest(), weights_sha256=weights_checksum(W),
             torch_version=torch.__version__)
    print("wrote", t.GOLDEN_NAME)


if __name__ == "__main__":
    main()
