"""Rounds 2-4's small config-5 geometry ([512,128,64] / 8 in words, ~1.1 k-word documents, hashing tokenizer) alone: docs/s by worker count.
Usage: python tools/probes/ingest_small_geometry.py [n_docs] [workers ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    from tensor_truth_amd.embedding import HipHuggingFaceEmbedding
    from tensor_truth_amd.encoder import BGE_M3
    from tensor_truth_amd.index_builder import build_index
    from tensor_truth_amd.schema import TextNode

    n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    rng = np.random.default_rng(55)
    w = bench._words()
    docs = []
    for d in range(n_docs):
        sents = []
        for block in range(4):
            band = int(rng.integers(0, 40)) * 1000
            for _ in range(int(rng.integers(12, 20))):
                k = int(rng.integers(10, 24))
                sents.append(" ".join(w[band + int(j)] for j in rng.integers(0, 1000, size=k)) + ".")
        docs.append(TextNode(text=" ".join(sents), metadata={"title": f"doc {d}"}))
    emb = HipHuggingFaceEmbedding("BAAI/bge-m3", device="cuda", embed_batch_size=128,
                                  model_kwargs={"encoder_config": BGE_M3, "synthetic_seed": 1, "torch_dtype": "bfloat16"})
    kw = dict(chunking_strategy="semantic_hierarchical", chunk_sizes=[512, 128, 64], chunk_overlap=8)
    for W in [int(x) for x in sys.argv[2:]] or [8, 16]:
        build_index(docs[:96], emb, workers=W, **kw)
        torch.cuda.synchronize()
        st0 = dict(emb.stats)
        t0 = time.perf_counter()
        idx = build_index(docs, emb, workers=W, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"workers {W:3d}: {n_docs} docs -> {idx.n} leaves in {dt:.2f} s = {n_docs / dt:.1f} docs/s, {(emb.stats['tokens'] - st0['tokens']) / dt / 1e6:.2f} M tokens/s", flush=True)


if __name__ == "__main__":
    main()
