"""Where does a plugin-surface query spend its host time?  Wraps the main host-side steps of retrieve() /
postprocess_nodes() with wall-clock accumulators (all threads) and runs bench.py's surface leg at a reduced corpus."""
import os
import sys
import threading
import time
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402

acc = defaultdict(float)
cnt = defaultdict(int)
lock = threading.Lock()


def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name

    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            dt = time.perf_counter() - t0
            with lock:
                acc[label] += dt
                cnt[label] += 1

    setattr(obj, name, timed)


def main():
    from tensor_truth_amd import encoder as enc_mod, rerank as rr_mod, vector_index as vi
    from tensor_truth_amd.encoder import BGE_M3, BGE_RERANKER_V2_M3, Encoder
    from tensor_truth_amd.tokenization import HashTokenizer, HFTokenizer

    wrap(HashTokenizer, "encode_pair_batch")
    wrap(HashTokenizer, "encode_batch")
    wrap(HFTokenizer, "encode_pair_batch", "hf.encode_pair_batch")
    wrap(HFTokenizer, "encode_batch", "hf.encode_batch")
    wrap(vi.HipVectorRetriever, "nodes_from_hits")
    wrap(vi.HipVectorRetriever, "_query_matrix")
    wrap(Encoder, "rerank_packed")
    wrap(Encoder, "_upload")
    wrap(rr_mod.HipSentenceTransformerRerank, "_pack")
    wrap(rr_mod.HipSentenceTransformerRerank, "_enqueue_many"); wrap(rr_mod.HipSentenceTransformerRerank, "_collect_many")
    wrap(rr_mod.HipSentenceTransformerRerank, "_prepare_many")
    wrap(rr_mod.HipSentenceTransformerRerank, "postprocess_nodes")
    wrap(vi.HipVectorRetriever, "retrieve")
    from tensor_truth_amd.sharded_index import ShardedHipVectorIndex, ShardedHipVectorRetriever
    wrap(ShardedHipVectorIndex, "search", "index.search")
    wrap(ShardedHipVectorRetriever, "_retrieve_batch")

    class A:
        pass

    args = A()
    args.top_k, args.top_n, args.chunk_len, args.query_len = 50, 10, 256, 32
    args.surface_threads = int(os.environ.get("THREADS", "32"))
    args.surface_queries = int(os.environ.get("QUERIES", "256"))
    dev = torch.device("cuda", 0)
    rows = bench.synth_corpus_shard(int(os.environ.get("ROWS", "1000000")), 1024, 1234, dev)
    t0 = time.perf_counter()
    # TOKENIZER=unigram-250k (default: the trained sub-word model, round 5) | hash
    res = bench.surface_leg(args, dev, rows, BGE_M3, BGE_RERANKER_V2_M3, tokenizer=os.environ.get("TOKENIZER", "unigram-250k"))
    print({k: v for k, v in res.items() if k != "what"})
    print(f"leg wall {time.perf_counter() - t0:.2f}s (includes model construction + warm-up)")
    for k in sorted(acc, key=lambda k: -acc[k]):
        print(f"{k:28s} calls {cnt[k]:6d}  total {acc[k] * 1e3:9.1f} ms  per call {acc[k] / cnt[k] * 1e3:8.3f} ms")


if __name__ == "__main__":
    main()
