"""Round 6: how stable is reference_defaults' 8-thread rate?  Five timed bursts in one process (3 modules, bf16)."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
sys.argv = [sys.argv[0], "--corpus-rows", "10000000"]
import bench as B  # noqa: E402
from tensor_truth_amd import model_manager as mm  # noqa: E402
from tensor_truth_amd.encoder import BGE_M3, BGE_RERANKER_V2_M3  # noqa: E402
from tensor_truth_amd.retrieval_service import build_retrieval_service  # noqa: E402
from tensor_truth_amd.vector_index import HipVectorIndex  # noqa: E402

args = B.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
shard = B.synth_corpus_shard(10_000_000, 1024, 1234, dev)
texts = B.surface_texts(args, "unigram-250k")
params = {"reranker_top_n": 5, "confidence_cutoff": 0.35, "confidence_cutoff_hard": 0.05, "balance_strategy": "top_k_per_index"}
mgr = mm.ModelManager.get_instance()
kw = {"torch_dtype": "bfloat16"}
mgr.model_kwargs_overrides["BAAI/bge-m3"] = {"encoder_config": BGE_M3, "synthetic_seed": 1, "tokenizer": texts.tokenizer, **kw}
mgr.model_kwargs_overrides["BAAI/bge-reranker-v2-m3"] = {"encoder_config": BGE_RERANKER_V2_M3, "synthetic_seed": 2, "tokenizer": texts.tokenizer, **kw}
emb = mgr.get_embedder("BAAI/bge-m3", str(dev))
n_rows, D = shard.shape
bounds = [n_rows * i // 3 for i in range(4)]
docstore = B._SynthDocstore(args.chunk_len, texts.chunk)
indexes = []
for lo, hi in zip(bounds[:-1], bounds[1:]):
    ix = HipVectorIndex(D, dev, emb)
    ix._mat, ix.n, ix.leaf_ids, ix.docstore = shard[lo:hi], hi - lo, B._RowIds(hi - lo, lo), docstore
    ix._mark_written()
    indexes.append(ix)
svc = build_retrieval_service(indexes, params, device=str(dev), manager=mgr)
rr = mgr.get_reranker(None, top_n=5, device=str(dev))
B.wait_pair_pool(rr)
lat = []
phase_max = {}


def _timed(obj, attr, label):
    fn = getattr(obj, attr, None)
    if fn is None:
        return

    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            dt = time.perf_counter() - t0
            if dt > phase_max.get(label, (0.0, 0))[0]:
                phase_max[label] = (dt, len(a[0]) if a and hasattr(a[0], "__len__") else -1)

    setattr(obj, attr, wrapper)


from tensor_truth_amd import ingest_workers as _iw  # noqa: E402

_timed(rr, "_tokenize_pairs", "rerank.tokenize_pairs")
_timed(_iw.PairTokenizerPool, "encode", "pool.encode")
_timed(rr._front, "_run", "rerank.prepare")
_timed(rr._front, "_execute", "rerank.execute")
_timed(rr._front, "_finish", "rerank.finish")
_timed(svc._retriever._scan_front, "_run", "group.scan")
_timed(emb, "get_agg_embedding_from_queries", "query.embed")


def one(q):
    t0 = time.perf_counter()
    n = len(svc.retrieve(q).source_nodes)
    lat.append(time.perf_counter() - t0)
    return n


for i in range(4):
    one(texts.query(9_000_000_000 + i))
out = []
for burst in range(6):
    qs = [texts.query(9_100_000_000 + 1000 * burst + i) for i in range(64)]
    torch.cuda.synchronize(dev)
    del lat[:]
    phase_max.clear()
    dt, _ = B._run_threads(8, qs, one)
    slow = sorted(lat)[-3:]
    front = svc._retriever._scan_front
    out.append({"burst": burst, "queries_per_s": 64 / dt, "scan_batches": front.batches, "scan_items": front.items,
                "rerank_batches": rr._front.batches if rr._front else None, "slowest_calls_ms": [round(x * 1e3, 1) for x in slow],
                "median_call_ms": round(sorted(lat)[len(lat) // 2] * 1e3, 1),
                "slowest_phase_ms": {k: (round(v[0] * 1e3, 1), v[1]) for k, v in phase_max.items()}})
print(json.dumps(out))
