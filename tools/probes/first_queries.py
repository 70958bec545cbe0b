"""Round 6: what do the FIRST queries of a fresh service cost?  (build_retrieval_service over 1 and 3 modules of a 10 M-row corpus, bf16;
each of the first six calls timed on its own, then the slowest phases.)"""
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
out = {}
for label, kw in (("bf16", {"torch_dtype": "bfloat16"}), ("default_precision", {})):
    for n_mod in (1, 3):
        mm.ModelManager.reset_instance()
        mgr = mm.ModelManager.get_instance()
        mgr.model_kwargs_overrides["BAAI/bge-m3"] = {"encoder_config": BGE_M3, "synthetic_seed": 1, "tokenizer": texts.tokenizer, **kw}
        mgr.model_kwargs_overrides["BAAI/bge-reranker-v2-m3"] = {"encoder_config": BGE_RERANKER_V2_M3, "synthetic_seed": 2, "tokenizer": texts.tokenizer, **kw}
        t0 = time.perf_counter()
        emb = mgr.get_embedder("BAAI/bge-m3", str(dev))
        n_rows, D = shard.shape
        bounds = [n_rows * i // n_mod for i in range(n_mod + 1)]
        docstore = B._SynthDocstore(args.chunk_len, texts.chunk)
        indexes = []
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            ix = HipVectorIndex(D, dev, emb)
            ix._mat, ix.n, ix.leaf_ids, ix.docstore = shard[lo:hi], hi - lo, B._RowIds(hi - lo, lo), docstore
            ix._mark_written()
            indexes.append(ix)
        svc = build_retrieval_service(indexes, params, device=str(dev), manager=mgr)
        torch.cuda.synchronize()
        build_s = time.perf_counter() - t0
        calls = []
        for i in range(6):
            t1 = time.perf_counter()
            r = svc.retrieve(texts.query(7_000_000_000 + 1000 * n_mod + i))
            calls.append(round((time.perf_counter() - t1) * 1e3, 1))
        out[f"{label}/{n_mod}"] = {"build_s": round(build_s, 2), "call_ms": calls, "sources": r.num_sources}
        del svc, indexes
print(json.dumps(out))
