"""Where the host time of a config-5 query goes: cProfile of svc.retrieve() (auto-merging retriever + reranker) from ONE thread,
and the wall time per query from 1 and 32 threads."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from tensor_truth_amd import model_manager as mm
from tensor_truth_amd.encoder import BGE_M3, BGE_RERANKER_V2_M3
from tensor_truth_amd.index_builder import build_index
from tensor_truth_amd.retrieval_service import build_retrieval_service
from tensor_truth_amd.schema import TextNode

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(55)
w = bench._words()
docs = []
for d in range(n_docs):
    sents = []
    for block in range(4):
        band = int(rng.integers(0, 40)) * 1000
        for _ in range(int(rng.integers(12, 20))):
            sents.append(" ".join(w[band + int(j)] for j in rng.integers(0, 1000, size=int(rng.integers(10, 24)))) + ".")
    docs.append(TextNode(text=" ".join(sents), metadata={"title": f"doc {d}"}))
mgr = mm.ModelManager.get_instance()
mgr.model_kwargs_overrides["BAAI/bge-m3"] = {"encoder_config": BGE_M3, "synthetic_seed": 1, "torch_dtype": "bfloat16"}
mgr.model_kwargs_overrides["BAAI/bge-reranker-v2-m3"] = {"encoder_config": BGE_RERANKER_V2_M3, "synthetic_seed": 2, "torch_dtype": "bfloat16"}
emb = mgr.get_embedder("BAAI/bge-m3", "cuda")
index = build_index(docs, emb, chunking_strategy="semantic_hierarchical", chunk_sizes=[512, 128, 64], chunk_overlap=8)
svc = build_retrieval_service([index], {"reranker_top_n": 10, "similarity_top_k": 50, "confidence_cutoff": 0.35}, device="cuda", manager=mgr)
qs = [" ".join(w[int(j)] for j in rng.integers(0, 40000, size=32)) for _ in range(64 + 384)]
for q in qs[:8]:
    svc.retrieve(q)
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for q in qs[8:64]:
    svc.retrieve(q)
pr.disable()
dt1 = (time.perf_counter() - t0) / 56
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print("\n".join(ln[:170] for ln in s.getvalue().splitlines()[:60]))
dt, _ = bench._run_threads(32, qs[64:], lambda q: svc.retrieve(q).num_sources)
print(f"one thread: {dt1 * 1e3:.2f} ms per query ({1 / dt1:.0f} q/s); 32 threads: {384 / dt:.0f} q/s")
