"""Where does the host time of the composed config-5 ingest go?  cProfile of build_index(semantic_hierarchical) over
synthetic documents (the bench's generator), bge-m3-shaped embedder.  Usage: python tools/probes/ingest_profile.py [docs]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from tensor_truth_amd import model_manager as mm  # noqa: E402
from tensor_truth_amd.encoder import BGE_M3  # noqa: E402
from tensor_truth_amd.index_builder import build_index  # noqa: E402
from tensor_truth_amd.schema import TextNode  # noqa: E402


def main():
    n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    windows = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [512]
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
    mgr = mm.ModelManager.get_instance()
    _kw = {"encoder_config": BGE_M3, "synthetic_seed": 1}
    if os.environ.get("INGEST_PRECISION", "bf16") != "default":      # INGEST_PRECISION=default: the constructor without a dtype (reference precision)
        _kw["torch_dtype"] = "bfloat16"
    mgr.model_kwargs_overrides["BAAI/bge-m3"] = _kw
    emb = mgr.get_embedder("BAAI/bge-m3", "cuda")
    build_index(docs[:64], emb, chunking_strategy="semantic_hierarchical", chunk_sizes=[512, 128, 64], chunk_overlap=8)   # warm-up
    if os.environ.get("SWITCH_INTERVAL"):
        sys.setswitchinterval(float(os.environ["SWITCH_INTERVAL"]))
        print("switch interval", sys.getswitchinterval())
    for win in windows:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        index = build_index(docs, emb, chunking_strategy="semantic_hierarchical", chunk_sizes=[512, 128, 64], chunk_overlap=8,
                            window_docs=win)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        from tensor_truth_amd.ingest_workers import default_workers

        print(f"{n_docs} docs -> {index.n} leaves in {dt:.2f} s = {n_docs / dt:.1f} docs/s (unprofiled, window_docs={win}, "
              f"host workers {default_workers()})")
    if os.environ.get("NO_PROFILE") == "1":
        return
    pr = cProfile.Profile()
    pr.enable()
    build_index(docs, emb, chunking_strategy="semantic_hierarchical", chunk_sizes=[512, 128, 64], chunk_overlap=8)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr, stream=sys.stdout)
    st.sort_stats("cumulative").print_stats(45)
    st.sort_stats("tottime").print_stats(25)


if __name__ == "__main__":
    main()
