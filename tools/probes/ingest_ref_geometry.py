"""Where does build_index spend its wall time at the REFERENCE's chunk geometry ([2048, 512, 256] / 64, sub-word counted, Unigram
tokenizer, 4-8 k-word documents)?  Wraps the feeder's steps with wall-clock accumulators and reads the kernels' HIP-event totals.
Usage: python tools/probes/ingest_ref_geometry.py [n_docs] [workers | -] [words lo-hi]"""
import ctypes
import os
import sys
import time
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402

acc, cnt = defaultdict(float), defaultdict(int)


def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name

    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label] += time.perf_counter() - t0
            cnt[label] += 1

    setattr(obj, name, timed)


def main():
    from tensor_truth_amd import _lib, ingest_workers as iw, vector_index as vi
    from tensor_truth_amd.embedding import HipHuggingFaceEmbedding
    from tensor_truth_amd.encoder import BGE_M3
    from tensor_truth_amd.index_builder import build_index

    n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    workers = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "-" else None
    lo_w, hi_w = (int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "4000-8000").split("-"))

    class A:
        query_len, chunk_len = 32, 256

    texts = bench.surface_texts(A(), "unigram-250k")
    docs, n_sent, n_words = bench._c5_docs_reference_geometry(n_docs, lo_w, hi_w, np.random.default_rng(55))
    emb = HipHuggingFaceEmbedding("BAAI/bge-m3", device="cuda", embed_batch_size=128,
                                  model_kwargs={"encoder_config": BGE_M3, "synthetic_seed": 1, "torch_dtype": "bfloat16", "tokenizer": texts.tokenizer})
    kw = dict(chunking_strategy="semantic_hierarchical", chunk_sizes=None, chunk_overlap=None, token_counter="embedder", workers=workers)
    build_index(docs[:64], emb, **kw)
    torch.cuda.synchronize()
    wrap(iw.IngestWorkers, "wait", "feeder: blocked waiting for a worker reply")
    wrap(iw.IngestWorkers, "poll", "feeder: poll (recv + unpickle of ready replies)")
    wrap(iw._PipeConn, "send", "feeder: send to worker (pickle + write)")
    wrap(HipHuggingFaceEmbedding, "embed_token_batches", "feeder: embed_token_batches (sort, pack, upload, enqueue)")
    wrap(HipHuggingFaceEmbedding, "embed_flat", "feeder: embed_flat (argsort, pack, upload, enqueue)")
    wrap(vi.HipVectorIndex, "add", "feeder: index.add")
    wrap(vi.HipVectorIndex, "add_to_docstore", "feeder: index.add_to_docstore")
    lib = _lib.load_library()
    lib.tt_prof_enable(1)
    st0 = dict(emb.stats)
    prof = None
    if os.environ.get("CPROFILE") == "1":
        import cProfile

        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    index = build_index(docs, emb, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if prof is not None:
        import pstats

        prof.disable()
        pstats.Stats(prof).sort_stats("cumulative").print_stats(28)
    names = {1: "scan_filter", 2: "scan_sample", 3: "select", 4: "gemm", 5: "attention", 6: "rowops", 7: "scan_tail"}
    gpu_ms = 0.0
    for kid in range(8):
        ms, n = ctypes.c_double(), ctypes.c_int()
        if lib.tt_prof_read(kid, ctypes.byref(ms), ctypes.byref(n)) == 0 and n.value:
            print(f"  kernels[{names.get(kid, kid)}]: {ms.value:9.1f} ms in {n.value} launches")
            gpu_ms += ms.value
    lib.tt_prof_enable(0)
    tok = emb.stats["tokens"] - st0["tokens"]
    print(f"{n_docs} docs ({n_words} words, {n_sent} sentences) -> {index.n} leaves in {dt:.2f} s = {n_docs / dt:.1f} docs/s, {tok / dt / 1e6:.2f} M tokens/s; "
          f"kernel time (HIP events, library kernels only) {gpu_ms / 1e3:.2f} s = {gpu_ms / 1e3 / dt:.2f} of the wall")
    for k in sorted(acc, key=lambda k: -acc[k]):
        print(f"  {k:70s} calls {cnt[k]:6d}  total {acc[k]:7.2f} s")


if __name__ == "__main__":
    main()
