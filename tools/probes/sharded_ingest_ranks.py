"""BASELINE config 5 on N ranks, end to end on the real kernels: every rank ingests documents rank, rank + N, ... with
`build_index_sharded` (semantic-hierarchical), the rank-local rows become the shards of one index, and the same queries go
through the auto-merging retriever on every rank.  Launched by tools/gpu_sharded_ingest.sh with N ranks sharing GPU 0 over
gloo (a 1-GPU box; on an N-GPU node set TT_ONE_DEVICE=0 and it runs one rank per GPU over RCCL).  Rank 0 also builds the
single-device index over ALL documents and checks that the sharded answers are the same nodes (by text) with the same scores."""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from tensor_truth_amd.embedding import HipHuggingFaceEmbedding  # noqa: E402
from tensor_truth_amd.encoder import BGE_M3, EncoderConfig  # noqa: E402
from tensor_truth_amd.index_builder import build_index, build_index_sharded  # noqa: E402
from tensor_truth_amd.retrievers import AutoMergingRetriever  # noqa: E402
from tensor_truth_amd.schema import TextNode  # noqa: E402


def main():
    n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    layers = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    one_device = os.environ.get("TT_ONE_DEVICE", "1") == "1"
    local = 0 if one_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if one_device else "nccl")
    rng = np.random.default_rng(55)                    # the same documents on every rank
    w = bench._words()
    docs = []
    for d in range(n_docs):
        sents = []
        for block in range(4):
            band = int(rng.integers(0, 40)) * 1000
            for _ in range(int(rng.integers(12, 20))):
                sents.append(" ".join(w[band + int(j)] for j in rng.integers(0, 1000, size=int(rng.integers(10, 24)))) + ".")
        docs.append(TextNode(text=" ".join(sents), id_=f"doc{d}", metadata={"title": f"doc {d}"}))
    cfg = EncoderConfig(**{**BGE_M3.__dict__, "layers": layers})
    emb = HipHuggingFaceEmbedding("BAAI/bge-m3", device=f"cuda:{local}", embed_batch_size=128,
                                  model_kwargs={"encoder_config": cfg, "synthetic_seed": 1})
    kw = dict(chunking_strategy="semantic_hierarchical", chunk_sizes=[512, 128, 64], chunk_overlap=8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    index = build_index_sharded(docs, emb, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    queries = [" ".join(w[int(j)] for j in rng.integers(0, 40000, size=32)) for _ in range(16)]
    retr = AutoMergingRetriever(index.as_retriever(similarity_top_k=20), index.docstore)
    got = [[(h.node.text, round(h.score, 6)) for h in retr.retrieve(q)] for q in queries]
    n_local = sum(r.shape[0] for r, _ in index._shards)
    print(f"rank {rank}/{world}: {n_local} local rows of {index.n_total} (rows {index.row_lo}..), "
          f"{len(index.docstore)} nodes in the docstore, ingest {dt:.2f} s", flush=True)
    if world > 1:
        all_got = [None] * world
        dist.all_gather_object(all_got, got)
        assert all(g == all_got[0] for g in all_got), "ranks disagree on replicated queries"
    if rank == 0:
        full = build_index(docs, emb, **kw)
        assert full.n == index.n_total, (full.n, index.n_total)
        want = [[(h.node.text, round(h.score, 6)) for h in AutoMergingRetriever(full.as_retriever(similarity_top_k=20), full.docstore).retrieve(q)]
                for q in queries]
        # same multiset of leaf rows in both indexes (the order differs: rank-major vs document-major)
        same = sum(1 for g, wv in zip(got, want) if g == wv)
        print(f"{n_docs} documents over {world} rank(s): {index.n_total} leaves; {same}/{len(queries)} queries return the same nodes "
              f"(text) with the same scores as the single-device index over all documents", flush=True)
        assert same == len(queries)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
