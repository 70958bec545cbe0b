"""Does the plugin surface hold its memory under sustained load?  12 rounds x 192 queries from 24 request threads through
retrieve() + postprocess_nodes() (1M x 1024 corpus, 6-layer models so a round takes seconds): device memory allocated /
reserved, pinned staging slots and host RSS after every round, and that every round returns the same answers."""
import os
import resource
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from tensor_truth_amd import encoder as enc_mod  # noqa: E402
from tensor_truth_amd.embedding import HipHuggingFaceEmbedding  # noqa: E402
from tensor_truth_amd.encoder import BGE_M3, BGE_RERANKER_V2_M3, EncoderConfig  # noqa: E402
from tensor_truth_amd.rerank import HipSentenceTransformerRerank  # noqa: E402
from tensor_truth_amd.schema import QueryBundle  # noqa: E402
from tensor_truth_amd.sharded_index import ShardedHipVectorIndex  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    layers = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    emb_cfg = EncoderConfig(**{**BGE_M3.__dict__, "layers": layers})
    rr_cfg = EncoderConfig(**{**BGE_RERANKER_V2_M3.__dict__, "layers": layers})
    rows = bench.synth_corpus_shard(1_000_000, 1024, 1234, dev)
    # TOKENIZER=unigram-250k (round 5): the trained sub-word tokenizer; coalesced batches then tokenise their pairs in worker processes
    # (ingest_workers.PairTokenizerPool) -- the answers must still be the first round's, bit for bit, whatever batches form
    class A:
        query_len, chunk_len = 32, 256

    texts = bench.surface_texts(A(), os.environ.get("TOKENIZER", "hash"))
    tk_kw = {"tokenizer": texts.tokenizer} if texts.tokenizer is not None else {}
    emb = HipHuggingFaceEmbedding("BAAI/bge-m3", device=str(dev), embed_batch_size=128,
                                  model_kwargs={"encoder_config": emb_cfg, "synthetic_seed": 1, **tk_kw})
    rr = HipSentenceTransformerRerank(model="BAAI/bge-reranker-v2-m3", top_n=10, device=str(dev), batch_pairs=4096,
                                      model_kwargs={"encoder_config": rr_cfg, "synthetic_seed": 2, **tk_kw})
    index = ShardedHipVectorIndex(1024, rows, 0, rows.shape[0], bench._RowIds(rows.shape[0]), bench._SynthDocstore(256, texts.chunk),
                                  embed_model=emb, score_mode="cosine")
    retr = index.as_retriever(similarity_top_k=50, max_batch=64)
    queries = [texts.query(10_000_000_000 + i) for i in range(192)]

    def one(q):
        nodes = retr.retrieve(q)
        return [(x.node.id_, round(x.score, 6)) for x in rr.postprocess_nodes(nodes, query_bundle=QueryBundle(query_str=q))]

    first = None
    for rnd in range(12):
        dt, res = bench._run_threads(24, queries, one)
        torch.cuda.synchronize()
        same = "first" if first is None else ("same answers" if res == first else "ANSWERS DIFFER")
        first = first or res
        print(f"round {rnd:2d}: {len(queries) / dt:6.1f} q/s  device allocated {torch.cuda.memory_allocated(dev) / 2**30:6.2f} GiB  "
              f"reserved {torch.cuda.memory_reserved(dev) / 2**30:6.2f} GiB  staging slots {len(enc_mod._stager.slots)}  "
              f"host RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20:5.2f} GiB  {same}", flush=True)
        assert same != "ANSWERS DIFFER"


if __name__ == "__main__":
    main()
