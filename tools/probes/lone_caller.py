"""Round 6: the reference's own call -- ONE caller, one un-batched query through the plugin surface (retrieve -> postprocess_nodes)
over a resident 10M x 1024 corpus -- with bench.py's per-stage breakdown, in bf16 and in the default (reference) precision.
Usage: python tools/probes/lone_caller.py [rows] [top_k] [top_n] [modes: bf16,default]"""
import json
import sys

import torch

sys.path.insert(0, ".")
argv = sys.argv[1:]
rows = int(argv[0]) if argv else 10_000_000
top_k = int(argv[1]) if len(argv) > 1 else 50
top_n = int(argv[2]) if len(argv) > 2 else 10
modes = (argv[3] if len(argv) > 3 else "bf16,default").split(",")
sys.argv = [sys.argv[0], "--corpus-rows", str(rows), "--top-k", str(top_k), "--top-n", str(top_n), "--surface-threads", "1",
            "--surface-queries", "40"]
import bench as B  # noqa: E402
from tensor_truth_amd.encoder import BGE_M3, BGE_RERANKER_V2_M3  # noqa: E402

args = B.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
shard = B.synth_corpus_shard(rows, 1024, 1234, dev)
for mode in modes:
    r = B.surface_leg(args, dev, shard, BGE_M3, BGE_RERANKER_V2_M3, default_precision=(mode == "default"), n_queries=40)
    keep = {k: r[k] for k in ("precision", "single_caller_ms_per_query", "single_caller_ms_per_query_with_leaf_token_ids", "lone_caller_breakdown")}
    print(json.dumps({"mode": mode, "rows": rows, "top_k": top_k, "top_n": top_n, **keep}))
