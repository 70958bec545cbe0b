"""Round 6: bench.py's reference_defaults block alone (the reference's session defaults over 1 and 3 modules of a resident corpus).
Usage: python tools/probes/reference_defaults.py [rows]"""
import json
import sys

import torch

sys.path.insert(0, ".")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
sys.argv = [sys.argv[0], "--corpus-rows", str(rows)]
import bench as B  # noqa: E402
from tensor_truth_amd.encoder import BGE_M3, BGE_RERANKER_V2_M3  # noqa: E402

args = B.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
shard = B.synth_corpus_shard(rows, 1024, 1234, dev)
print(json.dumps(B.reference_defaults_leg(args, dev, shard, BGE_M3, BGE_RERANKER_V2_M3)))
