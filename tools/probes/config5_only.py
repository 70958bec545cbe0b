"""Round 6: bench.py's composed BASELINE config 5 leg alone (reference chunk geometry; bf16 pass + default-precision pass).
Usage: python tools/probes/config5_only.py [docs] [small_docs] [words lo-hi] [queries]"""
import json
import sys

import torch

sys.path.insert(0, ".")
docs = sys.argv[1] if len(sys.argv) > 1 else "256"
small = sys.argv[2] if len(sys.argv) > 2 else "0"
words = sys.argv[3] if len(sys.argv) > 3 else "4000-8000"
queries = sys.argv[4] if len(sys.argv) > 4 else "256"
sys.argv = [sys.argv[0], "--config5-docs", docs, "--config5-small-docs", small, "--config5-doc-words", words, "--config5-queries", queries]
import bench as B  # noqa: E402
from tensor_truth_amd.encoder import BGE_M3, BGE_RERANKER_V2_M3  # noqa: E402

args = B.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
r = B.config5_leg(args, dev, BGE_M3, BGE_RERANKER_V2_M3)
print(json.dumps(r))
