"""Multi-module retrieval latency: one segmented pass vs one tt_scan_topk per module (same stream, no threads).
Usage: python tools/probes/segmented_scan.py [rows_per_module] [modules] [queries]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from oracle import scan as osc  # noqa: E402  (synthetic data only)
from tensor_truth_amd import scan as tscan  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
mods = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
corpus = torch.randn(rows * mods, 1024, generator=g, device=dev)
corpus = (corpus / corpus.norm(dim=1, keepdim=True)).to(torch.bfloat16)
q = corpus[torch.arange(nq, device=dev) * 977 % corpus.shape[0]].contiguous()
offs = [i * rows for i in range(mods + 1)]
parts = [corpus[offs[i]:offs[i + 1]] for i in range(mods)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


t_seg = timed(lambda: tscan.scan_topk_segmented(corpus, q, 10, offs))
t_per = timed(lambda: [tscan.scan_topk(p, q, 10, check_overflow=False) for p in parts])
t_chk = timed(lambda: [tscan.scan_topk(p, q, 10) for p in parts])
gb = corpus.numel() * 2 / 1e9
print(f"{mods} modules x {rows} rows, {nq} queries, k=10: segmented {t_seg:.3f} ms ({gb / t_seg * 1e3:.0f} GB/s), "
      f"per-module {t_per:.3f} ms, per-module with overflow check (product retriever path) {t_chk:.3f} ms")
