"""256-query scan over 10M x 1024: time per batch and per stage under the current environment switches."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from tensor_truth_amd import _lib, scan as tscan

dev = torch.device("cuda", 0)
n = int(os.environ.get("ROWS", "10000000"))
corpus = bench.synth_corpus_shard(n, 1024, 1234, dev)
lib = _lib.load_library()
for nq in (256, 128, 96):
    q = torch.nn.functional.normalize(torch.randn(nq, 1024, device=dev, generator=torch.Generator(device=dev).manual_seed(1)), dim=1).to(torch.bfloat16)
    tscan.scan_topk(corpus, q, 50); torch.cuda.synchronize()
    lib.tt_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(5):
        tscan.scan_topk(corpus, q, 50)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    out = []
    for name, kid in (("filter", 1), ("sample", 2), ("select", 3), ("tail", 7)):
        ms, cnt = ctypes.c_double(0), ctypes.c_int(0)
        lib.tt_prof_read(kid, ctypes.byref(ms), ctypes.byref(cnt))
        out.append(f"{name} {ms.value / 5:.3f} ms ({cnt.value // 5})")
    lib.tt_prof_enable(0)
    print(f"Q={nq}: {dt * 1e3:.3f} ms per batch | " + " | ".join(out) + f" | env PERSIST={os.environ.get('TT_SCAN_GEMM_PERSIST')} N0={os.environ.get('TT_SCAN_GEMM_N0')} GEMM={os.environ.get('TT_SCAN_GEMM')}")
