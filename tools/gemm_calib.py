"""Calibration only (not a product path): time torch's library GEMM (hipBLASLt / rocBLAS) beside
tt_gemm_bf16 on the encoder-layer shapes.  python tools/gemm_calib.py [M]"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from tensor_truth_amd import _lib  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 236800 // 256 * 256
lib = _lib.load_library()
dev = torch.device("cuda:0")
shapes = [(3072, 1024, 0, "qkv"), (1024, 1024, 2, "o-proj"), (4096, 1024, 1, "ffn-up"), (1024, 4096, 2, "ffn-down")]
st = torch.cuda.current_stream().cuda_stream
for n, k, epi, name in shapes:
    a = (torch.rand(M, k, device=dev) * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand(n, k, device=dev) * 2 - 1) * 0.05).to(torch.bfloat16)
    bias = torch.zeros(n, device=dev)
    res = torch.randn(M, n, device=dev).to(torch.bfloat16)
    c = torch.empty(M, n, device=dev, dtype=torch.bfloat16)
    bb = bias.to(torch.bfloat16)

    def lib_gemm():
        torch.nn.functional.linear(a, w, bb, )

    def mine():
        rc = lib.tt_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), res.data_ptr() if epi == 2 else None,
                              c.data_ptr(), M, n, k, epi, st)
        assert rc == 0

    out = {}
    for label, fn in (("hipblaslt", lib_gemm), ("tt_gemm", mine)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        out[label] = 2.0 * M * n * k / ms / 1e9
    print(f"{name:9s} M={M} N={n} K={k}: hipblaslt(bias only) {out['hipblaslt']:.0f} TF/s   tt_gemm(epi {epi}) {out['tt_gemm']:.0f} TF/s",
          flush=True)
