"""Per-shape time of the encoder's five GEMMs in the three arithmetic modes: bf16 (tt_gemm_bf16), split-bf16 (tt_gemm_x3) and
f16c (tt_gemm_f16c), at the bench's row count / 2 (236 800 rows).  "units" = time relative to the bf16 GEMM of the same shape
(split-bf16 is 3 units of MFMA work, f16c 2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tensor_truth_amd import _lib
from tensor_truth_amd.encoder_f16c import quantize_planes
from tensor_truth_amd.encoder_x3 import split_planes

dev = torch.device("cuda", 0)
lib = _lib.load_library()
st = torch.cuda.current_stream(dev).cuda_stream
M = int(sys.argv[1]) if len(sys.argv) > 1 else 236800
M = M // 256 * 256
REPS = 5


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS


g = torch.Generator(device=dev).manual_seed(0)
for name, N, K, epi in (("q,k proj (bias -> 16-bit)", 2048, 1024, 0), ("attn out (residual)", 1024, 1024, 2), ("ffn up (GELU)", 4096, 1024, 1),
                        ("ffn down (residual)", 1024, 4096, 2)):
    a = torch.randn((M, K), device=dev, generator=g)
    w = torch.randn((N, K), device=dev, generator=g) * 0.03
    bias = torch.randn(N, device=dev, generator=g)
    res32 = torch.randn((M, N), device=dev, generator=g) if epi == 2 else None
    flops = 2.0 * M * N * K
    # bf16
    a16, w16 = a.to(torch.bfloat16), w.to(torch.bfloat16)
    res16 = res32.to(torch.bfloat16) if epi == 2 else None
    c16 = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    t_b = timed(lambda: lib.tt_gemm_bf16(a16.data_ptr(), w16.data_ptr(), bias.data_ptr(), res16.data_ptr() if epi == 2 else None, c16.data_ptr(), M, N, K, epi, st))
    del a16, w16, c16, res16
    # split-bf16
    ap3, wp3 = split_planes(a), split_planes(w)
    c3 = torch.empty((M, 2 * N), dtype=torch.bfloat16, device=dev) if epi != 2 else None
    c32 = torch.empty((M, N), dtype=torch.float32, device=dev) if epi == 2 else None
    t_3 = timed(lambda: lib.tt_gemm_x3(ap3.data_ptr(), wp3.data_ptr(), bias.data_ptr(), res32.data_ptr() if epi == 2 else None,
                                       c3.data_ptr() if c3 is not None else None, c32.data_ptr() if c32 is not None else None, M, N, K, epi, st))
    del ap3, wp3, c3
    # f16c
    apc, asc = quantize_planes(a, False)
    wpc, wsc = quantize_planes(w, True)
    if epi == 1:
        cc = torch.empty((M, 4 * N), dtype=torch.uint8, device=dev)
        cs = torch.empty(int(lib.tt_f16c_scale_bytes(M, N, 0)), dtype=torch.uint8, device=dev)
    elif epi == 0:
        cc, cs = torch.empty((M, N), dtype=torch.float16, device=dev), None
    else:
        cc, cs = c32, None
    t_c = timed(lambda: lib.tt_gemm_f16c(apc.data_ptr(), asc.data_ptr(), wpc.data_ptr(), wsc.data_ptr(), bias.data_ptr(),
                                         res32.data_ptr() if epi == 2 else None, cc.data_ptr(), cs.data_ptr() if cs is not None else None, M, N, K, epi, st))
    print(f"{name:28s} M={M} N={N} K={K}: bf16 {t_b:.3f} ms ({flops / t_b / 1e9:.0f} TF/s) | split-bf16 {t_3:.3f} ms = {t_3 / t_b:.2f} units "
          f"({3 * flops / t_3 / 1e9:.0f} TF/s of bf16 MFMA work) | f16c {t_c:.3f} ms = {t_c / t_b:.2f} units ({2 * flops / t_c / 1e9:.0f} TF/s-equivalent)", flush=True)
    del apc, asc, wpc, wsc, cc, cs, c32, a, w, res32
    torch.cuda.empty_cache()
