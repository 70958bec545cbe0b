"""The vendor library's bf16 GEMM (torch.matmul -> hipBLASLt / rocBLAS) on the ffn-up shape under the power cap, clock and power sampled
like tools/probes/gemm_sustain.py: is the cap specific to this repo's kernel?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

dev = torch.device("cuda", 0)
M, N, K = 236800 // 256 * 256, 4096, 1024
g = torch.Generator(device=dev).manual_seed(0)
for data in ("randn", "const"):
    a = torch.randn((M, K), device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn((N, K), device=dev, generator=g) * 0.03).to(torch.bfloat16)
    if data == "const":
        a.fill_(1.0); w.fill_(0.03125)
    c = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    wt = w.t()
    fn = lambda: torch.matmul(a, wt, out=c)
    fn(); torch.cuda.synchronize()
    flops = 2.0 * M * N * K
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < 1.5:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
    sampler = bench.ClockSampler(0, period_s=0.2).start()
    marks, t0 = [], time.perf_counter()
    while time.perf_counter() - t0 < 5:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); e1.synchronize()
        marks.append(e0.elapsed_time(e1) / 20)
    clock = sampler.stop()
    print(f"torch.matmul bf16 ({data} operands) M={M} N={N} K={K}: {flops / (sum(marks) / len(marks) * 1e-3) / 1e12:.0f} TF/s, sclk median {clock['sclk_mhz_median']} MHz, "
          f"socket power {clock['socket_power_w_mean'] and round(clock['socket_power_w_mean'])} W", flush=True)
