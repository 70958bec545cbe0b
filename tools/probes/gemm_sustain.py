"""Keep ONE GEMM shape running for a few seconds (for clock / power sampling beside it): python gemm_sustain.py bf16|x3 SECONDS.
Prints the average rate over the whole window and over its last third."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tensor_truth_amd import _lib
from tensor_truth_amd.encoder_x3 import split_planes

mode, secs = sys.argv[1], float(sys.argv[2])
dev = torch.device("cuda", 0)
lib = _lib.load_library()
st = torch.cuda.current_stream(dev).cuda_stream
M, N, K = 236800 // 256 * 256, 4096, 1024
g = torch.Generator(device=dev).manual_seed(0)
a = torch.randn((M, K), device=dev, generator=g)
w = torch.randn((N, K), device=dev, generator=g) * 0.03
data = sys.argv[3] if len(sys.argv) > 3 else "randn"
if data == "zero":          # no operand toggling at all: what the kernel's STRUCTURE sustains when the MFMA data path draws no switching power
    a.zero_(); w.zero_()
elif data == "const":       # every element the same value
    a.fill_(1.0); w.fill_(0.03125)
bias = torch.randn(N, device=dev, generator=g)
if mode == "bf16":
    a16, w16 = a.to(torch.bfloat16), w.to(torch.bfloat16)
    c = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    fn = lambda: lib.tt_gemm_bf16(a16.data_ptr(), w16.data_ptr(), bias.data_ptr(), None, c.data_ptr(), M, N, K, 0, st)
    units = 1.0
else:
    ap, wp = split_planes(a), split_planes(w)
    c = torch.empty((M, 2 * N), dtype=torch.bfloat16, device=dev)
    fn = lambda: lib.tt_gemm_x3(ap.data_ptr(), wp.data_ptr(), bias.data_ptr(), None, c.data_ptr(), None, M, N, K, 0, st)
    units = 3.0
fn(); torch.cuda.synchronize()
flops = 2.0 * M * N * K * units
import bench
sampler = bench.ClockSampler(0, period_s=0.2)
t_warm = time.perf_counter()
while time.perf_counter() - t_warm < 1.5:      # let the clock settle under the cap before sampling
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
sampler.start()
t0 = time.perf_counter()
marks = []
while time.perf_counter() - t0 < secs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); e1.synchronize()
    marks.append(e0.elapsed_time(e1) / 20)
clock = sampler.stop()
third = marks[len(marks) * 2 // 3:]
print(f"{mode} ({data} operands): {len(marks) * 20} launches, {flops / (sum(marks) / len(marks) * 1e-3) / 1e12:.0f} TF/s of MFMA work over the window, "
      f"{flops / (sum(third) / len(third) * 1e-3) / 1e12:.0f} in its last third; sclk median {clock['sclk_mhz_median']} MHz "
      f"({clock['sclk_mhz_min']}-{clock['sclk_mhz_max']}), socket power {clock['socket_power_w_mean'] and round(clock['socket_power_w_mean'])} W "
      f"[{os.environ.get('TT_LIB_NAME', 'libtt_hip.so')} ENERGY={os.environ.get('TT_GEMM_ENERGY')} A0={os.environ.get('TT_GEMM_DEBUG_A0')} "
      f"TRAFFIC={os.environ.get('TT_GEMM_DEBUG_TRAFFIC')}]")
