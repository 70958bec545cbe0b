"""Shader clock and socket power while the varlen attention kernel runs for ~3 s (1600 sequences x 292 tokens x 16 heads x 64): is the kernel
at the power cap, and does removing its memory operations (TT_ATT_RES_ABL, diagnostic library) raise the clock?
Usage (one process per variant; the diagnostic library reads its switches once):
  TT_LIB_NAME=libtt_hip_diag.so TT_ATT_RESIDENT=1 TT_ATT_RES_ABL=7 python tools/probes/attention_power.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from tensor_truth_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lib = _lib.load_library()
n_seq, L, heads, dh = 1600, int(os.environ.get("LEN", "292")), 16, 64
H = heads * dh
stride = (L + 7) // 8 * 8
T = (n_seq * stride + 255) // 256 * 256
g = torch.Generator(device=dev).manual_seed(1)
qk = (torch.rand((T, 2 * H), generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
vt = (torch.rand((T // 8, H * 8), generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
out = torch.empty((T, H), dtype=torch.bfloat16, device=dev)
ss = (torch.arange(n_seq, dtype=torch.int32) * stride).to(dev)
sl = torch.full((n_seq,), L, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream


def call():
    rc = lib.tt_attention_varlen(qk.data_ptr(), 2 * H, 0, H, vt.data_ptr(), 8 * H, out.data_ptr(), H, ss.data_ptr(), sl.data_ptr(), n_seq, heads, dh, L, st)
    assert rc == 0, lib.tt_last_error()


for _ in range(50):
    call()
torch.cuda.synchronize()
t_w = time.perf_counter()
while time.perf_counter() - t_w < 1.5:
    for _ in range(100):
        call()
    torch.cuda.synchronize()
s = bench.ClockSampler(0, period_s=0.2).start()
n, t0 = 0, time.perf_counter()
while time.perf_counter() - t0 < 3.0:
    for _ in range(100):
        call()
    n += 100
    torch.cuda.synchronize()
dt = time.perf_counter() - t0
c = s.stop()
tag = f"TT_ATT_RESIDENT={os.environ.get('TT_ATT_RESIDENT', '-')} TT_ATT_RES_ABL={os.environ.get('TT_ATT_RES_ABL', '-')}"
print(f"{tag}: {dt / n * 1e3:.3f} ms per launch | sclk median {c['sclk_mhz_median']} MHz ({c['sclk_mhz_min']}-{c['sclk_mhz_max']}), "
      f"socket power {c['socket_power_w_mean'] and round(c['socket_power_w_mean'])} W", flush=True)
