"""Debug probe of the fp8 shadow prefilter: survivors per query / per wave, bounds, thresholds (reads the call's workspace)."""
import sys

import torch

sys.path.insert(0, ".")
from tensor_truth_amd import _lib, scan as tscan  # noqa: E402

dev = torch.device("cuda", 0)
n, d, k, nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1_300_000, 1024, 50, 1
g = torch.Generator(device=dev).manual_seed(99)
x = torch.randn((n, d), generator=g, device=dev)
c = (x / x.norm(dim=1, keepdim=True)).to(torch.bfloat16)
del x
q = torch.randn((nq, d), generator=g, device=dev)
q = (q / q.norm(dim=1, keepdim=True)).to(torch.bfloat16)
mods = sys.argv[2] if len(sys.argv) > 2 else ""
if "d" in mods:
    c[5] = c[1_200_000]; c[900_001] = c[123]
if "n" in mods:
    c[40_000:40_064] = float("nan")
if "s" in mods:
    c[1000:1100] *= 3.0
if "t" in mods:
    c[2000:2100] *= 0.01
if "p" in mods:
    sh = tscan.ScanShadow(c[: n // 2].contiguous(), cap_rows=n)
    sh.extend(c, n)
else:
    sh = tscan.ScanShadow(c)
lib = _lib.load_library()
al = lambda b: (b + 255) // 256 * 256  # noqa: E731
off_be = al(n * d)
r16 = (n + 15) // 16 * 16
base = sh.ptr - sh._raw.data_ptr()
be = sh._raw[base + off_be: base + off_be + 4 * n].view(torch.float32)
dn = sh._raw[base + off_be + al(r16 * 4): base + off_be + al(r16 * 4) + 4 * n].view(torch.float32)
print("be mean/max", be.mean().item(), be.max().item(), "dn mean", dn.mean().item())
s8 = sh._raw[base: base + n * d].view(torch.float8_e4m3fn)[: 4 * d].float().view(4, d) / 256
print("dequant vs bf16 row 0..3: max |diff|", (s8 - c[:4].float()).abs().max().item(), " row norm err", (s8 - c[:4].float()).norm(dim=1).tolist(), be[:4].tolist())
s, i, flag = tscan.scan_topk(c, q, k, return_flag=True, shadow=sh)
torch.cuda.synchronize()
print("flag", flag)
# workspace layout (scan_api.hip shadow_plan)
n0 = max(32768, 128 * k, n // 32 // 32 * 32)
n0 = min(n0, n)
stride = ((n0 + 31) // 32 + 31) // 32 * 32
blocks = 256
n_waves = blocks * 8
cap = min(max(n // 16, 65536), 1 << 20, (n + 31) // 32 * 32)
cap = (cap + 31) // 32 * 32
capw = max(64, 4 * cap // n_waves)
ws = tscan._ws.get(dev, 0)
b0 = (ws.data_ptr() + 255) // 256 * 256 - ws.data_ptr()
off = 0
def take(nb):
    global off
    o = off
    off += al(nb)
    return o
o_sample = take(64 * stride * 4); o_ss = take(64 * k * 4); o_si = take(64 * k * 4); o_thr = take(64 * 4)
o_pl = take(2 * (d // 128) * 2048); o_qi = take(32 * 4); o_list = take(nq * n_waves * capw * 4); o_wc = take(nq * n_waves * 4)
o_tab = take(nq * cap * 4); o_tc = take(64 * 4)
f = lambda o, cnt, dt: ws[b0 + o: b0 + o + cnt * 4].view(dt)  # noqa: E731
thr = f(o_thr, 4, torch.float32)
wc = f(o_wc, n_waves, torch.int32)
tc = f(o_tc, 4, torch.int32)
qi = f(o_qi, 32, torch.float32)
print("n0", n0, "cap", cap, "capw", capw, "thr", thr.tolist(), "||q||", qi[0].item())
print("top waves", wc.topk(5))
print("wave counts: sum", int(wc.sum()), "max", int(wc.max()), "mean", wc.float().mean().item(), "table cnt", tc.tolist())
exact = (c.float() @ q[0].float())
print("exact: k-th best", exact.topk(k).values[-1].item(), "rows >= thr", int((exact >= thr[0]).sum()), "rows >= thr - 0.04", int((exact >= thr[0] - 0.04).sum()))
# determinism + reference survivor count from the dequantised shadow
d8 = sh._raw[base: base + n * d].view(torch.float8_e4m3fn)
qf = q[0].float()
s8_ref = torch.empty(n, device=dev)
for lo in range(0, n, 262144):
    hi = min(n, lo + 262144)
    s8_ref[lo:hi] = (d8[lo * d: hi * d].float().view(hi - lo, d) @ qf) / 256.0
ub = s8_ref + qi[0] * be + 1e-4
print("reference survivors (s8_ref + bound >= thr):", int((ub >= thr[0]).sum()))
for rep in range(3):
    tscan.scan_topk(c, q, k, return_flag=True, shadow=sh)
    torch.cuda.synchronize()
    print("rep", rep, "wave sum", int(f(o_wc, n_waves, torch.int32).sum()), "table cnt", int(f(o_tc, 4, torch.int32)[0]))
tab = f(o_tab, cap, torch.int32)[: int(tc[0])].long()
want = (ub >= thr[0]).nonzero().flatten()
got = torch.zeros(n, dtype=torch.bool, device=dev)
got[tab] = True
missing = want[~got[want]]
extra = (got & ~(ub >= thr[0])).nonzero().flatten()
print("missing", missing.numel(), missing[:10].tolist(), "extra", extra.numel(), extra[:10].tolist())
if missing.numel():
    m = missing[:6]
    print("missing rows: s8_ref", s8_ref[m].tolist(), "ub", ub[m].tolist(), "row % 16", (m % 16).tolist(), "group % n_waves", ((m // 16) % n_waves).tolist())
