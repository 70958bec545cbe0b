"""Bring-up diagnostic of tt_gemm_f16c: which part of the K stream is wrong?  (a) operands exactly representable in fp16: only
the hi.hi part contributes; (b) general a, fp16-exact w: + lo8(a).x8(w); (c) fp16-exact a, general w: + x8(a).lo8(w)."""
import sys

sys.path.insert(0, ".")
import torch

from oracle import f16c as of
from tensor_truth_amd import _lib
from tensor_truth_amd.encoder_f16c import quantize_planes

dev = torch.device("cuda:0")
lib = _lib.load_library()
st = torch.cuda.current_stream(dev).cuda_stream
m, n, k = 256, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator().manual_seed(1)
a = torch.randn(m, k, generator=g)
w = torch.randn(n, k, generator=g) * 0.05
bias = torch.zeros(n)
res = torch.zeros(m, n)
for name, aa, ww in (("fp16-exact a and w (hi.hi only)", a.half().float(), w.half().float()),
                     ("general a, fp16-exact w", a, w.half().float()),
                     ("fp16-exact a, general w", a.half().float(), w),
                     ("general", a, w)):
    ap, asc = quantize_planes(aa.to(dev), False)
    wp, wsc = quantize_planes(ww.to(dev), True)
    out = torch.empty((m, n), dtype=torch.float32, device=dev)
    rc = lib.tt_gemm_f16c(ap.data_ptr(), asc.data_ptr(), wp.data_ptr(), wsc.data_ptr(), bias.to(dev).data_ptr(), res.to(dev).data_ptr(),
                          out.data_ptr(), None, m, n, k, 2, st)
    assert rc == 0, lib.tt_last_error()
    torch.cuda.synchronize()
    want = of.matmul(aa, ww)
    ah, a8, al = of.dequant(of.quantize(aa, False), False)
    wh, w8, wl = of.dequant(of.quantize(ww, True), True)
    parts = {"hi.hi": ah @ wh.T, "x8.lo8w": a8 @ wl.T, "lo8.x8w": al @ w8.T}
    got = out.cpu()
    d = got - want
    print(f"{name}: max |got - want| {d.abs().max():.3e} (|want| max {want.abs().max():.3e}); parts max: " +
          ", ".join(f"{k_} {v.abs().max():.2e}" for k_, v in parts.items()))
    # is the difference explained by a missing / doubled part?
    for k_, v in parts.items():
        for f in (-1.0, 1.0):
            r = (d - f * v).abs().max().item()
            if r < 1e-3 * max(v.abs().max().item(), 1e-9) + 1e-6:
                print(f"    -> difference = {f:+.0f} x {k_}")
    print("    first row got", got[0, :4].tolist(), "want", want[0, :4].tolist())

print("---- per-block magnitudes")
for name, sa_, sw_ in (("a blocks vary", True, False), ("w blocks vary", False, True), ("a rows vary", "rows", False), ("w rows vary", False, "rows")):
    aa, ww = a.clone(), w.clone()
    if sa_ is True:
        aa = aa * torch.exp2(torch.randint(-6, 6, (m, k // 32), generator=g).float()).repeat_interleave(32, 1)
    elif sa_ == "rows":
        aa = aa * torch.exp2(torch.randint(-6, 6, (m, 1), generator=g).float())
    if sw_ is True:
        ww = ww * torch.exp2(torch.randint(-4, 4, (n, k // 32), generator=g).float()).repeat_interleave(32, 1)
    elif sw_ == "rows":
        ww = ww * torch.exp2(torch.randint(-4, 4, (n, 1), generator=g).float())
    ap, asc = quantize_planes(aa.to(dev), False)
    wp, wsc = quantize_planes(ww.to(dev), True)
    out = torch.empty((m, n), dtype=torch.float32, device=dev)
    rc = lib.tt_gemm_f16c(ap.data_ptr(), asc.data_ptr(), wp.data_ptr(), wsc.data_ptr(), bias.to(dev).data_ptr(), res.to(dev).data_ptr(),
                          out.data_ptr(), None, m, n, k, 2, st)
    assert rc == 0, lib.tt_last_error()
    torch.cuda.synchronize()
    got = out.cpu()
    want = of.matmul(aa, ww)
    exact = (aa.double() @ ww.double().T).float()
    h16 = (aa.half().double() @ ww.half().double().T).float()
    sc = (aa.abs().double() @ ww.abs().double().T).float()
    print(f"{name}: vs planes oracle {((got - want).abs() / sc).max():.2e}, vs exact {((got - exact).abs() / sc).max():.2e}, "
          f"hi.hi alone vs exact {((h16 - exact).abs() / sc).max():.2e}, oracle vs exact {((want - exact).abs() / sc).max():.2e}")
    bad = ((got - exact).abs() / sc)
    i, j = divmod(int(bad.argmax()), n)
    print(f"    worst at row {i} col {j}: got {got[i, j]:.6f} exact {exact[i, j]:.6f} hi.hi {h16[i, j]:.6f}")
