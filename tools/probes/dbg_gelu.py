import sys, torch
sys.path.insert(0, "/root/repo")
from oracle import encoder as oe
from tensor_truth_amd import _lib
lib=_lib.load_library(); dev=torch.device("cuda:0"); st=torch.cuda.current_stream().cuda_stream
M, N, K = 256, 256, 256
xs = torch.linspace(-8, 8, M).to(torch.bfloat16)
a = torch.zeros(M, K, dtype=torch.bfloat16); a[:, 0] = xs
w = torch.zeros(N, K, dtype=torch.bfloat16); w[:, 0] = 1
bias = torch.linspace(0, 0.06, N)          # sub-bf16 offsets so many distinct pre-activations are hit
c = torch.empty(M, N, dtype=torch.float32)
ad, wd, bd = a.to(dev), w.to(dev), bias.to(dev)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for var in ("4", "5"):
    import os
    lib.tt_gemm_bf16(ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), None, out.data_ptr(), M, N, K, 1, st)
    torch.cuda.synchronize()
    pre = xs.float()[:, None] + bias[None, :]
    want = oe.gelu_erf(pre)
    got = out.float().cpu()
    err = (got - want).abs()
    rel = err / want.abs().clamp_min(1e-30)
    bad = (err > 2.0 ** -7 * want.abs() + 1e-6)
    print("max abs err", err.max().item(), "n bad", bad.sum().item())
    idx = bad.nonzero()[:8].tolist()
    for r, cc in idx: print("x", pre[r, cc].item(), "want", want[r, cc].item(), "got", got[r, cc].item())
    break
