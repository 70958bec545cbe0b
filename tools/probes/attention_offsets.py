import sys, math, torch
sys.path.insert(0, "/root/repo")
from tensor_truth_amd import _lib
lib=_lib.load_library(); dev=torch.device("cuda:0"); st=torch.cuda.current_stream().cuda_stream
heads, dh, n = 16, 64, 292
H = heads*dh
g = torch.Generator().manual_seed(0)
q = torch.randn(n, H, generator=g).to(torch.bfloat16); k = torch.randn(n, H, generator=g).to(torch.bfloat16); v = torch.randn(n, H, generator=g).to(torch.bfloat16)
qq = q.float().view(n, heads, dh).transpose(0,1); kk = k.float().view(n, heads, dh).transpose(0,1); vv = v.float().view(n, heads, dh).transpose(0,1)
ref = (torch.softmax(qq @ kk.transpose(-1,-2) / math.sqrt(dh), dim=-1) @ vv).transpose(0,1).reshape(n, H)
outs = {}
for off in (0, 4, 1, 7):
    T = 1024
    Q = torch.randn(T, H, generator=g).to(torch.bfloat16); K = torch.randn(T, H, generator=g).to(torch.bfloat16); V = torch.randn(T, H, generator=g).to(torch.bfloat16)
    s0 = 296 + off        # some other sequence's rows in front
    Q[s0:s0+n], K[s0:s0+n], V[s0:s0+n] = q, k, v
    qk = torch.cat([Q, K], 1).contiguous().to(dev); vt = V.view(T // 8, 8, H).permute(0, 2, 1).contiguous().to(dev)
    out = torch.zeros(T, H, dtype=torch.bfloat16, device=dev)
    stt = torch.tensor([0, s0], dtype=torch.int32, device=dev); ln = torch.tensor([s0, n], dtype=torch.int32, device=dev)
    rc = lib.tt_attention_varlen(qk.data_ptr(), 2*H, 0, H, vt.data_ptr(), 8*H, out.data_ptr(), H, stt.data_ptr(), ln.data_ptr(), 2, heads, dh, max(s0, n), st)
    torch.cuda.synchronize()
    o = out[s0:s0+n].float().cpu(); outs[off] = o
    print("off", off, "max err vs fp32 ref", (o - ref).abs().max().item(), "mean", (o - ref).abs().mean().item())
print("off0 vs off4: max", (outs[0]-outs[4]).abs().max().item(), "mean", (outs[0]-outs[4]).abs().mean().item(), "frac differing", (outs[0]!=outs[4]).float().mean().item())
