import sys, torch
sys.path.insert(0, "/root/repo")
from oracle import encoder as oe
from tensor_truth_amd import _lib
lib=_lib.load_library(); dev=torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
x = (torch.randn(300, 1024, generator=g) * torch.rand(300, 1, generator=g) * 3).to(torch.bfloat16)
want_q, want_s = oe.quantize_rows_e4m3(x.float())
xd=x.to(dev); q=torch.empty(300,1024,dtype=torch.uint8,device=dev); s=torch.empty(300,device=dev)
lib.tt_quantize_rows_fp8(xd.data_ptr(),300,1024,q.data_ptr(),s.data_ptr(),torch.cuda.current_stream().cuda_stream)
got=q.cpu().view(torch.float8_e4m3fn).float()
bad=(got!=want_q).nonzero()
print(len(bad))
amax=x.float().abs().amax(1,keepdim=True); inv=448.0/amax
for r,c in bad[:20].tolist():
    print(r,c,"x",x[r,c].item(),"scaled",(x[r,c].float()*inv[r,0]).item(),"got",got[r,c].item(),"want",want_q[r,c].item())
