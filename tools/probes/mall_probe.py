"""Does a consumer that starts where its producer just finished hit the memory-side cache?  Producer = a device copy
that writes x front to back; consumer = LayerNorm reading x front-to-back or back-to-front."""
import sys, os, subprocess
import torch
sys.path.insert(0, ".")
from tensor_truth_amd import _lib
lib=_lib.load_library(); dev=torch.device("cuda:0"); st=torch.cuda.current_stream().cuda_stream
H=1024
for T in (118400, 236800, 473600):
    src=torch.randn(T,H,device=dev).to(torch.bfloat16); x=torch.empty_like(src); y=torch.empty_like(src)
    g=torch.ones(H,device=dev); b=torch.zeros(H,device=dev)
    junk=torch.empty(2_000_000_000, dtype=torch.uint8, device=dev)
    def run():
        tot=0
        for _ in range(10):
            junk.fill_(1)            # flush
            x.copy_(src)             # producer
            e0,e1=torch.cuda.Event(True),torch.cuda.Event(True)
            e0.record(); lib.tt_layernorm_bf16(x.data_ptr(), y.data_ptr(), g.data_ptr(), b.data_ptr(), T, H, 1e-5, st); e1.record()
            torch.cuda.synchronize(); tot+=e0.elapsed_time(e1)
        return tot/10
    run()
    print(f"T={T} ({T*H*2/1e6:.0f} MB) TT_LN_NT={os.environ.get('TT_LN_NT','5')}: {run():.4f} ms", flush=True)
