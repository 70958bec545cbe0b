"""Diagnostic: where do two tt_gemm_x3 runs over row-permuted inputs differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tensor_truth_amd import _lib
from tensor_truth_amd.encoder_x3 import split_planes
lib = _lib.load_library()
dev = torch.device("cuda:0")
m, n, k = 2048, 1024, 1024
g = torch.Generator().manual_seed(9)
ap, wp = split_planes(torch.randn(m, k, generator=g)).to(dev), split_planes(torch.randn(n, k, generator=g) * 0.03).to(dev)
bias = torch.randn(n, generator=g).to(dev)
res = torch.randn(m, n, generator=g).to(dev)
perm = torch.randperm(m, generator=g).to(dev)
for epi in (0, 1, 2):
    outs = []
    for rep in range(2):
        for a, r in ((ap, res), (ap[perm].contiguous(), res[perm].contiguous())):
            c = torch.empty((m, 2 * n), dtype=torch.bfloat16, device=dev)
            c32 = torch.empty((m, n), device=dev)
            _lib.check(lib.tt_gemm_x3(a.data_ptr(), wp.data_ptr(), bias.data_ptr(), r.data_ptr() if epi == 2 else None,
                                      c.data_ptr() if epi != 2 else None, c32.data_ptr() if epi == 2 else None, m, n, k, epi, None), "x3")
            outs.append(c32 if epi == 2 else c)
    torch.cuda.synchronize()
    same_rerun = torch.equal(outs[0], outs[2])
    d = outs[0][perm] != outs[1]
    print(f"epi {epi}: rerun identical {same_rerun}; permuted mismatches {int(d.sum())} of {d.numel()}",
          "" if epi == 2 else f"(hi plane {int(d[:, :n].sum())}, lo plane {int(d[:, n:].sum())})")
    if d.any():
        idx = d.nonzero()[:5]
        for i, j in idx.tolist():
            print("   ", i, j, float(outs[0][perm][i, j]), float(outs[1][i, j]), "row in tile", int(perm[i]) % 256, i % 256)
