import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, bench
from tensor_truth_amd import scan as tscan
import test_scan_gpu as t
dev = torch.device("cuda:0")
for n in (1_250_000,):
    corpus = bench.synth_corpus_shard(n, 1024, 1234, dev)
    queries, planted = t._planted_queries(corpus, 256, 4321 + n % 97, dev)
    s, i, ov = tscan.scan_topk(corpus, queries, 50, return_flag=True)
    bs, bi = t._running_topk_checker(corpus, queries, 51)
    gap50 = (bs[:, :49] - bs[:, 1:50]).min(dim=1).values
    gap51 = (bs[:, :50] - bs[:, 1:51]).min(dim=1).values
    bad = [q for q in range(256) if not torch.equal(i[q].long(), bi[q, :50])]
    print("overflow", ov, "mismatching queries", bad)
    for q in bad:
        pos = (i[q].long() != bi[q, :50]).nonzero().flatten().tolist()
        print(q, "positions", pos, "gap within top-50 %.2e, incl. 50/51 %.2e" % (gap50[q].item(), gap51[q].item()),
              "ours", i[q, pos].tolist(), s[q, pos].tolist(), "checker", bi[q, pos].tolist(), bs[q, pos].tolist(), "51st", bi[q, 50].item(), bs[q, 50].item(),
              "same set", set(i[q].tolist()) == set(bi[q, :50].tolist()))
