"""How many rows would an fp8 shadow of the corpus pass on to an exact re-score?  (Round 4 costing; BUILT in round 6: csrc/shadow.hip)
Emulation with torch on the bench's synthetic corpus: rows and queries quantised to e4m3 with one scale per vector, the
prefilter score s8 = <q8, c8> (exact in fp32 up to accumulation rounding), the RIGOROUS upper bound on the true bf16 score
    s <= s8 + |q - q8| . |c|  + |q8| . |c - c8|          (Cauchy-Schwarz on both error terms; all four norms known exactly)
and the number of rows whose bound reaches the exact top-k threshold.  Usage: python tools/probes/fp8_prefilter_bound.py [rows] [queries] [k]"""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 32
k = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = torch.device("cuda", 0)
corpus = bench.synth_corpus_shard(rows, 1024, 1234, dev)                      # bf16, unit rows
g = torch.Generator(device=dev).manual_seed(4321)
# half planted neighbours (cos ~ 0.89 to a corpus row), half random directions: SURVEY.md section 8d's query mix
qs = torch.nn.functional.normalize(torch.randn(nq, 1024, device=dev, generator=g), dim=1)
planted = corpus[torch.randint(0, rows, (nq // 2,), device=dev, generator=g)].float()
qs[: nq // 2] = torch.nn.functional.normalize(planted + 0.5 * qs[: nq // 2], dim=1)
q = qs.to(torch.bfloat16)


def quant(x):          # e4m3 with one scale per row: absmax -> 448
    xf = x.float()
    s = xf.abs().amax(dim=1, keepdim=True).clamp_min(1e-30) / 448.0
    x8 = (xf / s).to(torch.float8_e4m3fn).float() * s
    return x8, (xf - x8).norm(dim=1), xf.norm(dim=1), x8.norm(dim=1)


q8, eq, _, nq8 = quant(q)
exact_thr = torch.empty(nq, device=dev)
surv = torch.zeros(nq, device=dev)
surv_loose = torch.zeros(nq, device=dev)
chunk = 250_000
# pass A: the exact top-k threshold of every query (what the sampled threshold converges to from below)
tops = []
for lo in range(0, rows, chunk):
    s = q.float() @ corpus[lo:lo + chunk].float().T
    tops.append(s.topk(k, dim=1).values)
thr = torch.cat(tops, dim=1).topk(k, dim=1).values[:, -1]
err_c = []
for lo in range(0, rows, chunk):
    c = corpus[lo:lo + chunk]
    c8, ec, nc, _ = quant(c)
    s8 = q8 @ c8.T
    bound = s8 + eq[:, None] * nc[None, :] + nq8[:, None] * ec[None, :]
    true = q.float() @ c.float().T
    assert (bound >= true - 1e-5).all(), "bound violated"
    surv += (bound >= thr[:, None]).sum(dim=1)
    surv_loose += (bound >= (thr - 0.02)[:, None]).sum(dim=1)          # a sampled threshold sits a little below the exact one
    err_c.append(ec)
ec = torch.cat(err_c)
print(f"{rows} rows x 1024, {nq} queries (half planted), top-{k}; quantisation error norms: rows mean {ec.mean():.4f} max {ec.max():.4f}, queries mean {eq.mean():.4f}")
print(f"exact top-{k} thresholds: min {thr.min():.3f} mean {thr.mean():.3f} max {thr.max():.3f}")
print(f"rows whose rigorous bound reaches the exact threshold: mean {surv.mean():.0f} per query ({100 * surv.mean() / rows:.3f} % of the rows), max {surv.max():.0f}")
print(f"... a threshold 0.02 lower (sampled): mean {surv_loose.mean():.0f} per query ({100 * surv_loose.mean() / rows:.3f} %), max {surv_loose.max():.0f}")
print(f"re-score traffic at the mean: {surv_loose.mean() * 2048 / 1e6:.1f} MB per query against {rows * 1024 / 1e6:.0f} MB of fp8 rows streamed")
