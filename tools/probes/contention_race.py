"""Do the kernels with hand-counted waits stay bit-reproducible when ANOTHER stream loads the memory system?  (Found by the
plugin-surface soak after the retrievers got their own stream: reranked scores changed by ~5e-3 run to run.)  A scan loop on a
second stream keeps HBM busy while each building block is repeated on the main stream and compared with its first result."""
import sys, threading, time
sys.path.insert(0, ".")
import numpy as np, torch
import bench
from tensor_truth_amd import _lib, scan as tscan
from tensor_truth_amd.encoder import BGE_RERANKER_V2_M3, Encoder, EncoderConfig, EncoderWeights, pack_token_matrix, synthetic_state_device

lib = _lib.load_library(); dev = torch.device("cuda", 0); torch.cuda.set_device(0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
corpus = bench.synth_corpus_shard(rows, 1024, 1234, dev)
q = torch.nn.functional.normalize(torch.randn(32, 1024, device=dev), dim=1).to(torch.bfloat16)
stop = False
hi = torch.cuda.Stream(device=dev, priority=-1)
mode = sys.argv[2] if len(sys.argv) > 2 else "scan"

def load():
    torch.cuda.set_device(0)
    with torch.cuda.stream(hi):
        while not stop:
            if mode == "scan":
                tscan.scan_topk(corpus, q, 50)
            hi.synchronize()

def st(): return torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(1)
results = {}
def check(name, fn, reps):
    ref = fn().clone(); bad = 0
    for _ in range(reps):
        if not torch.equal(fn(), ref): bad += 1
    torch.cuda.synchronize()
    results[name] = results.get(name, []) + [bad]
    print(f"  {name:36s} {reps} repetitions, differing from the first: {bad}", flush=True)

M = 14848
a = torch.randn(M, 1024, device=dev, generator=g).to(torch.bfloat16); a4 = torch.randn(M, 4096, device=dev, generator=g).to(torch.bfloat16)
w3 = (torch.randn(3072, 1024, device=dev, generator=g) * 0.03).to(torch.bfloat16); w1 = (torch.randn(1024, 1024, device=dev, generator=g) * 0.03).to(torch.bfloat16)
wu = (torch.randn(4096, 1024, device=dev, generator=g) * 0.03).to(torch.bfloat16); wd = (torch.randn(1024, 4096, device=dev, generator=g) * 0.03).to(torch.bfloat16)
b3, b1, bu = torch.randn(3072, device=dev), torch.randn(1024, device=dev), torch.randn(4096, device=dev)
r = torch.randn(M, 1024, device=dev, generator=g).to(torch.bfloat16)
def gemm(A, W, B, R, epi):
    m, k = A.shape; n = W.shape[0]
    c = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    rc = lib.tt_gemm_bf16(A.data_ptr(), W.data_ptr(), B.data_ptr(), R.data_ptr() if epi == 2 else None, c.data_ptr(), m, n, k, epi, st())
    assert rc == 0
    return c
gamma, beta = torch.ones(1024, device=dev), torch.zeros(1024, device=dev)
def ln():
    o = torch.empty_like(a)
    assert lib.tt_layernorm_bf16(a.data_ptr(), o.data_ptr(), gamma.data_ptr(), beta.data_ptr(), M, 1024, 1e-5, st()) == 0
    return o
cfg = EncoderConfig(**{**BGE_RERANKER_V2_M3.__dict__, "layers": 6})
rr = Encoder(EncoderWeights(cfg, synthetic_state_device(cfg, dev, seed=2), dev))
rng = np.random.default_rng(3)
pairs = rng.integers(4, cfg.vocab_size, size=(350, 292), dtype=np.int32); pairs[:, 0] = 0; pairs[:, -1] = 2
batch = pack_token_matrix(pairs, cfg)

# attention on the rerank shape: 350 sequences x 292 tokens, 16 heads x 64 (Q, K planes [T][2H], V in the V8 layout)
T = batch.n_rows
qk = (torch.randn(T, 2048, device=dev, generator=g) * 0.8).to(torch.bfloat16)
vt = torch.randn(T // 8, 1024, 8, device=dev, generator=g).to(torch.bfloat16)
starts = torch.from_numpy(batch.seq_start).to(dev); lens = torch.from_numpy(batch.seq_len).to(dev)
def att():
    o = torch.zeros(T, 1024, dtype=torch.bfloat16, device=dev)
    assert lib.tt_attention_varlen(qk.data_ptr(), 2048, 0, 1024, vt.data_ptr(), 8192, o.data_ptr(), 1024, starts.data_ptr(), lens.data_ptr(),
                                   350, 16, 64, 292, st()) == 0
    return o
idle_ref = {}
for phase in ("idle", "loaded"):
    print(f"== {phase} ({'nothing else on the GPU' if phase == 'idle' else 'scan loop over %d rows on a high-priority stream' % rows})", flush=True)
    t = None
    if phase == "loaded":
        t = threading.Thread(target=load); t.start(); time.sleep(0.3)
    check("gemm qkv-shape bias (one-tile)", lambda: gemm(a, w3, b3, r, 0), 60)
    check("gemm o-proj + residual", lambda: gemm(a, w1, b1, r, 2), 60)
    check("gemm ffn-up gelu (persistent)", lambda: gemm(a, wu, bu, r, 1), 60)
    check("gemm ffn-down + residual (K=4096)", lambda: gemm(a4, wd, b1, r, 2), 60)
    check("layernorm", ln, 60)
    check("attention 350 x 292", att, 60)
    check("full forward (hidden states)", lambda: rr.forward_packed(batch)[0], 30)
    check("cls forward (CLS tail)", lambda: rr.cls_hidden_packed(batch)[0][:350], 30)
    check("reranker forward, 350 pairs x 292", lambda: rr.rerank_packed(batch), 40)
    cur = rr.rerank_packed(batch).clone()
    if phase == "idle":
        idle_ref["s"] = cur
    else:
        d = (cur - idle_ref["s"]).abs()
        print(f"  loaded vs idle reranker scores: {int((d > 0).sum())} of {d.numel()} differ, max |diff| {d.max().item():.2e}", flush=True)
    if t is not None:
        stop = True; t.join()
