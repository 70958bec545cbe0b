"""Does a HIP graph make ONE query's embedding faster?  The forward of a 34-token query is ~170 launches of 5-10 us kernels
(tools/probes/query_embed_launch_bound.py).  Captures tt_encoder_forward_cls + tt_embed_pool for a fixed (1 sequence, 64 rows)
shape on static buffers with torch.cuda.CUDAGraph and compares replay against the eager path, bit for bit and in time.
Usage: python tools/probes/query_embed_graph.py [tokens]"""
import ctypes
import statistics
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from tensor_truth_amd import _lib  # noqa: E402
from tensor_truth_amd.encoder import BGE_M3, Encoder, EncoderWeights, pack_tokens, synthetic_state_device  # noqa: E402

tokens = int(sys.argv[1]) if len(sys.argv) > 1 else 34
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lib = _lib.load_library()
cfg = BGE_M3
enc = Encoder(EncoderWeights(cfg, synthetic_state_device(cfg, dev, seed=1), dev))
rng = np.random.default_rng(5)


def query(n):
    return [0] + rng.integers(4, cfg.vocab_size, n - 2).tolist() + [2]


def timed(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts), min(ts)


q = query(tokens)
batch = pack_tokens([q], cfg)
eager = lambda: enc.embed_packed(batch)[0]            # noqa: E731
ref = eager().clone()
print(f"one query of {tokens} tokens: n_rows {batch.n_rows}, max_len {batch.max_len}")
print("eager embed_packed (upload + forward + pool), sync after each: median %.3f ms, min %.3f ms" % timed(eager))

# static buffers for a graph of shape (1 sequence, n_rows rows, keys up to MAXLEN)
MAXLEN = 64
H = cfg.hidden
n_rows = batch.n_rows
ints = torch.zeros(2 * n_rows + 2 * 64, dtype=torch.int32, device=dev)
ids_d, pos_d, st_d, ln_d = ints[:n_rows], ints[n_rows:2 * n_rows], ints[2 * n_rows:2 * n_rows + 64], ints[2 * n_rows + 64:]
host = torch.empty(ints.numel(), dtype=torch.int32, pin_memory=True)
cls = torch.empty((256, H), dtype=torch.bfloat16, device=dev)
out = torch.empty((1, H), dtype=torch.float32, device=dev)
out16 = torch.empty((1, H), dtype=torch.bfloat16, device=dev)
rows = torch.zeros(1, dtype=torch.int32, device=dev)
need = lib.tt_encoder_cls_workspace_bytes(ctypes.byref(enc.w.struct), n_rows, 1)
ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
base = (ws.data_ptr() + 255) // 256 * 256


def fill(b):
    hn = host.numpy()
    hn[:] = 0
    hn[:n_rows] = b.ids
    hn[n_rows:2 * n_rows] = b.pos
    hn[2 * n_rows] = b.seq_start[0]
    hn[2 * n_rows + 64] = b.seq_len[0]
    ints.copy_(host, non_blocking=True)


def launch(stream):
    rc = lib.tt_encoder_forward_cls(ctypes.byref(enc.w.struct), ids_d.data_ptr(), pos_d.data_ptr(), None, st_d.data_ptr(), ln_d.data_ptr(),
                                    1, n_rows, MAXLEN, cls.data_ptr(), base, need, stream)
    assert rc == 0, lib.tt_last_error()
    rc = lib.tt_embed_pool(cls.data_ptr(), H, rows.data_ptr(), 1, H, out.data_ptr(), out16.data_ptr(), stream)
    assert rc == 0, lib.tt_last_error()


fill(batch)
launch(torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("static buffers, max_len 64 instead of %d: %s" % (batch.max_len, "same bits as embed_packed" if torch.equal(out, ref) else "DIFFERENT from embed_packed"))
static_eager = lambda: (fill(batch), launch(torch.cuda.current_stream().cuda_stream))      # noqa: E731
print("eager on static buffers (copy + two C calls):           median %.3f ms, min %.3f ms" % timed(static_eager))

g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    launch(side.cuda_stream)             # warm-up on the capture stream
    side.synchronize()
    with torch.cuda.graph(g, stream=side):
        launch(side.cuda_stream)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
out.zero_()
replay = lambda: (fill(batch), g.replay())            # noqa: E731
replay()
torch.cuda.synchronize()
print("graph replay: %s" % ("same bits as embed_packed" if torch.equal(out, ref) else "DIFFERENT from embed_packed"))
print("graph replay (copy + replay):                           median %.3f ms, min %.3f ms" % timed(replay))
# a different query through the same graph
q2 = query(max(8, tokens - 9))
b2 = pack_tokens([q2], cfg)
want = enc.embed_packed(b2)[0].clone()
fill(b2)
g.replay()
torch.cuda.synchronize()
print("another query (%d tokens) through the same graph: %s" % (len(q2), "same bits as embed_packed" if torch.equal(out, want) else "DIFFERENT"))
