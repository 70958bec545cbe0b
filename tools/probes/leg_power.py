"""Shader clock and socket power while ONE leg of the path runs for ~4 s: the reranker forward per precision, the streaming scan
(64 queries), the tiled scan (256 queries).  Which legs sit at the 1400 W cap?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from tensor_truth_amd import scan as tscan
from tensor_truth_amd.encoder import BGE_RERANKER_V2_M3, Encoder, EncoderWeights, pack_token_matrix, synthetic_state_device
from tensor_truth_amd.encoder_x3 import EncoderWeightsX3, EncoderX3
from tensor_truth_amd.encoder_f16c import EncoderF16C, EncoderWeightsF16C

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
cfg = BGE_RERANKER_V2_M3
rng = np.random.default_rng(3)
pairs = rng.integers(4, cfg.vocab_size, size=(1600, 292), dtype=np.int32); pairs[:, 0] = 0; pairs[:, -1] = 2
batch = pack_token_matrix(pairs, cfg)


def sustain(name, fn, secs=4.0, unit=None, per_call=1.0):
    fn(); torch.cuda.synchronize()
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 1.5:
        fn()
        torch.cuda.synchronize()
    s = bench.ClockSampler(0, period_s=0.2).start()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < secs:
        fn(); n += 1
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c = s.stop()
    print(f"{name}: {dt / n * 1e3:.2f} ms per call" + (f" = {per_call * n / dt:.3g} {unit}" if unit else "") +
          f" | sclk median {c['sclk_mhz_median']} MHz ({c['sclk_mhz_min']}-{c['sclk_mhz_max']}), socket power {c['socket_power_w_mean'] and round(c['socket_power_w_mean'])} W", flush=True)


state16 = synthetic_state_device(cfg, dev, seed=2)
rr = Encoder(EncoderWeights(cfg, state16, dev))
sustain("reranker forward bf16, 1600 x 292 tok", lambda: rr.rerank_packed(batch), unit="M tok/s", per_call=1600 * 292 / 1e6)
rr.calibrate_fp8(pack_token_matrix(pairs[:64], cfg)); rr.w.set_gemm_dtype("fp8")
sustain("reranker forward fp8 projections", lambda: rr.rerank_packed(batch), unit="M tok/s", per_call=1600 * 292 / 1e6)
del rr
rr16 = Encoder(EncoderWeights(cfg, state16, dev, dtype=torch.float16))
sustain("reranker forward fp16", lambda: rr16.rerank_packed(batch), unit="M tok/s", per_call=1600 * 292 / 1e6)
del rr16, state16
state32 = synthetic_state_device(cfg, dev, seed=2, dtype=torch.float32)
e3 = EncoderX3(EncoderWeightsX3(cfg, state32, dev, dtype=torch.float16))
sustain("reranker forward f16x3 (reference precision)", lambda: e3.rerank_packed(batch), unit="M tok/s", per_call=1600 * 292 / 1e6)
del e3
ec = EncoderF16C(EncoderWeightsF16C(cfg, state32, dev))
sustain("reranker forward f16c", lambda: ec.rerank_packed(batch), unit="M tok/s", per_call=1600 * 292 / 1e6)
del ec, state32
corpus = bench.synth_corpus_shard(10_000_000, 1024, 1234, dev)
for nq in (1, 64, 256):
    q = torch.nn.functional.normalize(torch.randn(nq, 1024, device=dev, generator=torch.Generator(device=dev).manual_seed(nq)), dim=1).to(torch.bfloat16)
    sustain(f"scan 10M x 1024, {nq} queries, top-50", lambda: tscan.scan_topk(corpus, q, 50), unit="GB/s (algorithmic)", per_call=20.48)
