"""bf16x3 (reference precision on the bf16 matrix cores): per-stage time of 50 / 200 / 1600 pairs x 292 tokens through 24 layers."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tensor_truth_amd import _lib
from tensor_truth_amd.encoder import BGE_RERANKER_V2_M3, EncoderConfig, pack_token_matrix, synthetic_state_device
from tensor_truth_amd.encoder_x3 import EncoderWeightsX3, EncoderX3

dev = torch.device("cuda", 0)
cfg = EncoderConfig(**{**BGE_RERANKER_V2_M3.__dict__, "vocab_size": 8192, "max_pos": 514})
enc = EncoderX3(EncoderWeightsX3(cfg, synthetic_state_device(cfg, dev, seed=2, dtype=torch.float32), dev))
rng = np.random.default_rng(0)
lib = _lib.load_library()
for n_pairs in (50, 200, 1600):
    pairs = rng.integers(4, cfg.vocab_size, size=(n_pairs, 292), dtype=np.int32)
    pairs[:, 0], pairs[:, -1] = 0, 2
    b = pack_token_matrix(pairs, cfg)
    enc.rerank_packed(b); torch.cuda.synchronize()
    lib.tt_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(3):
        enc.rerank_packed(b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    out = []
    for name, kid in (("gemm", 4), ("attention", 5), ("rowops", 6)):
        ms, cnt = ctypes.c_double(0), ctypes.c_int(0)
        lib.tt_prof_read(kid, ctypes.byref(ms), ctypes.byref(cnt))
        out.append(f"{name} {ms.value / 3:.2f} ms ({cnt.value // 3} launches)")
    lib.tt_prof_enable(0)
    gflop = n_pairs * 292 * 24 * 24 * 1024 * 1024 * 3          # three bf16 MFMA products per product
    gemm_ms = float(out[0].split()[1])
    print(f"{n_pairs} pairs x 292 tok: {dt * 1e3:.1f} ms = {dt * 1e3 / (n_pairs / 50):.1f} ms per 50-pair query | " + " | ".join(out)
          + f" | GEMM {gflop / (gemm_ms * 1e-3) / 1e12:.0f} TF/s of bf16 MFMA work")
