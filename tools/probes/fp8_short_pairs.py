"""Reranker forward on SHORT pairs (config 5's shape: ~100 tokens) vs the bench's 292-token pairs, bf16 and fp8, per stage."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tensor_truth_amd import _lib
from tensor_truth_amd.encoder import BGE_RERANKER_V2_M3, Encoder, EncoderWeights, pack_token_matrix, synthetic_state_device
lib = _lib.load_library(); dev = torch.device("cuda", 0); torch.cuda.set_device(0)
cfg = BGE_RERANKER_V2_M3
rr = Encoder(EncoderWeights(cfg, synthetic_state_device(cfg, dev, seed=2), dev))
rng = np.random.default_rng(3)
KIDS = (("gemm", 4), ("attention", 5), ("rowops", 6))
for n_pairs, L in ((1600, 292), (1600, 100), (400, 100), (4800, 100)):
    pairs = rng.integers(4, cfg.vocab_size, size=(n_pairs, L), dtype=np.int32); pairs[:, 0] = 0; pairs[:, -1] = 2
    batch = pack_token_matrix(pairs, cfg)
    for mode in ("bf16", "fp8"):
        if mode == "fp8":
            rr.calibrate_fp8(pack_token_matrix(pairs[:64], cfg)); rr.w.set_gemm_dtype("fp8")
        else:
            rr.w.set_gemm_dtype("bf16")
        rr.rerank_packed(batch); torch.cuda.synchronize()
        lib.tt_prof_enable(1)
        t0 = time.perf_counter()
        for _ in range(3):
            rr.rerank_packed(batch)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        st = []
        for name, kid in KIDS:
            ms, cnt = ctypes.c_double(0), ctypes.c_int(0)
            lib.tt_prof_read(kid, ctypes.byref(ms), ctypes.byref(cnt))
            st.append(f"{name} {ms.value / 3:.2f} ms ({cnt.value // 3})")
        lib.tt_prof_enable(0)
        print(f"{n_pairs} pairs x {L} tok, {mode}: {dt * 1e3:.2f} ms per forward = {n_pairs * L / dt / 1e6:.2f} M tok/s | " + " | ".join(st), flush=True)
rr.w.set_gemm_dtype("bf16")
