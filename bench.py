#!/usr/bin/env python3
"""End-to-end retrieval benchmark (BASELINE.json metric) on MI355X.

One step = one pass of the hot path over one batch of synthetic queries per GPU:
    embed queries (bge-m3 shape) -> all-gather query embeddings -> exact top-50 scan of this
    rank's corpus shard -> all-gather partial top-k + merge -> rerank 50 (query, chunk) pairs
    per query with the bge-reranker-v2-m3-shaped cross-encoder -> top-10.
Nothing is skipped or cached inside the timed region; inputs are resident in HBM (corpus,
weights) or host token arrays (what the reference's tokenizer would hand over).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see the driver contract in the task description).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL otherwise fails in hipIpcGetMemHandle); must be set before HIP starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--corpus-rows", type=int, default=10_000_000, help="total rows, sharded over the GPUs")
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--queries-per-gpu", type=int, default=32,
                    help="query batch per GPU and step (measured on one MI355X: 16 -> 107, 32 -> 112, 64 -> 111, 128 -> 114 q/s: "
                         "the per-batch embed / scan latency and the GEMMs' tail rounds amortise)")
    ap.add_argument("--top-k", type=int, default=50)
    ap.add_argument("--top-n", type=int, default=10)
    ap.add_argument("--query-len", type=int, default=32)
    ap.add_argument("--chunk-len", type=int, default=256)
    ap.add_argument("--embed-chunks", type=int, default=1024, help="chunks per GPU in the ingest (chunks embedded/s) leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp8-leg", action="store_true", help="skip the extra fp8-reranker timing (BASELINE config 5)")
    ap.add_argument("--headline-only", action="store_true",
                    help="run only the warm-up and timed steps (no scan-only / fp8 / ingest legs): every kernel of the process then "
                         "belongs to the timed workload, so a rocprofv3 --stats summary of this command can be compared with "
                         "roofline.avg_launch_ms directly (tools/rocprof_vs_bench.py)")
    ap.add_argument("--layers", type=int, default=24, help="encoder depth (24 = the named models; for debugging only)")
    return ap.parse_args()


def synth_corpus_shard(n_rows, dim, seed, device):
    """randn -> L2 normalise -> bf16, generated on the device in 512k-row pieces."""
    out = torch.empty((n_rows, dim), dtype=torch.bfloat16, device=device)
    g = torch.Generator(device=device).manual_seed(seed)
    step = 524288
    for lo in range(0, n_rows, step):
        hi = min(n_rows, lo + step)
        x = torch.randn((hi - lo, dim), generator=g, device=device)
        x = x / x.norm(dim=1, keepdim=True)
        out[lo:hi] = x.to(torch.bfloat16)
    return out


def passage_tokens(idx: np.ndarray, length: int, vocab: int) -> np.ndarray:
    """Deterministic synthetic chunk tokens: ids uniform in [4, vocab) as a hash of (row, pos)."""
    i = idx.astype(np.uint64)[:, None]
    p = np.arange(length, dtype=np.uint64)[None, :]
    h = (i * np.uint64(0x9E3779B97F4A7C15) + p * np.uint64(0xC2B2AE3D27D4EB4F)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    h ^= h >> np.uint64(29)
    h = (h * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    h ^= h >> np.uint64(32)
    return (h % np.uint64(vocab - 4) + np.uint64(4)).astype(np.int32)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched through torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback in tensor_truth_amd)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from tensor_truth_amd import _lib, scan as tscan
    from tensor_truth_amd.encoder import (BGE_M3, BGE_RERANKER_V2_M3, Encoder, EncoderConfig, EncoderWeights,
                                          pack_token_matrix, pack_tokens, synthetic_state_device)
    from tensor_truth_amd.sharded import ShardedCorpus, gather_queries, shard_bounds

    lib = _lib.load_library()
    Bq, K, topn, D = args.queries_per_gpu, args.top_k, args.top_n, args.dim
    emb_cfg, rr_cfg = BGE_M3, BGE_RERANKER_V2_M3
    if args.layers != 24:
        emb_cfg = EncoderConfig(**{**BGE_M3.__dict__, "layers": args.layers})
        rr_cfg = EncoderConfig(**{**BGE_RERANKER_V2_M3.__dict__, "layers": args.layers})

    # ---- resident state: corpus shard + both models -------------------------------------------
    lo, hi = shard_bounds(args.corpus_rows, world, rank)
    corpus = ShardedCorpus(synth_corpus_shard(hi - lo, D, 1234 + rank, dev), lo, args.corpus_rows)
    embedder = Encoder(EncoderWeights(emb_cfg, synthetic_state_device(emb_cfg, dev, seed=1), dev))
    reranker = Encoder(EncoderWeights(rr_cfg, synthetic_state_device(rr_cfg, dev, seed=2), dev))
    vocab = emb_cfg.vocab_size
    rng = np.random.default_rng(777 + rank)

    def make_queries():
        return rng.integers(4, vocab, size=(Bq, args.query_len), dtype=np.int32)

    tokens_step = {"embed": 0, "rerank": 0}

    def step(q_tok):
        # 1. embed this rank's queries: <s> q </s>
        q_ids = np.empty((Bq, args.query_len + 2), dtype=np.int32)
        q_ids[:, 0], q_ids[:, 1:-1], q_ids[:, -1] = 0, q_tok, 2
        batch = pack_token_matrix(q_ids, emb_cfg)
        _, q16 = embedder.embed_packed(batch)
        tokens_step["embed"] = batch.n_tokens
        # 2.-4. every shard scans the gathered query batch; partial top-k all-gathered + merged
        all_q = gather_queries(q16)
        s, i = corpus.search(all_q, K)
        mine = i[rank * Bq:(rank + 1) * Bq].cpu().numpy()          # candidate rows of my queries (host)
        # 5. rerank: <s> q </s></s> chunk </s>, 50 pairs per query
        flat = mine.reshape(-1)
        ptok = passage_tokens(np.maximum(flat, 0), args.chunk_len, vocab)
        QL = args.query_len
        pair_ids = np.empty((Bq * K, QL + args.chunk_len + 4), dtype=np.int32)
        pair_ids[:, 0] = 0
        pair_ids[:, 1:1 + QL] = np.repeat(q_tok, K, axis=0)
        pair_ids[:, 1 + QL:3 + QL] = 2
        pair_ids[:, 3 + QL:-1] = ptok
        pair_ids[:, -1] = 2
        rb = pack_token_matrix(pair_ids, rr_cfg)
        tokens_step["rerank"] = rb.n_tokens
        scores = reranker.rerank_packed(rb).view(Bq, K)
        # 6. top-n by rerank score (host-visible result, as the postprocessor returns it)
        top_s, top_j = torch.topk(scores, topn, dim=1)
        rows = torch.gather(i[rank * Bq:(rank + 1) * Bq].long(), 1, top_j)
        return top_s.cpu(), rows.cpu()

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step(make_queries())
    queries = [make_queries() for _ in range(args.steps)]
    sync_all()
    KIDS = (("scan_filter", 1), ("scan_sample", 2), ("select", 3), ("gemm", 4), ("attention", 5), ("rowops", 6))

    def read_prof():
        out = {}
        for name, kid in KIDS:
            ms, n = ctypes.c_double(0), ctypes.c_int(0)
            lib.tt_prof_read(kid, ctypes.byref(ms), ctypes.byref(n))
            out[name] = (ms.value, n.value)
        return out

    # timed region: HIP events (on the launch stream, inside the library) around the two roofline kernels only --
    # an event pair per launch of all ~700 kernels of a step costs ~2 % of the step
    lib.tt_prof_enable((1 << 1) | (1 << 4))
    t0 = time.perf_counter()
    for q in queries:
        step(q)
    sync_all()
    dt = time.perf_counter() - t0
    prof = read_prof()
    lib.tt_prof_enable(0)
    # one more, untimed, step with every kernel family instrumented: the per-stage table
    lib.tt_prof_enable(1)
    step(queries[-1])
    sync_all()
    stage_prof = read_prof()
    lib.tt_prof_enable(0)

    scan_only = None
    if not args.headline_only:
        # ---- the similarity scan alone (BASELINE configs 2 / 4): 256 resident query embeddings (256 / world per GPU, gathered)
        # against the sharded corpus
        nq_scan = max(1, 256 // world)
        scan_q = torch.nn.functional.normalize(torch.randn(nq_scan, D, device=dev, generator=torch.Generator(device=dev).manual_seed(4321)), dim=1).to(torch.bfloat16)
        corpus.search(gather_queries(scan_q), K)
        sync_all()
        t3 = time.perf_counter()
        for _ in range(3):
            corpus.search(gather_queries(scan_q), K)
        sync_all()
        dt_scan = (time.perf_counter() - t3) / 3
        if world > 1:
            t = torch.tensor([dt_scan], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_scan = float(t.item())
        scan_only = {"queries_per_s": world * nq_scan / dt_scan, "ms_per_batch": dt_scan * 1e3,
                     "what": f"exact top-{K} of {world * nq_scan} queries over the {args.corpus_rows} x {D} corpus (gather + scan + merge), no encoders"}

    # ---- BASELINE config 5's "fp8 MFMA reranker": the same steps with the cross-encoder's Q/K/V and FFN-up
    # projections on e4m3 operands.  Reported beside the headline (which stays bf16), never as it.
    fp8_leg = None
    if not args.no_fp8_leg and not args.headline_only:
        # static scale of the FFN intermediate per layer: one bf16 calibration forward over 64 synthetic pairs
        cal = rng.integers(4, vocab, size=(64, args.query_len + args.chunk_len + 4), dtype=np.int32)
        cal[:, 0], cal[:, -1] = 0, 2
        reranker.calibrate_fp8(pack_token_matrix(cal, rr_cfg))
        reranker.w.set_gemm_dtype("fp8")
        step(queries[0])
        sync_all()
        t2 = time.perf_counter()
        for q in queries:
            step(q)
        sync_all()
        dt8 = time.perf_counter() - t2
        reranker.w.set_gemm_dtype("bf16")
        if world > 1:
            t = torch.tensor([dt8], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt8 = float(t.item())
        fp8_leg = {"queries_per_s": world * Bq * args.steps / dt8, "ms_per_step": dt8 / args.steps * 1e3,
                   "what": "all four projections of the reranker's layers in e4m3 (per-token activation / per-channel "
                           "weight scales, static calibrated scale for the FFN intermediate, fp32 accumulate); "
                           "embedder, scan, attention and the CLS tail unchanged"}
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    chunks_per_s = None
    if not args.headline_only:
        # ---- second half of the BASELINE metric: batch chunk embedding (ingest), separately timed ----
        chunk_tok = rng.integers(4, vocab, size=(args.embed_chunks, args.chunk_len), dtype=np.int32)
        chunk_seqs = [np.concatenate(([0], c, [2])) for c in chunk_tok]
        chunk_batch = pack_tokens(chunk_seqs, emb_cfg)
        embedder.embed_packed(chunk_batch)
        sync_all()
        t1 = time.perf_counter()
        for _ in range(2):
            embedder.embed_packed(chunk_batch)
        sync_all()
        dt_embed = (time.perf_counter() - t1) / 2
        if world > 1:
            t = torch.tensor([dt_embed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_embed = float(t.item())
        chunks_per_s = world * args.embed_chunks / dt_embed
        if fp8_leg is not None:   # the ingest leg with the bi-encoder's layer projections in e4m3, same protocol
            embedder.calibrate_fp8(chunk_batch)
            embedder.w.set_gemm_dtype("fp8")
            embedder.embed_packed(chunk_batch)
            sync_all()
            t1 = time.perf_counter()
            for _ in range(2):
                embedder.embed_packed(chunk_batch)
            sync_all()
            dt8e = (time.perf_counter() - t1) / 2
            embedder.w.set_gemm_dtype("bf16")
            if world > 1:
                t = torch.tensor([dt8e], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt8e = float(t.item())
            fp8_leg["chunks_embedded_per_s"] = world * args.embed_chunks / dt8e

    # ---- roofline of the dominant kernel (GEMM, MFMA-bound) and of the scan (HBM-bound) -------
    H, F, L = emb_cfg.hidden, emb_cfg.ffn, emb_cfg.layers
    # algorithmic GEMM flops of what is computed (real tokens only): L-1 full layers + the last layer's QKV
    # projection for every token, the last layer's output projection + FFN for the CLS rows only, the head
    full_layer = 2 * (3 * H * H + H * H + 2 * H * F)
    gemm_flops_per_token = (L - 1) * full_layer + 2 * 3 * H * H
    cls_tail_flops = 2 * (H * H + 2 * H * F)
    n_seq_step = Bq + Bq * K
    head_flops = 2 * H * H * (Bq * K)
    gemm_flops_step = (gemm_flops_per_token * (tokens_step["embed"] + tokens_step["rerank"])
                       + cls_tail_flops * n_seq_step + head_flops)
    gemm_ms, gemm_n = prof["gemm"]
    gemm_tf = gemm_flops_step * args.steps / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    scan_ms, scan_n = prof["scan_filter"]
    q_tiles = (Bq * world + 63) // 64
    scan_bytes = (hi - lo) * D * 2 * q_tiles                                    # per launch: shard read once per 64 queries
    scan_gbs = scan_bytes * scan_n / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0

    # HBM traffic per launch: PMC counters cannot be read from inside this process; the committed
    # profiles/r01_pmc_traffic.json holds them for exactly this default single-GPU command
    # (tools/gpu_pmc_bench.sh + tools/pmc_to_traffic.py), otherwise null.
    traffic = {"gemm": None, "scan_filter": None}
    default_cfg = (world == 1 and args.corpus_rows == 10_000_000 and D == 1024 and Bq == 32 and K == 50
                   and args.chunk_len == 256 and args.query_len == 32 and L == 24)
    tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if default_cfg and os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
        traffic = {k: tj[k]["hbm_bytes_per_launch"] for k in traffic if k in tj}

    out = {
        "metric": "queries/sec (embed+top-k+rerank) over 10M x 1024 corpus",
        "value": world * Bq * args.steps / dt,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic (seeded unit-norm corpus, random-init bge-m3 / bge-reranker-v2-m3 shaped weights, hashed token ids)",
        "config": {
            "workload": (f"{args.corpus_rows} x {D} bf16 corpus row-sharded over {world} GPU(s); per GPU and step "
                         f"{Bq} queries: embed ({args.query_len}+2 tok) + exact top-{K} scan + all-gather merge + "
                         f"rerank {K} pairs x {args.query_len + args.chunk_len + 4} tok -> top-{topn}"),
            "corpus_rows": args.corpus_rows, "dim": D, "queries_per_gpu_per_step": Bq, "top_k": K, "top_n": topn,
            "pair_tokens": args.query_len + args.chunk_len + 4, "encoder_layers": L,
            "parallelism": f"corpus row-sharded x{world}, encoders replicated",
            "chunks_reranked_per_s": world * Bq * K * args.steps / dt,
            "chunks_embedded_per_s": chunks_per_s,
            "embed_batch": f"{args.embed_chunks} chunks x {args.chunk_len + 2} tok per GPU (bge-m3 shape)",
            "fp8_reranker": fp8_leg,
            "scan_only": scan_only,
        },
        "roofline": {
            "kernel": "gemm_kernel (encoder GEMMs, bf16 MFMA)",
            "bound": "mfma", "achieved": gemm_tf, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
            "frac": gemm_tf / MFMA_BF16_PEAK_TF, "traffic": traffic.get("gemm"),
            "launches": gemm_n, "avg_launch_ms": gemm_ms / max(gemm_n, 1),
        },
        "roofline_scan": {
            "kernel": "scan_kernel (filter pass over the corpus shard)",
            "bound": "hbm", "achieved": scan_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": scan_gbs / HBM_PEAK_GBS, "traffic": traffic.get("scan_filter"),
            "launches": scan_n, "avg_launch_ms": scan_ms / max(scan_n, 1),
        },
        "stage_ms_per_step": {k: v[0] for k, v in stage_prof.items()},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, emb_cfg, rr_cfg, embedder, reranker, corpus, queries[0], vocab)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(args, emb_cfg, rr_cfg, embedder, reranker, corpus, q_tok, vocab):
    """The CPU oracle (oracle/, a port of the reference's upstream arithmetic) timed on this
    box's host cores on a BOUNDED sample of the same workload (kept to ~10-30 s):
      * encoders: CPU_LAYERS of the 24 identical layers at full width (1024 hidden, 16 heads,
        4096 FFN), fp32, scaled by 24/CPU_LAYERS; 1 query embedded, CPU_PAIRS of the 50 pairs
        reranked (scaled to 50);
      * scan: a 1M-row slice of the shard, fp32 matmul + stable top-k, scaled to the corpus.
    A reported baseline, not the optimisation target."""
    from oracle import encoder as oe
    from oracle import scan as osc

    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 64))
    torch.set_num_threads(cores)
    CPU_LAYERS, CPU_PAIRS, CPU_VOCAB = 2, 4, 4096
    K = args.top_k
    rows = min(1_000_000, corpus.shard.shape[0])
    small = {**emb_cfg.__dict__, "layers": CPU_LAYERS, "vocab_size": CPU_VOCAB}
    ocfg_e = oe.EncoderConfig(**small)
    ocfg_r = oe.EncoderConfig(**{**small, "num_labels": 1})
    W_r = oe.synth_weights(ocfg_r, seed=1)   # encoder tensors shared by both legs; values do not affect timing
    W_e = W_r
    host_corpus = corpus.shard[:rows].cpu()
    scale_layers = emb_cfg.layers / CPU_LAYERS
    q_ids = np.concatenate(([0], q_tok[0] % CPU_VOCAB, [2]))
    q = torch.from_numpy(q_ids).view(1, -1).long()
    with torch.no_grad():
        oe.embed(q, torch.ones_like(q), W_e, ocfg_e)  # warm the thread pool
        t0 = time.perf_counter()
        e = oe.embed(q, torch.ones_like(q), W_e, ocfg_e)
        t_embed = (time.perf_counter() - t0) * scale_layers
        t0 = time.perf_counter()
        _, idx, _ = osc.scan_topk(host_corpus, e.to(torch.bfloat16), K)
        t_scan = time.perf_counter() - t0
        ptok = passage_tokens(idx[0, :CPU_PAIRS].numpy(), args.chunk_len, CPU_VOCAB)
        head = np.concatenate(([0], q_tok[0] % CPU_VOCAB, [2, 2]))
        ids = torch.from_numpy(np.stack([np.concatenate((head, p, [2])) for p in ptok])).long()
        t0 = time.perf_counter()
        oe.rerank_scores(ids, torch.ones_like(ids), W_r, ocfg_r)
        t_rr = (time.perf_counter() - t0) * scale_layers
    t_query = t_embed + t_scan * (args.corpus_rows / rows) + t_rr * (K / CPU_PAIRS)
    return {
        "value": 1.0 / t_query, "unit": "queries/s", "cores": cores, "kind": "port",
        "sample": (f"fp32 CPU oracle, torch threads={cores} of {avail} visible: per query = embed 1 query x "
                   f"{q.shape[1]} tok ({t_embed:.2f}s) + scan {args.corpus_rows} rows ({t_scan:.2f}s measured on {rows} rows, "
                   f"x{args.corpus_rows / rows:.0f}) + rerank {K} pairs x {ids.shape[1]} tok ({t_rr:.2f}s measured on "
                   f"{CPU_PAIRS} pairs, x{K / CPU_PAIRS:.1f}); encoder legs time {CPU_LAYERS} of {emb_cfg.layers} "
                   f"identical full-width layers and scale x{scale_layers:.0f}"),
    }


if __name__ == "__main__":
    main()
