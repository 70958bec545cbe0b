#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes of `bench.py --steps 1 --warmup 1` (FETCH_SIZE, WRITE_SIZE; see
tools/gpu_pmc_bench.sh) into per-launch HBM traffic of the kernels bench.py reports rooflines for.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): counters are in KiB; FETCH_SIZE reports exactly half
of the bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled; WRITE_SIZE is exact.
Only the dispatches of the TIMED step are used (the warm-up step and the ingest leg are skipped)."""
import csv
import glob
import json
import sys


def load(dirname, counter):
    import os
    # newest file: gpurun merges every call's outputs into the same local directory
    f = max(glob.glob(f"{dirname}/*/*counter_collection.csv"), key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def main(fetch_dir, write_dir, out):
    res = {}
    # (the tiled GEMM kernels: gemm_kernel / gemm_kernel_v3 / gemm_kernel_p and, since round 6, gemm_staged_kernel for one-round grids)
    for kind, match in (("gemm", ("gemm_kernel", "gemm_staged_kernel")), ("scan_filter", ("scan_kernel<1024, 1, 0",))):
        fr = [r for r in load(fetch_dir, "FETCH_SIZE") if any(m in r["Kernel_Name"] for m in match)]
        wr = [r for r in load(write_dir, "WRITE_SIZE") if any(m in r["Kernel_Name"] for m in match)]
        if kind == "gemm":
            # rerank forward: 24 layers x 5 launches (QKV as two) + the head GEMM; query-embedding forward (small grid,
            # 128x128 kernel, fused QKV): 24 x 4
            per_step = 121 + 96
            fr, wr = fr[per_step:2 * per_step], wr[per_step:2 * per_step]
        else:
            fr, wr = fr[1:2], wr[1:2]
        fetch = sum(float(r["Counter_Value"]) for r in fr) * 1024 * 2
        write = sum(float(r["Counter_Value"]) for r in wr) * 1024
        res[kind] = {"launches": len(fr), "fetch_bytes_per_launch": fetch / len(fr),
                     "write_bytes_per_launch": write / len(wr), "hbm_bytes_per_launch": (fetch + write) / len(fr)}
    # attention (its "3.8 GB per launch = algorithmic" was arithmetic, not a counter): the varlen kernel's launches of the
    # timed step's rerank forward -- the 23 largest by fetched bytes in the second half of the process's dispatches (the first half is warm-up;
    # the query-embedding forward's 23 launches over 34-token sequences move a few MB each)
    fa = [r for r in load(fetch_dir, "FETCH_SIZE") if "attention_kernel" in r["Kernel_Name"] and "cls" not in r["Kernel_Name"]]
    wa = [r for r in load(write_dir, "WRITE_SIZE") if "attention_kernel" in r["Kernel_Name"] and "cls" not in r["Kernel_Name"]]
    if fa and len(fa) == len(wa):
        half = len(fa) // 2
        idx = sorted(range(half, len(fa)), key=lambda i: -float(fa[i]["Counter_Value"]))[:23]
        fetch = sum(float(fa[i]["Counter_Value"]) for i in idx) * 1024 * 2
        write = sum(float(wa[i]["Counter_Value"]) for i in idx) * 1024
        res["attention"] = {"launches": len(idx), "fetch_bytes_per_launch": fetch / len(idx), "write_bytes_per_launch": write / len(idx),
                            "hbm_bytes_per_launch": (fetch + write) / len(idx),
                            "algorithmic_bytes_per_launch_note": "Q + K + V read once + O written once = 4 x tokens x 1024 x 2 bytes (1600 x 296 rows: 3.88 GB)"}
    # the 256-query tiled filter pass of the scan-only leg (gemm_kernel_v3<TT_EPI_SCAN = 6, ...>): last of its launches
    is_scan = lambda r: "gemm_kernel_v3<6" in r["Kernel_Name"] or "gemm_kernel_p<6" in r["Kernel_Name"]   # noqa: E731 (one-tile / persistent form)
    fr = [r for r in load(fetch_dir, "FETCH_SIZE") if is_scan(r)]
    wr = [r for r in load(write_dir, "WRITE_SIZE") if is_scan(r)]
    if fr and wr:
        # the process runs the tiled pass at two sizes (same kernel, same grid in the persistent form): over the whole corpus
        # (scan_only leg) and over corpus / 8 rows (scan_only_shard leg) -- told apart by the bytes they fetch
        # (round 4: the threshold SAMPLE runs on the same kernel over n0 <= 131072 rows -- < 0.3 GB; a filter pass reads >= 2.5 GB)
        order = sorted((i for i in range(len(fr)) if float(fr[i]["Counter_Value"]) * 2048 >= 1e9), key=lambda i: float(fr[i]["Counter_Value"]))
        samples = sorted((i for i in range(len(fr)) if float(fr[i]["Counter_Value"]) * 2048 < 1e9), key=lambda i: float(fr[i]["Counter_Value"]))
        picks = [("scan_tiled_256q", order[-1]), ("scan_tiled_256q_shard", order[0])] if order else []
        if samples:
            picks += [("scan_tiled_sample", samples[-1]), ("scan_tiled_sample_shard", samples[0])]
        for name, i in picks:
            fetch = float(fr[i]["Counter_Value"]) * 1024 * 2
            write = float(wr[i]["Counter_Value"]) * 1024
            res[name] = {"launches": 1, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                         "hbm_bytes_per_launch": fetch + write}
    res["source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/gpu_pmc_bench_r04.sh) of `python3 bench.py --steps 1 "
                     "--warmup 1 --no-cpu-baseline --no-fp8-leg --no-reference-leg --no-surface-leg --no-config5-leg`; FETCH_SIZE doubled (gfx950: the "
                     "counter tallies 128-B requests at 64 B), counters in KiB")
    # ties the file to the kernel sources it was measured on: bench.py refuses it when they differ
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    res["csrc_sha256"] = bench.csrc_sha256()
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
