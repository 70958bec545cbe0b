#!/usr/bin/env python3
"""Matrix-core utilisation of the bench step's kernels from one rocprofv3 --pmc pass (tools/gpu_pmc_bench_r04.sh):

    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs),   kernel cycles = GRBM_GUI_ACTIVE / 8
                (the counter sums cycles over all SIMDs -- 16 per v_mfma_f32_16x16x32_bf16, 32 per 32x32x16 -- and
                 GRBM_GUI_ACTIVE sums over the 8 XCDs: MI355X_MICROARCH.md, cycle-constants / DVFS sections)

per GEMM shape of the rerank forward (dispatch order inside a layer: Q,K projection | V projection | [attention] |
attention output + residual | FFN-up + GELU | FFN-down + residual), for the attention kernel, and for the 256-query tiled
filter pass of the scan-only leg.  Raw counter sums are kept beside the ratio (SQ_BUSY_CYCLES, SQ_INSTS_MFMA, wave cycles)."""
import collections
import csv
import glob
import json
import os
import sys


def main(pmc_dir, out):
    f = max(glob.glob(f"{pmc_dir}/*/*counter_collection.csv"), key=os.path.getmtime)
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"]), "wg": int(r["Workgroup_Size"])})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    order = sorted(disp)
    fam = collections.defaultdict(list)
    # walk the dispatches in order; the big-batch rerank forward is the run of layers whose GEMM grids are the largest
    resid_toggle = 0
    big = max((d["grid"] for d in disp.values() if "gemm_kernel" in d["name"]), default=0)
    for i in order:
        d = disp[i]
        n = d["name"]
        if "GRBM_GUI_ACTIVE" not in d:
            continue
        if "gemm_kernel_v3<6" in n or "gemm_kernel_p<6" in n:
            # (round 4: the threshold sample runs on the same kernel with a grid of n0 / 256 <= 512 row tiles; a filter pass has
            #  one workgroup per CU in the persistent form or >= 4883 tiles)
            fam["scan_tiled_filter_pass" if ("gemm_kernel_p<6" in n or d["grid"] > 1024 * 512) else "scan_tiled_sample"].append(d)
        elif "attention_kernel" in n and "cls" not in n:
            if d["grid"] * 1 >= 1000 * 256:
                fam["attention (rerank batch)"].append(d)
        elif "gemm_kernel" in n and "skinny" not in n:
            if "gemm_kernel_p<1" in n:                     # (persistent: grid = one workgroup per CU; only big batches take it)
                fam["gemm ffn-up + gelu (persistent)"].append(d)
                continue
            if d["grid"] < big // 8:
                continue                                   # query-embedding / CLS-tail sized launches
            if False:
                pass
            elif "gemm_kernel_v3<5" in n:
                fam["gemm v projection (V^T epilogue)"].append(d)
            elif "gemm_kernel_v3<2" in n:
                fam["gemm o-proj + residual" if resid_toggle == 0 else "gemm ffn-down + residual (K = 4096)"].append(d)
                resid_toggle ^= 1
            elif "gemm_kernel_v3<0" in n:
                fam["gemm q,k projection (bias)"].append(d)
    res = {}
    for k, ds in fam.items():
        gui = sum(d["GRBM_GUI_ACTIVE"] for d in ds)
        mf = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for d in ds)
        res[k] = {"launches": len(ds), "mfma_busy": mf / (gui / 8.0 * 1024.0) if gui else None,
                  "SQ_VALU_MFMA_BUSY_CYCLES": mf, "SQ_BUSY_CYCLES": sum(d.get("SQ_BUSY_CYCLES", 0.0) for d in ds),
                  "SQ_INSTS_MFMA": sum(d.get("SQ_INSTS_MFMA", 0.0) for d in ds),
                  "mfma_busy_over_sq_busy": (mf / sum(d.get("SQ_BUSY_CYCLES", 0.0) for d in ds)) if sum(d.get("SQ_BUSY_CYCLES", 0.0) for d in ds) else None,
                  "wave_cycles_waiting_frac": (sum(d.get("SQ_WAIT_ANY", 0.0) for d in ds) / sum(d.get("SQ_WAVE_CYCLES", 1.0) for d in ds)) if ds else None,
                  "kernel_cycles_per_launch": gui / 8.0 / len(ds)}
    gem = [v for k, v in res.items() if k.startswith("gemm")]
    if gem:
        tot_m = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] for v in gem)
        tot_c = sum(v["kernel_cycles_per_launch"] * v["launches"] for v in gem)
        res["gemm (all shapes of the rerank forward)"] = {"mfma_busy": tot_m / (tot_c * 1024.0)}
    res["source"] = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY "
                     "GRBM_GUI_ACTIVE --kernel-trace of `python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fp8-leg "
                     "--no-reference-leg --no-surface-leg --no-config5-leg` (tools/gpu_pmc_bench_r04.sh); mfma_busy = MFMA busy cycles / "
                     "(GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); profiled passes clock ~3 % lower than un-profiled ones")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    res["csrc_sha256"] = bench.csrc_sha256()
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
