#!/bin/bash
# How busy is the GPU under the plugin surface (32 request threads, coalesced)?  Kernel trace of tools/probes/surface_profile.py:
# sum of kernel durations vs the span between the first and the last kernel of the measured window, and the largest idle gaps.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/surf_trace
ROWS=10000000 THREADS=32 QUERIES=768 timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/surf_trace -- python3 tools/probes/surface_profile.py > gpurun_out/surface_busy_run.log 2>&1
grep "queries_per_s" gpurun_out/surface_busy_run.log | cut -c1-300
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/surf_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# the measured window = the last 60 % of the trace span that contains encoder GEMMs (skips model construction / warm-up)
g = [r for r in rows if "gemm_kernel" in r[2]]
t0, t1 = g[0][0], g[-1][1]
w0 = t0 + (t1 - t0) * 4 // 10
sel = [r for r in rows if r[0] >= w0 and r[1] <= t1]
busy = 0; cur_s, cur_e = sel[0][0], sel[0][1]; gaps = []
for s, e, _ in sel[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = sel[-1][1] - sel[0][0]
print(f"window {span/1e6:.1f} ms, GPU busy (union of kernel intervals) {busy/1e6:.1f} ms = {busy/span:.3f}; {len(sel)} kernels")
gaps.sort(reverse=True)
print("idle gaps: total %.1f ms; > 1 ms: %d (%.1f ms); 0.1-1 ms: %d (%.1f ms); < 0.1 ms: %d (%.1f ms)" % (
    sum(gaps)/1e6, sum(1 for x in gaps if x > 1e6), sum(x for x in gaps if x > 1e6)/1e6,
    sum(1 for x in gaps if 1e5 < x <= 1e6), sum(x for x in gaps if 1e5 < x <= 1e6)/1e6,
    sum(1 for x in gaps if x <= 1e5), sum(x for x in gaps if x <= 1e5)/1e6))
by = {}
for s, e, n in sel:
    k = "gemm" if "gemm" in n else "attention" if "attention" in n else "scan" if "scan" in n or "select" in n else "rowops/other"
    by[k] = by.get(k, 0) + e - s
print({k: round(v/1e6, 1) for k, v in by.items()})
PY
find gpurun_out/surf_trace -name "*.csv" -delete
