#!/bin/bash
# How busy is the GPU during the composed config-5 ingest at the REFERENCE chunk geometry (Unigram tokenizer, sub-word counted [2048,512,256]/64)?  Kernel trace of
# tools/probes/ingest_ref_geometry.py: union of kernel intervals vs the span of the ingest, the largest idle gaps.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/ingest_trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ingest_trace -- python3 tools/probes/ingest_ref_geometry.py ${DOCS:-3000} - ${WORDS:-700-1500} > gpurun_out/ingest_busy_run.log 2>&1
grep -i "docs/s\|ingest\|leaves" gpurun_out/ingest_busy_run.log | head -5 | cut -c1-240
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ingest_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
g = [r for r in rows if "gemm" in r[2]]
t0, t1 = g[0][0], g[-1][1]
# skip the first 15 % (model construction, warm-up forward)
w0 = t0 + (t1 - t0) * 20 // 100
sel = [r for r in rows if r[0] >= w0 and r[1] <= t1]
busy = 0; cs, ce = sel[0][0], sel[0][1]; gaps = []
for s, e, _ in sel[1:]:
    if s > ce:
        busy += ce - cs; gaps.append((s - ce, ce)); cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
span = sel[-1][1] - sel[0][0]
print(f"window {span/1e6:.1f} ms, GPU busy {busy/1e6:.1f} ms = {busy/span:.3f}; {len(sel)} kernels")
gs = sorted(g_ for g_, _ in gaps)[::-1]
print("idle gaps: total %.1f ms; > 5 ms: %d (%.1f ms); 1-5 ms: %d (%.1f ms); 0.1-1 ms: %d (%.1f ms); < 0.1 ms: %d (%.1f ms)" % (
    sum(gs)/1e6, sum(1 for x in gs if x > 5e6), sum(x for x in gs if x > 5e6)/1e6,
    sum(1 for x in gs if 1e6 < x <= 5e6), sum(x for x in gs if 1e6 < x <= 5e6)/1e6,
    sum(1 for x in gs if 1e5 < x <= 1e6), sum(x for x in gs if 1e5 < x <= 1e6)/1e6,
    sum(1 for x in gs if x <= 1e5), sum(x for x in gs if x <= 1e5)/1e6))
print("largest gaps (ms at offset ms):", [(round(a/1e6,1), round((b-sel[0][0])/1e6)) for a, b in sorted(gaps, reverse=True)[:12]])
PY
find gpurun_out/ingest_trace -name "*.csv" -delete
