#!/bin/bash
# Lone caller: where do the 1.27 ms of ONE query's embedding go -- kernel time or the gaps between ~170 dependent launches?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/qe_trace
python3 tools/probes/query_embed_only.py 2>&1 | tail -1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/qe_trace -- python3 tools/probes/query_embed_only.py 2>&1 | grep "query embedding"
python3 - <<'PY'
import csv, glob, collections, re
f = glob.glob("gpurun_out/qe_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
# forwards = runs of kernels separated by > 150 us of idle (the host sync + pack between two embeddings)
runs, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - cur[-1][1] > 40_000:
        runs.append(cur); cur = [r]
    else:
        cur.append(r)
runs.append(cur)
runs = [r for r in runs if len(r) > 100][-40:]
n = sorted(len(r) for r in runs)[len(runs) // 2]
runs = [r for r in runs if len(r) == n]
span = sorted(r[-1][1] - r[0][0] for r in runs)[len(runs) // 2] / 1e3
busy = sorted(sum(e - s for s, e, _ in r) for r in runs)[len(runs) // 2] / 1e3
gaps = sorted(sum(max(0, b[0] - a[1]) for a, b in zip(r, r[1:])) for r in runs)[len(runs) // 2] / 1e3
print(f"{len(runs)} forwards of {n} kernels: first start -> last end {span:.1f} us; sum of kernel durations {busy:.1f} us; sum of gaps {gaps:.1f} us ({gaps / (n - 1):.2f} us per gap)")
by = collections.defaultdict(list)
for r in runs:
    for s, e, k in r:
        m = re.search(r"(\w+)(<[^(]*)?\(", k.replace("(anonymous namespace)::", ""))
        by[(m.group(1) + (m.group(2) or "")) if m else k[:60]].append((e - s) / 1e3)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {sum(v) / len(runs):8.1f} us per forward  {len(v) // len(runs):4d} x {sum(v) / len(v):6.2f} us  {k}")
PY
find gpurun_out/qe_trace -name "*.csv" -delete
