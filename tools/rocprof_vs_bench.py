"""Cross-check of bench.py's live HIP-event timing against rocprofv3 (the judge's agreement test, in one place).

    rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 bench.py --headline-only ... > bench.json
    python tools/rocprof_vs_bench.py DIR/**/*kernel_stats.csv bench.json

With --headline-only every kernel of the process belongs to warm-up / timed / instrumented steps of the SAME shape, so
rocprof's per-kernel averages are averages over identical steps.  Compared: the GEMM family (all gemm_* kernels: calls
and total duration -> average launch) with roofline.avg_launch_ms, and the scan filter kernel with
roofline_scan.avg_launch_ms."""
import csv
import json
import sys


def main():
    stats_csv, bench_json = sys.argv[1], sys.argv[2]
    with open(bench_json) as f:
        bench = json.loads([line for line in f if line.startswith("{")][-1])
    gemm_calls = gemm_ns = scan_calls = scan_ns = 0
    rows = []
    with open(stats_csv) as f:
        for r in csv.DictReader(f):
            name, calls, tot = r["Name"], int(r["Calls"]), int(r["TotalDurationNs"])
            if "gemm_kernel" in name or "gemm_skinny" in name or "gemm_staged" in name or "gemm_relay" in name:
                gemm_calls += calls
                gemm_ns += tot
                rows.append((name.split("::")[-1][:60], calls, tot / calls / 1e6))
            elif "scan_kernel<" in name and ", 0, 0>" in name:      # OUT = 0: the filter pass
                scan_calls += calls
                scan_ns += tot
    steps_total = bench["steps"] + bench["warmup"] + 1                # + the untimed instrumented step
    out = []
    out.append(f"bench.py --headline-only: {bench['value']:.1f} {bench['unit']}, {bench['ms_per_step']:.1f} ms/step, "
               f"{steps_total} identical steps in the process")
    r = bench["roofline"]
    out.append(f"GEMM family  rocprofv3: {gemm_calls} launches ({gemm_calls / steps_total:.0f}/step), avg {gemm_ns / gemm_calls / 1e6:.4f} ms"
               f"   | bench HIP events: {r['launches']} launches ({r['launches'] / bench['steps']:.0f}/step), avg {r['avg_launch_ms']:.4f} ms"
               f"   | ratio {gemm_ns / gemm_calls / 1e6 / r['avg_launch_ms']:.4f}")
    s = bench["roofline_scan"]
    out.append(f"scan filter  rocprofv3: {scan_calls} launches, avg {scan_ns / scan_calls / 1e6:.4f} ms"
               f"   | bench HIP events: {s['launches']} launches, avg {s['avg_launch_ms']:.4f} ms"
               f"   | ratio {scan_ns / scan_calls / 1e6 / s['avg_launch_ms']:.4f}")
    for name, calls, avg in sorted(rows, key=lambda t: -t[1] * t[2]):
        out.append(f"    {name:60s} {calls:6d} x {avg:8.4f} ms")
    print("\n".join(out))


if __name__ == "__main__":
    main()
