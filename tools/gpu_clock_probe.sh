#!/bin/bash
# Is the GEMM clock-limited?  Sample the GPU's shader clock and power (rocm-smi) while ONE workload runs: the bf16 encoder GEMM,
# the split-plane GEMM, and the MFMA-only instruction stream (tools/probes/mfma_rate).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/clock_probe.log
: > $out
sample() {   # $1 = pid to watch
    while kill -0 "$1" 2>/dev/null; do
        rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ' '
        echo
        sleep 0.4
    done
}
echo "== idle" >> $out
rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|Power" >> $out
for mode in bf16 x3; do
    echo "== $mode GEMM (ffn-up shape, 236 800 rows), 8 s" >> $out
    timeout 120 python tools/probes/gemm_sustain.py $mode 8 > gpurun_out/clock_probe_$mode.txt 2>&1 &
    pid=$!
    sleep 6    # import + set-up
    sample $pid | sort | uniq -c | sort -rn | head -8 >> $out
    wait $pid
    tail -1 gpurun_out/clock_probe_$mode.txt >> $out
done
echo "== MFMA-only stream (tools/probes/mfma_rate, repeated)" >> $out
( for i in 1 2 3 4 5 6; do timeout 60 tools/probes/mfma_rate | head -1; done > gpurun_out/clock_probe_mfma.txt 2>&1 ) &
pid=$!
sleep 1
sample $pid | sort | uniq -c | sort -rn | head -8 >> $out
wait $pid
tail -2 gpurun_out/clock_probe_mfma.txt >> $out
cat $out
