#!/bin/bash
# What the chip sustains on MFMA-only streams under its power cap, per instruction shape, with the shader clock / power beside.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/mfma_power.log
: > $out
for mode in 0 1 2 0 1; do
    timeout 60 tools/probes/mfma_power $mode 4 > gpurun_out/mfma_power_$mode.txt 2>&1 &
    pid=$!
    sleep 1.5
    while kill -0 $pid 2>/dev/null; do
        rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | sed 's/^GPU\[0\]\s*: //; s/clock level: 1: //; s/Current Socket Graphics Package //' | tr '\n' ' '; echo
        sleep 0.4
    done | tail -4 >> $out
    wait $pid
    cat gpurun_out/mfma_power_$mode.txt >> $out
done
cat $out
