#!/bin/bash
# The bf16 encoder GEMM under the power cap with random / constant / zero operands: how much of the cap is operand switching in the MFMA data path?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/clock_probe2.log
: > $out
for data in randn const zero; do
  for mode in bf16 x3; do
    timeout 120 python tools/probes/gemm_sustain.py $mode 6 $data > gpurun_out/cp2.txt 2>&1 &
    pid=$!
    sleep 7
    while kill -0 $pid 2>/dev/null; do
        rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | sed 's/^GPU\[0\]\s*: //; s/clock level: 1: //; s/Current Socket Graphics Package //' | tr '\n' ' '; echo
        sleep 0.4
    done | head -6 | tail -3 >> $out
    wait $pid
    tail -1 gpurun_out/cp2.txt >> $out
  done
done
cat $out
