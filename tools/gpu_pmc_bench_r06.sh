#!/bin/bash
# Round-5 evidence run on the FINAL tree: PMC passes of the default single-GPU bench step -- HBM traffic (FETCH_SIZE,
# WRITE_SIZE: separate passes) and matrix-core utilisation (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_INSTS_MFMA,
# GRBM_GUI_ACTIVE: one more pass) -- then rocprofv3 --stats of the headline-only bench + the agreement check against the
# live HIP-event timing, and the per-shape GEMM rates.  Everything lands in gpurun_out/; copy the summaries to profiles/r06_*.
# (--pmc passes carry --kernel-trace only: no sys / hip / memory-copy trace domains on this pool.)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
LEGS="--no-cpu-baseline --no-fp8-leg --no-reference-leg --no-fp16-leg --no-surface-leg --no-config5-leg"
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma gpurun_out/prof_headline
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 $LEGS > gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 1 --warmup 1 $LEGS > gpurun_out/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -- python3 bench.py --steps 1 --warmup 1 $LEGS > gpurun_out/pmc_mfma.log 2>&1
python tools/pmc_to_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/r06_pmc_traffic.json
python tools/pmc_to_mfma.py gpurun_out/pmc_mfma gpurun_out/r06_pmc_mfma.json
find gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma -name "*.csv" -delete
# (headline stats + gemm shapes: see below)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_headline -- python3 bench.py --headline-only --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r06_bench_headline.json 2> gpurun_out/bench_headline.err
tail -2 gpurun_out/bench_headline.err
S=$(find gpurun_out/prof_headline -name "*kernel_stats.csv" | head -1)
cp "$S" gpurun_out/r06_bench_headline_kernel_stats.csv
find gpurun_out/prof_headline -name "*kernel_trace.csv" -delete
python tools/rocprof_vs_bench.py gpurun_out/r06_bench_headline_kernel_stats.csv gpurun_out/r06_bench_headline.json | tee gpurun_out/r06_rocprof_vs_bench.txt
./tools/gemm_bench 473600 10 | tee gpurun_out/r06_gemm_shapes.log
