#!/bin/bash
# round 5: the kernels' A/B switches exist in the diagnostic library only (csrc: make DIAG=1); the product library reads no environment
export TT_LIB_NAME=${TT_LIB_NAME:-libtt_hip_diag.so}
# Super-tile shape sweep (TT_GEMM_SN = column tiles per 32-tile super-tile of an XCD): time and L2-miss traffic per shape.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/sn
export LD_LIBRARY_PATH=$PWD/tensor-truth_amd:$LD_LIBRARY_PATH
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
{
if [ -z "$SKIP_TIME" ]; then for rep in 1 2; do for sn in 1 2 4 8 16; do
  echo "== time, TT_GEMM_SN=$sn (round $rep)"
  TT_GEMM_SN=$sn timeout 120 $R/tools/gemm_bench 473600 10 | sed -n 2,8p
done; done; fi
for sn in 1 2 4 8 16; do
  export TT_GEMM_SN=$sn
  rm -rf /tmp/sn_$sn
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/sn_$sn -- $R/tools/gemm_bench 473600 2 > /dev/null 2>&1
  echo "== FETCH_SIZE, TT_GEMM_SN=$sn"
  python3 - /tmp/sn_$sn <<'PY'
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if r.get("Counter_Name") == "FETCH_SIZE" and "gemm_" in r["Kernel_Name"] and "fill" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
shapes = ["qkv N=3072 K=1024 (alg. reads 0.976 GB)", "o-proj + res N=1024 K=1024 (1.94)", "ffn-up gelu N=4096 K=1024 (0.978)", "ffn-down + res N=1024 K=4096 (4.86)",
          "bias N=1024 K=1024 (0.972)", "bias N=4096 K=1024 (0.978)", "bias N=1024 K=4096 (3.89)"]
per = 5          # 3 warm-up + 2 timed launches per shape
for i, name in enumerate(shapes):
    v = [float(r["Counter_Value"]) for r in rows[i * per:(i + 1) * per]]
    if v:
        print(f"  {name:52s} fetch {2 * 1024 * sum(v) / len(v) / 1e9:7.3f} GB per launch   ({rows[i * per]['Kernel_Name'][:48]})")
PY
done
} 2>&1 | tee $R/gpurun_out/gemm_sn_sweep.log
