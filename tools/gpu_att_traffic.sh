#!/bin/bash
# round 5: the kernels' A/B switches exist in the diagnostic library only (csrc: make DIAG=1); the product library reads no environment
export TT_LIB_NAME=${TT_LIB_NAME:-libtt_hip_diag.so}
# HBM traffic of the attention kernel with and without the XCD-aware workgroup order (separate --pmc passes).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for x in 1 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/atr_${x}_$c
    TT_ATT_XCD=$x rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/atr_${x}_$c -- ./tools/att_bench 800 292 3 > gpurun_out/atr.log 2>&1
  done
  TT_ATT_XCD=$x ./tools/att_bench 800 292 20
  TT_ATT_XCD=$x ./tools/att_bench 400 512 20
done
python3 - <<'PY'
import csv, glob, collections
for x in (1, 0):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob(f"gpurun_out/atr_{x}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "attention_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    vals.append(float(r["Counter_Value"]))
        if vals:
            print(f"TT_ATT_XCD={x} {c}: mean {sum(vals)/len(vals):.0f} (raw counter units, n={len(vals)})")
PY
