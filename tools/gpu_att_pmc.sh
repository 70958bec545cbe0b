#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/apmc1 gpurun_out/apmc2
./tools/att_bench 800 292 10
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/apmc1 -- ./tools/att_bench 800 292 3 > gpurun_out/apmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d gpurun_out/apmc2 -- ./tools/att_bench 800 292 3 > gpurun_out/apmc2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/apmc1", "gpurun_out/apmc2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "attention" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(d, k, "per-dispatch mean", sum(v) / len(v), "n", len(v))
PY
