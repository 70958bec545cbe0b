#!/bin/bash
# round 6: what does the staged split-plane kernel wait for?  PMC passes over tools/gemm_bench 3072 (kernel-trace only beside --pmc)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/spmc_*
P1="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
P2="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
P3="TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"
P4="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS"
P5="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/spmc_$i -- ./tools/gemm_bench 3072 10 > gpurun_out/spmc_$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 - <<'PY' | tee gpurun_out/r06_staged_pmc.log
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/spmc_*')):
    if not d[-1].isdigit(): continue
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            key = 'staged' if 'gemm_staged' in k else ('v3_x3' if 'gemm_kernel_v3' in k and 'Lb1E' in k.split('gemm_kernel_v3')[1][:20] else None)
            if key is None: continue
            key += ' grid=' + r.get('Grid_Size', '?')
            agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, cs in sorted(agg.items()):
            print(d, k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, 'n=', len(next(iter(cs.values()))))
PY
find gpurun_out -path "*spmc_*" -name "*.csv" -delete
