#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d $OUT/pmc_sq_c5 -o c5 --output-format csv -- python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import csv, collections
rows=list(csv.DictReader(open('gpurun_out/pmc_sq_c5/c5_counter_collection.csv')))
agg=collections.defaultdict(list)
for r in rows:
    if r['Kernel_Name'].startswith('gdv_k_'):
        agg[(r['Dispatch_Id'], r['Counter_Name'])].append(float(r['Counter_Value']))
byd=collections.defaultdict(dict)
for (d,c),v in agg.items(): byd[d][c]=sum(v)
for d in sorted(byd, key=int)[-2:]:
    print(d, {k: f"{v:.3g}" for k,v in byd[d].items()})
PY
rocprofv3 --pmc TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum --kernel-trace -d $OUT/pmc_ta_c5 -o c5 --output-format csv -- python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import csv, collections
try:
    rows=list(csv.DictReader(open('gpurun_out/pmc_ta_c5/c5_counter_collection.csv')))
    byd=collections.defaultdict(dict)
    for r in rows:
        if r['Kernel_Name'].startswith('gdv_k_'):
            byd[r['Dispatch_Id']][r['Counter_Name']]=byd[r['Dispatch_Id']].get(r['Counter_Name'],0)+float(r['Counter_Value'])
    for d in sorted(byd, key=int)[-2:]:
        print(d, {k: f"{v:.3g}" for k,v in byd[d].items()})
except Exception as e:
    print("ta counters failed", e)
PY
