#!/bin/bash
# Where do the C5 kernel's instructions go?  SQ_INSTS_VALU / SALU per wave and device time with
# parts of the generated kernel switched off (GDV_ABL bits; results are wrong on purpose).
export TMPDIR=/tmp GDV_NO_DISK_CACHE=1
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/abl; rm -rf $OUT; mkdir -p $OUT
cd /tmp
for abl in ${ABLS:-0 1 2 4 8 16 31}; do
  GDV_RTC_OPT="-DGDV_ABL=$abl" rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --kernel-trace -d $OUT/a$abl -o c5 --output-format csv -- python $R/bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  t=$(GDV_RTC_OPT="-DGDV_ABL=$abl" GDV_TRACE=1 python $R/bench.py --workload c5 --no-cpu-baseline --steps 6 --warmup 2 2>&1 | grep "^\[gdv\]" | tail -1 | sed 's/.*device_ms=\([0-9.]*\).*/\1/')
  python3 - $OUT/a$abl $abl $t <<'PY'
import csv, glob, sys
acc = {}
for p in glob.glob(sys.argv[1] + "/**/c5_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if r["Kernel_Name"].startswith("gdv_k_"):
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
w = acc.get("SQ_WAVES", 1)
print(f"abl={int(sys.argv[2]):2d}  ms={sys.argv[3]}  per wave: VALU {acc.get('SQ_INSTS_VALU',0)/w:7.1f}  SALU {acc.get('SQ_INSTS_SALU',0)/w:7.1f}  LDS {acc.get('SQ_INSTS_LDS',0)/w:6.1f}")
PY
done
