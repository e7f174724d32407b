#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
for cfg in "A GDV_SWEEP_GROUP=1" "D GDV_SWEEP_GROUP=4 GDV_RTC_OPT=-DGDV_OUT_WIN=2048"; do
  set -- $cfg; tag=$1; shift
  OUT=$R/gpurun_out/c5g_$tag; rm -rf $OUT; mkdir -p $OUT
  env "$@" GDV_NO_DISK_CACHE=1 C5_VARIANTS=all3 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace -d $OUT/p1 -o c5 --output-format csv -- python $R/tools/c5_valu_breakdown.py > /dev/null 2>&1
  env "$@" GDV_NO_DISK_CACHE=1 C5_VARIANTS=all3 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace -d $OUT/p2 -o c5 --output-format csv -- python $R/tools/c5_valu_breakdown.py > /dev/null 2>&1
  env "$@" GDV_NO_DISK_CACHE=1 C5_VARIANTS=all3 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --kernel-trace -d $OUT/p3 -o c5 --output-format csv -- python $R/tools/c5_valu_breakdown.py > /dev/null 2>&1
  python - $OUT $tag <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("gdv_k_"):
            acc[r["Kernel_Name"][:22]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    w = sum(c["SQ_WAVES"]) / max(len(c["SQ_WAVES"]), 1)
    print(tag, k, "waves %.0f" % w, " ".join(f"{n}={sum(v)/len(v)/w:.0f}" for n, v in sorted(c.items()) if n != "SQ_WAVES"))
PY
done
