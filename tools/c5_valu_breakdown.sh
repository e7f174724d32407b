#!/bin/bash
# instruction counts per wave of every C5 variant's kernels: two rocprofv3 --pmc passes (kernel-trace only)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/${1:-c5_valu}
rm -rf $OUT; mkdir -p $OUT
cd /tmp
python $R/tools/c5_valu_breakdown.py 2>/dev/null | grep VARIANT > $OUT/variants.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace -d $OUT/pmc1 -o c5 --output-format csv -- python $R/tools/c5_valu_breakdown.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $OUT/pmc2 -o c5 --output-format csv -- python $R/tools/c5_valu_breakdown.py > /dev/null 2>&1
cd $R
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("gdv_k_"):
            acc[r["Kernel_Name"][:22]][r["Counter_Name"]].append(float(r["Counter_Value"]))
per = {}
for k, c in acc.items():
    waves = sum(c["SQ_WAVES"]) / max(len(c["SQ_WAVES"]), 1)
    per[k] = {n: (sum(v) / len(v)) / waves for n, v in c.items() if n != "SQ_WAVES"}
    per[k]["waves"] = waves
lines = []
for ln in open(out + "/variants.txt"):
    parts = ln.split()
    name, ms, kernels = parts[1], parts[2], parts[5:]
    lines.append(f"{name:14s} {ms} ms")
    for kn in kernels:
        p = per.get(kn)
        if p:
            lines.append(f"    {kn}: waves {p['waves']:.0f}  per wave: VALU {p.get('SQ_INSTS_VALU', 0):.0f}  SALU {p.get('SQ_INSTS_SALU', 0):.0f}  LDS {p.get('SQ_INSTS_LDS', 0):.0f}  "
                         f"VMEM rd {p.get('SQ_INSTS_VMEM_RD', 0):.1f} wr {p.get('SQ_INSTS_VMEM_WR', 0):.1f}  wave cycles {p.get('SQ_WAVE_CYCLES', 0):.0f}")
open(out + "/c5_valu_breakdown.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
