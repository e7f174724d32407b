#!/bin/bash
# HBM traffic of the fused filter -> project kernel against the Filter + selection-mode Projector chain
# (two rocprofv3 --pmc passes, kernel-trace only) -> $1/fp_traffic.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=${1:-$R/gpurun_out/fp_traffic}; mkdir -p $OUT
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o fp --output-format csv -- python $R/tools/fused_fp_pmc_run.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o fp --output-format csv -- python $R/tools/fused_fp_pmc_run.py > /dev/null 2>&1
cd $R
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
def per_kernel(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and (r["Kernel_Name"].startswith("gdv_k_") or "EmitIndices" in r["Kernel_Name"]):
            acc.setdefault(r["Kernel_Name"][:60], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}
f, nf = per_kernel(glob.glob(out + "/fetch/**/fp_counter_collection.csv", recursive=True)[0], "FETCH_SIZE")
w, _ = per_kernel(glob.glob(out + "/write/**/fp_counter_collection.csv", recursive=True)[0], "WRITE_SIZE")
with open(out + "/fp_traffic.txt", "w") as o:
    o.write("# per launch, GB: reads = FETCH_SIZE KiB x 1024 x 2 (gfx950: 64-B units for 128-B requests), writes = WRITE_SIZE KiB x 1024\n")
    o.write("# 10^9 int64 rows x 2, a > 499 AND b < 250 (1/8 selected), a + b projected, uint32 selection vector\n")
    for k in sorted(f, key=lambda k: -f[k]):
        rd, wr = f[k] * 1024 * 2 / 1e9, w.get(k, 0) * 1024 / 1e9
        o.write(f"{k:62s} launches {nf[k]:2d}   read {rd:7.3f}   write {wr:6.3f}   total {rd + wr:7.3f}\n")
print(open(out + "/fp_traffic.txt").read())
PY
