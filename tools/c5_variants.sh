#!/bin/bash
# A/B of compile-time variants of the C5 kernels on one box: device ms per Evaluate (GDV_TRACE) and
# L2-miss reads (FETCH_SIZE raw) of the main kernel.  Every step under its own timeout.
#   gpurun --timeout 600 -- 'bash tools/c5_variants.sh'
export GDV_NO_DISK_CACHE=1 TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/c5v; rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
run() {  # name, env assignments...
  local name=$1; shift; i=$((i+1))
  local t=$(timeout 120 env "$@" GDV_TRACE=1 python $R/bench.py --workload c5 --no-cpu-baseline --steps 8 --warmup 2 2>&1 | grep "^\[gdv\]" | tail -3 | sed 's/.*device_ms=\([0-9.]*\).*/\1/' | tr '\n' ' ')
  timeout 120 env "$@" rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/v$i -o c5 --output-format csv -- python $R/bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python3 - $OUT/v$i "$name" "$t" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for p in glob.glob(sys.argv[1] + "/**/c5_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if r["Kernel_Name"].startswith("gdv_k_"):
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], "| device ms", sys.argv[3], "| FETCH raw GB", sorted(round(sum(v) / len(v) * 1024 / 1e9, 3) for v in acc.values()))
PY
}
run "base (U8)" A=1
run "U8, 4 waves/SIMD" "GDV_RTC_OPT=-DGDV_STRING_KERNEL_ATTR=__attribute__((amdgpu_waves_per_eu(4,4)))"
run "U8, 4 waves/SIMD, NT stores" "GDV_RTC_OPT=-DGDV_STRING_KERNEL_ATTR=__attribute__((amdgpu_waves_per_eu(4,4))) -DGDV_NT_STRING_STORES"
run "U8, 5 waves/SIMD, NT stores" "GDV_RTC_OPT=-DGDV_STRING_KERNEL_ATTR=__attribute__((amdgpu_waves_per_eu(5,5))) -DGDV_NT_STRING_STORES"
run "U8, NT stores" "GDV_RTC_OPT=-DGDV_NT_STRING_STORES"
