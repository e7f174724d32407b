#!/bin/bash
# Per-sub-tile byte sweep + LDS mirror (round 3): string parity on the GPU, then device ms per
# Evaluate (GDV_TRACE) and L2-miss reads (FETCH_SIZE raw) of the C5 kernels, with and without the
# mirror.  Every step under its own timeout.
#   gpurun --timeout 780 -- 'bash tools/c5_mirror_check.sh'
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/c5m; rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 420 python -m pytest tests/test_strings.py tests/test_registry_tail.py tests/test_fuzz_trees.py tests/test_golden.py tests/test_full_size.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest.txt
export GDV_NO_DISK_CACHE=1
cd /tmp
i=0
run() {  # name, env assignments...
  local name=$1; shift; i=$((i+1))
  local t=$(timeout 100 env "$@" GDV_TRACE=1 python $R/bench.py --workload c5 --no-cpu-baseline --steps 8 --warmup 2 2>&1 | grep "^\[gdv\]" | tail -3 | sed 's/.*device_ms=\([0-9.]*\).*/\1/' | tr '\n' ' ')
  timeout 100 env "$@" rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/v$i -o c5 --output-format csv -- python $R/bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python3 - $OUT/v$i "$name" "$t" <<'PY' | tee -a $OUT/variants.txt
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for p in glob.glob(sys.argv[1] + "/**/c5_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if r["Kernel_Name"].startswith("gdv_k_"):
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], "| device ms", sys.argv[3], "| FETCH raw GB", sorted(round(sum(v) / len(v) * 1024 / 1e9, 3) for v in acc.values()))
PY
}
run "sub-tile sweep + LDS mirror" A=1
run "mirror + NT stores" "GDV_RTC_OPT=-DGDV_NT_STRING_STORES"
(cd $R && PYTHONPATH=$R timeout 150 python tools/registry_tail_timing.py 2>&1 | tail -8 | tee $OUT/registry_tail_timing.txt)
(cd $R && PYTHONPATH=$R timeout 60 python tools/flat_only_timing.py 2>&1 | tail -4 | tee $OUT/flat_only.txt)
timeout 100 python $R/bench.py --workload c5 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_c5.json
cat $OUT/bench_c5.json | cut -c1-400
