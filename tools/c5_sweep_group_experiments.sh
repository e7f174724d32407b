#!/bin/bash
# Round 6 (profiles/r06_c5_budget.md): C5 with the byte sweep over groups of sub-tiles (GDV_SWEEP_GROUP) and with other LDS budgets.
#   bash tools/c5_sweep_group_experiments.sh bench      three repeats of bench.py --workload c5 per configuration (the figures of the budget file)
#   bash tools/c5_sweep_group_experiments.sh counters   SQ counters of group = 1 against group = 4 at equal occupancy
#   bash tools/c5_sweep_group_experiments.sh occupancy  one-shot timings of six LDS / group configurations (synchronous calls: relative only)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
case "${1:-bench}" in
bench)
  run() { echo "$1: $(env $2 GDV_NO_DISK_CACHE=1 python bench.py --workload c5 --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['verified'], d['roofline']['frac'])")"; }
  for rep in 1 2 3; do
    run "A group=1 (26 KB LDS)            " "GDV_SWEEP_GROUP=1"
    run "D group=4, OUT_WIN 2048 (27 KB)  " "GDV_SWEEP_GROUP=4 GDV_RTC_OPT=-DGDV_OUT_WIN=2048"
    run "G group=4, OUT_WIN 1536 (25 KB)  " "GDV_SWEEP_GROUP=4 GDV_RTC_OPT=-DGDV_OUT_WIN=1536"
  done;;
occupancy)
  run() { echo "$1: $(env $2 GDV_NO_DISK_CACHE=1 C5_VARIANTS=all3 PYTHONPATH=. python tools/c5_valu_breakdown.py 2>/dev/null | grep VARIANT | awk '{printf "%s %s ms", $2, $3}')"; }
  for rep in 1 2; do
    run "A group=1 (26 KB LDS)                 " "GDV_SWEEP_GROUP=1"
    run "B group=1, SUB_SPAN 4096 (35 KB)      " "GDV_SWEEP_GROUP=1 GDV_RTC_OPT=-DGDV_SUB_SPAN=4096"
    run "C group=4 (35 KB)                     " "GDV_SWEEP_GROUP=4"
    run "D group=4, OUT_WIN 2048 (27 KB)       " "GDV_SWEEP_GROUP=4 GDV_RTC_OPT=-DGDV_OUT_WIN=2048"
    run "E group=2 (26 KB)                     " "GDV_SWEEP_GROUP=2"
    run "F group=1, OUT_WIN 2048 (18 KB)       " "GDV_SWEEP_GROUP=1 GDV_RTC_OPT=-DGDV_OUT_WIN=2048"
  done;;
counters)
  cd /tmp
  for cfg in "A GDV_SWEEP_GROUP=1" "D GDV_SWEEP_GROUP=4 GDV_RTC_OPT=-DGDV_OUT_WIN=2048"; do
    set -- $cfg; tag=$1; shift
    OUT=$R/gpurun_out/c5g_$tag; rm -rf $OUT; mkdir -p $OUT
    for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
      env "$@" GDV_NO_DISK_CACHE=1 C5_VARIANTS=all3 rocprofv3 --pmc $set --kernel-trace -d $OUT/p_$(echo $set | cut -d' ' -f1) -o c5 --output-format csv -- python $R/tools/c5_valu_breakdown.py > /dev/null 2>&1
    done
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
  done;;
esac
