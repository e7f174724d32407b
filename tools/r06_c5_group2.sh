#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}
run() { echo "$1: $(env $2 GDV_NO_DISK_CACHE=1 C5_VARIANTS=all3 PYTHONPATH=. python tools/c5_valu_breakdown.py 2>/dev/null | grep VARIANT | awk '{printf "%s %s ms", $2, $3}')"; }
for rep in 1 2; do
run "A group=1 (26 KB LDS)                 " "GDV_SWEEP_GROUP=1"
run "B group=1, SUB_SPAN 4096 (35 KB)      " "GDV_SWEEP_GROUP=1 GDV_RTC_OPT=-DGDV_SUB_SPAN=4096"
run "C group=4 (35 KB)                     " "GDV_SWEEP_GROUP=4"
run "D group=4, OUT_WIN 2048 (27 KB)       " "GDV_SWEEP_GROUP=4 GDV_RTC_OPT=-DGDV_OUT_WIN=2048"
run "E group=2 (26 KB)                     " "GDV_SWEEP_GROUP=2"
run "F group=1, OUT_WIN 2048 (18 KB)       " "GDV_SWEEP_GROUP=1 GDV_RTC_OPT=-DGDV_OUT_WIN=2048"
done
