#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}
run() { echo "$1: $(env $2 GDV_NO_DISK_CACHE=1 python bench.py --workload c5 --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['verified'], d['roofline']['frac'])")"; }
for rep in 1 2 3; do
run "A group=1 (26 KB LDS)            " "GDV_SWEEP_GROUP=1"
run "D group=4, OUT_WIN 2048 (27 KB)  " "GDV_SWEEP_GROUP=4 GDV_RTC_OPT=-DGDV_OUT_WIN=2048"
run "G group=4, OUT_WIN 1536 (25 KB)  " "GDV_SWEEP_GROUP=4 GDV_RTC_OPT=-DGDV_OUT_WIN=1536"
done
