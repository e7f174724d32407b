#!/bin/bash
# Which earlier workload of the driver's line stretches single steps of a later one?   bash tools/bench_step_times.sh
cd ${GRAFT_REPO_ROOT:-$PWD}
for extras in "c5" "k2f,c5" "c3,c5" "c1,c5" "c1,c3,k2f,c5" "c1,c3,k2f,c5,c4"; do
GDV_BENCH_STEP_TIMES=1 python bench.py --no-cpu-baseline --extras $extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
out=[]
for k,w in d['workloads'].items():
    r=w['roofline']; st=r['kernel_ms_steps']; med=r['kernel_ms_median']
    out.append('%s mean %.3f median %.3f outliers %s' % (k, r['kernel_ms'], med, [(i,x) for i,x in enumerate(st) if x > 1.3*med]))
print('extras $extras:', ' | '.join(out))
"; done
