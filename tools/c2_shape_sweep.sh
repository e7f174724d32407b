#!/bin/bash
# Round 6: the projection kernel's tile shape (GDV_U sub-tiles per wave x GDV_WAVES waves per workgroup) with the output columns
# from the device pool — alternating runs on one box.   bash tools/c2_shape_sweep.sh [workload]
cd ${GRAFT_REPO_ROOT:-$PWD}
W=${1:-c2}
run() { echo "$1: $(env $2 GDV_NO_TIER0=1 python bench.py --workload $W --no-extras --no-cpu-baseline --data philox --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['ms_per_step'], r['kernel_ms'], d['verified'], r['frac'], r['placement']['sets'][0]['rates_gbs'])")"; }
for rep in 1 2 3 4 5; do
run "U=4 W=4 (rounds 1-5)" "GDV_U=4 GDV_WAVES=4"
run "U=8 W=4             " "GDV_U=8 GDV_WAVES=4"
run "U=16 W=4            " "GDV_U=16 GDV_WAVES=4"
done
