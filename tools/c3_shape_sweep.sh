#!/bin/bash
# Round 6: C3's predicate kernel against the two-stream read skeleton (6.85 TB/s at 2 workgroups per CU, 4 sub-tiles, non-temporal;
# the kernel: 6.07 at 8 per CU, 16 sub-tiles) — grid x sub-tiles x waves, alternating on one box.   bash tools/c3_shape_sweep.sh
cd ${GRAFT_REPO_ROOT:-$PWD}
run() { echo "$1: $(env $2 GDV_NO_TIER0=1 python bench.py --workload c3 --no-extras --no-cpu-baseline --data philox --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['ms_per_step'], r['kernel_ms'], d['verified'], r['frac'])")"; }
for rep in 1 2 3; do
run "default (U=16 W=4 grid x8)" ""
run "grid x2                   " "GDV_GRID_MULT=2"
run "grid x4                   " "GDV_GRID_MULT=4"
run "grid x16                  " "GDV_GRID_MULT=16"
run "U=4 grid x2               " "GDV_U=4 GDV_GRID_MULT=2"
run "U=4 grid x8               " "GDV_U=4 GDV_GRID_MULT=8"
run "U=8 grid x4               " "GDV_U=8 GDV_GRID_MULT=4"
run "U=16 W=8 grid x4          " "GDV_WAVES=8 GDV_GRID_MULT=4"
run "U=16 W=2 grid x16         " "GDV_WAVES=2 GDV_GRID_MULT=16"
done
