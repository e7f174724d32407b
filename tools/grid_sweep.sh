#!/bin/bash
# Round 6: workgroups per CU of the grid-stride launch (GDV_GRID_MULT), alternating on one box.   bash tools/grid_sweep.sh <workload> <reps> "<cfg>;<cfg>;..."
cd ${GRAFT_REPO_ROOT:-$PWD}
W=${1:-c3}; REPS=${2:-3}; CFGS=${3:-"GDV_GRID_MULT=8;GDV_GRID_MULT=2;GDV_GRID_MULT=1"}
run() { echo "$W [$1]: $(env $1 GDV_NO_TIER0=1 python bench.py --workload $W --no-extras --no-cpu-baseline --data philox --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['ms_per_step'], r['kernel_ms'], d['verified'], r['frac'])")"; }
for rep in $(seq $REPS); do
  IFS=';' read -ra L <<< "$CFGS"
  for c in "${L[@]}"; do run "$c"; done
done
