#!/bin/bash
# Round 6: C5 with the byte sweep over groups of 4 sub-tiles (product) vs one sub-tile at a time (rounds 3-5), same box
cd ${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2; do
  for g in 4 1; do
    echo "GDV_SWEEP_GROUP=$g: $(GDV_SWEEP_GROUP=$g C5_VARIANTS=all3,like,like+upper,substr PYTHONPATH=. python tools/c5_valu_breakdown.py 2>/dev/null | grep VARIANT | awk '{printf "%s %s ms | ", $2, $3}')"
  done
done
