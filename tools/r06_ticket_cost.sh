#!/bin/bash
# Round 6: what the atomic ticket of the fused filter-project kernel costs.  Same box, same process order, three repeats each:
# GDV_FP_EXPERIMENT=3 = tile from blockIdx (rounds 4-5), default = tile from the ticket.
cd ${GRAFT_REPO_ROOT:-$PWD}
for rep in 1 2 3; do
  echo "ticket (product)     : $(PYTHONPATH=. python tools/fused_fp_timing.py 2>/dev/null | tail -1)"
  echo "blockIdx (rounds 4-5): $(GDV_FP_EXPERIMENT=3 PYTHONPATH=. python tools/fused_fp_timing.py 2>/dev/null | tail -1)"
done
