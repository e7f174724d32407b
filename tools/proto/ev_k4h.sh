#!/bin/bash
# A/B of the round-3 C5 prototype (pre-pass + independent waves) against k4e (the structure of the
# round-2 product kernel) on one box.  Binaries are built in the container and travel.
#   gpurun --timeout 300 -- 'bash tools/proto/ev_k4h.sh'
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/k4h; mkdir -p $OUT
cd $R/tools/proto
timeout 60 ./k4e_proto 100000000 x > $OUT/k4e.txt 2>&1; tail -1 $OUT/k4e.txt
for b in k4h_u*_proto; do
  echo "== $b" | tee -a $OUT/k4h.txt
  timeout 90 ./$b 100000000 >> $OUT/k4h.txt 2>&1
done
grep -E "^==|best|check|k4h W" $OUT/k4h.txt
