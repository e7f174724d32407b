#!/bin/bash
# The ring-mirror C5 prototype (tools/proto/k4r_proto.hip) against the k4h structure in the same binary,
# and the product's own C5 line on the same box.
#   gpurun --timeout 200 -- 'bash tools/proto/ev_k4r.sh'
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/k4r; mkdir -p $OUT
cd $R/tools/proto
for b in ${K4R_BINARIES:-k4r_u8w4_proto}; do
  echo "== $b" | tee -a $OUT/k4r.txt
  timeout 70 ./$b 100000000 >> $OUT/k4r.txt 2>&1
done
grep -E "^==|best|check|k4r W" $OUT/k4r.txt
[ -n "$K4R_WITH_BENCH" ] && (cd $R && timeout 60 python bench.py --workload c5 --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | cut -c1-200)
