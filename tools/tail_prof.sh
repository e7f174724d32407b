#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/tailprof; rm -rf $OUT; mkdir -p $OUT
cd /tmp
PYTHONPATH=$R timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/p -o tail --output-format csv -- python $R/tools/registry_tail_timing.py > $OUT/timing.txt 2>&1
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT -name "*kernel_trace.csv" -size +3000k -delete
head -30 $OUT/kernel_stats.csv | cut -c1-160
tail -8 $OUT/timing.txt
