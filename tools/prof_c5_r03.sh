#!/bin/bash
# C5 on the wave shape: string parity first, then kernel stats + tile-shape sweep (round 3)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03a; mkdir -p $OUT
cd $R
python -m pytest tests/test_strings.py tests/test_registry_tail.py tests/test_fuzz_trees.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -3
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_c5 -o c5 --output-format csv -- python $R/bench.py --workload c5 --no-cpu-baseline > $OUT/prof_c5_bench.json 2> /dev/null
find $OUT -name "*kernel_stats.csv" | head -1 | xargs head -4 | cut -c1-120
cd $R
for cfg in "4 4" "4 2" "4 8" "8 4" "8 2" "8 8" "2 4"; do set -- $cfg; echo "U=$1 W=$2 $(GDV_U=$1 GDV_WAVES=$2 python bench.py --workload c5 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; done
find $OUT -name "*kernel_trace.csv" -size +1000k -delete
