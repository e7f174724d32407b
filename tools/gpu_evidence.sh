#!/bin/bash
# end-of-round evidence: full GPU suite, smoke, default bench, rocprof of the same command,
# PMC passes (separate, kernel-trace only), the other workloads' bench lines + kernel stats
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
python -m pytest tests -m gpu -q --timeout 1800 2>&1 > $OUT/pytest_gpu_full.log
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu_full.log | tail -10
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err; cut -c1-2300 $OUT/bench_c2.json
rocprofv3 --kernel-trace --stats -d $OUT/prof_c2 -o c2 --output-format csv -- python bench.py --no-cpu-baseline > $OUT/prof_c2_bench.json 2> /dev/null
grep -E "^\"?(gdv_k)" $OUT/prof_c2/c2_kernel_stats.csv | cut -c1-200
for w in c2 c5; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch_$w -o $w --output-format csv -- python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write_$w -o $w --output-format csv -- python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
for w in c3 c4 c5 c1; do python bench.py --workload $w --no-cpu-baseline > $OUT/bench_$w.json 2>/dev/null; python -c "import json; d=json.load(open('$OUT/bench_$w.json')); print('$w', d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['frac'])"; done
for w in c3 c5; do
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$w -o $w --output-format csv -- python bench.py --workload $w --no-cpu-baseline > $OUT/prof_${w}_bench.json 2> /dev/null
  grep -E "gdv|Scan|Emit" $OUT/prof_$w/${w}_kernel_stats.csv | cut -c1-160
done
find $OUT -name "*.csv" -size +2000k -delete   # raw traces are large; the stats/counter files stay
ls $OUT
