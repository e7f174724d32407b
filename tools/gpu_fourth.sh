#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
python -m pytest tests -m gpu -q --timeout 1800 2>&1 > $OUT/pytest_gpu_full.log
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu_full.log | tail -30
for w in c4 c5 c3; do
  timeout 900 python bench.py --workload $w --steps 10 --warmup 2 > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  cut -c1-900 $OUT/bench_$w.json; tail -2 $OUT/bench_$w.err
done
# per-kernel breakdown of the filter and string paths
rocprofv3 --kernel-trace --stats -d $OUT/prof_c3 -o c3 --output-format csv -- python bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/prof_c3_bench.json 2> /dev/null
rocprofv3 --kernel-trace --stats -d $OUT/prof_c5 -o c5 --output-format csv -- python bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/prof_c5_bench.json 2> /dev/null
rocprofv3 --kernel-trace --stats -d $OUT/prof_c4 -o c4 --output-format csv -- python bench.py --workload c4 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/prof_c4_bench.json 2> /dev/null
for w in c3 c4 c5; do echo "== $w"; grep -E "^\"?(gdv_k|void gdv|gdv::)" $OUT/prof_$w/${w}_kernel_stats.csv | cut -c1-160; done
