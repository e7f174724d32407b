#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
python -m pytest tests -m gpu -q --timeout 1800 2>&1 > $OUT/pytest_gpu_full.log
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu_full.log | tail -30
python bench.py --workload c5 --steps 10 --warmup 2 > $OUT/bench_c5.json 2>/dev/null; cut -c1-400 $OUT/bench_c5.json
rocprofv3 --kernel-trace --stats -d $OUT/prof_c5 -o c5 --output-format csv -- python bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/prof_c5_bench.json 2> /dev/null
grep -E "^\"?(gdv_k|void gdv|gdv::)" $OUT/prof_c5/c5_kernel_stats.csv | cut -c1-200
