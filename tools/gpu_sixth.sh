#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
python -m pytest tests -m gpu -q --timeout 1800 2>&1 > $OUT/pytest_gpu_full.log
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu_full.log | tail -30
python -m pytest tests -m "not gpu" -q --timeout 1800 2>&1 | tail -3
python bench.py --steps 20 --warmup 3 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; cut -c1-1600 $OUT/bench_c2.json
python bench.py --workload c3 --steps 20 --warmup 3 > $OUT/bench_c3.json 2> $OUT/bench_c3.err; cut -c1-400 $OUT/bench_c3.json
nproc; free -g | head -2
