#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
python -m pytest tests -m gpu -q --timeout 1200 -s 2>&1 > $OUT/pytest_gpu_full.log
tail -25 $OUT/pytest_gpu_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py --steps 20 --warmup 3 > $OUT/bench_c2.json 2> $OUT/bench_c2.err
cat $OUT/bench_c2.json
python bench.py --workload c3 --steps 10 --warmup 2 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
cat $OUT/bench_c3.json; tail -3 $OUT/bench_c3.err
