#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
python -m pytest tests -m gpu -q --timeout 1200 2>&1 | tail -60 > $OUT/pytest_gpu.log
# per-kernel timing of the bench command
rocprofv3 --kernel-trace --stats -d $OUT/prof_c2 -o c2 --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/prof_c2_bench.json 2> $OUT/prof_c2.err
# HBM traffic counters, one pass each (TCC slot limits)
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o c2 --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o c2 --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err
find $OUT -name "*.csv" | head -20
tail -15 $OUT/pytest_gpu.log
