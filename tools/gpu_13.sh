#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
python -m pytest tests/test_strings.py tests/test_reference_kats.py -m gpu -q --timeout 900 2>&1 | tail -3
for U in 2 4; do GDV_U=$U python bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5 U=$U', d['ms_per_step'], d['value'], d['roofline']['achieved'])"; done
rocprofv3 --kernel-trace --stats -d $OUT/prof_c5 -o c5 --output-format csv -- python bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/prof_c5_bench.json 2> /dev/null
grep -E "^\"?(gdv_k)" $OUT/prof_c5/c5_kernel_stats.csv | cut -c1-200
