#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
python -m pytest tests -m gpu -q --timeout 1800 2>&1 > $OUT/pytest_gpu_full.log
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu_full.log | tail -30
rm -f $OUT/sweep_c3.log
for U in 4 8 16; do for WV in 4 8; do for GM in 8 16; do
  echo "U=$U WAVES=$WV GRID_MULT=$GM" >> $OUT/sweep_c3.log
  GDV_U=$U GDV_WAVES=$WV GDV_GRID_MULT=$GM timeout 300 python bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['achieved'])" >> $OUT/sweep_c3.log
done; done; done
cat $OUT/sweep_c3.log
rocprofv3 --kernel-trace --stats -d $OUT/prof_c3 -o c3 --output-format csv -- python bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/prof_c3_bench.json 2> /dev/null
grep -E "^\"?(gdv_k|void gdv|gdv::)" $OUT/prof_c3/c3_kernel_stats.csv | cut -c1-200
python bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-200
