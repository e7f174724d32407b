#!/bin/bash
# run-to-run variance study of the C2 kernel
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | head -30
for i in 1 2 3; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('c2 full', d['ms_per_step'], r['kernel_ms'], r['kernel_ms_min'], r['kernel_ms_max'], r['achieved'])"
done
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --rows $((1<<27)) 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('c2 2^27', d['ms_per_step'], r['kernel_ms'], r['kernel_ms_min'], r['kernel_ms_max'], r['achieved'])"
GDV_NT=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('c2 NT=0', d['ms_per_step'], r['kernel_ms'], r['kernel_ms_min'], r['kernel_ms_max'], r['achieved'])"
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | head -30
