#!/bin/bash
# first GPU contact: parity tests, then bench + a knob sweep on C2
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 10 --warmup 2 --rows $((1<<27)) --no-cpu-baseline > gpurun_out/bench_c2_small.json 2> gpurun_out/bench_c2_small.err
for U in 2 4 8; do for WV in 4 8; do for NT in 0 1; do
  echo "U=$U WAVES=$WV NT=$NT" >> gpurun_out/sweep.log
  GDV_U=$U GDV_WAVES=$WV GDV_NT=$NT timeout 300 python bench.py --steps 10 --warmup 2 --rows $((1<<27)) --no-cpu-baseline >> gpurun_out/sweep.log 2>> gpurun_out/sweep.err
done; done; done
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
tail -5 gpurun_out/pytest_gpu.log
cat gpurun_out/bench_c2.json
