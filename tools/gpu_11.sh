#!/bin/bash
export TMPDIR=/tmp
python -m pytest tests/test_c_device_interface.py tests/test_parity_gpu.py -m gpu -q --timeout 900 -k "device or offsets" 2>&1 | tail -5
python bench.py --workload c1 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-900
for U in 8 16; do GDV_U=$U python bench.py --workload c1 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('U=$U', d['ms_per_step'], d['roofline']['achieved'])"; done
