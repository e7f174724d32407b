#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
python -m pytest tests/test_parity_gpu.py -m gpu -q --timeout 900 -k "bitwise" 2>&1 | tail -2
for NTL in 0 1; do for w in c2 c3 c1; do
 GDV_NTLOAD=$NTL python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w NTLOAD=$NTL', d['ms_per_step'], d['roofline']['achieved'])"
done; done
python tools/host_path_rate.py 2>/dev/null
for w in c3 c4; do
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch_$w -o $w --output-format csv -- python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write_$w -o $w --output-format csv -- python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
ls $OUT/pmc_fetch_c3 $OUT/pmc_write_c4
