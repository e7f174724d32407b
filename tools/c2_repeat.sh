#!/bin/bash
# C2, the same command three times plain and three times under rocprofv3, back to back on one box:
# how far apart two runs of one kernel on one box are (the box alternates between two states ~15 % apart
# at constant sclk / mclk), and that bench.py's kernel_ms agrees with rocprofv3's average WITHIN a run.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/c2rep; rm -rf $OUT; mkdir -p $OUT
cd /tmp
for i in 1 2 3; do python $R/bench.py --no-cpu-baseline --no-verify --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; b=r['box']; print('plain   run $i: ms_per_step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'min', r['kernel_ms_min'], 'max', r['kernel_ms_max'], 'sclk', b.get('sclk_mhz',{}).get('mean'), 'power_w', b.get('power_w',{}).get('mean'), 'ceiling on its own buffers', r.get('measured_ceiling'), 'frac', r.get('frac_of_measured_ceiling'))"; done
for i in 1 2 3; do rocprofv3 --kernel-trace --stats -d $OUT/r$i -o c2 --output-format csv -- python $R/bench.py --no-cpu-baseline --no-verify --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); r=d['roofline']; b=r['box']; print('rocprof run $i: ms_per_step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'sclk', b.get('sclk_mhz',{}).get('mean'), 'power_w', b.get('power_w',{}).get('mean'), 'ceiling', r.get('measured_ceiling'), 'frac', r.get('frac_of_measured_ceiling'))"
  grep '^"gdv_k_' $(find $OUT/r$i -name "*kernel_stats.csv" | head -1) | awk -F, '{printf "          rocprofv3 average of %s: %.4f ms over %s calls (pre-warm + warm-up + timed steps)\n", $1, $4/1e6, $2}'; done
