#!/bin/bash
# A/B of planner variants of the C5 kernels on one box: bench ms per Evaluate + per-kernel rocprof averages.
#   gpurun --timeout 900 -- 'bash tools/c5_variants.sh'
export GDV_NO_DISK_CACHE=1 TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/c5v; rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
run() {  # name, env assignments...
  local name=$1; shift; i=$((i+1))
  local ms=$(timeout 200 env "$@" python $R/bench.py --workload c5 --no-cpu-baseline --steps 12 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms_min'], d['verified'])")
  timeout 200 env "$@" rocprofv3 --kernel-trace --stats -d $OUT/v$i -o c5 --output-format csv -- python $R/bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline --no-verify > /dev/null 2>&1
  local k=$(find $OUT/v$i -name "*kernel_stats.csv" | head -1 | xargs grep "^\"gdv_k_" | awk -F, '{printf "%s=%.1fus ", substr($1,2,12), $4/1000}')
  find $OUT/v$i -name "*kernel_trace.csv" -delete
  echo "$name | bench ms/step, min kernel ms, verified: $ms | rocprof: $k"
}
run "product (needle baked, pre-pass unrolled)" A=1
run "needle at run time" GDV_RUNTIME_NEEDLES=1
run "pre-pass rolled" GDV_PREPASS_ROLLED=1
run "product again" A=2
