#!/bin/bash
# A/B of compile-time variants of the C5 kernels on one box: device ms per Evaluate (GDV_TRACE).
#   gpurun --timeout 600 -- 'bash tools/c5_variants.sh'
export GDV_NO_DISK_CACHE=1
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
run() {  # name, env assignments...
  local name=$1; shift
  local t=$(env "$@" GDV_TRACE=1 python bench.py --workload c5 --no-cpu-baseline --steps 8 --warmup 2 2>&1 | grep "^\[gdv\]" | tail -4 | sed 's/.*device_ms=\([0-9.]*\).*/\1/' | tr '\n' ' ')
  echo "$name: $t"
}
run "base U8" A=1
run "base U4" GDV_U=4
run "flush aligned U8" "GDV_RTC_OPT=-DGDV_FLUSH_ALIGNED"
run "flush aligned U4" GDV_U=4 "GDV_RTC_OPT=-DGDV_FLUSH_ALIGNED"
run "U8 small LDS (6 B/row out, 24 B/row span)" "GDV_RTC_OPT=-DGDV_OUT_WIN=(GDV_U*64*6) -DGDV_SPAN_MAX=(GDV_U*64*24)"
run "U8 small LDS + aligned flush" "GDV_RTC_OPT=-DGDV_FLUSH_ALIGNED -DGDV_OUT_WIN=(GDV_U*64*6) -DGDV_SPAN_MAX=(GDV_U*64*24)"
run "U16" GDV_U=16
run "U8 W8" GDV_WAVES=8
run "U8 W2" GDV_WAVES=2
