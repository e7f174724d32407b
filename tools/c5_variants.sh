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
run "base (U8)" A=1
run "row loop unroll 2" "GDV_RTC_OPT=-DGDV_ROW_UNROLL2"
run "non-temporal string stores" "GDV_RTC_OPT=-DGDV_NT_STRING_STORES"
run "unroll 2 + NT stores" "GDV_RTC_OPT=-DGDV_ROW_UNROLL2 -DGDV_NT_STRING_STORES"
run "window 4 B/row (8 waves/SIMD)" "GDV_RTC_OPT=-DGDV_OUT_WIN=(GDV_U*64*4) -DGDV_SPAN_MAX=(GDV_U*64*24)"
run "window 4 B/row + NT" "GDV_RTC_OPT=-DGDV_OUT_WIN=(GDV_U*64*4) -DGDV_SPAN_MAX=(GDV_U*64*24) -DGDV_NT_STRING_STORES"
run "U4" GDV_U=4
run "U4 + NT" GDV_U=4 "GDV_RTC_OPT=-DGDV_NT_STRING_STORES"
run "base again" A=1
