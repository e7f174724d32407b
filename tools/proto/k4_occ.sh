export GDV_NO_DISK_CACHE=1
run() { echo "--- $1"; env $2 GDV_TRACE=1 python bench.py --workload c5 --no-cpu-baseline --steps 6 --warmup 2 2>&1 | grep -E "^\[gdv\]|rror" | tail -1 | cut -c1-100; }
run "U8 W4 (default)" "X=1"
run "U4 W4" "GDV_U=4"
run "U16 W4" "GDV_U=16"
run "U8 W4 wpe6" "GDV_RTC_OPT=-DGDV_STRING_KERNEL_ATTR=__attribute__((amdgpu_waves_per_eu(6,8)))"
run "U8 W8" "GDV_WAVES=8"
run "U8 W4 again" "X=1"
