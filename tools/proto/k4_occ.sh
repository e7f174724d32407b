export GDV_NO_DISK_CACHE=1
run() { echo "--- $1"; env $2 GDV_TRACE=1 python bench.py --workload c5 --no-cpu-baseline --steps 8 --warmup 2 2>&1 | grep -E "^\[gdv\]|rror" | tail -1 | cut -c1-100; }
run "wsleep 4 (default)" "X=1"
run "wsleep 1" "GDV_RTC_OPT=-DGDV_LB_WSLEEP=1"
run "wsleep 2" "GDV_RTC_OPT=-DGDV_LB_WSLEEP=2"
run "wsleep 8" "GDV_RTC_OPT=-DGDV_LB_WSLEEP=8"
run "wsleep 16" "GDV_RTC_OPT=-DGDV_LB_WSLEEP=16"
run "wsleep 4 again" "X=1"
