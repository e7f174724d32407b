export GDV_NO_DISK_CACHE=1
run() { echo "--- $1"; env $2 GDV_TRACE=1 python bench.py --workload c5 --no-cpu-baseline --steps 8 --warmup 2 2>&1 | grep -E "^\[gdv\]|rror" | tail -1 | cut -c1-100; }
for i in 1 2 3; do
run "flat copy in shadow" "X=1"
run "flat copy after sweep" "GDV_RTC_OPT=-DGDV_ABL=128"
done
