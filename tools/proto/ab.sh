export GDV_NO_DISK_CACHE=1
run() { echo "--- $1"; env $2 GDV_TRACE=1 python bench.py --workload c5 --no-cpu-baseline --steps 8 --warmup 2 2>&1 | grep -E "^\[gdv\]|rror" | tail -1 | cut -c1-100; }
for i in 1 2; do
run "U4 W4" "X=1"
run "U4 W8" "GDV_WAVES=8"
run "U4 W16" "GDV_WAVES=16"
done
