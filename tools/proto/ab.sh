export GDV_NO_DISK_CACHE=1
run() { echo "--- $1"; env $2 GDV_TRACE=1 python bench.py --workload c5 --no-cpu-baseline --steps 8 --warmup 2 2>&1 | grep -E "^\[gdv\]|rror" | tail -1 | cut -c1-100; }
for i in 1 2 3; do
run "aligned copies (head/tail bytes)" "X=1"
run "unaligned copies" "GDV_RTC_OPT=-DGDV_COPY_UNALIGNED=1"
done
GDV_RTC_OPT=-DGDV_COPY_UNALIGNED=1 timeout 300 python -m pytest tests/test_golden.py "tests/test_strings.py::test_hip_strings_match_oracle" -m gpu -x -q 2>&1 | tail -1
