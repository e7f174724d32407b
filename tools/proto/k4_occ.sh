export GDV_NO_DISK_CACHE=1
for w in "" 4 5 6 7 8; do
  if [ -z "$w" ]; then opt=""; else opt="-DGDV_STRING_KERNEL_ATTR=__attribute__((amdgpu_waves_per_eu($w,8)))"; fi
  echo "--- waves_per_eu=$w"
  GDV_RTC_OPT="$opt" GDV_TRACE=1 python bench.py --workload c5 --no-cpu-baseline --steps 6 --warmup 2 2>&1 | grep "^\[gdv\]" | tail -2 | cut -c1-90
done
