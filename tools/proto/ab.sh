export GDV_NO_DISK_CACHE=1
one() { env $2 python bench.py --workload $1 --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])"; }
for w in c3 c2; do for i in 1 2; do
  echo "$w plain word stores $(one $w X=1) | non-temporal $(one $w GDV_RTC_OPT=-DGDV_WORD_NT=1)"
done; done
