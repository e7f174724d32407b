export GDV_NO_DISK_CACHE=1
one() { env $2 python bench.py --workload $1 --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['roofline']['frac'])"; }
for w in c3 c2 c1 c4; do
  for i in 1 2; do
    echo "$w default: $(one $w X=1)   load fence: $(one $w GDV_LOAD_FENCE=1)"
  done
done
