timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_decimal.py tests/test_golden.py tests/test_fuzz_trees.py tests/test_api.py tests/test_c_device_interface.py -m gpu -x -q --timeout 600 2>&1 | grep -vE "^Extension modules" | tail -4
for w in c1 c2 c4; do
  for i in 1 2; do
    a=$(GDV_NO_DISK_CACHE=1 GDV_NO_WIDE=1 python bench.py --workload $w --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['roofline']['frac'])")
    b=$(GDV_NO_DISK_CACHE=1 python bench.py --workload $w --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['roofline']['frac'])")
    echo "$w row-per-lane: $a   wide: $b"
  done
done
