#!/bin/bash
# round-3 mid-way check: GPU suite (minus full-size), bench line with box telemetry, registry-tail timing
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03b; mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -x -q --ignore=tests/test_full_size.py 2>&1 | tail -4
python bench.py --no-cpu-baseline > $OUT/bench_c2.json 2>$OUT/bench_c2.err; python - <<PY
import json
d=json.load(open("$OUT/bench_c2.json"))
print(d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_of_measured_ceiling"))
print(json.dumps(d["roofline"].get("box"))[:1500])
PY
PYTHONPATH=$R timeout 120 python tools/registry_tail_timing.py 2>&1 | tail -12
tools/small_batch_bench | tail -6
