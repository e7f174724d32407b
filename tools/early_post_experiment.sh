#!/bin/bash
# First experiment of the next round (written at the end of round 2, when no GPU minutes were
# left): the planner's early-post shape for string plans (GDV_EARLY_POST=1: lengths pass -> post ->
# staging pass -> wait -> flush), parity first, then an A/B against the default shape on C5.
#   gpurun --timeout 420 -- 'bash tools/early_post_experiment.sh'
# Every step runs under its own timeout: a hang in the new shape must not take the box with it.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/early_post; rm -rf $OUT; mkdir -p $OUT
cd $R
GDV_EARLY_POST=1 timeout 240 python -m pytest tests/test_strings.py tests/test_registry_tail.py tests/test_fuzz_trees.py tests/test_golden.py \
    -m gpu -x -q > $OUT/parity_early_post.log 2>&1
tail -3 $OUT/parity_early_post.log
if ! grep -q " passed" $OUT/parity_early_post.log || grep -q "failed\|error" $OUT/parity_early_post.log; then
  echo "early-post shape is NOT parity-green: no timing taken"; exit 1
fi
for i in 1 2 3; do
  timeout 60 python bench.py --workload c5 --no-cpu-baseline > $OUT/c5_default_$i.json 2>/dev/null
  GDV_EARLY_POST=1 timeout 60 python bench.py --workload c5 --no-cpu-baseline > $OUT/c5_early_$i.json 2>/dev/null
done
python - <<PY
import glob, json
for tag in ("default", "early"):
    ms = sorted(json.load(open(f))["ms_per_step"] for f in glob.glob("$OUT/c5_%s_*.json" % tag))
    print(tag, ms)
PY
