#!/bin/bash
# round 5, last GPU minutes: the regular-expression GPU test on the final automaton layout
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r5k; mkdir -p $OUT; cd $R
timeout 170 python -m pytest tests/test_registry_tail.py -m gpu -q --timeout 160 -k "regular or to_date or real or per_row or regexp" > $OUT/pytest_regex.log 2>&1; tail -4 $OUT/pytest_regex.log
