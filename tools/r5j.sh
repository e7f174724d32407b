#!/bin/bash
# round 5, late: regular expressions, three-stage asynchronous plans
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r5j; mkdir -p $OUT; cd $R
export GDV_EVIDENCE_PENDING=1
timeout 900 python -m pytest tests/test_registry_tail.py tests/test_strings.py -m gpu -q --timeout 600 -k "regular or regexp or stage or to_date or real or per_row" > $OUT/pytest2.log 2>&1; tail -8 $OUT/pytest2.log
