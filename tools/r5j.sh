#!/bin/bash
# round 5, late: the string suites (exact variant's pre-pass), the updated round-3 tail test, the non-ASCII sweep
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r5j; mkdir -p $OUT; cd $R
export GDV_EVIDENCE_PENDING=1
timeout 900 python -m pytest tests/test_strings.py tests/test_registry_tail_r3.py -m gpu -q -x --timeout 600 > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
PYTHONPATH=$R timeout 400 python tools/c5_nonascii_sweep.py 2>&1 | grep -v amdgpu.ids | tee $OUT/c5_nonascii_sweep2.txt
( cd /tmp; rocprofv3 --kernel-trace --stats -d $OUT/prof_c5na -o na --output-format csv -- python $R/tools/c5_nonascii_profile.py > /dev/null 2>&1 )
grep '^"gdv_k_' $(find $OUT/prof_c5na -name "*kernel_stats.csv" | head -1) | cut -c1-120
