#!/bin/bash
# round 5, last GPU minutes: the offline fuzz with the late-round generator (general regular expressions, to_date, float text,
# per-row replace / pad arguments), row mode + filter + UINT32 selection mode, bit for bit against the oracle
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r5k; mkdir -p $OUT; cd $R
FUZZ_ONLY=late PYTHONPATH=$R timeout 330 python tools/fuzz_offline.py 200 22 2>&1 | grep -v amdgpu.ids | tee $OUT/fuzz_late.txt | tail -12
