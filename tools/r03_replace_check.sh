#!/bin/bash
# replace() answered by the byte sweep: GPU parity of every test that touches replace / two-stage plans /
# random string trees, then the registry-tail timings.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/repl; rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 400 python -m pytest tests/test_registry_tail.py tests/test_strings.py tests/test_fuzz_trees.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -5 | tee $OUT/pytest.txt
PYTHONPATH=$R timeout 150 python tools/registry_tail_timing.py 2>&1 | grep -v amdgpu.ids | tee $OUT/registry_tail_timing.txt
