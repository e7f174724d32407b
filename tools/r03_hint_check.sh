#!/bin/bash
# Host-side change only (var-len capacity hint): the GPU tests that exercise var-len outputs, then the
# registry-tail timings again (each Evaluate after the first now runs ONCE: no capacity retry).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/hint; rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 400 python -m pytest tests/test_registry_tail.py tests/test_strings.py tests/test_c_device_interface.py tests/test_cxx_api.py tests/test_pyarrow_gandiva.py tests/test_jni_flat.py -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest.txt
PYTHONPATH=$R timeout 150 python tools/registry_tail_timing.py 2>&1 | grep -v amdgpu.ids | tee $OUT/registry_tail_timing.txt
