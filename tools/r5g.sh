export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5g; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_registry_tail.py tests/test_registry_tail_r3.py tests/test_decimal.py tests/test_host_registered.py tests/test_strings.py -m gpu -q --timeout 900 > $O/pytest.log 2>&1; tail -8 $O/pytest.log | cut -c1-300
