export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5f; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_strings.py tests/test_registry_tail.py tests/test_registry_tail_r3.py tests/test_fuzz_trees.py tests/test_parity_gpu.py -m gpu -q --timeout 900 -x > $O/pytest.log 2>&1; tail -6 $O/pytest.log
PYTHONPATH=$R timeout 200 python tools/registry_tail_timing.py > $O/registry_tail_timing.txt 2>&1; cat $O/registry_tail_timing.txt
GDV_NO_ASYNC_TWO_STAGE=1 PYTHONPATH=$R timeout 200 python tools/registry_tail_timing.py 2>&1 | sed 's/^/[stage by stage, rounds 3-4] /' | tee -a $O/registry_tail_timing.txt
