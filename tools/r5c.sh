export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5c; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_filter_project.py -m gpu -q --timeout 600 -x > $O/pytest_fp.log 2>&1; tail -4 $O/pytest_fp.log
FP_VARIANTS=0,1,11,12,13,14,15 timeout 600 python tools/fused_fp_sweep.py 1000000000 250 2>&1 | grep -v amdgpu.ids > $O/fp_sweep.txt; cat $O/fp_sweep.txt
timeout 900 python -m pytest tests/test_strings.py tests/test_fuzz_trees.py -m gpu -q --timeout 600 -x > $O/pytest_str.log 2>&1; tail -4 $O/pytest_str.log
PYTHONPATH=. timeout 300 python tools/filter_string_chain.py 2>&1 | grep -v amdgpu.ids > $O/filter_string_chain.txt; cat $O/filter_string_chain.txt
