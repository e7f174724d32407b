export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5d; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_filter_project.py -m gpu -q --timeout 600 -x > $O/pytest_fp.log 2>&1; tail -4 $O/pytest_fp.log
timeout 600 python tools/fused_fp_sweep.py 1000000000 250 2>&1 | grep -v amdgpu.ids > $O/fp_sweep.txt; cat $O/fp_sweep.txt
FP_VARIANTS=0,3 timeout 600 python tools/fused_fp_sweep.py 1000000000 1000 60 2>&1 | grep -v amdgpu.ids > $O/fp_sweep_other_selectivities.txt; cat $O/fp_sweep_other_selectivities.txt
PYTHONPATH=. timeout 500 python tools/box_states.py 268435456 2 2>&1 | grep -v amdgpu.ids > $O/box_states_stagger.txt; cat $O/box_states_stagger.txt
