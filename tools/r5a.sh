export TMPDIR=/tmp
O=gpurun_out/r5a; mkdir -p $O
python -m pytest tests/test_filter_project.py tests/test_registry_tail.py tests/test_registry_tail_r3.py tests/test_strings.py -m gpu -q --timeout 900 -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python tools/fused_fp_sweep.py 1000000000 250 1000 2>&1 | grep -v amdgpu.ids > $O/fp_sweep.txt; cat $O/fp_sweep.txt
PYTHONPATH=. timeout 300 python tools/filter_string_chain.py 2>&1 | grep -v amdgpu.ids > $O/filter_string_chain.txt; cat $O/filter_string_chain.txt
PYTHONPATH=. timeout 400 python tools/box_states.py 2>&1 | grep -v amdgpu.ids > $O/box_states.txt; cat $O/box_states.txt
