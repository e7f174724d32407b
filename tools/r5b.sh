export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5b; mkdir -p $O
cd $R
python -m pytest tests/test_strings.py tests/test_fuzz_trees.py tests/test_registry_tail_r3.py tests/test_filter_project.py -m gpu -q --timeout 900 -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
PYTHONPATH=. timeout 300 python tools/filter_string_chain.py 2>&1 | grep -v amdgpu.ids > $O/filter_string_chain.txt; cat $O/filter_string_chain.txt
FP_VARIANTS=0,1,8,9,10 python tools/fused_fp_sweep.py 1000000000 250 2>&1 | grep -v amdgpu.ids > $O/fp_sweep.txt; cat $O/fp_sweep.txt
cd /tmp
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  PYTHONPATH=$R timeout 400 rocprofv3 --pmc $set --kernel-trace -d $O/pmc_$tag -o bs --output-format csv -- python $R/tools/box_states.py 268435456 4 --no-carve 2>&1 | grep -v amdgpu.ids | grep "^round\|^#" > $O/box_states_$tag.txt
  cat $O/box_states_$tag.txt
  python $R/tools/box_states_pmc.py $O/pmc_$tag | tee $O/box_states_pmc_$tag.txt
  find $O/pmc_$tag -name "*.csv" -size +2000k -delete
done
