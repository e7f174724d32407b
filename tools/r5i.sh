export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5i; mkdir -p $O
cd $R
PYTHONPATH=$R timeout 1200 python tools/fuzz_offline.py 12 44 2>&1 | grep -v amdgpu.ids > $O/fuzz_offline.txt; tail -15 $O/fuzz_offline.txt | cut -c1-500
