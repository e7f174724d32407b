export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5h; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_filter_project.py tests/test_parity_gpu.py -m gpu -q --timeout 600 -x -k "asynchronous_evaluations or filter or c3 or C3 or selection" > $O/pytest.log 2>&1; tail -4 $O/pytest.log | cut -c1-300
for v in "" "GDV_EMIT_WGS_PER_CU=4" "GDV_EMIT_WGS_PER_CU=3" "GDV_EMIT_WORDS=32" "GDV_EMIT_WORDS=32 GDV_EMIT_WGS_PER_CU=6" "GDV_EMIT_WORDS=32 GDV_EMIT_WGS_PER_CU=5"; do
  echo "== emit variant: [$v]"
  ( cd /tmp; env $v rocprofv3 --kernel-trace --stats -d $O/emit_$(echo $v | tr ' =' '__') -o c3 --output-format csv -- python $R/bench.py --workload c3 --no-cpu-baseline --steps 30 > $O/bench_c3_$(echo $v | tr ' =' '__').json 2>/dev/null )
  f=$(find $O/emit_$(echo $v | tr ' =' '__') -name "*kernel_stats.csv" | head -1)
  grep -a "EmitIndices\|^\"gdv_k_\|^gdv_k_" $f | awk -F, '{printf "   %s  avg %.4f ms x%s\n", substr($1,1,60), $4/1e6, $2}'
  python3 -c "
import json; d=json.loads([l for l in open('$O/bench_c3_$(echo $v | tr ' =' '__').json') if l.startswith('{')][-1]); print('   bench ms_per_step', d['ms_per_step'], 'verified', d['verified'])"
  find $O -name "*kernel_trace.csv" -delete
done
