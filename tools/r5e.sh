export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r5e; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu_full.log 2>&1; tail -6 $O/pytest_gpu_full.log
python bench.py --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err; python3 -c "
import json,sys
d=json.loads([l for l in open('$O/bench_c2.json') if l.startswith('{')][-1]); r=d['roofline']
print('C2', d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('placement_trials_ms'), r.get('frac_of_measured_ceiling'), d['verified'])"
python bench.py --workload c4 --no-cpu-baseline > $O/bench_c4.json 2>/dev/null; python3 -c "
import json,sys
d=json.loads([l for l in open('$O/bench_c4.json') if l.startswith('{')][-1]); r=d['roofline']
print('C4', d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('placement_trials_ms'), d['verified'])"
