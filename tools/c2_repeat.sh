# Three consecutive driver-style runs of the default bench line on ONE box (round 5: with the placement search) — do they agree?
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/c2rep5; mkdir -p $O
cd $R
for i in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/run$i.json 2>/dev/null
  python3 -c "
import json; d=json.loads([l for l in open('$O/run$i.json') if l.startswith('{')][-1]); r=d['roofline']
print('run $i: ms_per_step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'frac_of_measured_ceiling', r.get('frac_of_measured_ceiling'), 'placements', r.get('placement_trials_ms'), 'traffic', r.get('traffic') is not None, 'verified', d['verified'])"
done | tee $O/c2_repeat.txt
for i in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --placements 1 > $O/single$i.json 2>/dev/null
  python3 -c "
import json; d=json.loads([l for l in open('$O/single$i.json') if l.startswith('{')][-1]); r=d['roofline']
print('--placements 1, run $i: ms_per_step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'])"
done | tee -a $O/c2_repeat.txt
