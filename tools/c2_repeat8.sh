export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/c2rep8; mkdir -p $O
cd $R
for i in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/run$i.json 2>/dev/null
  python3 -c "
import json; d=json.loads([l for l in open('$O/run$i.json') if l.startswith('{')][-1]); r=d['roofline']
print('eight placements, run $i: ms_per_step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'placements', r.get('placement_trials_ms'), 'verified', d['verified'])"
done | tee $O/c2_repeat8.txt
