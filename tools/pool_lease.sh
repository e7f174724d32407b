python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; p=r.get('placement',{}); s=(p.get('sets') or [{}])[0]
print('ms_per_step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'verified', d['verified'], 'frac', r['frac'], '| first plain allocation frac', r.get('frac_first_plain_allocation'), '| pool candidates GB/s', s.get('rates_gbs'), 'kept', s.get('kept'), 'reserve_ms', s.get('reserve_ms'), '| frac_of_measured_ceiling', r.get('frac_of_measured_ceiling'), '| sub-tiles per wave of the ceiling instrument', (r.get('box',{}).get('ceiling_same_shape') or {}).get('subtiles_per_wave'))"
