set -e
python bench.py > gpurun_out/bench_r04_full.json 2> gpurun_out/bench_r04_full.err || (tail -30 gpurun_out/bench_r04_full.err; exit 1)
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r04_full.json'))
print('value', d['value'], 'ms', d['ms_per_step'])
r=d['roofline']; print('roofline', r['bound'], round(r['frac'],4), r['achieved'], 'fabric', r['fabric'] and {k:r['fabric'][k] for k in ('achieved','bytes_per_step','measured_over_algorithmic','flop_per_byte')})
for k,v in d['also'].items(): print(k, {kk:(round(vv,3) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ('fps','conv_frac_of_peak','conv_launches_per_step','error')})
print(d['cpu_baseline']['value'], d['psnr_db_vs_cpu_ref'])
PY
