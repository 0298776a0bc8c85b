set -e
mkdir -p gpurun_out
timeout -k 10 500 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
r=d['reference_api_loop']
for k,v in r.items():
    if k=='what': continue
    if k=='restart_mode':
        for kk,vv in v.items(): print(kk, round(vv['value']), round(vv['ms_per_batch'],4), 'loop', round(vv['per_batch_loop']['value']), vv['ap'], vv['per_batch_loop']['ap'], vv['restarted_nodes'])
    else: print(k, round(v['value']), round(v['ms_per_batch'],4), 'loop', round(v['per_batch_loop']['value']))
PY
