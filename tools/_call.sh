set -e
mkdir -p gpurun_out
for r in seq static; do
python bench.py --train --train-restarter $r --train-restart-prob 0.01 --no-cpu-baseline > gpurun_out/r05_train_c2_${r}_lazy_restart_v2.json 2> gpurun_out/train_${r}_lazy.err
done
python bench.py --train --train-restarter seq --no-cpu-baseline > gpurun_out/r05_train_c2_seq_v2.json 2>> gpurun_out/train_seq_lazy.err
python - <<'PY'
import json
for f in ('r05_train_c2_seq_lazy_restart_v2','r05_train_c2_static_lazy_restart_v2','r05_train_c2_seq_v2'):
    d=json.loads(open(f'gpurun_out/{f}.json').read().strip().splitlines()[-1])
    print(f,round(d['value']),round(d['ms_per_step'],3),d.get('lazy_restart_loop'))
PY
