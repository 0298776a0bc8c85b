#!/bin/bash
# GPU box: the multi-rank bench flow rehearsed with N ranks on the one GPU (window exchange), phase marks on stderr
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/q3; mkdir -p $O; cd $R
N=${1:-2}
TG_DIST_DEBUG=1 TG_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus $N --steps 10 --warmup 3 --preroll 12 --no-cpu-baseline > $O/rehearsal$N.json 2> $O/rehearsal$N.err
echo rc $?
grep "dist rank" $O/rehearsal$N.err | tail -20
python - $O/rehearsal$N.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('rehearsal:', round(d['value']/1e6,3), 'M ev/s', round(d['ms_per_step'],4), 'ms', d['config']['exchange'], '|', d['config']['launch'][:90], d['config']['exchange_rows_per_step_rank0'])
except Exception as e: print('no line', e)
PY
