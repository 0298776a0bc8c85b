#!/bin/bash
# quick check on the GPU box: C2 bench lines under knob settings given as arguments ("NAME=VAL,NAME=VAL" per run)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/q; mkdir -p $O; cd $R
P="--no-cpu-baseline --no-c5s-leg --no-dist-leg --no-api-loop"
i=0
for cfg in "$@"; do
  i=$((i+1))
  envs=$(echo "$cfg" | tr ',' ' ')
  [ "$cfg" = "-" ] && envs=""
  env $envs timeout -k 10 300 python bench.py --steps 100 --warmup 20 $P > $O/run$i.json 2> $O/err$i.log || { tail -20 $O/err$i.log; exit 1; }
  python - "$cfg" $O/run$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d['value']/1e6,3), round(d['ms_per_step']*1e3,2), {k[:14]:round(v*1e3,1) for k,v in d.get('stages_ms',{}).items() if v>0.002}, d.get('replay_self_check'))
PY
done
