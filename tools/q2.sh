#!/bin/bash
# quick check on the GPU box: the one-rank point of the partitioned multi-GPU form (window exchange vs RCCL)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/q2; mkdir -p $O; cd $R
for ex in ipc rccl; do
  timeout -k 10 300 python bench.py --gpus 1 --force-dist --dist-exchange $ex --steps 100 --warmup 20 --no-cpu-baseline > $O/part_$ex.json 2> $O/part_$ex.err || { tail -30 $O/part_$ex.err; exit 1; }
  python - $O/part_$ex.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d['value']/1e6,3), round(d['ms_per_step']*1e3,2), 'host', round(d['host_enqueue_ms_per_step_rank0']*1e3,2), d['config']['launch'][:80], d['stages_ms_rank0'])
PY
done
