#!/bin/bash
# quick C2 bench line: tools/q.sh TAG [ENV=VAL ...] -- [bench args]; prints events/s, ms/step and the stage times
tag=$1; shift
envs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
[ "$1" == "--" ] && shift
mkdir -p gpurun_out
env "${envs[@]}" python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-c5s-leg --no-dist-leg --no-self-check "$@" > gpurun_out/q_$tag.json 2> gpurun_out/q_$tag.err || { tail -5 gpurun_out/q_$tag.err; exit 1; }
python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
r = json.loads(open(f'gpurun_out/q_{tag}.json').read().strip().splitlines()[-1])
st = r.get('stages_ms') or r.get('stage_ms') or {}
print(tag, f"{r['value']/1e6:.3f} M ev/s  {r['ms_per_step']:.4f} ms", {k: round(v*1e3, 1) for k, v in st.items() if v > 0.0005} if isinstance(st, dict) else '')
PY
