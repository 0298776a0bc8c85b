set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05d; mkdir -p $O
cd $R
N="--no-cpu-baseline --no-c5s-leg --no-dist-leg --no-api-loop --no-self-check"
for v in 0 1; do
HIP_FORCE_DEV_KERNARG=$v python bench.py --steps 100 --warmup 20 $N > $O/k$v.json 2>/dev/null; python -c "import json;j=json.load(open('$O/k$v.json'));print('DEV_KERNARG=$v', round(j['ms_per_step'],5), int(j['value']), j['timed_region']['ms_per_step_all'], {k:v['avg_ms'] for k,v in j['kernels_ms'].items()})"
done
python bench.py --steps 100 --warmup 20 $N > $O/kd.json 2>/dev/null; python -c "import json;j=json.load(open('$O/kd.json'));print('default', round(j['ms_per_step'],5), int(j['value']), j['timed_region']['ms_per_step_all'])"
