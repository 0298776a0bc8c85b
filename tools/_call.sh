set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05c; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_train.py tests/test_hip_eval.py tests/test_hip_parity.py -m gpu -x -q -k "seq or mutual or train or restart or eval or mlp_merge" > $O/pytest.log 2>&1; echo pytest rc $?
tail -2 $O/pytest.log
python bench.py --train --train-restarter seq --no-cpu-baseline > $O/t.json 2> $O/t.err; python -c "import json;j=json.load(open('$O/t.json'));print('train seq',round(j['ms_per_step'],4),int(j['value']))"
