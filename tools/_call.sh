set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05b; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_train.py tests/test_hip_eval.py tests/test_hip_parity.py -m gpu -x -q -k "seq or mutual or train or restart or eval or mlp_merge" > $O/pytest.log 2>&1; echo pytest rc $?
tail -15 $O/pytest.log
python bench.py --train --train-restarter seq --no-cpu-baseline > $O/train_seq.json 2> $O/train_seq.err; tail -c 600 $O/train_seq.json
