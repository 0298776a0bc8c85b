set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05d; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_train.py tests/test_hip_baseline_shapes.py tests/test_hip_parity.py -m gpu -x -q -k "long_history or c1_wikipedia or anonym or sampler" 2>&1 | tee $O/pytest_new.log | tail -6
timeout -k 10 500 python tools/period_drift.py 2 48 2048 > $O/period_drift.json 2> $O/period_drift.err; tail -c 1500 $O/period_drift.json
