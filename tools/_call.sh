set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05e; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_eval.py tests/test_hip_train.py tests/test_hip_parity.py -m gpu -x -q -k "eval or restart or seq" 2>&1 | tee $O/pytest_eval.log | tail -3
python tools/prof_restart_loop.py 200 2>&1 | head -3
