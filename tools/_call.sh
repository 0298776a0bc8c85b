set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05c; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_eval.py -m gpu -x -q > $O/pytest.log 2>&1; echo pytest rc $?
tail -2 $O/pytest.log
python tools/prof_restart_loop.py > $O/prof_restart.txt 2>&1; head -4 $O/prof_restart.txt
