set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05e; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_train.py -m gpu -x -q -k "empty_and_minimal" 2>&1 | tee $O/pytest_edge.log | tail -12
