set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05d; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_train.py tests/test_hip_parity.py tests/test_dist.py tests/test_hip_eval.py -m gpu -x -q -k "recent_nodes or two_layer or window or partitioned or eval or sharded" 2>&1 | tee $O/pytest_new2.log | tail -6
