set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_hip_eval.py -x -q -m gpu -k "restart" 2>&1 | tee gpurun_out/t_eval.log | tail -15
for G in 4 1 2 8; do echo G=$G; TG_EVAL_RESTART_GROUP=$G timeout -k 10 200 python tools/prof_restart_loop.py 500 2>&1 | grep "ms per batch"; done
