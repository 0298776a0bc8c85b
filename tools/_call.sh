set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_hip_eval.py -x -q -m gpu 2>&1 | tee gpurun_out/t_eval.log | tail -5
