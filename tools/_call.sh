set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_hip_train.py -x -q -m gpu -k "follow_the_device_optimizer or mutual_trajectory" 2>&1 | tee gpurun_out/t_tr.log | tail -25
