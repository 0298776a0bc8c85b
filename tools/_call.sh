set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_hip_eval.py tests/test_cabi_host.py -x -q -m gpu -k "restart or layout" 2>&1 | tee gpurun_out/t_eval.log | tail -15
timeout -k 10 200 python tools/prof_restart_loop.py 500 2>&1 | grep "ms per batch"
