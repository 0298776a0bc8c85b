set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_dist.py -m gpu -x -q -k "exchange_period" 2>&1 | tail -4
