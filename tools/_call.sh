set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05b; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_hip_train.py tests/test_hip_eval.py tests/test_hip_parity.py -m gpu -x -q -k "seq or mutual or train or restart or eval or mlp_merge" > $O/pytest.log 2>&1; echo pytest rc $?
tail -4 $O/pytest.log
python bench.py --train --train-restarter seq --no-cpu-baseline > $O/train_seq.json 2> $O/train_seq.err; tail -c 400 $O/train_seq.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_seq -o seq -- python $R/bench.py --train --train-restarter seq --steps 30 --warmup 10 --no-cpu-baseline > $O/prof_seq.log 2>&1
python $R/tools/rocpd_stats.py $(find $O/prof_seq -name '*.db' | head -1) $O/r05_train_c2_seq_kernel_stats_v6.csv > /dev/null && echo stats ok
rm -rf $O/prof_seq
