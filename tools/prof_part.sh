#!/bin/bash
# rocprofv3 kernel statistics of the one-rank partitioned form (window exchange)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_part; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/p -o part -- python $R/bench.py --gpus 1 --force-dist --steps 100 --warmup 20 --no-cpu-baseline > $O/run.log 2>&1
python $R/tools/rocpd_stats.py $(find $O/p -name '*.db' | head -1) $O/part_kernel_stats.csv | head -30
rm -rf $O/p
