#!/bin/bash
# usage: tools/prof_bench.sh TAG [extra bench args]   -> gpurun_out/prof_TAG/ (rocpd db) + gpurun_out/TAG.csv (per-kernel summary)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o $TAG -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline "$@" > $R/gpurun_out/$TAG.log 2>&1
DB=$(find $R/gpurun_out/prof_$TAG -name '*.db' | head -1)
python $R/tools/rocpd_stats.py $DB $R/gpurun_out/$TAG.csv | cut -c1-150 | head -16
grep '"metric"' $R/gpurun_out/$TAG.log | tail -1 | cut -c1-200
