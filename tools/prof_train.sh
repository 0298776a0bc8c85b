#!/bin/bash
# usage: tools/prof_train.sh TAG [extra bench args]   -> gpurun_out/prof_TAG/TAG_results.db + gpurun_out/TAG.log
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o $TAG -- python $R/bench.py --train --steps 50 --warmup 10 --no-graph "$@" > $R/gpurun_out/$TAG.log 2>&1
grep '"metric"' $R/gpurun_out/$TAG.log | tail -1 | cut -c1-260
