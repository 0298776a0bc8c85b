#!/bin/bash
# Round-end measurement set (1x MI355X): bench lines, kernel statistics, PMC traffic passes -> gpurun_out/final/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
python bench.py --steps 100 --warmup 20 > $O/bench_c2.json 2> $O/bench_c2.err && echo bench ok
python bench.py --steps 100 --warmup 20 --no-fuse --no-cpu-baseline > $O/bench_c2_nofuse.json 2>> $O/bench_c2.err
python bench.py --workload c5s --steps 30 --warmup 20 --no-cpu-baseline > $O/bench_c5s.json 2>> $O/bench_c2.err && echo c5s ok
python bench.py --workload c3 --steps 30 --warmup 8 --no-cpu-baseline > $O/bench_c3.json 2>> $O/bench_c2.err
python bench.py --workload c4 --steps 30 --warmup 8 --no-cpu-baseline > $O/bench_c4.json 2>> $O/bench_c2.err
for r in none static seq; do python bench.py --train --train-restarter $r --no-cpu-baseline > $O/train_$r.json 2>> $O/bench_c2.err; done; echo train ok
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_c2 -o c2 -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $O/prof_c2.log 2>&1
python $R/tools/rocpd_stats.py $(find $O/prof_c2 -name '*.db' | head -1) $O/kernel_stats_c2.csv > /dev/null && echo stats ok
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o f -- python $R/bench.py --steps 10 --warmup 25 --no-cpu-baseline --no-graph > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o w -- python $R/bench.py --steps 10 --warmup 25 --no-cpu-baseline --no-graph > $O/pmc_w.log 2>&1
python $R/tools/pmc_traffic.py $(find $O/pmc_f -name '*counter_collection.csv' | head -1) $(find $O/pmc_w -name '*counter_collection.csv' | head -1) $O/hbm_traffic.json | grep -i "gru\|attn_core" 
rm -rf $O/prof_c2/*/*.db.tmp; find $O -name '*.db' -size +20M -delete; echo done
