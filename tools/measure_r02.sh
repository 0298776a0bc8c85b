#!/bin/bash
# Round-2 measurement set (1x MI355X): bench lines, rocprofv3 kernel statistics, PMC traffic passes -> gpurun_out/r02_final/
# usage (on the GPU box): bash tools/measure_r02.sh [TAG]   (files are named r02_*_TAG)
TAG=${1:-v1}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_final
mkdir -p $O
cd $R
set -e
python bench.py --steps 100 --warmup 20 > $O/r02_bench_c2_$TAG.json 2> $O/err.log && echo bench ok
python bench.py --steps 100 --warmup 20 --no-eager --no-cpu-baseline --no-c5s-leg > $O/r02_bench_c2_lazy_$TAG.json 2>> $O/err.log
python bench.py --steps 100 --warmup 20 --no-fuse --no-eager --no-cpu-baseline --no-c5s-leg > $O/r02_bench_c2_nofuse_lazy_$TAG.json 2>> $O/err.log
python bench.py --workload c1 --steps 300 --warmup 50 --no-cpu-baseline > $O/r02_bench_c1_$TAG.json 2>> $O/err.log
python bench.py --workload c3 --steps 100 --warmup 20 --no-cpu-baseline > $O/r02_bench_c3_$TAG.json 2>> $O/err.log
python bench.py --workload c4 --steps 100 --warmup 20 --no-cpu-baseline > $O/r02_bench_c4_$TAG.json 2>> $O/err.log
python bench.py --workload c5s --steps 20 --warmup 10 --no-cpu-baseline > $O/r02_bench_c5s_$TAG.json 2>> $O/err.log && echo workloads ok
for r in none static seq; do python bench.py --train --train-restarter $r --no-cpu-baseline > $O/r02_train_c2_${r}_$TAG.json 2>> $O/err.log; done; echo train ok
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_c2 -o c2 -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-c5s-leg > $O/prof_c2.log 2>&1
python $R/tools/rocpd_stats.py $(find $O/prof_c2 -name '*.db' | head -1) $O/r02_bench_c2_kernel_stats_$TAG.csv > /dev/null && echo stats c2 ok
rocprofv3 --kernel-trace --stats -d $O/prof_c5s -o c5s -- python $R/bench.py --workload c5s --steps 10 --warmup 10 --no-cpu-baseline > $O/prof_c5s.log 2>&1
python $R/tools/rocpd_stats.py $(find $O/prof_c5s -name '*.db' | head -1) $O/r02_bench_c5s_kernel_stats_$TAG.csv > /dev/null && echo stats c5s ok
for W in c2 c5s; do
  EXTRA="--steps 10 --warmup 25 --preroll 100"; [ $W = c5s ] && EXTRA="--steps 4 --warmup 4 --preroll 16"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f_$W -o f -- python $R/bench.py --workload $W $EXTRA --no-cpu-baseline --no-graph --no-c5s-leg > $O/pmc_f_$W.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w_$W -o w -- python $R/bench.py --workload $W $EXTRA --no-cpu-baseline --no-graph --no-c5s-leg > $O/pmc_w_$W.log 2>&1
  F=$(find $O/pmc_f_$W -name '*counter_collection.csv' | head -1); Wf=$(find $O/pmc_w_$W -name '*counter_collection.csv' | head -1)
  python $R/tools/pmc_traffic.py $F $Wf $O/r02_hbm_traffic_${W}.json | grep -i "gru\|attn_core\|gather" || true
  cp $F $O/r02_pmc_FETCH_SIZE_${W}_$TAG.csv; cp $Wf $O/r02_pmc_WRITE_SIZE_${W}_$TAG.csv
done
# the copy form of the eager step (stand-alone memory-gather launch) at C5 shape: traffic of that kernel
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f_copy -o f -- python $R/bench.py --workload c5s --steps 4 --warmup 4 --preroll 16 --no-cpu-baseline --no-graph --eager-copy > $O/pmc_f_copy.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w_copy -o w -- python $R/bench.py --workload c5s --steps 4 --warmup 4 --preroll 16 --no-cpu-baseline --no-graph --eager-copy > $O/pmc_w_copy.log 2>&1
python $R/tools/pmc_traffic.py $(find $O/pmc_f_copy -name '*counter_collection.csv' | head -1) $(find $O/pmc_w_copy -name '*counter_collection.csv' | head -1) $O/r02_hbm_traffic_c5s_copy.json | grep -i "gather" || true
rm -rf $O/pmc_f_copy $O/pmc_w_copy
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_m -o m -- python $R/bench.py --steps 10 --warmup 25 --preroll 100 --no-cpu-baseline --no-graph --no-c5s-leg > $O/pmc_m.log 2>&1 || true
cp $(find $O/pmc_m -name '*counter_collection.csv' | head -1) $O/r02_pmc_mfma_c2_$TAG.csv 2>/dev/null || true
rm -rf $O/prof_c2 $O/prof_c5s $O/pmc_f_* $O/pmc_w_* $O/pmc_m; echo done
