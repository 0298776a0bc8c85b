#!/bin/bash
# Round-3 measurement set (1x MI355X): bench lines, rocprofv3 kernel statistics, PMC traffic passes -> gpurun_out/r03_final/
# usage (on the GPU box): bash tools/measure_r03.sh [TAG]   (files are named r03_*_TAG)
TAG=${1:-v6}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_final
mkdir -p $O
cd $R
set -e
python bench.py --steps 100 --warmup 20 > $O/r03_bench_c2_$TAG.json 2> $O/err.log && echo bench ok
python bench.py --steps 20 --warmup 5 --no-c5s-leg --no-dist-leg --no-cpu-baseline > $O/r03_bench_c2_driver_steps_$TAG.json 2>> $O/err.log
python bench.py --steps 100 --warmup 20 --no-eager --no-cpu-baseline --no-c5s-leg --no-dist-leg > $O/r03_bench_c2_lazy_$TAG.json 2>> $O/err.log
TG_GTAB=0 TG_ATTN_TILE=1 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-c5s-leg --no-dist-leg > $O/r03_bench_c2_attn_tile_$TAG.json 2>> $O/err.log
python bench.py --workload c1 --steps 300 --warmup 50 --no-cpu-baseline > $O/r03_bench_c1_$TAG.json 2>> $O/err.log
python bench.py --workload c3 --steps 100 --warmup 20 --no-cpu-baseline > $O/r03_bench_c3_$TAG.json 2>> $O/err.log
python bench.py --workload c4 --steps 100 --warmup 20 --no-cpu-baseline > $O/r03_bench_c4_$TAG.json 2>> $O/err.log
python bench.py --workload c5s --steps 30 --warmup 4 --no-cpu-baseline > $O/r03_bench_c5s_$TAG.json 2>> $O/err.log
python bench.py --workload c5 --steps 30 --warmup 4 --no-cpu-baseline > $O/r03_bench_c5_counter_$TAG.json 2>> $O/err.log && echo workloads ok
python bench.py --gpus 1 --force-dist --steps 100 --warmup 20 --no-cpu-baseline > $O/r03_bench_c2_partitioned_1rank_$TAG.json 2>> $O/err.log
python bench.py --gpus 1 --force-dist --dist-graphs --steps 100 --warmup 20 --no-cpu-baseline > $O/r03_bench_c2_partitioned_1rank_graphs_$TAG.json 2>> $O/err.log
TG_BENCH_REHEARSAL=1 python bench.py --gpus 4 --steps 10 --warmup 3 --preroll 20 --no-cpu-baseline > $O/r03_rehearsal_4ranks_one_gpu_gloo_$TAG.json 2>> $O/err.log && echo dist ok
for r in none static seq; do python bench.py --train --train-restarter $r --no-cpu-baseline > $O/r03_train_c2_${r}_$TAG.json 2>> $O/err.log; done; echo train ok
cd /tmp && export TMPDIR=/tmp
P="--no-cpu-baseline --no-c5s-leg --no-dist-leg --no-self-check"
rocprofv3 --kernel-trace --stats -d $O/prof_c2 -o c2 -- python $R/bench.py --steps 100 --warmup 20 $P > $O/prof_c2.log 2>&1
python $R/tools/rocpd_stats.py $(find $O/prof_c2 -name '*.db' | head -1) $O/r03_bench_c2_kernel_stats_$TAG.csv > /dev/null && echo stats c2 ok
rocprofv3 --kernel-trace --stats -d $O/prof_c5s -o c5s -- python $R/bench.py --workload c5s --steps 10 --warmup 4 --preroll 40 $P > $O/prof_c5s.log 2>&1
python $R/tools/rocpd_stats.py $(find $O/prof_c5s -name '*.db' | head -1) $O/r03_bench_c5s_kernel_stats_$TAG.csv > /dev/null && echo stats c5s ok
for W in c2 c5s; do
  EXTRA="--steps 10 --warmup 25 --preroll 100"; [ $W = c5s ] && EXTRA="--steps 4 --warmup 4 --preroll 40"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f_$W -o f -- python $R/bench.py --workload $W $EXTRA $P --no-graph > $O/pmc_f_$W.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w_$W -o w -- python $R/bench.py --workload $W $EXTRA $P --no-graph > $O/pmc_w_$W.log 2>&1
  F=$(find $O/pmc_f_$W -name '*counter_collection.csv' | head -1); Wf=$(find $O/pmc_w_$W -name '*counter_collection.csv' | head -1)
  python $R/tools/pmc_traffic.py $F $Wf $O/r03_hbm_traffic_${W}.json | grep -i "gru\|attn_core\|gather\|gemm" || true
  cp $F $O/r03_pmc_FETCH_SIZE_${W}_$TAG.csv; cp $Wf $O/r03_pmc_WRITE_SIZE_${W}_$TAG.csv
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_m -o m -- python $R/bench.py --steps 10 --warmup 25 --preroll 100 $P --no-graph > $O/pmc_m.log 2>&1 || true
cp $(find $O/pmc_m -name '*counter_collection.csv' | head -1) $O/r03_pmc_mfma_c2_$TAG.csv 2>/dev/null || true
rm -rf $O/prof_c2 $O/prof_c5s $O/pmc_f_* $O/pmc_w_* $O/pmc_m; echo done
