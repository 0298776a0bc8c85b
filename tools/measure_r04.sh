#!/bin/bash
# Round-4 measurement set (1x MI355X): bench lines, rocprofv3 kernel statistics, PMC traffic passes -> gpurun_out/r04_final/
# usage (on the GPU box): bash tools/measure_r04.sh [TAG] [PART]   (files are named r04_*_TAG; PART: a = lines, b = profiles)
TAG=${1:-v2}; PART=${2:-ab}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_final
mkdir -p $O
cd $R
set -e
N="--no-cpu-baseline --no-c5s-leg --no-dist-leg --no-api-loop"
if [[ $PART == *a* ]]; then
python bench.py --steps 100 --warmup 20 > $O/r04_bench_c2_$TAG.json 2> $O/err.log && echo bench ok
python bench.py --steps 20 --warmup 5 $N > $O/r04_bench_c2_driver_steps_$TAG.json 2>> $O/err.log
python bench.py --steps 100 --warmup 20 --no-eager $N > $O/r04_bench_c2_lazy_$TAG.json 2>> $O/err.log
TG_GRU_SPLIT=1 python bench.py --steps 100 --warmup 20 $N > $O/r04_bench_c2_gru_split1_$TAG.json 2>> $O/err.log
TG_GRU_SPLIT=2 python bench.py --steps 100 --warmup 20 $N > $O/r04_bench_c2_gru_split2_$TAG.json 2>> $O/err.log
TG_PREFETCH_SPLIT=1 python bench.py --steps 100 --warmup 20 $N > $O/r04_bench_c2_prefetch_split_$TAG.json 2>> $O/err.log
python bench.py --workload c1 --steps 300 --warmup 50 --no-cpu-baseline > $O/r04_bench_c1_$TAG.json 2>> $O/err.log
python bench.py --workload c3 --steps 100 --warmup 20 --no-cpu-baseline > $O/r04_bench_c3_$TAG.json 2>> $O/err.log
python bench.py --workload c4 --steps 100 --warmup 20 --no-cpu-baseline > $O/r04_bench_c4_$TAG.json 2>> $O/err.log
python bench.py --workload c5s --steps 30 --warmup 4 --no-cpu-baseline > $O/r04_bench_c5s_$TAG.json 2>> $O/err.log && echo workloads ok
python bench.py --gpus 1 --force-dist --steps 100 --warmup 20 --no-cpu-baseline > $O/r04_bench_c2_partitioned_1rank_$TAG.json 2>> $O/err.log
python bench.py --gpus 1 --force-dist --dist-exchange rccl --steps 100 --warmup 20 --no-cpu-baseline > $O/r04_bench_c2_partitioned_1rank_rccl_$TAG.json 2>> $O/err.log
TG_BENCH_REHEARSAL=1 python bench.py --gpus 2 --steps 20 --warmup 5 --preroll 40 --no-cpu-baseline > $O/r04_rehearsal_2ranks_one_gpu_windows_$TAG.json 2>> $O/err.log && echo dist ok
for r in none static seq; do python bench.py --train --train-restarter $r --no-cpu-baseline > $O/r04_train_c2_${r}_$TAG.json 2>> $O/err.log; done; echo train ok
fi
if [[ $PART == *b* ]]; then
cd /tmp && export TMPDIR=/tmp
P="--no-cpu-baseline --no-c5s-leg --no-dist-leg --no-api-loop --no-self-check --repeats 0"
rocprofv3 --kernel-trace --stats -d $O/prof_c2 -o c2 -- python $R/bench.py --steps 100 --warmup 20 $P > $O/prof_c2.log 2>&1
python $R/tools/rocpd_stats.py $(find $O/prof_c2 -name '*.db' | head -1) $O/r04_bench_c2_kernel_stats_$TAG.csv > /dev/null && echo stats c2 ok
rocprofv3 --kernel-trace --stats -d $O/prof_c5s -o c5s -- python $R/bench.py --workload c5s --steps 10 --warmup 4 --preroll 100 $P > $O/prof_c5s.log 2>&1
python $R/tools/rocpd_stats.py $(find $O/prof_c5s -name '*.db' | head -1) $O/r04_bench_c5s_kernel_stats_$TAG.csv > /dev/null && echo stats c5s ok
rocprofv3 --kernel-trace --stats -d $O/prof_part -o part -- python $R/bench.py --gpus 1 --force-dist --steps 100 --warmup 20 --no-cpu-baseline > $O/prof_part.log 2>&1
python $R/tools/rocpd_stats.py $(find $O/prof_part -name '*.db' | head -1) $O/r04_bench_c2_partitioned_1rank_kernel_stats_$TAG.csv > /dev/null && echo stats part ok
for W in c2 c5s; do
  EXTRA="--steps 10 --warmup 25 --preroll 100"; [ $W = c5s ] && EXTRA="--steps 4 --warmup 4 --preroll 40"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f_$W -o f -- python $R/bench.py --workload $W $EXTRA $P --no-graph > $O/pmc_f_$W.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w_$W -o w -- python $R/bench.py --workload $W $EXTRA $P --no-graph > $O/pmc_w_$W.log 2>&1
  F=$(find $O/pmc_f_$W -name '*counter_collection.csv' | head -1); Wf=$(find $O/pmc_w_$W -name '*counter_collection.csv' | head -1)
  python $R/tools/pmc_traffic.py $F $Wf $O/r04_hbm_traffic_${W}.json $O/r04_bench_${W}_kernel_stats_$TAG.csv | grep -i "gru\|attn_core\|gather\|gemm" || true
  cp $F $O/r04_pmc_FETCH_SIZE_${W}_$TAG.csv; cp $Wf $O/r04_pmc_WRITE_SIZE_${W}_$TAG.csv
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_m -o m -- python $R/bench.py --steps 10 --warmup 25 --preroll 100 $P --no-graph > $O/pmc_m.log 2>&1 || true
cp $(find $O/pmc_m -name '*counter_collection.csv' | head -1) $O/r04_pmc_mfma_c2_$TAG.csv 2>/dev/null || true
rm -rf $O/prof_c2 $O/prof_c5s $O/prof_part $O/pmc_f_* $O/pmc_w_* $O/pmc_m
fi
echo done
