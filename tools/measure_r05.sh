#!/bin/bash
# Round-5 measurement set (1x MI355X): bench lines, rocprofv3 kernel statistics, PMC traffic / MFMA-busy passes -> gpurun_out/r05_final/
# usage (on the GPU box): bash tools/measure_r05.sh [TAG] [PART]   (files are named r05_*_TAG; PART: a = lines, b = stream profiles, c = restarter profiles, d = other workloads, e = restart-mode evaluation run)
TAG=${1:-v1}; PART=${2:-abc}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_final
mkdir -p $O
cd $R
set -e
N="--no-cpu-baseline --no-c5s-leg --no-dist-leg --no-api-loop"
if [[ $PART == *a* ]]; then
python bench.py --steps 100 --warmup 20 > $O/r05_bench_c2_$TAG.json 2> $O/err.log && echo bench ok
python bench.py --steps 20 --warmup 5 $N > $O/r05_bench_c2_driver_steps_$TAG.json 2>> $O/err.log
python bench.py --workload c1 --steps 300 --warmup 50 --no-cpu-baseline > $O/r05_bench_c1_$TAG.json 2>> $O/err.log
python bench.py --workload c3 --steps 100 --warmup 20 --no-cpu-baseline > $O/r05_bench_c3_$TAG.json 2>> $O/err.log
python bench.py --workload c4 --steps 100 --warmup 20 --no-cpu-baseline > $O/r05_bench_c4_$TAG.json 2>> $O/err.log
python bench.py --workload c5s --steps 30 --warmup 4 --no-cpu-baseline > $O/r05_bench_c5s_$TAG.json 2>> $O/err.log && echo workloads ok
for r in none static seq; do python bench.py --train --train-restarter $r --no-cpu-baseline > $O/r05_train_c2_${r}_$TAG.json 2>> $O/err.log; done; echo train ok
fi
cd /tmp && export TMPDIR=/tmp
if [[ $PART == *b* ]]; then
P="--no-cpu-baseline --no-c5s-leg --no-dist-leg --no-api-loop --no-self-check --repeats 0"
rocprofv3 --kernel-trace --stats -d $O/prof_c2 -o c2 -- python $R/bench.py --steps 100 --warmup 20 $P > $O/prof_c2.log 2>&1
python $R/tools/rocpd_stats.py $(find $O/prof_c2 -name '*.db' | head -1) $O/r05_bench_c2_kernel_stats_$TAG.csv > /dev/null && echo stats c2 ok
rocprofv3 --kernel-trace --stats -d $O/prof_c5s -o c5s -- python $R/bench.py --workload c5s --steps 10 --warmup 4 $P > $O/prof_c5s.log 2>&1
python $R/tools/rocpd_stats.py $(find $O/prof_c5s -name '*.db' | head -1) $O/r05_bench_c5s_kernel_stats_$TAG.csv > /dev/null && echo stats c5s ok
for W in c2 c5s; do
  EXTRA="--steps 10 --warmup 25 --preroll 100"; [ $W = c5s ] && EXTRA="--steps 4 --warmup 4 --preroll 60"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f_$W -o f -- python $R/bench.py --workload $W $EXTRA $P --no-graph > $O/pmc_f_$W.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w_$W -o w -- python $R/bench.py --workload $W $EXTRA $P --no-graph > $O/pmc_w_$W.log 2>&1
  F=$(find $O/pmc_f_$W -name '*counter_collection.csv' | head -1); Wf=$(find $O/pmc_w_$W -name '*counter_collection.csv' | head -1)
  python $R/tools/pmc_traffic.py $F $Wf $O/r05_hbm_traffic_${W}.json $O/r05_bench_${W}_kernel_stats_$TAG.csv | grep -i "gru\|attn_core\|gather\|gemm\|sample" || true
  cp $F $O/r05_pmc_FETCH_SIZE_${W}_$TAG.csv; cp $Wf $O/r05_pmc_WRITE_SIZE_${W}_$TAG.csv
done
rm -rf $O/prof_c2 $O/prof_c5s $O/pmc_f_* $O/pmc_w_*
fi
if [[ $PART == *c* ]]; then
# the SeqRestarter's kernels (VERDICT r04 task 2a): kernel statistics and the matrix-pipe occupancy of the training iteration
rocprofv3 --kernel-trace --stats -d $O/prof_seq -o seq -- python $R/bench.py --train --train-restarter seq --steps 30 --warmup 10 --no-cpu-baseline > $O/prof_seq.log 2>&1
python $R/tools/rocpd_stats.py $(find $O/prof_seq -name '*.db' | head -1) $O/r05_train_c2_seq_kernel_stats_$TAG.csv > /dev/null && echo stats seq ok
python $R/tools/rocpd_timeline.py $(find $O/prof_seq -name '*.db' | head -1) > $O/r05_train_c2_seq_timeline_$TAG.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_m_seq -o m -- python $R/bench.py --train --train-restarter seq --steps 4 --warmup 6 --no-graph --no-cpu-baseline > $O/pmc_m_seq.log 2>&1 || true
cp $(find $O/pmc_m_seq -name '*counter_collection.csv' | head -1) $O/r05_pmc_mfma_train_seq_$TAG.csv 2>/dev/null || true
rm -rf $O/prof_seq $O/pmc_m_seq
fi
if [[ $PART == *d* ]]; then
# kernel statistics of the configurations that had none: C3 as written, C4, the evaluation harness in
# restart mode (seq restarter), and the 2-rank rehearsal with the plain hash owner table
P="--no-cpu-baseline --no-self-check --repeats 0"
for W in c3 c4; do
  rocprofv3 --kernel-trace --stats -d $O/prof_$W -o $W -- python $R/bench.py --workload $W --steps 50 --warmup 10 $P > $O/prof_$W.log 2>&1
  python $R/tools/rocpd_stats.py $(find $O/prof_$W -name '*.db' | head -1) $O/r05_bench_${W}_kernel_stats_$TAG.csv > /dev/null && echo stats $W ok
  rm -rf $O/prof_$W
done
TG_EVAL_RESTART_GRAPH=0 rocprofv3 --kernel-trace --stats -d $O/prof_ev -o ev -- python $R/tools/prof_restart_loop.py 200 > $O/prof_ev.log 2>&1
python $R/tools/rocpd_stats.py $(find $O/prof_ev -name '*.db' | head -1) $O/r05_eval_restart_seq_bs200_kernel_stats_$TAG.csv > /dev/null && echo stats eval ok
rm -rf $O/prof_ev
cd $R
TG_BENCH_REHEARSAL=1 python bench.py --gpus 2 --steps 20 --warmup 5 --preroll 40 --no-cpu-baseline > $O/r05_rehearsal_2ranks_one_gpu_windows_$TAG.json 2>> $O/err.log
TG_BENCH_REHEARSAL=1 python bench.py --gpus 2 --steps 20 --warmup 5 --preroll 40 --no-cpu-baseline --dist-owner hash > $O/r05_rehearsal_2ranks_one_gpu_hash_owner_$TAG.json 2>> $O/err.log && echo rehearsal ok
fi
if [[ $PART == *e* ]]; then
# restart-mode evaluation as one library call on two streams (tg_eval_restart_run): kernel statistics of the pass (the stream's
# last 200 batches of 200), per-batch time by group size, the training iteration with the lazy-restart loop in front of it
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof_ev2 -o ev -- python $R/tools/prof_restart_loop.py 200 > $O/prof_ev2.log 2>&1
python $R/tools/rocpd_stats.py $(find $O/prof_ev2 -name '*.db' | head -1) $O/r05_eval_restart_seq_bs200_kernel_stats_$TAG.csv > /dev/null && echo stats eval ok
rm -rf $O/prof_ev2
cd $R
{
for G in 1 2 4 8; do echo "tg_eval_restart_run, group $G:"; TG_EVAL_RESTART_GROUP=$G python tools/prof_restart_loop.py 500 2>&1 | grep "ms per batch" | tail -1; done
echo "host-sequenced pipeline, two streams:"; TG_EVAL_RESTART_RUN=0 python tools/prof_restart_loop.py 500 2>&1 | grep "ms per batch" | tail -1
echo "host-sequenced pipeline, one stream:"; TG_EVAL_RESTART_RUN=0 TG_EVAL_RESTART_OVERLAP=0 python tools/prof_restart_loop.py 500 2>&1 | grep "ms per batch" | tail -1
echo "no pipeline (count read back before anything else is enqueued):"; TG_EVAL_RESTART_PIPELINE=0 python tools/prof_restart_loop.py 500 2>&1 | grep "ms per batch" | tail -1
} > $O/r05_eval_restart_seq_bs200_forms_$TAG.txt
for r in seq static; do python bench.py --train --train-restarter $r --train-restart-prob 0.01 --no-cpu-baseline > $O/r05_train_c2_${r}_lazy_restart_$TAG.json 2>> $O/err.log; done
echo part e ok
fi
echo done
