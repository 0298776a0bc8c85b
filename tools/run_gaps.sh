cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_g -o g -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-c5s-leg --no-dist-leg --no-self-check > $R/gpurun_out/g.log 2>&1
python $R/tools/rocpd_gaps.py $(find $R/gpurun_out/prof_g -name '*.db' | head -1)
rm -rf $R/gpurun_out/prof_g
