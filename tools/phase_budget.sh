#!/bin/bash
# Builds the library with the phase stamps compiled in (in the GPU box's scratch copy of the repo) and prints the s_memtime
# budget of every launch of the C2 step and of the one-launch attention kernel -> gpurun_out/r05_final/r05_phase_budget_c2.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_final; mkdir -p $O
cd $R
set -e
make -C www2023tiger_amd/csrc clean > /dev/null
make -C www2023tiger_amd/csrc -j16 EXTRA='-DTG_PHASE_TRACE -DTG_CORE_TRACE' > $O/phase_build.log 2>&1
echo built
: > $O/r05_phase_budget_c2.txt
for w in core fc1 fc2 updater qrows tile; do
  python tools/phase_budget.py $w >> $O/r05_phase_budget_c2.txt 2>> $O/phase_err.log
  echo >> $O/r05_phase_budget_c2.txt
  echo done $w
done
cat $O/r05_phase_budget_c2.txt
