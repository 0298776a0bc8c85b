set -e
cd $GRAFT_REPO_ROOT
for v in 0 1 2; do
  touch www2023tiger_amd/csrc/tg_gemm.hip
  make -C www2023tiger_amd/csrc -j16 EXTRA=-DTG_RB_VAR=$v > /dev/null 2>&1
  echo variant $v; python tools/micro/sgemm_ceiling.py | head -2
done
