#!/bin/bash
# Build profiling variants of the HIP library with one phase removed and count VALU instructions of each
# (results of these builds are wrong by construction; only the counters matter).  Run on the GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export MOCCA_ALLOW_DIAGNOSTIC_BUILD=1
for v in NONE SOLVE COLLIDE ABA; do
  flag=""; [ $v != NONE ] && flag="-DMOCCA_SKIP_$v"
  python -m mocca_envs_amd.build --out /tmp/libmocca_$v.so $flag > /dev/null || exit 1
  echo "== variant skip=$v"
  MOCCA_LIB_PATH=/tmp/libmocca_$v.so $R/tools/pmc.sh abl_$v SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAVES
done
