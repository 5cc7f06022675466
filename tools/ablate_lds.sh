#!/bin/bash
# Which phases cause the LDS bank conflicts?  Diagnostic builds with one phase removed, LDS counters of each (run on the GPU box).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export MOCCA_ALLOW_DIAGNOSTIC_BUILD=1
for v in NONE SOLVE COLLIDE ABA; do
  flag=""; [ $v != NONE ] && flag="-DMOCCA_SKIP_$v"
  python -m mocca_envs_amd.build --out /tmp/libmocca_$v.so $flag > /dev/null || exit 1
  echo "== variant skip=$v"
  MOCCA_LIB_PATH=/tmp/libmocca_$v.so $R/tools/pmc.sh abl_lds_$v SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
done
