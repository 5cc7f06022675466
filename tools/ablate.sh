#!/bin/bash
# Build profiling variants of the HIP library with one phase removed and count VALU instructions of each
# (results of these builds are wrong by construction; only the counters matter).  Run on the GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/mocca_envs_amd/csrc
for v in NONE SOLVE COLLIDE ABA; do
  flag=""; [ $v != NONE ] && flag="-DMOCCA_SKIP_$v"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -shared -fPIC -I../../include $flag -o /tmp/libmocca_$v.so mocca_api.hip || exit 1
  echo "== variant skip=$v"
  MOCCA_LIB_PATH=/tmp/libmocca_$v.so $R/tools/pmc.sh abl_$v SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAVES
done
