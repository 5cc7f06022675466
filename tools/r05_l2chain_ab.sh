#!/bin/bash
# is the substep's chain waiting on per-lane model-blob loads (self-collision pair records, joint limits)?  A/B of builds of
# the SAME tree in one gpurun call, interleaved rounds.  Variants are built beforehand into .ab/ (they travel with the snapshot):
#   python -m mocca_envs_amd.build --out .ab/libPAIRABL.so -DMOCCA_ABL_PAIRLOAD    broad phase: pair records made up instead of loaded (no survivors)
#   python -m mocca_envs_amd.build --out .ab/libNOPASS2.so -DMOCCA_ABL_NOPASS2     broad phase as built, its survivors dropped (no narrow phase, no self contacts)
#   python -m mocca_envs_amd.build --out .ab/libNOHITS.so  -DMOCCA_ABL_NOHITS      narrow phase as built, its contacts dropped (no self-contact rows)
#   python -m mocca_envs_amd.build --out .ab/libBOTH.so    -DMOCCA_ABL_PAIRLOAD -DMOCCA_ABL_NOPASS2
# (ablations: results wrong by construction, lib.load() refuses them without MOCCA_ALLOW_DIAGNOSTIC_BUILD)
# usage: VARIANTS="MAIN NOPASS2 BOTH" tools/r05_l2chain_ab.sh [rounds] [bench args...]      results: profiles/r05_l2chain_ab.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
rounds=${1:-4}; shift
export MOCCA_ALLOW_DIAGNOSTIC_BUILD=1
for i in $(seq $rounds); do
  for v in ${VARIANTS:-MAIN PAIRABL NOPASS2 NOHITS BOTH}; do
    if [ $v = MAIN ]; then unset MOCCA_LIB_PATH; else export MOCCA_LIB_PATH=$R/.ab/lib$v.so; fi
    python bench.py --steps 1000 --warmup 200 --no-cpu-baseline --no-physics-bracket "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['roofline']['kernel_ms']*1000,2), 'us', d['kernel_info'])"
  done
done
