#!/bin/bash
# VERDICT r4 item 7, the register lever: CassieEnv-v0 at 2048 envs (two resident waves per SIMD) with the step kernel compiled for a
# two-waves-per-SIMD register budget (256 VGPRs), without (W2) and with (W2WIN) the solver's register window widened to every row
# (32 rows + 8 contacts' friction rows in registers: no Delassus reads from LDS inside the five sweeps).  Builds (travel with the snapshot):
#   python -m mocca_envs_amd.build --out .ab/libW2.so -DMOCCA_WAVES_PER_EU=2
#   python -m mocca_envs_amd.build --out .ab/libW2WIN.so -DMOCCA_WAVES_PER_EU=2 -DMOCCA_PGS_REG_ROWS=32 -DMOCCA_PGS_REG_CONTACTS=8
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for i in 1 2 3; do
  for v in MAIN W2 W2WIN; do
    if [ $v = MAIN ]; then unset MOCCA_LIB_PATH; else export MOCCA_LIB_PATH=$R/.ab/lib$v.so; fi
    python bench.py --env-id CassieEnv-v0 --envs 2048 --action-scale 0.1 --steps 100 --warmup 30 --no-cpu-baseline --no-physics-bracket 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(1e3*d['roofline']['kernel_ms'],1), 'us', d['kernel_info'])"
  done
done
