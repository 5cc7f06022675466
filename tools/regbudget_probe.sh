#!/bin/bash
# Timing of .ab/lib_dyn{4,5,6}.so -- the step kernel built with dynamic LDS (`extern __shared__`) and __launch_bounds__(64, W): the compiler then
# sizes its register budget for W waves per SIMD (96 VGPRs at 5, 80 at 6) while the run-time occupancy stays at 4.  The 8-line source change is
# described in profiles/r03_regbudget_probe.txt and was not kept (the dynamic form alone costs 28 %).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
 for v in base dyn4 dyn5 dyn6; do
  if [ $v = base ]; then unset MOCCA_LIB_PATH; else export MOCCA_LIB_PATH=$R/.ab/lib_$v.so; fi
  for envs in 4096 16384; do
   python bench.py --envs $envs --steps 300 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_info']; print('$v envs $envs:', round(d['roofline']['kernel_ms']*1000,1), 'us  vgprs', k['vgprs'], 'scratch', k['scratch_bytes'], 'blocks/CU', k['max_blocks_per_cu'])"
  done
 done
done
