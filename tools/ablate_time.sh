#!/bin/bash
# Kernel time of profiling variants with one phase removed (results are wrong by construction; only the time matters),
# at 1 wave/SIMD (1024 envs: the single-wave critical path) and at the headline 4096 envs.  Run on the GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export MOCCA_ALLOW_DIAGNOSTIC_BUILD=1
for v in NONE SOLVE COLLIDE ABA "SOLVE -DMOCCA_SKIP_COLLIDE -DMOCCA_SKIP_ABA" $EXTRA; do
  flag=""; [ "$v" != NONE ] && flag="-DMOCCA_SKIP_$v"
  python -m mocca_envs_amd.build --out /tmp/libmocca_abl.so $flag > /dev/null || exit 1
  for n in 1024 4096; do
    MOCCA_LIB_PATH=/tmp/libmocca_abl.so python $R/bench.py --envs $n --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('skip=$v', $n, 'envs', round(d['roofline']['kernel_ms']*1000,1), 'us')"
  done
done
