#!/bin/bash
# like flag_sweep.sh for another env id: tools/flag_sweep_env.sh <env-id> <envs> "<flags A>" "<flags B>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export MOCCA_ALLOW_DIAGNOSTIC_BUILD=1
envid=$1; envs=$2; shift; shift
i=0
for f in "$@"; do python -m mocca_envs_amd.build --out /tmp/libsweep_$i.so $f > /dev/null || exit 1; i=$((i+1)); done
for round in 1 2; do
  i=0
  for f in "$@"; do
    MOCCA_LIB_PATH=/tmp/libsweep_$i.so python bench.py --env-id $envid --envs $envs --steps 100 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$f]', round(d['roofline']['kernel_ms']*1000,1), 'us')"
    i=$((i+1))
  done
done
