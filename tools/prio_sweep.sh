#!/bin/bash
# kernel time of one env id under several issue-priority threshold sets (MOCCA_PARAM_ISSUE_PRIORITY), two interleaved rounds
# usage: tools/prio_sweep.sh <env-id> <envs> "<extra bench args>" t1,t2,t3 [t1,t2,t3 ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
envid=$1; envs=$2; extra=$3; shift; shift; shift
steps=300; [[ $envid == Cassie* ]] && steps=60
for round in 1 2; do
  for p in "$@"; do
    python bench.py --env-id $envid --envs $envs $extra --steps $steps --warmup 30 --prio $p --no-cpu-baseline 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$envid $extra [$p]', round(d['roofline']['kernel_ms']*1000,1), 'us')"
  done
done
