#!/bin/bash
# A/B two builds of the HIP library in ONE gpurun call, interleaved rounds (methodology rule: same device, same process env).
# usage: tools/ab.sh [rounds] [bench args...]   -> A = .ab/libA.so (tools/ab_save.sh), B = the working tree's library
R=${GRAFT_REPO_ROOT:-/root/repo}
rounds=${1:-4}; shift
cd $R
[ -f $R/.ab/libA.so ] || { echo "no .ab/libA.so: run tools/ab_save.sh first"; exit 1; }
for i in $(seq $rounds); do
  for v in A B; do
    if [ $v = A ]; then export MOCCA_LIB_PATH=$R/.ab/libA.so; else unset MOCCA_LIB_PATH; fi
    python bench.py --steps 400 --warmup 100 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['roofline']['kernel_ms']*1000,1), 'us')"
  done
done
