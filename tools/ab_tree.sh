#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for i in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then d=$R/.ab/r02; else d=$R; fi
    (cd $d && python bench.py --steps 400 --warmup 100 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['roofline']['kernel_ms']*1000,1), 'us')")
  done
done
