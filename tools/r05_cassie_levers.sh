#!/bin/bash
# config 4 (CassieEnv-v0, 2048 envs = 2 resident waves per SIMD) with the cheap levers for an under-filled chip:
# pace priorities off / row-count priorities off (prio thresholds out of reach), and the batch sizes around it.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python bench.py --env-id CassieEnv-v0 --action-scale 0.1 --steps 100 --warmup 30 --no-cpu-baseline --no-physics-bracket "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'args': '$*', 'kernel_us': round(1000*d['roofline']['kernel_ms'],1), 'env_steps_per_s': round(d['value'])}))"; }
for rep in 1 2; do
  run --envs 2048
  run --envs 2048 --pace 0
  run --envs 2048 --pace 0 --prio 63,63,63
  run --envs 2048 --pace -16
  run --envs 2048 --pace -20
done
run --envs 1024
run --envs 3072
run --envs 4096
run --envs 8192
