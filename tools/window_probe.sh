#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for s in 20 50 100 400; do
 for rep in 1 2; do
  python bench.py --steps $s --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps $s warmup 5:', round(d['roofline']['kernel_ms']*1000,1), 'us kernel,', round(d['ms_per_step']*1000,1), 'us/step')"
 done
done
python bench.py --steps 20 --warmup 200 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps 20 warmup 200:', round(d['roofline']['kernel_ms']*1000,1), 'us kernel,', round(d['ms_per_step']*1000,1), 'us/step')"
python tools/ramp_probe.py 2>/dev/null | cut -c1-230
