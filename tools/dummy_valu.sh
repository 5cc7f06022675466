#!/bin/bash
# experiment: add N x 4 dependent FMAs per substep and watch the launch time
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export MOCCA_ALLOW_DIAGNOSTIC_BUILD=1
for n in 0 125 250 500; do
  flag=""; [ $n != 0 ] && flag="-DMOCCA_DUMMY_VALU=$n"
  python -m mocca_envs_amd.build --out /tmp/libdv_$n.so $flag > /dev/null || exit 1
done
cd $R
for r in 1 2; do for n in 0 125 250 500; do
  MOCCA_LIB_PATH=/tmp/libdv_$n.so python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dummy FMAs/substep', $n*4, round(d['roofline']['kernel_ms']*1000,1), 'us')"
done; done
