#!/bin/bash
# A/B two builds of the HIP library in ONE gpurun call, interleaved rounds (methodology rule: same device, same process env).
# usage: tools/ab.sh <git-rev-A> [rounds]   -> A = that revision's csrc, B = working tree
R=${GRAFT_REPO_ROOT:-/root/repo}
rev=$1; rounds=${2:-4}
mkdir -p /tmp/abA && cd /tmp/abA
for f in mocca_api.hip mocca_device.h topo_walker3d.h topo_cassie.h topo_walker2d.h topo_crab2d.h topo_laikago.h; do cp $R/.ab_src/$f . ; done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -shared -fPIC -I$R/include -I. -o /tmp/libA.so mocca_api.hip || exit 1
cd $R
for i in $(seq $rounds); do
  for v in A B; do
    if [ $v = A ]; then export MOCCA_LIB_PATH=/tmp/libA.so; else unset MOCCA_LIB_PATH; fi
    python bench.py --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['roofline']['kernel_ms']*1000,1), 'us')"
  done
done
