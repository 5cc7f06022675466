#!/bin/bash
# How much launch time does one more vector instruction per wave cost?  N x 4 extra FMAs per substep (16 N per env.step), as ONE dependent
# chain and as four independent chains, at 4096 envs (four waves per SIMD) and at 1024 (one wave per SIMD).  The slope calibrates what
# removing instructions from the step kernel could buy: result in profiles/r06_dummy_valu.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export MOCCA_ALLOW_DIAGNOSTIC_BUILD=1
for n in 0 125 250 500; do
  for ilp in 0 1; do
    [ $n = 0 ] && [ $ilp = 1 ] && continue
    flag=""; [ $n != 0 ] && flag="-DMOCCA_DUMMY_VALU=$n"; [ $ilp = 1 ] && flag="$flag -DMOCCA_DUMMY_ILP"
    python -m mocca_envs_amd.build --out /tmp/libdv_${n}_$ilp.so $flag > /dev/null || exit 1
  done
done
for envs in 4096 1024; do for r in 1 2; do for n in 0 125 250 500; do for ilp in 0 1; do
  [ $n = 0 ] && [ $ilp = 1 ] && continue
  MOCCA_LIB_PATH=/tmp/libdv_${n}_$ilp.so python bench.py --envs $envs --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('envs', $envs, 'extra FMAs per env.step', $n*16, 'independent chains' if $ilp else 'one dependent chain', round(d['roofline']['kernel_ms']*1000,1), 'us')"
done; done; done; done
