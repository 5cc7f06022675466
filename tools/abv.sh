#!/bin/bash
# builds every kernel-source variant under .abv/<name>/ and runs the substep parity test + a short bench on each (bisecting a change)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for d in .abv/*/; do
  v=$(basename $d)
  python -m mocca_envs_amd.build --src $R/.abv/$v --out /tmp/lib_$v.so > /dev/null 2>&1 || { echo "$v: build failed"; continue; }
  export MOCCA_LIB_PATH=/tmp/lib_$v.so
  t=$(python bench.py --steps 300 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms']*1000,1))")
  r=$(timeout -k 10 300 python -m pytest "tests/test_gpu_substep.py::test_single_substep_parity_with_matching_active_sets" -x -q 2>&1 | tail -1)
  echo "$v: $t us | $r"
done
