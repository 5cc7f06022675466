#!/bin/bash
# A/B of the product library against the builds under .ab/lib_*.so (other compiler flags, MOCCA_HIPCC_FLAGS), interleaved, one call
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
  for lib in base $(ls .ab/lib_*.so 2>/dev/null); do
    if [ $lib = base ]; then unset MOCCA_LIB_PATH; else export MOCCA_LIB_PATH=$R/$lib; fi
    python bench.py --steps 400 --warmup 100 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['roofline']['kernel_ms']*1000,1), 'us')"
  done
done
