#!/bin/bash
# usage: tools/pmc.sh <outdir-name> <counter> [<counter> ...]   (one rocprofv3 --pmc pass over a short bench run)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/$1; shift
cd /tmp && rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --preroll-seconds 0 > $out.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "mocca_step" not in k: continue
    print(k, {c: (sum(v) / len(v)) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
