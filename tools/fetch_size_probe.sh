#!/bin/bash
# builds and runs tools/fetch_size_probe.hip under two rocprofv3 counter passes; writes gpurun_out/r05_fetch_size_probe.txt
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_size_probe tools/fetch_size_probe.hip
out=gpurun_out/r05_fetch_size_probe.txt
/tmp/fetch_size_probe > $out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/fsp_$c
  (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/fsp_$c -- /tmp/fetch_size_probe > /tmp/fsp_$c.log 2>&1)
  python3 - "$c" /tmp/fsp_$c >> $out <<'PY'
import csv, glob, sys
c, d = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows:
    if r["Counter_Name"] == c and "memset" not in r["Kernel_Name"].lower() and "fill" not in r["Kernel_Name"].lower():
        print(c, r["Kernel_Name"].split("(")[0], float(r["Counter_Value"]))
PY
done
cat $out
