#!/bin/bash
# A/B of pace-priority build variants (band width, extra checkpoint) on three env ids; one box, one call.
set -o pipefail
OUT=${1:-gpurun_out/r04_pace_variants.jsonl}; : > "$OUT"
run() {
  local lib=$1; shift
  local line
  line=$(MOCCA_LIB_PATH=$lib python bench.py --no-cpu-baseline --steps 300 --warmup 50 "$@" 2>/dev/null | tail -1)
  python - "$(basename ${lib:-product}) $*" "$line" >> "$OUT" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print(json.dumps({"args": sys.argv[1], "kernel_ms": round(d["roofline"]["kernel_ms"], 5), "env_steps_per_s": round(d["value"])}))
PY
  tail -1 "$OUT"
}
for rep in 1 2; do
for lib in "" $PWD/.ab/lib_s3.so $PWD/.ab/lib_s5.so $PWD/.ab/lib_extra.so; do
  for k in -18 -19; do
    run "$lib" --pace $k
    run "$lib" --env-id Walker3DStepperEnv-v0 --curriculum 9 --pace $k
    run "$lib" --env-id Crab2DCustomEnv-v0 --pace $k
  done
done
done
