#!/bin/bash
# Compact (32-row) step-kernel instance against the 48-row instance, on one box in one call: Walker3DCustomEnv-v0 at 4096 / 8192 / 16384
# envs, the same blob (max_rows 32), product build (5 waves per SIMD) and the A/B builds with a 4- and a 6-wave register budget.
#   tools/r04_compact_probe.sh [out.jsonl]       (run on the GPU box from the repo root)
set -o pipefail
OUT=${1:-gpurun_out/r04_compact_probe.jsonl}
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
run() {   # label, lib ('' = product), extra bench args
  local label=$1 lib=$2; shift 2
  local line
  line=$(MOCCA_LIB_PATH=$lib python bench.py --no-cpu-baseline --steps 300 --warmup 50 "$@" 2>/dev/null | tail -1)
  python - "$label" "$line" >> "$OUT" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print(json.dumps({"label": sys.argv[1], "envs": d["config"]["envs_per_gpu"], "max_rows": d["config"]["max_rows"], "kernel_ms": d["roofline"]["kernel_ms"],
                  "ms_per_step": d["ms_per_step"], "env_steps_per_s": d["value"], "kernel_info": d["kernel_info"]}))
PY
  tail -1 "$OUT"
}
for n in 4096 8192 16384; do
  run "full48 blob48" "" --envs $n
  run "full48 blob32 (forced)" "" --envs $n --max-rows 32 --kernel-variant 1
  run "compact w5" "" --envs $n --max-rows 32
  for w in 4 6; do
    [ -f .ab/lib_r32w$w.so ] && run "compact w$w" "$PWD/.ab/lib_r32w$w.so" --envs $n --max-rows 32
  done
done
