#!/bin/bash
# Pace priorities (MOCCA_PARAM_PACE_TICKS) against the row-count priorities: sweep of the pace, one box, one call.
set -o pipefail
OUT=${1:-gpurun_out/r04_pace_probe.jsonl}
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
run() {
  local label=$1; shift
  local line
  line=$(python bench.py --no-cpu-baseline --steps 300 --warmup 50 "$@" 2>/dev/null | tail -1)
  python - "$label" "$line" >> "$OUT" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print(json.dumps({"label": sys.argv[1], "env_id": d["config"]["workload"].split(",")[0], "envs": d["config"]["envs_per_gpu"], "max_rows": d["config"]["max_rows"],
                  "kernel_ms": d["roofline"]["kernel_ms"], "ms_per_step": d["ms_per_step"], "env_steps_per_s": d["value"]}))
PY
  tail -1 "$OUT"
}
for n in ${ENVS:-4096 8192}; do
  for p in ${PACES:-0 140000 160000 175000 190000 205000 220000 240000 270000}; do
    run "pace=$p" --envs $n --pace $p ${EXTRA}
  done
done
