#!/bin/bash
# Heaviest-first launch order (MOCCA_PARAM_ORDER_EVERY) against index order, 48-row and compact instance, one box, one call.
set -o pipefail
OUT=${1:-gpurun_out/r04_order_probe.jsonl}
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
for n in 4096 8192 16384; do
  for k in 0 4 16 64; do
    run "full48 order_every=$k" --envs $n --order-every $k
    run "compact order_every=$k" --envs $n --max-rows 32 --order-every $k
  done
done
run "stepper c9 8192 order 0" --env-id Walker3DStepperEnv-v0 --curriculum 9 --envs 8192 --order-every 0
run "stepper c9 8192 order 16" --env-id Walker3DStepperEnv-v0 --curriculum 9 --envs 8192 --order-every 16
