#!/bin/bash
# Self-calibrating pace priorities against the tuned row-count priorities, per env id (one box, one call).
set -o pipefail
OUT=${1:-gpurun_out/r04_pace_envs.jsonl}
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
run() {
  local line
  line=$(python bench.py --no-cpu-baseline --steps ${STEPS:-200} --warmup 50 "$@" 2>/dev/null | tail -1)
  python - "$*" "$line" >> "$OUT" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print(json.dumps({"args": sys.argv[1], "kernel_ms": round(d["roofline"]["kernel_ms"], 5), "env_steps_per_s": round(d["value"])}))
PY
  tail -1 "$OUT"
}
for p in 0 -17 -18 -19; do
  run --env-id Walker3DStepperEnv-v0 --curriculum 0 --pace $p
  run --env-id Walker3DStepperEnv-v0 --curriculum 9 --pace $p
  run --env-id LaikagoCustomEnv-v0 --pace $p
  run --env-id LaikagoStepperEnv-v0 --pace $p
  run --env-id Walker2DCustomEnv-v0 --pace $p
  run --env-id Crab2DCustomEnv-v0 --pace $p
  run --env-id Child3DCustomEnv-v0 --pace $p
  run --env-id MikeStepperEnv-v0 --pace $p
  run --env-id Walker3DPlannerEnv-v0 --pace $p
  run --env-id MikePlannerEnv-v0 --pace $p
  STEPS=60 run --env-id CassieEnv-v0 --envs 2048 --action-scale 0.1 --pace $p
  STEPS=60 run --env-id CassieEnv-v0 --envs 2048 --pace $p
  STEPS=60 run --env-id Cassie2DEnv-v0 --envs 2048 --action-scale 0.1 --pace $p
done
