#!/bin/bash
# Everything profiles/ holds for a round: usage  tools/round_artifacts.sh <tag> [1|2]   (two gpurun calls of < 20 minutes each: part 1 = the headline's
# profile + every bench line + the trainer loops, part 2 = timelines, micro-benchmarks and the other configs' profiles; no part = both)
tag=${1:-r02}
part=${2:-12}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
if [[ $part == *1* ]]; then
# 0. rocprofv3 on the headline config first: its PMC passes refresh profiles/traffic.json, which the bench lines below quote
tools/profile_round.sh ${tag} Walker3DCustomEnv-v0 4096 > /dev/null 2>&1
[ -s $O/${tag}_traffic.json ] && cp $O/${tag}_traffic.json profiles/traffic.json
# 1. the driver's own invocation (what BENCH_rNN.json records) + a long steady-state line (with the CPU baselines)
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${tag}_bench_driver_protocol.json 2> /dev/null
python bench.py --steps 1000 --warmup 200 > $O/${tag}_bench_full.json 2> $O/${tag}_bench_full.err
# 2. batch-size sweep of the headline env (1 / 2 / 4 waves per SIMD resident, then 2 and 4 rounds of waves)
for n in 512 1024 2048 4096 8192 16384; do
  python bench.py --envs $n --steps 400 --warmup 200 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'envs': $n, 'kernel_us': round(1000*d['roofline']['kernel_ms'],1), 'env_steps_per_s': round(d['value'])}))"
done > $O/${tag}_batch_sweep.jsonl
# 3. the other configs / env ids (untraced bench lines)
python bench.py --envs 8192 --steps 400 --warmup 200 --no-cpu-baseline --no-physics-bracket > $O/${tag}_custom8192_untraced_bench.json 2>/dev/null
python bench.py --env-id Walker3DStepperEnv-v0 --curriculum 0 --steps 400 --warmup 200 --no-cpu-baseline > $O/${tag}_stepper_c0_bench.json 2>/dev/null
python bench.py --env-id Walker3DStepperEnv-v0 --curriculum 9 --steps 400 --warmup 200 --no-cpu-baseline > $O/${tag}_stepper_c9_bench.json 2>/dev/null
python bench.py --env-id CassieEnv-v0 --envs 2048 --steps 100 --warmup 30 --no-cpu-baseline > $O/${tag}_cassie_bench.json 2>/dev/null
for e in Child3DCustomEnv-v0 MikeStepperEnv-v0 Walker2DCustomEnv-v0 Crab2DCustomEnv-v0 LaikagoCustomEnv-v0 LaikagoStepperEnv-v0 Walker3DPlannerEnv-v0 MikePlannerEnv-v0; do
  python bench.py --env-id $e --steps 300 --warmup 100 --no-cpu-baseline > $O/${tag}_$(echo $e | tr 'A-Z' 'a-z' | sed 's/env-v0//')_bench.json 2>/dev/null
done
python bench.py --env-id Cassie2DEnv-v0 --envs 2048 --steps 100 --warmup 30 --no-cpu-baseline > $O/${tag}_cassie2d_bench.json 2>/dev/null
python bench.py --env-id CassiePhaseMocca2DEnv-v0 --envs 2048 --steps 100 --warmup 30 --no-cpu-baseline > $O/${tag}_cassiephasemocca2d_bench.json 2>/dev/null
python bench.py --env-id CassiePhaseMirror2DEnv-v0 --envs 2048 --steps 100 --warmup 30 --no-cpu-baseline > $O/${tag}_cassiephasemirror2d_bench.json 2>/dev/null
# 3b. the trainer-facing surface (mocca_envs_amd.trainer_api): PPO collection loops at the metric's batch and at config 5's shard
python tools/trainer_loop_bench.py --envs 4096 > $O/${tag}_trainer_loop_4096.json 2>/dev/null
python tools/trainer_loop_bench.py --envs 8192 > $O/${tag}_trainer_loop_8192.json 2>/dev/null
python tools/trainer_loop_bench.py --envs 4096 --env-id Walker3DStepperEnv-v0 --skip-verbatim > $O/${tag}_trainer_loop_stepper4096.json 2>/dev/null
fi
if [[ $part == *2* ]]; then
# 4. per-phase timelines (diagnostic build)
python tools/stamps.py Walker3DCustomEnv-v0 4096 > $O/${tag}_stamps_custom4096.txt 2>&1
python tools/stamps.py Walker3DCustomEnv-v0 1024 > $O/${tag}_stamps_custom1024.txt 2>&1
python tools/param_time.py 4096 > $O/${tag}_param_time_4096.txt 2>&1
# 4b. cap pressure (how often the 12-contact / 48-row caps drop something)
python tools/cap_pressure.py 300 > $O/${tag}_cap_pressure.jsonl 2>/dev/null
# 5. record layouts
hipcc --offload-arch=gfx950 -O3 -o /tmp/layout_bench tools/layout_bench.hip && /tmp/layout_bench 4096 > $O/${tag}_layout_bench.json && /tmp/layout_bench 65536 >> $O/${tag}_layout_bench.json
# 5b. microbenchmarks behind DESIGN.md section 6: dependent-issue latencies, PGS visit forms, workgroup placement
hipcc --offload-arch=gfx950 -O3 -o /tmp/lat_bench tools/lat_bench.hip && /tmp/lat_bench > $O/${tag}_lat_bench.jsonl
hipcc --offload-arch=gfx950 -O3 -o /tmp/pgs_chain_bench tools/pgs_chain_bench.hip && /tmp/pgs_chain_bench > $O/${tag}_pgs_chain_bench.jsonl
hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_width_bench tools/lds_width_bench.hip && /tmp/lds_width_bench > $O/${tag}_lds_width_bench.jsonl
hipcc --offload-arch=gfx950 -O3 -o /tmp/dispatch_probe tools/dispatch_probe.hip && (cd $R && /tmp/dispatch_probe > $O/${tag}_dispatch_probe.txt)
# 6. rocprofv3: kernel-trace stats + separate PMC passes at steady state, the other two configs
tools/profile_round.sh ${tag}_stepper Walker3DStepperEnv-v0 4096 > /dev/null 2>&1
tools/profile_round.sh ${tag}_cassie CassieEnv-v0 2048 > /dev/null 2>&1
# 7. config 5's per-GPU shard: 48-row and compact instance (counters at 4 and 5 resident waves per SIMD), and the pipelined two-sub-batch lines
tools/profile_round.sh ${tag}_custom8192 Walker3DCustomEnv-v0 8192 > /dev/null 2>&1
BENCH_EXTRA="--max-rows 32" tools/profile_round.sh ${tag}_custom8192compact Walker3DCustomEnv-v0 8192 > /dev/null 2>&1
python bench.py --envs 8192 --stagger 2 --steps 400 --warmup 100 --no-cpu-baseline > $O/${tag}_custom8192_staggered_bench.json 2>/dev/null
python bench.py --envs 8192 --stagger 2 --max-rows 32 --steps 400 --warmup 100 --no-cpu-baseline > $O/${tag}_custom8192_staggered_compact_bench.json 2>/dev/null
fi
ls $O | grep "^${tag}" | head -120
