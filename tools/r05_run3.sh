R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
tools/profile_round.sh r05 Walker3DCustomEnv-v0 4096 > /dev/null 2>&1
tools/profile_round.sh r05_stepper Walker3DStepperEnv-v0 4096 > /dev/null 2>&1
tools/profile_round.sh r05_cassie CassieEnv-v0 2048 > /dev/null 2>&1
tools/profile_round.sh r05_custom8192 Walker3DCustomEnv-v0 8192 > /dev/null 2>&1
BENCH_EXTRA="--max-rows 32" tools/profile_round.sh r05_custom8192compact Walker3DCustomEnv-v0 8192 > /dev/null 2>&1
for i in 1 2 3; do python bench.py --envs 8192 --steps 400 --warmup 200 --no-cpu-baseline --no-physics-bracket 2>/dev/null | tail -1 > $O/r05_custom8192_untraced_bench.json; python -c "import json; d=json.load(open('$O/r05_custom8192_untraced_bench.json')); print('8192 untraced', d['value']/1e6, d['roofline']['kernel_ms'])"; done
ls $O | grep "^r05" | wc -l
