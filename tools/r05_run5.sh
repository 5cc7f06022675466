R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
python -m pytest tests/test_gpu_substep.py -q -s -k "kw38 or kw39 or kw40 or kw41 or kw42 or kw43" > $O/retest4.log 2>&1; echo rc=$? >> $O/retest4.log; tail -3 $O/retest4.log
python -m pytest tests/test_gpu_compact.py -q > $O/retest5.log 2>&1; echo rc=$? >> $O/retest5.log; tail -2 $O/retest5.log
for i in 1 2 3; do python bench.py --steps 1000 --warmup 200 --no-cpu-baseline --no-physics-bracket 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('walker', round(1e3*d['roofline']['kernel_ms'],2))"; done
python bench.py --env-id CassieEnv-v0 --envs 2048 --action-scale 0.1 --steps 100 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cassie', round(1e3*d['roofline']['kernel_ms'],1))"
