R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
python -m pytest tests/test_gpu_substep.py -q -k "kw3 or kw4 or kw0 or kw1" > $O/retest3.log 2>&1; echo rc=$? >> $O/retest3.log; tail -3 $O/retest3.log
tools/profile_round.sh r05 Walker3DCustomEnv-v0 4096 > /dev/null 2>&1
[ -s $O/r05_traffic.json ] && cp $O/r05_traffic.json profiles/traffic.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r05_bench_driver_protocol.json 2> /dev/null
python bench.py --steps 1000 --warmup 200 > $O/r05_bench_full.json 2> $O/r05_bench_full.err
python -c "
import json
for f in ('r05_bench_driver_protocol','r05_bench_full'):
    d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); print(f, d['value']/1e6, d['roofline']['kernel_ms'], d['roofline']['traffic'], d['kernel_info'])
"
