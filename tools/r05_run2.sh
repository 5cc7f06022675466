set -x
python -m pytest tests/test_gpu_compact.py -x -q -s -k "wide or caps_above" > gpurun_out/wide1.log 2>&1; echo rc=$? >> gpurun_out/wide1.log
python -m pytest tests/test_gpu_substep.py -x -q -s -k "64" > gpurun_out/wide2.log 2>&1; echo rc=$? >> gpurun_out/wide2.log
python tools/cap_effect.py --seeds 2 > gpurun_out/r05_cap_effect.jsonl 2> gpurun_out/cap_effect.err
python tools/cap_effect.py --seeds 2 --env-id Walker3DCustomEnv-v0 >> gpurun_out/r05_cap_effect.jsonl 2>> gpurun_out/cap_effect.err
python bench.py --env-id Walker3DStepperEnv-v0 --curriculum 9 --max-rows 64 --steps 300 --warmup 50 --no-cpu-baseline > gpurun_out/r05_stepper_c9_wide_bench.json 2>> gpurun_out/err2.log
python bench.py --env-id Walker3DStepperEnv-v0 --curriculum 9 --steps 300 --warmup 50 --no-cpu-baseline --no-physics-bracket > gpurun_out/r05_stepper_c9_bench.json 2>> gpurun_out/err2.log
python bench.py --env-id CassieEnv-v0 --envs 2048 --action-scale 0.1 --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r05_cassie2048_bench.json 2>> gpurun_out/err2.log
python bench.py --env-id CassieEnv-v0 --envs 4096 --action-scale 0.1 --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r05_cassie4096_bench.json 2>> gpurun_out/err2.log
tail -3 gpurun_out/wide1.log gpurun_out/wide2.log
