set -x
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_protocol.json 2> gpurun_out/r05_bench_driver_protocol.err
python bench.py --envs 8192 --stagger 2 --max-rows 32 --steps 400 --warmup 50 --no-cpu-baseline > gpurun_out/r05_custom8192_staggered_api_bench.json 2>> gpurun_out/err.log
python bench.py --envs 8192 --steps 400 --warmup 50 --no-cpu-baseline --no-physics-bracket > gpurun_out/r05_custom8192_bench.json 2>> gpurun_out/err.log
python bench.py --env-id Walker3DStepperEnv-v0 --curriculum 9 --envs 8192 --steps 400 --warmup 50 --no-cpu-baseline --no-physics-bracket > gpurun_out/r05_stepper_c9_8192_bench.json 2>> gpurun_out/err.log
python bench.py --env-id Walker3DStepperEnv-v0 --curriculum 9 --envs 8192 --stagger 2 --max-rows 32 --steps 400 --warmup 50 --no-cpu-baseline > gpurun_out/r05_stepper_c9_8192_staggered_compact_bench.json 2>> gpurun_out/err.log
python tools/subbatch_loop_bench.py > gpurun_out/r05_subbatch_loop_8192.json 2>> gpurun_out/err.log
python tools/subbatch_loop_bench.py --envs 4096 --max-rows 0 > gpurun_out/r05_subbatch_loop_4096.json 2>> gpurun_out/err.log
echo done
