"""Per-launch HIP-event durations inside the driver's 20-step window (bench.py --steps 20 --warmup 5), after the same 1000-step preroll:
where the 20-step protocol loses its 3 - 9 % against a 1000-step window (the first launch after the mandatory synchronize).  Run on the GPU box."""
import torch, time, sys
sys.path.insert(0, "/root/repo")
from mocca_envs_amd.vec_env import VecEnv
env = VecEnv("Walker3DCustomEnv-v0", 4096, auto_reset=True, seed=1000)
env.reset()
tape = torch.rand(64, 4096, 21, device="cuda") * 2 - 1
for i in range(1000): env.step(tape[i % 64])
for rep in range(3):
    for i in range(5): env.step(tape[i % 64])
    torch.cuda.synchronize(); torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(20):
        env.step(tape[i % 64]); evs[i + 1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    d = [evs[i].elapsed_time(evs[i + 1]) * 1000 for i in range(20)]
    print("wall/step %.1f us; per-launch us:" % (wall * 1e6 / 20), " ".join("%.0f" % x for x in d))
