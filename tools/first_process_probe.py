"""Why is the FIRST bench.py process on a fresh box 5 - 9 % slower than the second (36.1 M against 39.6 M env-steps/s, same box, same command)?
Mimics bench.py's sequence (1000-step preroll, 5 warm-up steps with the reset-fraction reduction, synchronize, 20 timed launches) with one HIP
event per launch, three windows in a row.  Run as the first GPU process of a call."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mocca_envs_amd.vec_env import VecEnv
mode = sys.argv[1] if len(sys.argv) > 1 else "reduce"     # reduce | plain | item (only the D2H read) | sum (only the reduction kernels)
reduce_in_warmup = mode in ("reduce", "sum")
env = VecEnv("Walker3DCustomEnv-v0", 4096, auto_reset=True, seed=1000)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
tape = torch.rand(64, 4096, 21, device="cuda", generator=g) * 2 - 1
torch.cuda.synchronize()
for i in range(1000): env.step(tape[(i + 17) % 64])
for rep in range(3):
    n_done = torch.zeros((), device="cuda")
    for i in range(5):
        done = env.step(tape[i % 64])[2]
        if reduce_in_warmup:
            n_done += (done != 0).sum()
    if mode in ("reduce", "item"):
        frac = float(n_done.item()) / (5 * 4096)
    torch.cuda.synchronize(); torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(20):
        env.step(tape[i % 64]); evs[i + 1].record()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    d = [evs[i].elapsed_time(evs[i + 1]) * 1000 for i in range(20)]
    print("window %d (%s): wall/step %.1f us; enqueue of 20 took %.0f us; per-launch us:" % (rep, mode, wall * 1e6 / 20, t_enq * 1e6), " ".join("%.0f" % x for x in d), flush=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(1000): env.step(tape[i % 64])
e1.record(); torch.cuda.synchronize()
print("then 1000 launches back to back: %.1f us per launch" % (e0.elapsed_time(e1)), flush=True)
