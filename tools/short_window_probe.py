"""The 20-launch window of the driver's protocol (bench.py --steps 20 --warmup 5) runs 6 - 8 % slower per launch in the first GPU process of a
machine that has idled than in any later process, while 1000 launches back to back take the same 100 us in both (tools/first_process_probe.py).
Does GPU-busy time inside the process cure it?  windows -> N s of back-to-back launches -> windows -> 3 s of sleep -> windows."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mocca_envs_amd.vec_env import VecEnv
busy_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
env = VecEnv("Walker3DCustomEnv-v0", 4096, auto_reset=True, seed=1000)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
tape = torch.rand(64, 4096, 21, device="cuda", generator=g) * 2 - 1
torch.cuda.synchronize()
for i in range(1000): env.step(tape[(i + 17) % 64])

def windows(tag, k=3):
    for rep in range(k):
        n_done = torch.zeros((), device="cuda")
        for i in range(5):
            n_done += (env.step(tape[i % 64])[2] != 0).sum()
        n_done.item()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for i in range(20): env.step(tape[i % 64])
        e1.record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        print("%-28s window %d: wall/step %.1f us, events/launch %.1f us" % (tag, rep, wall * 1e6 / 20, e0.elapsed_time(e1) * 50), flush=True)

def busy(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): env.step(tape[i % 64])
    e1.record(); torch.cuda.synchronize()
    print("%d launches back to back: %.1f us per launch" % (n, e0.elapsed_time(e1) * 1e3 / n), flush=True)

windows("after the 1000-step preroll")
busy(busy_steps)
windows("after %.1f s more of launches" % (busy_steps * 1e-4))
time.sleep(3.0)
windows("after 3 s of sleep")
# how much GPU-busy time does the cure need?  (3 s of sleep re-creates the slow state every time)
for n in (1000, 2500, 5000, 10000, 20000, 40000):
    time.sleep(3.0)
    busy(n)
    windows("3 s sleep, then %.2f s busy" % (n * 1e-4), k=2)
