"""How long may the GPU idle between the warm-up and the timed 20-launch window before the window reads slow?  0.5 s of launches, a
synchronize, a sleep of g ms, then the window (HIP events around its 20 launches)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mocca_envs_amd.vec_env import VecEnv
env = VecEnv("Walker3DCustomEnv-v0", 4096, auto_reset=True, seed=1000)
env.reset()
tape = torch.rand(64, 4096, 21, device="cuda") * 2 - 1
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); e1.record(); torch.cuda.synchronize(); e0.elapsed_time(e1)
for i in range(1000): env.step(tape[i % 64])
torch.cuda.synchronize()
for g in (0, 0.5, 1, 2, 5, 10, 20, 40, 80, 160, 0):
    r = []
    for rep in range(4):
        for i in range(5000): env.step(tape[i % 64])
        torch.cuda.synchronize()
        if g: time.sleep(g * 1e-3)
        e0.record()
        for i in range(20): env.step(tape[i % 64])
        e1.record(); torch.cuda.synchronize()
        r.append(e0.elapsed_time(e1) * 50)
    print("idle gap %6.1f ms: events/launch of the 20-launch window: %s us" % (g, " ".join("%.1f" % x for x in r)), flush=True)
