#!/usr/bin/env python3
"""Per-launch duration of the first env.steps after a device synchronisation (why a 20-step bench reads ~9 % slower than a 500-step one)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from mocca_envs_amd.vec_env import VecEnv  # noqa: E402

n = 4096
env = VecEnv("Walker3DCustomEnv-v0", n, auto_reset=True, seed=1000)
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
tape = torch.rand(64, n, env.act_dim, device="cuda", generator=g) * 2 - 1
for i in range(300):
    env.step(tape[i % 64])
for label, idle in (("no idle", 0.0), ("10 ms idle", 0.01), ("no idle", 0.0), ("200 ms idle", 0.2)):
    torch.cuda.synchronize()
    time.sleep(idle)
    K = 48
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    host = []
    for i in range(K):
        h0 = time.perf_counter()
        env.step(tape[i % 64])
        ev[i + 1].record()
        host.append((time.perf_counter() - h0) * 1e6)
    torch.cuda.synchronize()
    us = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(K)]
    print(label, "| gpu us per step:", " ".join(f"{u:.0f}" for u in us))
    print(label, "| host us per step:", " ".join(f"{u:.0f}" for u in host))
