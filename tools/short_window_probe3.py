"""Is it the FIRST USE of torch's reduction kernels (lazy code-object load: the GPU idles while the host works) between the pre-roll and the
timed window that makes bench.py's window slow even after 2 s of pre-roll?  mode `late`: pre-roll, then the reductions for the first time in
the warm-up (bench.py as it was); mode `early`: the same reductions once BEFORE the pre-roll."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mocca_envs_amd.vec_env import VecEnv
mode = sys.argv[1]
env = VecEnv("Walker3DCustomEnv-v0", 4096, auto_reset=True, seed=1000)
env.reset()
g = torch.Generator(device="cuda"); g.manual_seed(1)
tape = torch.rand(64, 4096, 21, device="cuda", generator=g) * 2 - 1
torch.cuda.synchronize()
if mode == "early":
    n_done = torch.zeros((), device="cuda"); n_done += (env.done != 0).sum(); n_done.item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record(); e1.record(); torch.cuda.synchronize(); e0.elapsed_time(e1)
for i in range(1000): env.step(tape[(i + 17) % 64])
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 2.0:
    for i in range(256): env.step(tape[i % 64])
    n += 256
    torch.cuda.synchronize()
for rep in range(3):
    n_done = torch.zeros((), device="cuda")
    tg = time.perf_counter()
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
    print("%s: window %d: events/launch %.1f us (warm-up + gap before it took %.1f ms on the host)" % (mode, rep, e0.elapsed_time(e1) * 50, 1e3 * (t0 - tg)), flush=True)
