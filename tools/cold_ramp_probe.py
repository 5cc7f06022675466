"""How long does a cold MI355X take to reach its steady clocks?  Run as the FIRST GPU process of a fresh box: steps the headline batch in
chunks of 250 launches, one HIP event pair per chunk, and prints the mean launch time of every chunk against the GPU-busy time so far."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mocca_envs_amd.vec_env import VecEnv
n = 4096
env = VecEnv("Walker3DCustomEnv-v0", n, auto_reset=True, seed=1000)
env.reset()
tape = torch.rand(64, n, env.act_dim, device="cuda") * 2 - 1
torch.cuda.synchronize()
busy = 0.0
out = []
for c in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(250):
        env.step(tape[(c * 250 + i) % 64])
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    busy += ms
    out.append((busy, 1e3 * ms / 250))
print("GPU-busy ms -> us per launch (250-launch chunks):")
print(" ".join(f"{b:.0f}:{u:.1f}" for b, u in out))
