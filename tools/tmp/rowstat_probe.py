import sys, ctypes, torch
sys.path.insert(0, "/root/repo")
from mocca_envs_amd.vec_env import VecEnv
env = VecEnv("Walker3DCustomEnv-v0", 4096, auto_reset=True, seed=1000)
dbg = env.set_debug(True)
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
tape = torch.rand(64, 4096, env.act_dim, device="cuda", generator=g) * 2 - 1
for i in range(300):
    env.step(tape[i % 64])
torch.cuda.synchronize()
rows = dbg[:, 0].float()
print("rows of the last substep: mean", float(rows.mean()), "p50", float(rows.median()), "p90", float(rows.quantile(0.9)), "max", float(rows.max()))
