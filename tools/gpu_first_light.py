"""First-light check on the GPU box: reset + teacher-forced steps vs the f32 oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mocca_envs_amd import model as M
from mocca_envs_amd.vec_env import VecEnv, task_to_float64, task_from_float64
from oracle.oracle import Oracle, build
build()
task = int(sys.argv[1]) if len(sys.argv) > 1 else 0
env_id = "Walker3DCustomEnv-v0" if task == 0 else "Walker3DStepperEnv-v0"
N = 64
env = VecEnv(env_id, N, auto_reset=False, seed=5)
print("kernel info", env.kernel_info(), flush=True)
orc = Oracle(env.model.to_bytes(), task, N, "f32")
if task == 1:
    env.set_param(2, 5); orc.set_param(2, 5)
o_gpu = env.reset().cpu().numpy(); torch.cuda.synchronize()
o_cpu = orc.reset(seed=5)
print("reset obs max diff", np.abs(o_gpu - o_cpu).max(), flush=True)
st_g = env.get_state().cpu().numpy(); st_c = orc.get_state()
print("reset state max diff", np.abs(st_g - st_c).max())
tk_g = task_to_float64(env.get_task()); tk_c = orc.get_task()
print("reset task max diff", np.abs(tk_g - tk_c).max())
if task == 1:
    print("terrain diff", np.abs(env.get_terrain().cpu().numpy()[:, :124] - orc.get_terrain()).max())
rng = np.random.default_rng(0)
worst = {}
for t in range(60):
    # teacher forcing: GPU starts every step from the oracle's state
    env.set_state(orc.get_state().astype(np.float32))
    env.set_task(task_from_float64(orc.get_task()))
    if task == 1:
        ter = np.zeros((N, 128), np.float32); ter[:, :124] = orc.get_terrain(); env.set_terrain(ter)
    a = rng.uniform(-1, 1, (N, 21)).astype(np.float32)
    og, rg, dg, ig = env.step(torch.from_numpy(a).cuda())
    torch.cuda.synchronize()
    oc, rc, dc, ic = orc.step(a)
    sg = env.get_state().cpu().numpy(); sc = orc.get_state()
    d = dict(obs=np.nanmax(np.abs(og.cpu().numpy() - oc)), rew=np.nanmax(np.abs(rg.cpu().numpy() - rc)),
             state=np.nanmax(np.abs(sg[:, :55] - sc[:, :55])), warm=np.nanmax(np.abs(sg[:, 55:] - sc[:, 55:])),
             done=int((dg.cpu().numpy() != dc).sum()))
    for k, v in d.items(): worst[k] = max(worst.get(k, 0), v)
    if t < 5 or t % 10 == 0: print(t, d, flush=True)
    # keep the oracle's envs alive: reset the finished ones in both
    if dc.any():
        m = (dc != 0).astype(np.uint8)
        orc.reset(seed=5, mask=m); env.reset(torch.from_numpy(m).cuda())
print("WORST", worst)
# throughput smoke
env2 = VecEnv(env_id, 4096, auto_reset=True, seed=1)
env2.reset(); acts = torch.rand(4096, 21, device="cuda") * 2 - 1
for _ in range(5): env2.step(acts)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(20): env2.step(acts)
torch.cuda.synchronize(); dt = time.time() - t0
print("steps/s @4096:", 4096 * 20 / dt, "resets/step", float((env2.done != 0).float().mean()))
