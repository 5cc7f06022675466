"""Distribution of one-step (teacher-forced) errors: GPU vs f32 oracle vs f64 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mocca_envs_amd import model as M
from mocca_envs_amd.vec_env import VecEnv, task_from_float64
from oracle.oracle import Oracle, build
build()
N = 256
env = VecEnv("Walker3DCustomEnv-v0", N, auto_reset=False, seed=9)
o32 = Oracle(env.model.to_bytes(), 0, N, "f32"); o64 = Oracle(env.model.to_bytes(), 0, N, "f64")
env.reset(); o32.reset(seed=9); o64.reset(seed=9)
rng = np.random.default_rng(1)
E = {k: [] for k in ("g32", "g64", "c64")}
names = ["pos"]*3 + ["quat"]*4 + ["vel"]*3 + ["omg"]*3 + ["q"]*21 + ["qd"]*21
for t in range(100):
    o64.set_state(o32.get_state()); o64.set_task(o32.get_task())
    env.set_state(o32.get_state().astype(np.float32)); env.set_task(task_from_float64(o32.get_task()))
    a = rng.uniform(-1, 1, (N, 21)).astype(np.float32)
    env.step(torch.from_numpy(a).cuda()); _, _, dc, _ = o32.step(a); o64.step(a)
    sg, sc, s6 = env.get_state().cpu().numpy()[:, :55], o32.get_state()[:, :55], o64.get_state()[:, :55]
    sc_scale = 1e-3 + 1e-3 * np.abs(s6)
    E["g32"].append(np.abs(sg - sc) / sc_scale); E["g64"].append(np.abs(sg - s6) / sc_scale); E["c64"].append(np.abs(sc - s6) / sc_scale)
    if dc.any():
        o32.reset(seed=9, mask=(dc != 0).astype(np.uint8))
for k, v in E.items():
    v = np.nan_to_num(np.array(v).max(axis=2).ravel(), nan=0)  # worst component per (step, env), in units of (1e-3 + 1e-3|x|)
    print(k, "median %.3g p90 %.3g p99 %.3g p99.9 %.3g max %.3g  frac>1: %.4f  frac>10: %.4f" % (
        np.median(v), np.percentile(v, 90), np.percentile(v, 99), np.percentile(v, 99.9), v.max(), (v > 1).mean(), (v > 10).mean()))
