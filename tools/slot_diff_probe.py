"""Where do the contact-slot masks of the HIP path and the f32 oracle differ?  (debug aid)  usage: python tools/slot_diff_probe.py [env-id]"""
import os, sys
import numpy as np
import torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from mocca_envs_amd.vec_env import VecEnv, task_from_float64, TASKS
from oracle.oracle import Oracle
from test_gpu_substep import _one_substep_blob
env_id = sys.argv[1] if len(sys.argv) > 1 else "CassieEnv-v0"
n = 64
m = _one_substep_blob(env_id)
env = VecEnv(env_id, n, auto_reset=False, seed=4, model_blob=m.to_bytes())
dbg = env.set_debug(True)
orc = Oracle(m.to_bytes(), TASKS[env_id], n, "f32")
env.reset(); orc.reset(seed=4)
rng = np.random.default_rng(2)
shown = 0
for t in range(200):
    env.set_state(orc.get_state().astype(np.float32)); env.set_task(task_from_float64(orc.get_task()))
    a = (0.3 * rng.uniform(-1, 1, (n, env.act_dim))).astype(np.float32)
    env.step(torch.from_numpy(a).cuda()); _, _, dc, _ = orc.step(a)
    dg, do = dbg.cpu().numpy(), orc.get_debug()
    diff = ~(dg[:, :8] == do[:, :8]).all(axis=1)
    for e in np.nonzero(diff)[0][:2]:
        if shown < 10:
            mg = (int(dg[e, 3]) & 0xFFFFFFFF) | ((int(dg[e, 4]) & 0xFFFFFFFF) << 32)
            mo = (int(do[e, 3]) & 0xFFFFFFFF) | ((int(do[e, 4]) & 0xFFFFFFFF) << 32)
            print(f"t{t} env{e}: gpu rows {dg[e,0]} lim {dg[e,1]} c {dg[e,2]} | oracle rows {do[e,0]} lim {do[e,1]} c {do[e,2]} | slots gpu {mg:024b} oracle {mo:024b}")
            shown += 1
print("done")
