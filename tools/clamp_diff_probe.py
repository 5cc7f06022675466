"""Where do the solver's clamp masks of the HIP path and the f32 oracle differ?  (debug aid for tests/test_gpu_substep.py)
usage: python tools/clamp_diff_probe.py [env-id]"""
import os, sys
import numpy as np
import torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from mocca_envs_amd.vec_env import VecEnv, task_from_float64, TASKS
from oracle.oracle import Oracle
from test_gpu_substep import _one_substep_blob
env_id = sys.argv[1] if len(sys.argv) > 1 else "Walker2DCustomEnv-v0"
n = 256
m = _one_substep_blob(env_id)
env = VecEnv(env_id, n, auto_reset=False, seed=4, model_blob=m.to_bytes())
dbg = env.set_debug(True)
orc = Oracle(m.to_bytes(), TASKS[env_id], n, "f32")
env.reset(); orc.reset(seed=4)
rng = np.random.default_rng(2)
shown = 0
for t in range(60):
    env.set_state(orc.get_state().astype(np.float32)); env.set_task(task_from_float64(orc.get_task()))
    a = rng.uniform(-1, 1, (n, env.act_dim)).astype(np.float32)
    env.step(torch.from_numpy(a).cuda()); _, _, dc, _ = orc.step(a)
    dg, do = dbg.cpu().numpy(), orc.get_debug()
    rows_same = (dg[:, :8] == do[:, :8]).all(axis=1)
    diff = rows_same & ~(dg[:, 8:12] == do[:, 8:12]).all(axis=1)
    for e in np.nonzero(diff)[0][:3]:
        if shown < 12:
            mg = (int(dg[e, 8]) & 0xFFFFFFFF) | ((int(dg[e, 9]) & 0xFFFFFFFF) << 32)
            mo = (int(do[e, 8]) & 0xFFFFFFFF) | ((int(do[e, 9]) & 0xFFFFFFFF) << 32)
            print(f"t{t} env{e}: rows {dg[e,0]} limits {dg[e,1]} contacts {dg[e,2]} self {dg[e,7]} | last-iteration clamp mask gpu {mg:048b} oracle {mo:048b} xor {mg ^ mo:048b}")
            shown += 1
    if t % 8 == 7 and dc.any():
        orc.reset(seed=4, mask=(dc != 0).astype(np.uint8))
print("done")
