"""What a user of the reference's single-env API gets: steps/s of the gym classes (batch of one, numpy in / out, the reference's 4-tuple)
and of a bare VecEnv(n_envs=1) step + synchronize, beside the scalar CPU oracle on one host core (the stand-in for the reference's
single PyBullet client).  One env per launch is latency-bound: one wave on one SIMD of the 1024; the GPU path pays a launch, the
step's serial chain (~100 us) and the read-backs.  usage (GPU box): python tools/facade_latency.py [steps]"""
import json
import os
import sys
import time

import numpy as np
import torch

R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import mocca_envs_amd  # noqa: E402
from mocca_envs_amd.vec_env import VecEnv  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
out = []
for env_id in ("Walker3DCustomEnv-v0", "Walker3DStepperEnv-v0", "CassieEnv-v0"):
    env = mocca_envs_amd.make(env_id)
    env.seed(0)
    env.reset()
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, (steps, env.action_space.shape[0])) * (0.1 if "Cassie" in env_id else 1.0)
    n = steps if "Cassie" not in env_id else steps // 4
    for i in range(50):
        if env.step(acts[i])[2]:
            env.reset()
    t0 = time.perf_counter()
    resets = 0
    for i in range(n):
        if env.step(acts[i])[2]:
            env.reset(); resets += 1
    gym_s = (time.perf_counter() - t0) / n
    env.close()
    v = VecEnv(env_id, 1, auto_reset=True, seed=0)
    v.reset()
    a = torch.from_numpy(acts.astype(np.float32)).cuda()
    for i in range(50):
        v.step(a[i:i + 1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        v.step(a[i:i + 1])
        torch.cuda.synchronize()
    vec_s = (time.perf_counter() - t0) / n
    v.close()
    rec = {"env_id": env_id, "gym_class_us_per_step": round(1e6 * gym_s, 1), "gym_class_steps_per_s": round(1 / gym_s),
           "resets": resets, "vecenv1_step_sync_us": round(1e6 * vec_s, 1), "vecenv1_steps_per_s": round(1 / vec_s)}
    try:
        sys.path.insert(0, R)
        from oracle.oracle import Oracle  # the checker, timed as the CPU stand-in (like bench.py's cpu_baseline leg)
        from mocca_envs_amd.vec_env import compile_model_for, TASKS
        m = compile_model_for(env_id)
        o = Oracle(m.to_bytes(), TASKS[env_id], 1, "f64")
        o.reset(seed=0)
        k = max(50, n // 10)
        t0 = time.perf_counter()
        for i in range(k):
            _, _, d, _ = o.step(acts[i:i + 1].astype(np.float32))
            if d[0]:
                o.reset(seed=0)
        rec["cpu_oracle_f64_1core_steps_per_s"] = round(k / (time.perf_counter() - t0))
    except Exception as e:  # noqa
        rec["cpu_oracle_error"] = repr(e)
    print(json.dumps(rec), flush=True)
