"""How often do the 12-contact / 48-row caps of the solver -- which Bullet does not have -- drop something at steady state?
  Runs each env id with the debug record attached (MOCCA_DBG_CAP_*: cumulative per (env, substep)) after an
untimed pre-roll, and prints one JSON line per env id: fraction of substeps in which contacts / rows were dropped, fraction of envs
that ever hit a cap, the largest row count an uncapped solver would have held, and the row-count distribution of the last substeps.
usage: python tools/cap_pressure.py [steps [env_id]] > profiles/archive/r03_cap_pressure.jsonl"""
import json
import os
import sys

import numpy as np
import torch

R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
from mocca_envs_amd.vec_env import VecEnv  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
CASES = [("Walker3DCustomEnv-v0", 4096, None, 1.0), ("Walker3DStepperEnv-v0", 4096, 0, 1.0), ("Walker3DStepperEnv-v0", 4096, 9, 1.0),
         ("LaikagoCustomEnv-v0", 4096, None, 1.0), ("CassieEnv-v0", 2048, None, 0.1), ("CassieEnv-v0", 2048, None, 1.0),
         ("Cassie2DEnv-v0", 2048, None, 0.1)]
only = sys.argv[2] if len(sys.argv) > 2 else None     # optional: one env id
for env_id, n, cur, scale in CASES:
    if only and env_id != only:
        continue
    env = VecEnv(env_id, n, auto_reset=True, seed=1000)
    if cur is not None:
        env.set_param(2, cur)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    tape = scale * (torch.rand(64, n, env.act_dim, device="cuda", generator=g) * 2 - 1)
    pre = 1000 if "Cassie" not in env_id else 100
    for i in range(pre):
        env.step(tape[i % 64])
    dbg = env.set_debug(True)
    rows_hist = np.zeros(64, np.int64)
    for i in range(steps if "Cassie" not in env_id else max(20, steps // 10)):
        env.step(tape[i % 64])
        rows_hist += np.bincount(dbg[:, 0].cpu().numpy().clip(0, 63), minlength=64)
    d = dbg.cpu().numpy().astype(np.int64)
    sub = d[:, 14].sum()
    cdf = np.cumsum(rows_hist) / rows_hist.sum()
    out = {"env_id": env_id, "envs": n, "curriculum": cur, "action_scale": scale, "substeps": int(sub),
           "frac_substeps_contacts_dropped": float(d[:, 12].sum() / sub), "frac_substeps_rows_dropped": float(d[:, 13].sum() / sub),
           "frac_envs_ever_capped": float(((d[:, 12] + d[:, 13]) > 0).mean()), "max_rows_wanted": int(d[:, 15].max()),
           "rows_last_substep": {"mean": float((np.arange(64) * rows_hist).sum() / rows_hist.sum()),
                                 "p50": int(np.searchsorted(cdf, 0.5)), "p90": int(np.searchsorted(cdf, 0.9)),
                                 "p99": int(np.searchsorted(cdf, 0.99)), "max": int(np.nonzero(rows_hist)[0].max()),
                                 "frac_gt_32": float(rows_hist[33:].sum() / rows_hist.sum()), "frac_gt_24": float(rows_hist[25:].sum() / rows_hist.sum())},
           "max_contacts": int(env.model.max_contacts), "max_rows": int(env.model.max_rows)}
    print(json.dumps(out), flush=True)
    env.close()
