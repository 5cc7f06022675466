"""Diagnostic: one teacher-forced env.step of the Cassie phase envs from every recorded state of tests/golden/cassie_mocap_reference.npz,
on the HIP library and on the f32 / f64 oracle -- which of the three pairwise differences is large, and in which observation entry.
(The f64 oracle IS the recording's physics; f32 rounding can flip a clamp in one of the 50 substeps, on either implementation.)
usage (GPU box): python tools/mocap_step_probe.py"""
import os
import sys

import numpy as np
import torch

R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import test_golden_cassie_mocap as T  # noqa: E402
from mocca_envs_amd.vec_env import VecEnv, task_to_float64, task_from_float64  # noqa: E402

G = T.G
IDS = {"mocca": "CassiePhaseMocca2DEnv-v0", "mirror": "CassiePhaseMirror2DEnv-v0"}
for tag in ("mocca", "mirror"):
    env = VecEnv(IDS[tag], 1, auto_reset=False, seed=0)
    _, o32 = T._oracle(tag, "f32")
    env.reset()
    o32.reset(seed=0)
    rows = []
    for ep in range(3):
        gobs = G[f"{tag}_ep{ep}_obs"]
        for t, a in enumerate(G[f"{tag}_ep{ep}_actions"]):
            st = G[f"{tag}_ep{ep}_state"][t]
            s = np.zeros((1, env.state_dim), np.float32); s[0, :len(st)] = st
            env.set_state(torch.from_numpy(s))
            tk = task_to_float64(env.get_task()); tk[:, 24:38] = G[f"{tag}_ep{ep}_jvel"][t]; tk[:, 39] = G[f"{tag}_ep{ep}_istep"][t]; tk[:, 7] = 0
            env.set_task(task_from_float64(tk))
            og = env.step(torch.from_numpy(a[None].astype(np.float32)).cuda())[0].cpu().numpy()[0]
            so = o32.get_state(); so[:] = 0; so[0, :len(st)] = st; o32.set_state(so)
            to = o32.get_task(); to[0, 24:38] = G[f"{tag}_ep{ep}_jvel"][t]; to[0, 39] = G[f"{tag}_ep{ep}_istep"][t]; to[0, 7] = 0; o32.set_task(to)
            oo = o32.step(a[None].astype(np.float32))[0][0]
            w = gobs[t + 1]
            e1, e2, e3 = np.abs(og - w), np.abs(oo - w), np.abs(og - oo)
            rows.append((ep, t, int(e1.argmax()), float(e1.max()), int(e2.argmax()), float(e2.max()), float(e3.max())))
    rows.sort(key=lambda r: -r[3])
    print(tag, "worst five by |HIP - recording|: (ep, t, entry, |HIP-rec|, entry, |f32 oracle-rec|, |HIP-f32 oracle|)")
    for r in rows[:5]:
        print("  ", r)
    e = np.array([r[3] for r in rows]); f = np.array([r[5] for r in rows]); h = np.array([r[6] for r in rows])
    print(f"   median |HIP-rec| {np.median(e):.2e}  |f32-rec| {np.median(f):.2e}  |HIP-f32| {np.median(h):.2e}; steps {len(rows)}")
    env.close()
