#!/usr/bin/env python3
"""What the product's solver caps (48 rows / 12 contacts per env and substep; Bullet has none) change, measured against the 64-row / 20-contact
accuracy instance (mocca_r64.hip) on the same env id, seed and action stream.

  python tools/cap_effect.py [--env-id Walker3DStepperEnv-v0] [--curriculum 9] [--envs 4096] [--steps 1000] [--seeds 3]

Individual trajectories of a capped and an uncapped run part ways at the first dropped row (contact-rich random flailing is chaotic), so
the comparison is between DISTRIBUTIONS over envs x steps: episode length, reward per step, steps reached (info, Stepper), reset fraction,
the cap-pressure counters of the debug record -- each with the seed-to-seed spread of the capped run itself as the yardstick.
One JSON line per (caps, seed) and a summary line.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(env_id, n, steps, seed, curriculum, caps):
    import torch
    from mocca_envs_amd.vec_env import VecEnv
    env = VecEnv(env_id, n, auto_reset=True, seed=seed, **({"max_rows": caps[0], "max_contacts": caps[1]} if caps else {}))
    if curriculum is not None:
        env.set_param(2, curriculum)
    dbg = env.set_debug(True)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(1000 + seed)
    ep_len = torch.zeros(n, device="cuda")
    lens, rew_sum, n_done, info_done = [], 0.0, 0, []
    for t in range(steps):
        _, r, d, info = env.step(torch.rand(n, env.act_dim, device="cuda", generator=g) * 2 - 1)
        ep_len += 1
        fin = d != 0
        if t >= steps // 5:          # past the synchronised start
            rew_sum += float(r.sum())
            if fin.any():
                lens.append(ep_len[fin].clone())
                info_done.append(info[fin].clone().float())
            n_done += int(fin.sum())
        ep_len[fin] = 0
    d = dbg.cpu().numpy()
    lens = torch.cat(lens) if lens else torch.zeros(1, device="cuda")
    info_done = torch.cat(info_done) if info_done else torch.zeros(1, device="cuda")
    counted = n * (steps - steps // 5)
    out = dict(env_id=env_id, curriculum=curriculum, envs=n, steps=steps, seed=seed, max_rows=int(env.model.max_rows), max_contacts=int(env.model.max_contacts),
               episode_length_mean=float(lens.mean()), episode_length_median=float(lens.median()), episodes=int(lens.numel()),
               reward_per_step=rew_sum / counted, reset_fraction_per_step=n_done / counted, info_at_done_mean=float(info_done.mean()),
               substeps=int(d[:, 14].sum()), substeps_dropping_contacts=int(d[:, 12].sum()), substeps_dropping_rows=int(d[:, 13].sum()),
               envs_capped_at_least_once=int(((d[:, 12] + d[:, 13]) > 0).sum()), largest_row_count_wanted=int(d[:, 15].max()),
               lds_bytes=env.kernel_info()["lds_bytes"])
    env.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env-id", default="Walker3DStepperEnv-v0")
    ap.add_argument("--curriculum", type=int, default=9)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--seeds", type=int, default=3)
    args = ap.parse_args()
    cur = args.curriculum if "Stepper" in args.env_id else None
    rows = {"capped_48_12": [], "wide_64_20": []}
    for seed in range(args.seeds):
        for name, caps in (("capped_48_12", None), ("wide_64_20", (64, 20))):
            r = run(args.env_id, args.envs, args.steps, seed, cur, caps)
            rows[name].append(r)
            print(json.dumps(dict(r, caps=name)), flush=True)
    import numpy as np
    summ = {"summary": True, "env_id": args.env_id, "curriculum": cur}
    for key in ("episode_length_mean", "reward_per_step", "reset_fraction_per_step", "info_at_done_mean"):
        a, b = np.array([r[key] for r in rows["capped_48_12"]]), np.array([r[key] for r in rows["wide_64_20"]])
        summ[key] = {"capped": float(a.mean()), "wide": float(b.mean()), "difference": float(b.mean() - a.mean()),
                     "seed_spread_capped": float(a.max() - a.min()), "seed_spread_wide": float(b.max() - b.min())}
    for key in ("substeps_dropping_contacts", "substeps_dropping_rows", "envs_capped_at_least_once", "largest_row_count_wanted"):
        summ[key] = {"capped": [r[key] for r in rows["capped_48_12"]], "wide": [r[key] for r in rows["wide_64_20"]]}
    print(json.dumps(summ), flush=True)


if __name__ == "__main__":
    main()
