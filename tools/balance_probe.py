#!/usr/bin/env python3
"""Would a load-balancing block order shorten the launch?  (experiment, no kernel change)

tools/dispatch_probe.hip: with 4096 one-wave workgroups the hardware puts blocks b, b + 1024, b + 2048, b + 3072 on the same SIMD
(steady state).  A launch lasts as long as its slowest SIMD, and an env's cost follows its row count.  This script takes snapshots
of a steady-state batch, times ONE env.step from each snapshot (a) as it is and (b) with the envs PERMUTED in the state / task
buffers so that every SIMD's four envs have balanced row counts (heaviest with lightest: sorted rank g, 2047 - g, 2048 + g,
4095 - g share a SIMD), and prints both times.  The permutation moves whole envs, so the work is identical.
"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from mocca_envs_amd.vec_env import VecEnv  # noqa: E402


def time_step(env, act, st, tk, reps=5):
    ts = []
    for _ in range(reps):
        env.set_state(st); env.set_task(tk)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step(act); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))


def main():
    n = 4096
    env = VecEnv("Walker3DCustomEnv-v0", n, auto_reset=True, seed=1000)
    dbg = env.set_debug(True)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    tape = torch.rand(64, n, env.act_dim, device="cuda", generator=g) * 2 - 1
    for i in range(300):
        env.step(tape[i % 64])
    res = []
    for snap in range(12):
        for i in range(7):
            env.step(tape[(snap * 7 + i) % 64])
        st, tk = env.get_state().clone(), env.get_task().clone()
        rows = dbg[:, 0].cpu().numpy().astype(np.int64)          # rows of the last substep: the cost proxy
        act = tape[snap % 64]
        t_plain = time_step(env, act, st, tk)
        order = np.argsort(-rows, kind="stable")                  # heaviest first
        pos = np.empty(n, np.int64)                               # block position of sorted rank r
        gidx = np.arange(1024)
        pos[gidx] = gidx; pos[2047 - gidx] = gidx + 1024; pos[2048 + gidx] = gidx + 2048; pos[4095 - gidx] = gidx + 3072
        perm = np.empty(n, np.int64)                              # perm[block] = env that moves there
        perm[pos] = order
        p = torch.from_numpy(perm).cuda()
        t_bal = time_step(env, act[p], st[p], tk[p])
        worst = np.empty(n, np.int64)                             # adversarial: the four heaviest together, and so on
        worst[np.concatenate([gidx, gidx + 1024, gidx + 2048, gidx + 3072])] = order[np.concatenate([4 * gidx, 4 * gidx + 1, 4 * gidx + 2, 4 * gidx + 3])]
        w = torch.from_numpy(worst).cuda()
        t_worst = time_step(env, act[w], st[w], tk[w])
        rnd = torch.randperm(n, device="cuda")
        t_rnd = time_step(env, act[rnd], st[rnd], tk[rnd])
        res.append((t_plain, t_rnd, t_bal, t_worst))
        print(f"snapshot {snap}: as is {t_plain:.1f} us | random order {t_rnd:.1f} | balanced {t_bal:.1f} | heaviest together {t_worst:.1f} | rows mean {rows.mean():.1f} max {rows.max()}")
        env.set_state(st); env.set_task(tk)
    r = np.array(res)
    print("median: as is %.1f  random %.1f  balanced %.1f  heaviest-together %.1f us" % tuple(np.median(r, axis=0)))


if __name__ == "__main__":
    main()
