#!/usr/bin/env python3
"""The double-buffered trainer loop of INTEGRATION.md, timed through the product API (mocca_envs_amd.multi.SubBatchedVecEnv), against
the synchronous loop of one VecEnv -- with a policy in the loop (a small MLP on torch's current stream: obs -> 64 -> act), so the
actions of step t + 1 really depend on the observations of step t and every ordering the API promises is exercised.

  python tools/subbatch_loop_bench.py [--envs 8192] [--sub-batches 2] [--max-rows 32] [--steps 400] [--env-id ...] [--hidden 64]

Prints one JSON line per protocol: tape (pre-computed actions, step_async(ordered=False): what bench.py --stagger times),
policy_sync (one handle: policy -> step), policy_double_buffered (wait(i) -> policy -> step_async(i)), trainer_loop_double_buffered (the same
with the in-kernel masks and the rollout write of tools/trainer_loop_bench.py: the pipelined counterpart of its device-totals loop)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=8192)
    ap.add_argument("--sub-batches", type=int, default=2)
    ap.add_argument("--max-rows", type=int, default=32)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--preroll", type=int, default=1000)
    ap.add_argument("--env-id", default="Walker3DCustomEnv-v0")
    ap.add_argument("--hidden", type=int, default=64)
    args = ap.parse_args()
    import torch
    from mocca_envs_amd.multi import SubBatchedVecEnv
    from mocca_envs_amd.vec_env import VecEnv
    dev = torch.device("cuda", 0)
    mr = args.max_rows if args.max_rows > 0 else None
    one = VecEnv(args.env_id, args.envs, auto_reset=True, seed=1000, max_rows=mr)
    sub = SubBatchedVecEnv(args.env_id, args.envs, sub_batches=args.sub_batches, auto_reset=True, seed=1000, max_rows=mr)
    one.reset(); sub.reset()
    g = torch.Generator(device=dev).manual_seed(1)
    w1 = torch.randn(one.obs_dim, args.hidden, device=dev, generator=g) * 0.3
    w2 = torch.randn(args.hidden, one.act_dim, device=dev, generator=g) * 0.3
    policy = lambda obs: torch.tanh(torch.tanh(obs @ w1) @ w2)
    tape = torch.rand(64, args.envs, one.act_dim, device=dev, generator=g) * 2 - 1
    sub_tapes = [tape[:, sl].contiguous() for sl in sub.slices]
    k = sub.n_parts

    def timed(fn, steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(steps):
            fn(t)
        sub.synchronize(); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    def sync_tape(t):
        one.step(tape[t % 64])

    def sub_tape(t):
        for i in range(k):
            sub.step_async(i, sub_tapes[i][t % 64], ordered=False)

    def sync_policy(t):
        one.step(policy(one.obs))

    def sub_policy(t):
        for i in range(k):
            obs, _, _, _ = sub.wait(i)
            sub.step_async(i, policy(obs))

    # the PPO collection loop of tools/trainer_loop_bench.py (masked observation into a rollout buffer every step; Monitor / TimeLimitMask inside
    # the launch, ABI 7), double-buffered: while sub-batch i steps, the policy and the rollout write of the other one run
    ep = sub.episode_stats(True)
    masks, bad = ep["masks"].unsqueeze(1), ep["bad_masks"].unsqueeze(1)
    rollouts = torch.zeros(64, args.envs, one.obs_dim, device=dev)

    def sub_trainer(t):
        for i in range(k):
            sl = sub.slices[i]
            obs, _, _, _ = sub.wait(i)
            act = policy(obs)
            rollouts[t % 64, sl].copy_(obs * masks[sl] * bad[sl].clamp(min=1.0))
            sub.step_async(i, act)          # ordered after the policy AND the rollout write (both read rows the launch rewrites)

    results = {}
    for name, fn in (("tape_one_handle", sync_tape), ("tape_sub_batches", sub_tape), ("policy_one_handle", sync_policy),
                     ("policy_double_buffered", sub_policy), ("trainer_loop_double_buffered", sub_trainer)):
        timed(fn, args.preroll)          # age the batch, warm the clocks (and torch's GEMM kernels)
        s = min(timed(fn, args.steps) for _ in range(3))
        results[name] = {"ms_per_step": 1e3 * s, "env_steps_per_s": args.envs / s}
    print(json.dumps({"env_id": args.env_id, "envs": args.envs, "sub_batches": k, "max_rows": mr, "policy": f"MLP obs-{args.hidden}-act (tanh)",
                      "steps": args.steps, "results": results}))
    one.close(); sub.close()


if __name__ == "__main__":
    main()
