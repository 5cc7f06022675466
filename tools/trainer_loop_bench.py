#!/usr/bin/env python3
"""What a SymmetricRL / ALLSTEPS-style PPO collection loop gets out of mocca_envs_amd.trainer_api.TorchVecEnv on one MI355X.
Policy: MLP obs-64-act on the GPU; every loop writes the masked observation into a rollout buffer as the PPO storage does.
  trainer_loop_verbatim       the trainers' own loop, habits included: a Python list comprehension over `done` for the masks, a loop over the
                              N info dicts (both force the lazy `done` / `infos` of THIS step: one wait per step, then ~3 x N Python iterations)
  trainer_loop_device_masks   `envs.masks` / `envs.bad_masks` from the device, episode returns from `infos.episodes()` (arrays, no dict per env)
                              read ONE STEP LATE (after the next step has been issued: the wait never lets the GPU run dry)
  trainer_loop_device_totals  the same without `infos`: episode statistics from `envs.episode_totals` (device) at the end; built with
                              `record_events=False` (no per-step event record: nothing ever waits for a step)
  trainer_loop_graphed        policy -> mocca_step -> rollout write of `--chunk` consecutive steps captured in ONE torch.cuda.CUDAGraph
                              (`TorchVecEnv.capture_rollout`) and replayed: the collection phase without the host
  trainer_loop_in_place       PPO's own structure: the policy reads rollouts.obs[t]; the step kernel writes observation, reward, masks and
                              bad_masks of step t STRAIGHT into rows t + 1 / t of the storage (`step(action, into=...)`), the policy its action
                              into rollouts.actions[t]: `rollouts.insert` without a copy kernel; `_graphed`: the same from one CUDA graph
  python tools/trainer_loop_bench.py [--envs 4096] [--steps 300] [--env-id Walker3DCustomEnv-v0] [--sub-batches 1] [--chunk 10]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--env-id", default="Walker3DCustomEnv-v0")
    ap.add_argument("--sub-batches", type=int, default=1)
    ap.add_argument("--chunk", type=int, default=10)
    ap.add_argument("--skip-verbatim", action="store_true")
    args = ap.parse_args()
    import torch
    from mocca_envs_amd.trainer_api import make_vec_envs
    envs = make_vec_envs(args.env_id, seed=0, num_processes=args.envs, sub_batches=args.sub_batches)
    # the totals loop never looks at `done` / `infos`: its envs are built without the per-step event record (1 us of GPU time per step)
    envs_totals = make_vec_envs(args.env_id, seed=0, num_processes=args.envs, sub_batches=args.sub_batches, record_events=False)
    dev = envs.device
    g = torch.Generator(device=dev).manual_seed(1)
    w1 = torch.randn(envs.observation_space.shape[0], 64, device=dev, generator=g) * 0.3
    w2 = torch.randn(64, envs.action_space.shape[0], device=dev, generator=g) * 0.3
    policy = lambda o: torch.tanh(torch.tanh(o @ w1) @ w2)
    steps = (args.steps // args.chunk) * args.chunk
    rollouts = torch.zeros(max(steps, 50) + 1, args.envs, envs.observation_space.shape[0], device=dev)

    def verbatim(steps):
        obs = envs.reset(); ep = []
        for t in range(steps):
            with torch.no_grad():
                action = policy(obs)
            obs, reward, done, infos = envs.step(action)
            for info in infos:
                if "episode" in info.keys():
                    ep.append(info["episode"]["r"])
            masks = torch.FloatTensor([[0.0] if d else [1.0] for d in done]).to(dev)
            bad_masks = torch.FloatTensor([[0.0] if "bad_transition" in info.keys() else [1.0] for info in infos]).to(dev)
            rollouts[t + 1].copy_(obs * masks * bad_masks.clamp(min=1.0))
        return len(ep)

    def lean(steps):
        obs = envs.reset(); ep = []; prev = None
        for t in range(steps):
            with torch.no_grad():
                action = policy(obs)
            obs, reward, done, infos = envs.step(action)
            if prev is not None:      # logging one step late: waits for the launch BEFORE the one just issued
                ep.append(prev.episodes()["r"])
            prev = infos
            rollouts[t + 1].copy_(obs * envs.masks * envs.bad_masks.clamp(min=1.0))
        ep.append(prev.episodes()["r"])
        return sum(len(x) for x in ep)

    def totals(steps):
        e = envs_totals
        obs = e.reset(); e.episode_totals.zero_()
        for t in range(steps):
            with torch.no_grad():
                action = policy(obs)
            obs, reward, done, infos = e.step(action)
            rollouts[t + 1].copy_(obs * e.masks * e.bad_masks.clamp(min=1.0))
        return int(e.episode_totals[2].item())

    graph = None

    def graphed(steps):
        nonlocal graph
        if graph is None:       # (a trainer captures its whole num_steps rollout; the chunk writes rollout rows 1 .. chunk)
            graph = envs.capture_rollout(policy, args.chunk, sink=lambda t, obs, rew, masks, bad, act: rollouts[t + 1].copy_(obs * masks * bad.clamp(min=1.0)))
        envs.reset(); envs.episode_totals.zero_()
        for _ in range(steps // args.chunk):
            graph.replay()
        return int(envs.episode_totals[2].item())

    # ---- the kernel writes straight into the trainer's storage (TorchVecEnv.step(into=...)): PPO's rollouts.insert without copy kernels.
    # Storage like a2c-ppo-acktr's RolloutStorage: obs [T + 1], rewards [T], masks / bad_masks [T + 1], actions [T]; the policy reads obs[t].
    T = args.chunk
    od, ad = envs.observation_space.shape[0], envs.action_space.shape[0]
    S = {"obs": torch.zeros(T + 1, args.envs, od, device=dev), "reward": torch.zeros(T, args.envs, 1, device=dev),
         "masks": torch.ones(T + 1, args.envs, 1, device=dev), "bad_masks": torch.ones(T + 1, args.envs, 1, device=dev),
         "act": torch.zeros(T, args.envs, ad, device=dev)}
    row = lambda t: {"obs": S["obs"][t + 1], "reward": S["reward"][t], "masks": S["masks"][t + 1], "bad_masks": S["bad_masks"][t + 1]} if t >= 0 else {"obs": S["obs"][0]}
    policy_into = lambda o, t: torch.tanh(torch.tanh(o @ w1) @ w2, out=S["act"][t])

    def in_place(steps):
        e = envs_totals
        S["obs"][0].copy_(e.reset()); e.episode_totals.zero_()
        for r in range(steps // T):
            for t in range(T):
                with torch.no_grad():
                    action = policy_into(S["obs"][t], t)
                e.step(action, into=row(t))
            S["obs"][0].copy_(S["obs"][T]); S["masks"][0].copy_(S["masks"][T]); S["bad_masks"][0].copy_(S["bad_masks"][T])   # rollouts.after_update()
        return int(e.episode_totals[2].item())

    graph_ip = None

    def in_place_graphed(steps):
        nonlocal graph_ip
        e = envs_totals
        if graph_ip is None:
            graph_ip = e.capture_rollout(policy_into, T, into=row)
        S["obs"][0].copy_(e.reset()); e.episode_totals.zero_()
        for r in range(steps // T):
            graph_ip.replay()
            S["obs"][0].copy_(S["obs"][T]); S["masks"][0].copy_(S["masks"][T]); S["bad_masks"][0].copy_(S["bad_masks"][T])
        return int(e.episode_totals[2].item())

    out = {"env_id": args.env_id, "envs": args.envs, "sub_batches": args.sub_batches, "steps": steps, "graph_chunk": args.chunk}
    loops = [("trainer_loop_verbatim", verbatim), ("trainer_loop_device_masks", lean), ("trainer_loop_device_totals", totals)]
    if args.sub_batches == 1:
        loops += [("trainer_loop_graphed", graphed), ("trainer_loop_in_place", in_place), ("trainer_loop_in_place_graphed", in_place_graphed)]
    for name, fn in loops:
        if args.skip_verbatim and fn is verbatim:
            continue
        fn(50); torch.cuda.synchronize()
        t0 = time.perf_counter(); n_ep = fn(steps); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out[name] = {"ms_per_step": 1e3 * dt, "env_steps_per_s": args.envs / dt, "episodes": n_ep}
    print(json.dumps(out))
    envs.close(); envs_totals.close()


if __name__ == "__main__":
    main()
