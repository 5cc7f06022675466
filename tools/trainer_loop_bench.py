#!/usr/bin/env python3
"""What a SymmetricRL / ALLSTEPS-style PPO collection loop gets out of mocca_envs_amd.trainer_api.TorchVecEnv on one MI355X:
(a) the trainers' own loop, verbatim habits included (a Python list comprehension over `done` for the masks, a loop over the N info dicts),
(b) the same loop reading `envs.masks` / `envs.bad_masks` and `infos.finished()` instead.  Policy: MLP obs-64-act on the GPU.
  python tools/trainer_loop_bench.py [--envs 4096] [--steps 300] [--env-id Walker3DCustomEnv-v0] [--sub-batches 1]"""
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
    args = ap.parse_args()
    import torch
    from mocca_envs_amd.trainer_api import make_vec_envs
    envs = make_vec_envs(args.env_id, seed=0, num_processes=args.envs, sub_batches=args.sub_batches)
    dev = envs.device
    g = torch.Generator(device=dev).manual_seed(1)
    w1 = torch.randn(envs.observation_space.shape[0], 64, device=dev, generator=g) * 0.3
    w2 = torch.randn(64, envs.action_space.shape[0], device=dev, generator=g) * 0.3
    policy = lambda o: torch.tanh(torch.tanh(o @ w1) @ w2)
    rollouts = torch.zeros(args.steps + 1, args.envs, envs.observation_space.shape[0], device=dev)

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
        obs = envs.reset(); ep = []
        for t in range(steps):
            with torch.no_grad():
                action = policy(obs)
            obs, reward, done, infos = envs.step(action)
            ep += [info["episode"]["r"] for _, info in infos.finished()]
            rollouts[t + 1].copy_(obs * envs.masks * envs.bad_masks.clamp(min=1.0))
        return len(ep)

    out = {"env_id": args.env_id, "envs": args.envs, "sub_batches": args.sub_batches, "steps": args.steps}
    for name, fn in (("trainer_loop_verbatim", verbatim), ("trainer_loop_device_masks", lean)):
        fn(50); torch.cuda.synchronize()
        t0 = time.perf_counter(); n_ep = fn(args.steps); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        out[name] = {"ms_per_step": 1e3 * dt, "env_steps_per_s": args.envs / dt, "episodes": n_ep}
    print(json.dumps(out))
    envs.close()


if __name__ == "__main__":
    main()
