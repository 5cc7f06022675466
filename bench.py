#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of Walker3DCustomEnv-v0 at 4096 envs per MI355X.

  python bench.py --gpus N --steps K --warmup W
For N > 1 the driver launches one process per GPU with torch.distributed.run; env batches are
independent shards (no data-path collective; RCCL is only used for the barrier and the MAX of times).

One "step" = one env.step() of all envs of a rank = one launch of the step kernel (4 physics substeps,
observation, reward, termination, in-kernel auto-reset), inputs resident in HBM.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
ENV_ID = "Walker3DCustomEnv-v0"
# algorithmic HBM bytes of one env-step (state + task + action in, state + task +
# obs + reward + done out) for Walker3DCustomEnv, contact warm-start impulses persisted (34 slots)
ALGO_BYTES_PER_ENV_STEP = 836 + 2 * 34 * 4
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(env_id: str = ENV_ID, seconds_target: float = 12.0):
    """The CPU oracle (a scalar C port of the same algorithm; PyBullet is not installable here) on one host core."""
    import numpy as np
    from oracle.oracle import Oracle, PARAM_AUTO_RESET
    from mocca_envs_amd.vec_env import TASKS, compile_model_for
    task = TASKS[env_id]
    m = compile_model_for(env_id)
    n = 16
    orc = Oracle(m.to_bytes(), task, n, "f32")
    orc.set_param(PARAM_AUTO_RESET, 1)
    orc.reset(seed=0)
    rng = np.random.default_rng(0)
    tape = rng.uniform(-1, 1, (64, n, orc.act_dim)).astype(np.float32)
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < seconds_target:
        for k in range(64):
            orc.step(tape[k])
        steps += 64
    dt = time.perf_counter() - t0
    return {"value": n * steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{env_id}: {n} envs x {steps} steps, auto-reset, U(-1,1) actions, f32 C oracle (oracle/mocca_oracle.c)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--env-id", default=ENV_ID)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--curriculum", type=int, default=None, help="Stepper envs: curriculum 0..9 (SURVEY 8d config 3)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from mocca_envs_amd.vec_env import VecEnv
    from mocca_envs_amd import sharding
    lo, hi = sharding.env_range(rank, world, args.envs)
    # same seed on every rank, draws keyed by the GLOBAL env id: the job's result does not depend on how many GPUs share it
    env = VecEnv(args.env_id, args.envs, device=local_rank, auto_reset=True, seed=1000, env_offset=lo)
    if args.curriculum is not None:
        env.set_param(2, args.curriculum)  # MOCCA_PARAM_CURRICULUM: takes effect at reset
    env.reset()
    g = torch.Generator(device=dev)
    g.manual_seed(1 + rank)
    tape = torch.rand(64, args.envs, env.act_dim, device=dev, generator=g) * 2 - 1  # U(-1,1) action tape, looped

    n_done = torch.zeros((), device=dev)
    for i in range(args.warmup):
        _, _, done, _ = env.step(tape[i % 64])
        n_done += (done != 0).sum()  # reset fraction is sampled during warm-up, outside the timed region
    reset_frac = float(n_done.item()) / max(1, args.envs * args.warmup)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP events on the stream the kernel is launched on (torch's current stream is the one handed to mocca_step)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        env.step(tape[i % 64])
    ev1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed, dist, dev)
    kern_ms = ev0.elapsed_time(ev1) / args.steps  # back-to-back launches: average launch duration incl. launch gaps

    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and args.envs == ENVS_PER_GPU and args.env_id == ENV_ID:
            # HBM bytes per launch measured offline with rocprofv3 PMC passes (see the file's "method")
            traffic = json.load(open(tpath))["traffic_bytes_per_launch"]
        total_envs = args.envs * world
        value = sharding.aggregate_throughput(args.envs, world, args.steps, elapsed)
        # per env-step: state + task + action read, state + task + obs + reward + done written
        sd, td = env.state_dim * 4, 40 * 4
        algo = (sd + td + env.act_dim * 4) + (sd + td + env.obs_dim * 4 + 5)
        if args.env_id == ENV_ID:
            algo = ALGO_BYTES_PER_ENV_STEP  # the figure quoted in DESIGN.md (24-word task record of the metric env)
        achieved = algo * args.envs / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "env-steps/sec, Walker3DCustomEnv-v0 @ 4096 envs, 1/2/4/8 MI355X",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.env_id}, {args.envs} envs/GPU, "
                                   f"{'20 stepping planks' if 'Stepper' in args.env_id else 'flat ground'}, U(-1,1) action tape, auto-reset",
                       "envs_per_gpu": args.envs, "parallelism": f"independent env shards x{world}, no collective",
                       "reset_fraction_per_step": reset_frac},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "mocca_step_kernel", "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": algo * args.envs,
                         "note": "path is latency/VALU-bound by construction (SURVEY.md 8d); HBM fraction reported per contract"},
            "kernel_info": env.kernel_info(),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.env_id)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
