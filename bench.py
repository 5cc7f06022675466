#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of Walker3DCustomEnv-v0 at 4096 envs per MI355X.

  python bench.py --gpus N --steps K --warmup W            (BASELINE.json configs[1]; N = 1, 2, 4, 8)
  python bench.py --gpus 8 --envs 8192                     (configs[4]: 8 x 8192 envs, weak scaling)
  python bench.py --env-id Walker3DStepperEnv-v0 [--curriculum 9]     (configs[2])
  python bench.py --env-id CassieEnv-v0 --envs 2048 --action-scale 0.1          (configs[3]; SURVEY 8d: a ~ 0.1 U(-1,1))
  python bench.py --envs 8192 --stagger 2 --max-rows 32              (NOT the headline protocol: the GPU's batch as two independently stepping
                                                                       sub-batches on the compact kernel instance, config.pipelined = true)

One "step" = one env.step() of all envs of a rank = ONE launch of the step kernel (4 physics substeps, observation,
reward, termination, in-kernel auto-reset), inputs resident in HBM.

Multi-GPU: env batches are independent shards (each env is its own world in the reference, env_base.py:55): one
process per GPU, env_offset = rank x envs, NO data-path collective and no RCCL -- the start/stop barrier and the MAX of
the per-rank times go over a gloo (CPU) group.  Two ways in:
  * under torchrun / torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE set): this process is one rank;
  * plain `python bench.py --gpus N` with N > 1: this process touches no GPU, spawns N child ranks (subprocess, fresh
    interpreters) with the same environment variables torchrun would set, and relays rank 0's JSON line.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
ENV_ID = "Walker3DCustomEnv-v0"
# algorithmic HBM bytes of one env-step (state + task + action in, state + task + obs + reward +
# done out) for Walker3DCustomEnv: 352 read + 484 written.  A blob that warm-starts its contact rows (warmstart != 0; the compiled blobs do
# not) also reads and writes one impulse per terrain contact slot: algo_bytes_per_env_step() adds them only then.
ALGO_BYTES_PER_ENV_STEP = 836


def algo_bytes_per_env_step(env_id, model, obs_dim, act_dim, persist_impulses=False) -> int:
    nj, ns = int(model.n_joints), int(model.n_slots)
    warm_in = 4 * ns if model.warmstart != 0.0 else 0
    warm_out = 4 * ns if (model.warmstart != 0.0 or persist_impulses) else 0
    if env_id == ENV_ID:
        return ALGO_BYTES_PER_ENV_STEP + warm_in + warm_out   # the figure quoted in DESIGN.md (24-word task record of the metric env)
    sd, td = (13 + 2 * nj) * 4, 40 * 4                          # state (pose, velocity, q, qd) + task record
    return (sd + td + act_dim * 4 + warm_in) + (sd + td + obs_dim * 4 + 5 + warm_out)


HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_TFLOPS = 157.3   # same guide: peak FP32 vector rate
KERNEL_SOURCES = ["mocca_envs_amd/csrc/mocca_device.h", "mocca_envs_amd/csrc/mocca_kernels.h", "mocca_envs_amd/csrc/mocca_api.hip",
                  "mocca_envs_amd/csrc/mocca_r32.hip", "mocca_envs_amd/csrc/topo_walker3d.h", "include/mocca_model.h"]


def kernel_source_hash() -> str:
    """Identifies the step kernel the PMC numbers in profiles/traffic.json were measured on."""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _make_oracle(env_id, n_envs, blob=None):
    from oracle.oracle import Oracle, PARAM_AUTO_RESET
    from mocca_envs_amd.vec_env import TASKS, compile_model_for
    orc = Oracle(blob if blob is not None else compile_model_for(env_id).to_bytes(), TASKS[env_id], n_envs, "f32")
    if "Planner" in env_id:     # the planner envs stand on the height field
        from mocca_envs_amd.terrain import load_height_field
        orc.set_heightfield(*load_height_field())
    if "Phase" in env_id:       # the Cassie mocap / phase envs read their reference motion
        from mocca_envs_amd.trajectory import CassieTrajectory
        tr = CassieTrajectory()
        orc.set_trajectory(tr.table(), tr.max_time(), 0.03)
    orc.set_param(PARAM_AUTO_RESET, 1)
    orc.reset(seed=0)
    return orc


def _oracle_rate(env_id, n_envs, seconds_target, max_steps=None):
    import numpy as np
    orc = _make_oracle(env_id, n_envs)
    tape = np.random.default_rng(0).uniform(-1, 1, (64, n_envs, orc.act_dim)).astype(np.float32)
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < seconds_target and (max_steps is None or steps < max_steps):
        orc.step(tape[steps % 64])
        steps += 1
    return n_envs * steps / (time.perf_counter() - t0), steps


def usable_cpus() -> int:
    """CPUs this process may actually run on: the affinity mask, cut by the cgroup's CPU quota (a GPU box of the pool reports 256 host CPUs
    and grants a 16-CPU share: 256 threads there measure the quota, not the host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = max(1, min(n, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(env_id: str = ENV_ID, seconds_target: float = 10.0):
    """The CPU oracle (a scalar C port of the same algorithm; PyBullet is not installable here) on the host cores:
    one core (the headline `value`), BASELINE.json configs[0] (1 env x 1000 steps, 1 thread) and all cores."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from mocca_envs_amd.vec_env import compile_model_for
    n = 16
    rate1, steps1 = _oracle_rate(env_id, n, seconds_target)
    out = {"value": rate1, "unit": "env-steps/s", "cores": 1, "kind": "port",
           "sample": f"{env_id}: {n} envs x {steps1} steps, auto-reset, U(-1,1) actions, f32 C oracle (oracle/mocca_oracle.c)"}
    # configs[0]: "1 env, 1000 random-action steps" on one thread (the reference's own CPU-runnable case)
    r0, s0 = _oracle_rate(env_id, 1, 30.0, max_steps=1000)
    out["config0_1env_1000steps"] = {"value": r0, "unit": "env-steps/s", "cores": 1, "steps": s0}
    # every host core: one oracle instance per thread, each running its K steps inside ONE C call (orc_rollout: the interpreter lock is
    # released for the whole call; round 4 called orc_step once per step from 256 Python threads and measured the lock, 10.6 x one core)
    cores = usable_cpus()
    blob = compile_model_for(env_id).to_bytes()
    orcs = [_make_oracle(env_id, n, blob) for _ in range(cores)]
    tape = np.random.default_rng(0).uniform(-1, 1, (64, n, orcs[0].act_dim)).astype(np.float32)
    k = max(8, int(rate1 / n * 0.5 * seconds_target))      # ~ half the single-core sample's duration per thread
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(lambda o: o.rollout(tape, 2), orcs))   # threads started, pages touched
        t0 = time.perf_counter()
        list(ex.map(lambda o: o.rollout(tape, k), orcs))
        wall = time.perf_counter() - t0
    out["all_cores"] = {"value": cores * n * k / wall, "unit": "env-steps/s", "cores": cores, "host_cpus": os.cpu_count(),
                        "sample": f"{cores} threads (the CPUs this process may use: affinity mask and cgroup quota) x {n} envs x {k} steps, "
                                  "one C call per thread (orc_rollout)"}
    return out


WALKER3D_GAINS = (60, 80, 60, 80, 60, 100, 90, 60, 80, 60, 100, 90, 60, 60, 60, 50, 60, 60, 60, 50, 60)   # robots.py:168,234-256


def pybullet_baseline(n_steps: int = 1000, module=None):
    """BASELINE.md section 3, "opportunistic PyBullet baseline": if `import pybullet` works on this box AND the reference's model assets are
    reachable (MOCCA_REF_DATA = .../mocca_envs/data: the repo carries no copy of walker3d.xml), drive RAW pybullet with this loop -- never the
    reference's Python files -- on one core: plane + Walker3D, fixedTimeStep 1/60, 4 substeps, 5 solver iterations, contact ERP 0.9
    (bullet_utils.py:340-350), running-start pose at rest, n_steps of gains x U(-1,1)^21 torques from default_rng(0), and per step the eleven
    Python->C crossings of Walker3DCustomEnv.step (set torques, stepSimulation, getJointStates, 2 x getBasePositionAndOrientation +
    getEulerFromQuaternion, getBaseVelocity, 2 x getLinkState, 2 x getContactPoints); a fallen robot (z < 0.5) is put back.  Returns the
    `cpu_baseline["pybullet"]` object: value in env-steps/s, kind "reference", or value None with the reason.  `module`: tests hand in
    tests/fake_pybullet.py's stand-in."""
    import numpy as np
    p = module
    if p is None:
        try:
            import pybullet as p            # noqa: F811
        except ImportError:
            return {"value": None, "unit": "env-steps/s", "cores": 1, "kind": "reference", "sample": "n/a (pybullet not installed)"}
    data = os.environ.get("MOCCA_REF_DATA", "")
    xml, sdf = os.path.join(data, "robots", "walker3d.xml"), os.path.join(data, "objects", "misc", "plane_stadium.sdf")
    if module is None and not (os.path.isfile(xml) and os.path.isfile(sdf)):
        return {"value": None, "unit": "env-steps/s", "cores": 1, "kind": "reference",
                "sample": "n/a (pybullet is installed, but MOCCA_REF_DATA does not point at the reference's mocca_envs/data directory)"}
    cid = p.connect(p.DIRECT)
    try:
        p.setGravity(0, 0, -9.8)
        p.setDefaultContactERP(0.9)
        p.setPhysicsEngineParameter(fixedTimeStep=1 / 60, numSolverIterations=5, numSubSteps=4)
        plane = p.loadSDF(sdf)[0]
        p.changeDynamics(plane, -1, lateralFriction=0.8, restitution=0.5)
        robot = p.loadMJCF(xml, flags=p.MJCF_COLORS_FROM_FILE | p.URDF_USE_SELF_COLLISION | p.URDF_USE_SELF_COLLISION_EXCLUDE_ALL_PARENTS)[0]
        nj = p.getNumJoints(robot)
        info = [p.getJointInfo(robot, j) for j in range(nj)]
        act = [j for j in range(nj) if not info[j][1].decode().startswith(("jointfix", "ignore"))]
        links = [ji[12].decode() for ji in info]
        feet = [links.index(n) for n in ("right_foot", "left_foot")]
        for j in range(nj):
            p.setJointMotorControl2(robot, j, p.POSITION_CONTROL, positionGain=0.1, velocityGain=0.1, force=0)
        q0 = np.zeros(21)                                        # Walker3D.set_base_pose("running_start"), robots.py:296-302
        q0[[5, 6]] = -np.pi / 8; q0[10] = np.pi / 10; q0[[13, 17]] = np.pi / 3; q0[14] = -np.pi / 6; q0[18] = np.pi / 6; q0[[16, 20]] = np.pi / 3

        def place():
            p.resetBasePositionAndOrientation(robot, [0, 0, 1.32], [0, 0, 0, 1])
            p.resetBaseVelocity(robot, [0, 0, 0], [0, 0, 0])
            for k, j in enumerate(act):
                p.resetJointState(robot, j, float(q0[k]), 0.0)

        place()
        gains = np.asarray(WALKER3D_GAINS, float)
        actions = np.random.default_rng(0).uniform(-1, 1, (n_steps, 21))
        restarts, sink = 0, 0.0
        t0 = time.perf_counter()
        for t in range(n_steps):
            p.setJointMotorControlArray(robot, act, p.TORQUE_CONTROL, forces=list(gains * actions[t]))
            p.stepSimulation()
            js = p.getJointStates(robot, act)
            pos, orn = p.getBasePositionAndOrientation(robot)
            rpy = p.getEulerFromQuaternion(p.getBasePositionAndOrientation(robot)[1])
            lin, _ = p.getBaseVelocity(robot)
            fz = [p.getLinkState(robot, f)[0][2] for f in feet]
            fc = [len(p.getContactPoints(bodyA=robot, linkIndexA=f)) for f in feet]
            sink += js[0][0] + rpy[2] + lin[0] + min(fz) + fc[0]
            if pos[2] < 0.5:
                place(); restarts += 1
        wall = time.perf_counter() - t0
    finally:
        p.disconnect(cid)
    return {"value": n_steps / wall, "unit": "env-steps/s", "cores": 1, "kind": "reference",
            "sample": f"raw pybullet, 1 env x {n_steps} steps, Walker3D on the stadium plane, U(-1,1) torques x gains, 11 C-API calls per step, "
                      f"{restarts} restarts of a fallen robot" + (" [stand-in module: not a measurement]" if module is not None else "")}


PHYSICS_VARIANTS = ("limit_rows_from_predicted_gap", "absolute_2cm_margins", "pyramid_friction", "warmstart_0.85", "all_four")


def physics_variant_model(env_id, variant):
    """The blob with one (or all) of the [UNVERIFIED-BULLET] solver laws of DESIGN.md section 3 switched to its other reading -- each is a blob field
    that kernel and oracle honour (tests/test_gpu_substep.py runs every one against the oracle): limit rows from a predicted gap of
    limit_slack on instead of at the stop only; one absolute 2 cm contact margin instead of Bullet's relative breaking thresholds
    (millimetres); pyramid instead of cone friction; contact rows warm-started with 0.85 x the last impulse instead of from zero
    (bullet_utils.py:340-350 sets none of them: they are Bullet defaults as recalled)."""
    from mocca_envs_amd.vec_env import compile_model_for
    m = compile_model_for(env_id)
    if variant in ("limit_rows_from_predicted_gap", "all_four"):
        m.limit_at_violation = 0
    if variant in ("absolute_2cm_margins", "all_four"):
        for g in range(m.n_geoms):
            m.g_margin[g] = 0.0
        m.finalize_tables()
    if variant in ("pyramid_friction", "all_four"):
        m.friction_cone = 0
    if variant in ("warmstart_0.85", "all_four"):
        m.warmstart = 0.85
    return m


def physics_bracket(args, local_rank, lo, tape, steps=200, preroll=800):
    """The headline launch re-timed on blobs that read Bullet's unverifiable laws the OTHER way (untimed for `value`): how much of the
    rate is the builder's reading of Bullet.  Returns {variant: {ms_per_step, value, rows_per_substep}}."""
    import torch
    from mocca_envs_amd.vec_env import VecEnv
    out = {}
    for variant in ("as_built",) + PHYSICS_VARIANTS:
        m = physics_variant_model(args.env_id, variant) if variant != "as_built" else None
        env = VecEnv(args.env_id, args.envs, device=local_rank, auto_reset=True, seed=1000, env_offset=lo,
                     model_blob=m.to_bytes() if m is not None else None)
        if args.curriculum is not None:
            env.set_param(2, args.curriculum)
        env.reset()
        for i in range(preroll):
            env.step(tape[i % 64])
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        rows = torch.zeros((), device=tape.device)
        torch.cuda.synchronize()
        ev0.record()
        for i in range(steps):
            env.step(tape[i % 64])
        ev1.record()
        torch.cuda.synchronize()
        for i in range(16):     # rows of the last substep of 16 more steps (task word 23), outside the timed loop
            env.step(tape[i % 64])
            rows += env.get_task()[:, 23].float().mean()
        ms = ev0.elapsed_time(ev1) / steps
        out[variant] = {"ms_per_step": ms, "value": args.envs / (ms * 1e-3), "rows_per_substep": float(rows.item()) / 16}
        env.close()
    return out


WORKLOADS = ("uniform_0.3", "zero_actions", "pd_to_t_pose", "ppo_policy")
# tools/ppo_demo.py: plain PPO on this stepper (1.05 G env-steps = 7 minutes on one MI355X for the walker, 0.9 G = 6 minutes for the Stepper at curriculum 0, CassieEnv)
PPO_POLICIES = {ENV_ID: os.path.join(ROOT, "profiles", "ppo_policy_walker3d.npz"),
                "Walker3DStepperEnv-v0": os.path.join(ROOT, "profiles", "ppo_policy_stepper.npz"),
                "CassieEnv-v0": os.path.join(ROOT, "profiles", "ppo_policy_cassie.npz")}   # 0.6 G env-steps at 4096 envs, 6 minutes
PPO_POLICY = PPO_POLICIES[ENV_ID]
PD_KP, PD_KD = 2.0, 0.5     # action = clip(KP (theta_norm* - theta_norm) - KD (0.1 qdot), -1, 1): both terms in the observation's units (robots.py:46-50)


def workload_bracket(args, local_rank, lo, tape, steps=200, preroll=800):
    """The headline launch re-timed on OTHER BEHAVIOUR (untimed for `value`): random torques make robots that fall every ~22 steps, a trained
    policy (env_locomotion.py:111-141 is its reward) makes robots that stay up with their feet planted.  Three stand-ins: 0.3 x U(-1,1)
    torques; zero torques; a PD controller to the T-pose computed on the device from the observation (normalised joint angles and speeds,
    robots.py:46-50), which keeps robots standing for longer.  Closed-loop workloads cannot come from a tape, and timing the controller's
    torch kernels would not time the step kernel: each workload is therefore (1) pre-rolled closed-loop, (2) snapshotted (state, task
    record, terrain), (3) run closed-loop for `steps` steps while the actions are RECORDED, (4) restored and replayed from the recorded
    actions back to back under HIP events -- the same launches on the same states (`replay_exact`: final observations bit-identical).
    A fourth workload, Walker3DCustomEnv-v0, Walker3DStepperEnv-v0 and CassieEnv-v0: `ppo_policy` -- the policy `tools/ppo_demo.py` trained on this very stepper
    (weights in profiles/ppo_policy_*.npz: MLP obs-256-256-21 on normalised observations, with its training-time action noise): robots that
    WALK to their targets / over the planks for the full 1000 steps, what the batch looks like late in a trainer's run
    (env_locomotion.py:111-141 / :515-568 are its reward; the Stepper's policy was trained at curriculum 0).
    Returns {workload: {ms_per_step, value, rows_per_substep, reset_fraction_per_step, replay_exact}}."""
    import torch
    from mocca_envs_amd.vec_env import VecEnv
    out = {}
    stepper = "Stepper" in args.env_id
    for name in WORKLOADS:
        if name == "pd_to_t_pose" and "Cassie" in args.env_id:
            continue                      # (Cassie's action already is a PD target: env_cassie.py:380-393)
        if name == "ppo_policy" and not os.path.exists(PPO_POLICIES.get(args.env_id, "")):
            continue
        env = VecEnv(args.env_id, args.envs, device=local_rank, auto_reset=True, seed=1000, env_offset=lo)
        if args.curriculum is not None:
            env.set_param(2, args.curriculum)
        obs = env.reset()
        nj = env.act_dim
        if name == "pd_to_t_pose":        # theta_norm of q = 0: 2 (0 - lo) / (hi - lo) - 1 (robots.py:132)
            jlo = torch.tensor([env.model.jlo[1 + j] for j in range(nj)], device=obs.device)
            jhi = torch.tensor([env.model.jhi[1 + j] for j in range(nj)], device=obs.device)
            target = -(jhi + jlo) / (jhi - jlo)
        zero = torch.zeros(args.envs, nj, device=obs.device)
        if name == "ppo_policy":
            import numpy as np
            w = {k: torch.from_numpy(v).to(obs.device) for k, v in np.load(PPO_POLICIES[args.env_id]).items() if v.dtype.kind == "f"}
            std, gen = w["log_std"].exp(), torch.Generator(device=obs.device).manual_seed(7)

            def ppo_action():
                o = ((obs - w["obs_mean"]) / torch.sqrt(w["obs_var"] + 1e-8)).clamp(-10.0, 10.0)
                h = torch.tanh(torch.tanh(o @ w["pi_0_weight"].T + w["pi_0_bias"]) @ w["pi_2_weight"].T + w["pi_2_bias"])
                mu = h @ w["pi_4_weight"].T + w["pi_4_bias"]
                return mu + std * torch.randn(mu.shape, device=mu.device, generator=gen)

        def action(i):
            if name == "uniform_0.3":
                return 0.3 * tape[i % 64]
            if name == "zero_actions":
                return zero
            if name == "ppo_policy":
                return ppo_action()
            return (PD_KP * (target - obs[:, 6:6 + nj]) - PD_KD * obs[:, 6 + nj:6 + 2 * nj]).clamp_(-1.0, 1.0)

        for i in range(preroll):
            env.step(action(i))
        snap = (env.get_state(), env.get_task(), env.get_terrain() if stepper else None)
        rec = torch.empty(steps, args.envs, nj, device=obs.device)
        n_done = torch.zeros((), device=obs.device)
        for i in range(steps):
            rec[i].copy_(action(preroll + i))
            env.step(rec[i])
            n_done += (env.done != 0).sum()
        final = env.obs.clone()
        env.set_state(snap[0]); env.set_task(snap[1])
        if stepper:
            env.set_terrain(snap[2])
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        ev0.record()
        for i in range(steps):
            env.step(rec[i])
        ev1.record()
        torch.cuda.synchronize()
        exact = bool(torch.equal(final, env.obs))
        rows = torch.zeros((), device=obs.device)
        for i in range(16):     # rows of the last substep of 16 more steps (task word 23), outside the timed loop
            env.step(action(preroll + steps + i))
            rows += env.get_task()[:, 23].float().mean()
        ms = ev0.elapsed_time(ev1) / steps
        out[name] = {"ms_per_step": ms, "value": args.envs / (ms * 1e-3), "rows_per_substep": float(rows.item()) / 16,
                     "reset_fraction_per_step": float(n_done.item()) / (args.envs * steps), "replay_exact": exact}
        env.close()
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs", type=int, default=ENVS_PER_GPU, help="envs per GPU (4096 = the metric's config; 8192 = configs[4])")
    ap.add_argument("--env-id", default=ENV_ID)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--preroll", type=int, default=1000,
                    help="data preparation, before the W warm-up steps: env-steps that age the freshly reset batch into the steady-state mix of "
                         "episode ages and contact counts (SURVEY 8d config 2 asks for >= 200; 1000 = one max_episode_steps, so TimeLimit truncations "
                         "are in the mix) and bring the GPU to its operating clocks -- after idling the first ~50 ms of launches run 9 %% slow "
                         "(profiles/archive/r02_ramp_probe.txt); untimed, reported in config.preroll_steps")
    ap.add_argument("--preroll-seconds", type=float, default=0.5,
                    help="keep pre-rolling (same launches, untimed) until the GPU has been busy this long.  A 20-launch window that follows a "
                         "synchronize samples the power management's recovery as much as the kernel: after 3 s of idling it runs 108 us per "
                         "launch, with >= 0.1 s of launches behind it 100 - 102 us, although 1000 launches back to back take 100 us either way "
                         "(profiles/archive/r04_short_window.txt).  0 = steps only; reported in config.preroll_seconds")
    ap.add_argument("--curriculum", type=int, default=None, help="Stepper envs: curriculum 0..9 (SURVEY 8d config 3)")
    ap.add_argument("--prio", default=None,
                    help="TUNING ONLY: t1,t2,t3 row-count thresholds of the step kernel's issue priorities (MOCCA_PARAM_ISSUE_PRIORITY) in place of "
                         "the env id's default; timing only, results do not depend on it")
    ap.add_argument("--max-rows", type=int, default=None,
                    help="solver cap of the batch (MoccaModel.max_rows; default: the compiled blob's 48).  <= 32 selects the compact step-kernel "
                         "instance (less LDS per env, five resident waves per SIMD instead of four); reported in config.max_rows")
    ap.add_argument("--kernel-variant", type=int, default=0, help="TIMING ONLY: 1 forces the 48-row kernel instance for a blob that fits the compact one (A/B)")
    ap.add_argument("--order-every", type=int, default=None, help="TIMING ONLY: MOCCA_PARAM_ORDER_EVERY (heaviest envs first, re-sorted every K steps; 0 off; default: VecEnv's choice)")
    ap.add_argument("--pace", type=int, default=None, help="TIMING ONLY: MOCCA_PARAM_PACE_TICKS (pace priorities; 0 off)")
    ap.add_argument("--stagger", type=int, default=0,
                    help="NOT the headline protocol: split the batch into this many sub-batches (own handles, own HIP streams) whose steps are not "
                         "ordered against each other -- what a trainer does that computes the policy for one half while the other half steps "
                         "(every env still advances exactly K times inside the timed region).  The tail of one sub-batch's launch is filled by the "
                         "next launch of another; reported with config.pipelined = true, never as BENCH_rNN's value")
    ap.add_argument("--action-scale", type=float, default=1.0,
                    help="actions are action_scale x U(-1,1) (SURVEY 8d config 4 asks for 0.1 on Cassie: a robot that stays up instead of one that "
                         "falls every ~17 steps); reported in config.workload")
    ap.add_argument("--no-physics-bracket", action="store_true",
                    help="skip the sensitivity block: after the timed region the headline launch is re-timed (200 launches each, N = 1 only) on blobs "
                         "that read the unverifiable Bullet laws the other way -- limit rows from a predicted gap, 2 cm absolute margins, pyramid "
                         "friction, warm start 0.85, and all four; reported as `sensitivity`, never as `value`")
    ap.add_argument("--no-workload-bracket", action="store_true",
                    help="skip the workload block: after the timed region the headline launch is re-timed (N = 1 only) on other behaviour than falling "
                         "robots -- 0.3 x U(-1,1) torques, zero torques, a PD controller to the T-pose -- reported as `workload_sensitivity`, never as `value`")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: exercise launch / rendezvous / reporting only (CPU tests)")
    ap.add_argument("--test-barrier-delay", type=float, default=0.0,
                    help="TEST ONLY: every rank sleeps this many seconds inside each barrier (a slow rendezvous); the timed window must not see it")
    ap.add_argument("--host-io", action="store_true",
                    help="also time the loop with actions coming from pinned host memory and obs / reward / done copied back to the host "
                         "every step (the PCIe-inclusive rate quoted in DESIGN.md; reported beside `value`, never as it)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="TEST ONLY: let ranks share GPUs (device = rank %% visible GPUs) so the N-rank path can be exercised on a 1-GPU box; "
                         "the line is marked and is not a scaling measurement")
    args = ap.parse_args(argv)
    # The two untimed brackets re-run the headline launch ~8000 times on variant blobs and workloads.  Under a profiler they would swamp the
    # 25 headline launches in every per-kernel average (tools/pmc.sh averages over all launches named mocca_step), and an A/B library
    # (MOCCA_LIB_PATH) is there to time ONE kernel: both cases skip them without being told to.
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "") \
            or os.environ.get("MOCCA_LIB_PATH"):
        args.no_physics_bracket = args.no_workload_bracket = True
    if args.host_io and args.stagger > 1:
        ap.error("--host-io times the synchronous host loop of ONE handle; it does not combine with --stagger")
    return args


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpus() -> int:
    """GPUs of this node, counted WITHOUT loading a HIP runtime: the KFD topology lists one node per agent, and the GPU agents are
    the ones with SIMDs (CPU nodes report simd_count 0).  HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES narrow the set.  Returns -1 when
    the topology is not readable (the ranks then find out themselves and fail with a non-zero exit code)."""
    import glob
    n, seen = 0, False
    for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            with open(path) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            continue
        seen = True
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    if not seen:
        return -1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(args) -> int:
    """`python bench.py --gpus N` without torchrun: spawn the N ranks.  The parent's only job is to start children and relay rank
    0's line: it imports neither torch nor any HIP library (a process that has initialised HIP must not be replaced or forked on
    this pool), and counts the GPUs from the KFD topology in /sys."""
    if not args.dry_run:
        have = visible_gpus()
        if 0 <= have < args.gpus and not args.oversubscribe:
            print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible", file=sys.stderr)
            return 2
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    return max(rcs)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: note: WORLD_SIZE={world} overrides --gpus {args.gpus}", file=sys.stderr)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")   # barrier + MAX of times only: no RCCL on this path (independent shards)

    from mocca_envs_amd import sharding
    lo, hi = sharding.env_range(rank, world, args.envs)
    reset_frac, kern_ms, kinfo, env, host_io_ms = 0.0, 0.0, {}, None, None

    def barrier():
        # start / stop rendezvous over gloo (CPU, TCP): milliseconds across 8 processes -- always OUTSIDE the rank's clock
        if args.test_barrier_delay > 0:
            time.sleep(args.test_barrier_delay)
        if dist is not None:
            dist.barrier()

    if args.dry_run:
        barrier()
        t0 = time.perf_counter()
        time.sleep(1e-3 * args.steps * (1 + rank))   # stands in for the K launches + synchronize: rank r takes (r + 1) ms per step
        elapsed_rank = time.perf_counter() - t0
        barrier()
        kern_ms = 1.0 + rank
        preroll_steps, preroll_s = args.preroll, 0.0     # declared, not run
    else:
        import torch
        if args.oversubscribe:
            local_rank %= max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        from mocca_envs_amd.vec_env import VecEnv
        from mocca_envs_amd.multi import SubBatchedVecEnv
        # same seed on every rank, draws keyed by the GLOBAL env id: the job's result does not depend on how many GPUs share it
        kw = dict(device=local_rank, auto_reset=True, seed=1000, env_offset=lo, max_rows=args.max_rows)
        if args.stagger > 1:   # sub-batches with their own handles and HIP streams (same global env ids, same seed, same results as one handle)
            assert args.envs % args.stagger == 0, "--stagger must divide --envs"
            env = SubBatchedVecEnv(args.env_id, args.envs, sub_batches=args.stagger, **kw)
        else:
            env = VecEnv(args.env_id, args.envs, **kw)
        # handle parameters: set_param of a SubBatchedVecEnv reaches every sub-batch's handle
        if args.order_every is not None:
            env.set_param(12, args.order_every)  # MOCCA_PARAM_ORDER_EVERY
        if args.pace is not None:
            env.set_param(13, args.pace)  # MOCCA_PARAM_PACE_TICKS
        if args.kernel_variant:
            env.set_param(11, args.kernel_variant)  # MOCCA_PARAM_KERNEL_VARIANT
        if args.curriculum is not None:
            env.set_param(2, args.curriculum)  # MOCCA_PARAM_CURRICULUM: takes effect at reset
        if args.prio:
            t1, t2, t3 = (int(x) for x in args.prio.split(","))
            env.set_param(9, t1 + 64 * t2 + 4096 * t3)  # MOCCA_PARAM_ISSUE_PRIORITY
        subs = list(range(args.stagger)) if args.stagger > 1 else []
        env.reset()
        torch.cuda.synchronize()
        g = torch.Generator(device=dev)
        g.manual_seed(1 + rank)
        tape = (torch.rand(64, args.envs, env.act_dim, device=dev, generator=g) * 2 - 1) * args.action_scale  # action_scale x U(-1,1) action tape, looped

        # the synthetic input of this metric is a batch of envs in mid-episode, not 4096 identical first frames: age it
        sub_tapes = [tape[:, env.slices[k]].contiguous() for k in subs]
        torch.cuda.synchronize()

        def step_all(i, ordered=False):
            """one env.step of every env of this rank; returns the done flags of the batch (sub-batches: not yet ordered against the
            current stream -- env.wait(k) does that).  ordered=True: the sub-batches' launches wait for what the current stream has queued
            (the warm-up loop's reduction over `done` still reads the rows the next launch overwrites)"""
            if not subs:
                return env.step(tape[i % 64])[2]
            for k in subs:      # SubBatchedVecEnv.step_async: one launch on sub-batch k's own stream; the tape is resident, nothing to order
                env.step_async(k, sub_tapes[k][i % 64], ordered=ordered)
            return env.done

        # First use of everything the warm-up and the timed window call besides mocca_step, BEFORE the pre-roll: torch loads the code objects
        # of its reduction kernels on first use (40 - 150 ms on the host, the GPU idle meanwhile).  With that gap between the pre-roll and
        # the timed window, all K = 20 launches of the window run 7 - 14 % slow (profiles/archive/r04_short_window4.txt).
        n_done = torch.zeros((), device=dev)
        n_done += (env.done != 0).sum()
        n_done.item()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(); ev1.record()
        torch.cuda.synchronize()
        ev0.elapsed_time(ev1)
        barrier()   # N > 1: the ranks start their pre-roll together (and gloo's connections exist before the start barrier needs them)
        for i in range(args.preroll):
            step_all(i + 17)
        preroll_steps, preroll_s = args.preroll, 0.0
        if args.preroll_seconds > 0:
            # N > 1: every rank's timed pre-roll ends preroll_seconds after its release from this barrier, in chunks of 8 launches, so the
            # ranks reach the start barrier of the timed window within ~1 ms of each other.  A rank that waits there with an idle GPU for
            # 5 ms runs its whole 20-launch window 4 % slow, for >= 10 ms 9 % slow (profiles/archive/r04_idle_gap.txt),
            # and the job's time is the slowest rank's.
            torch.cuda.synchronize()
            barrier()
            t_pre = time.perf_counter()
            while preroll_s < args.preroll_seconds:
                for i in range(8):
                    step_all(preroll_steps + i + 17)
                preroll_steps += 8
                torch.cuda.synchronize()
                preroll_s = time.perf_counter() - t_pre
        n_done = torch.zeros((), device=dev)
        for i in range(args.warmup):
            done = step_all(i, ordered=True)
            for k in subs:
                env.wait(k)
            n_done += (done != 0).sum()  # reset fraction is sampled during warm-up, outside the timed region
        reset_frac = float(n_done.item()) / max(1, args.envs * args.warmup)
        # The K timed steps are bracketed by barrier + synchronize on both sides.  Each rank's clock runs from its release out of the
        # start barrier to the return of ITS OWN synchronize after the K-th launch; the stop barrier comes after the clock is read
        # (a gloo barrier over TCP costs 0.1 - 1 ms across 8 processes, the same order as the 2.6 ms window of --steps 20), and the
        # job's time is the MAX of the per-rank times.
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        # HIP events on the stream the kernel is launched on (torch's current stream is the one handed to mocca_step)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for k in subs:
            env._before(k)                                         # the sub-batches start after ev0
        for i in range(args.steps):
            step_all(i)
        for k in subs:
            env._after(k)                                          # ev1 after the last launch of every sub-batch
        ev1.record()
        torch.cuda.synchronize()
        elapsed_rank = time.perf_counter() - t0
        barrier()
        torch.cuda.synchronize()
        kern_ms = ev0.elapsed_time(ev1) / args.steps  # back-to-back launches: average launch duration incl. launch gaps
        kinfo = env.kernel_info()
        bracket = None
        if world == 1 and not args.no_physics_bracket and not subs and args.max_rows is None:
            try:      # (untimed extra: a failure here must not cost the line its headline)
                cassie = "Cassie" in args.env_id     # 50 physics steps per env.step: a shorter window
                bracket = physics_bracket(args, local_rank, lo, tape, steps=40 if cassie else 200, preroll=150 if cassie else 800)
            except Exception as e:      # noqa: BLE001
                bracket = {"error": f"{type(e).__name__}: {e}"}
        workloads = None
        if world == 1 and not args.no_workload_bracket and not subs and args.max_rows is None:
            try:
                cassie = "Cassie" in args.env_id
                workloads = workload_bracket(args, local_rank, lo, tape, steps=40 if cassie else 200, preroll=150 if cassie else 800)
            except Exception as e:      # noqa: BLE001
                workloads = {"error": f"{type(e).__name__}: {e}"}
        host_io_ms = None
        if args.host_io:   # a trainer on the host: actions up, obs / reward / done down, every step, through PCIe
            h_act = tape.cpu().pin_memory()
            h_obs = torch.empty(args.envs, env.obs_dim).pin_memory()
            h_rew, h_done = torch.empty(args.envs).pin_memory(), torch.empty(args.envs, dtype=torch.uint8).pin_memory()
            d_act = torch.empty(args.envs, env.act_dim, device=dev)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.steps):
                d_act.copy_(h_act[i % 64], non_blocking=True)
                o, r, d, _ = env.step(d_act)
                h_obs.copy_(o, non_blocking=True); h_rew.copy_(r, non_blocking=True); h_done.copy_(d, non_blocking=True)
                torch.cuda.synchronize()          # the host policy needs this step's observation before it can act
            host_io_ms = 1e3 * (time.perf_counter() - t1) / args.steps

    elapsed = sharding.max_over_ranks(elapsed_rank, dist)
    per_rank = sharding.gather_over_ranks([1e3 * elapsed_rank / args.steps, kern_ms], dist)   # [world][2]
    # which GPU each rank ran on: a scaling line must prove N distinct devices (uuid / PCI bus id from the runtime, host + local index beside it)
    ident = {"rank": rank, "host": socket.gethostname(), "local_index": local_rank, "name": None, "uuid": None, "pci_bus_id": None}
    if not args.dry_run:
        import torch
        pr = torch.cuda.get_device_properties(local_rank)
        ident["name"] = pr.name
        ident["uuid"] = str(getattr(pr, "uuid", "")) or None
        pci = [getattr(pr, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
        ident["pci_bus_id"] = "%04x:%02x:%02x" % tuple(pci) if all(isinstance(x, int) for x in pci) else None
    devices = sharding.gather_objects(ident, dist)
    if rank == 0:
        traffic, valu, pmc_note = None, None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and args.envs == ENVS_PER_GPU and args.env_id == ENV_ID:
            # HBM bytes / VALU work per launch measured offline with rocprofv3 PMC passes at steady state (tools/profile_round.sh).
            # The file names the kernel sources it was measured on; numbers of another kernel are not reported.
            tj = json.load(open(tpath))
            if tj.get("kernel_source_sha256") == kernel_source_hash():
                traffic = tj["traffic_bytes_per_launch"]
                valu = tj.get("valu")
            else:
                pmc_note = "profiles/traffic.json was measured on other kernel sources (stale): traffic not reported"
        value = sharding.aggregate_throughput(args.envs, world, args.steps, elapsed)
        # per env-step: state + task + action read, state + task + obs + reward + done written, derived from the blob
        algo = algo_bytes_per_env_step(args.env_id, env.model, env.obs_dim, env.act_dim) if env is not None else 0
        achieved = algo * args.envs / (kern_ms * 1e-3) / 1e9
        terrain = "20 stepping planks" if "Stepper" in args.env_id else ("height field" if "Planner" in args.env_id else "flat ground")
        scale_txt = "" if args.action_scale == 1.0 else f"{args.action_scale:g} x "
        if args.curriculum is not None:
            terrain += f", curriculum {args.curriculum}"
        out = {
            "metric": "env-steps/sec, Walker3DCustomEnv-v0 @ 4096 envs, 1/2/4/8 MI355X",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.env_id}, {args.envs} envs/GPU, {terrain}, {scale_txt}U(-1,1) action tape, auto-reset",
                       "max_rows": int(env.model.max_rows) if env is not None else None, "max_contacts": int(env.model.max_contacts) if env is not None else None,
                       "envs_per_gpu": args.envs, "parallelism": f"independent env shards x{world}, no collective",
                       "reset_fraction_per_step": reset_frac, "preroll_steps": preroll_steps, "preroll_seconds": round(preroll_s, 3)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "mocca_step_kernel", "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": algo * args.envs,
                         "note": "path is latency/VALU-bound by construction (SURVEY.md 8d); HBM fraction reported per contract"},
            "kernel_info": kinfo,
            # one entry per rank (rank order): wall ms per step of the rank's own K steps, and its HIP-event kernel time; `ms_per_step`
            # above is the max of the first list
            "per_rank": {"ms_per_step": [r[0] for r in per_rank], "kernel_ms": [r[1] for r in per_rank], "device": devices},
        }
        keys = [(d["host"], d["uuid"] or d["pci_bus_id"] or d["local_index"]) for d in devices]
        out["per_rank"]["distinct_devices"] = len(set(keys))
        if not args.dry_run and not args.oversubscribe and len(set(keys)) != world:
            print(f"bench.py: {world} ranks ran on {len(set(keys))} distinct GPU(s): {keys} -- not a scaling measurement (use --oversubscribe to rehearse)", file=sys.stderr)
            sys.exit(3)
        if pmc_note:
            out["roofline"]["traffic_note"] = pmc_note
        if valu:
            # secondary roofline: counted fp32 VALU lane-operations of one launch (SQ_THREAD_CYCLES_VALU, offline
            # PMC pass) over THIS run's launch time, against the 157.3 TFLOP/s vector peak (an FMA lane-op counted as 2 flop)
            lane_ops = valu["active_lane_ops_per_launch"]
            tf = 2.0 * lane_ops / (kern_ms * 1e-3) / 1e12
            out["roofline_valu"] = {"bound": "valu", "achieved": tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": tf / VALU_PEAK_TFLOPS, "valu_insts_per_env_step": valu["valu_insts_per_env_step"],
                                    "active_lanes_per_valu_inst": valu["active_lanes_per_valu_inst"],
                                    "note": "upper bound on useful flop: every active VALU lane-op counted as one FMA"}
        if not args.dry_run and host_io_ms is not None:
            out["host_io"] = {"ms_per_step": host_io_ms, "value": args.envs * world / (host_io_ms * 1e-3), "unit": "env-steps/s",
                              "note": "actions from pinned host memory, obs + reward + done copied to the host and waited for every step"}
        if not args.dry_run and bracket and "error" in bracket:
            out["sensitivity"] = bracket
        elif not args.dry_run and bracket:
            worst = min(bracket.values(), key=lambda v: v["value"])
            out["sensitivity"] = {"note": "the same launch on blobs that read Bullet's unverifiable solver laws the other way (DESIGN.md section 3; 200 launches "
                                          "each after an 800-step pre-roll -- Cassie: 40 after 150 --, kernel time by HIP events); `value` above is the as-built reading",
                                  "variants": bracket, "worst_case_value": worst["value"]}
        if not args.dry_run and workloads and "error" in workloads:
            out["workload_sensitivity"] = workloads
        elif not args.dry_run and workloads:
            vals = [v["value"] for v in workloads.values()] + [value]
            out["workload_sensitivity"] = {
                "note": "the same launch on other behaviour than the headline's falling robots (untimed for `value`): 0.3 x U(-1,1) torques, zero torques, "
                        f"a PD controller to the T-pose (kp {PD_KP}, kd {PD_KD} in observation units) computed on the device, and -- Walker3D Custom / Stepper -- the "
                        "walking policy tools/ppo_demo.py trained on this stepper (profiles/ppo_policy_*.npz); closed-loop runs are recorded and "
                        "replayed from a snapshot back to back under HIP events (bench.py workload_bracket)",
                "headline": {"ms_per_step": kern_ms, "value": args.envs / (kern_ms * 1e-3), "reset_fraction_per_step": reset_frac},
                "workloads": workloads, "range": [min(vals), max(vals)]}
        if args.stagger > 1:
            out["config"]["pipelined"] = True
            out["config"]["workload"] += f"; {args.stagger} sub-batches on their own streams (mocca_envs_amd.multi.SubBatchedVecEnv.step_async), their steps overlap (NOT the headline protocol)"
            out["roofline"]["kernel_ms_note"] = "time per step of the whole rank (all sub-batches), not one launch's duration"
        if args.dry_run:
            out["dry_run"] = True
        if args.oversubscribe:
            out["oversubscribed"] = True
        if world == 1 and not args.no_cpu_baseline and not args.dry_run:
            try:
                out["cpu_baseline"] = cpu_baseline(args.env_id)
            except Exception as e:      # noqa: BLE001  (the reported baseline must not cost the line its GPU numbers)
                out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port", "sample": f"failed: {type(e).__name__}: {e}"}
            try:      # the reference's own CPU path beside it, where this box has it
                out["cpu_baseline"]["pybullet"] = pybullet_baseline()
            except Exception as e:      # noqa: BLE001
                out["cpu_baseline"]["pybullet"] = {"value": None, "unit": "env-steps/s", "cores": 1, "kind": "reference", "sample": f"failed: {type(e).__name__}: {e}"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
