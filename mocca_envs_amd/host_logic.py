"""Host-side (numpy) pieces of the reference's env interface that do not touch physics.

The batched `VecEnv` resets on the device with Philox draws.  The single-env gym facade instead draws
from a host numpy RandomState *with the same calls in the same order as the reference*, so that for a
given seed it starts episodes from the very pose / target / terrain the reference would:
  Walker3DCustomEnv.reset    /root/reference/mocca_envs/env_locomotion.py:79-109
  WalkerBase.reset           /root/reference/mocca_envs/robots.py:179-210
  Walker3DStepperEnv.reset   env_locomotion.py:481-513, generate_step_placements :395-441
Everything here is pinned by tests/test_host_logic.py against vectors captured from the reference.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np

from . import model as M

DEG2RAD = np.pi / 180
N_STEPS = 20            # env_locomotion.py:343
STEP_RADIUS = 0.25      # :344
INIT_STEP_SEPARATION = 0.75  # :346
MAX_CURRICULUM = 9      # :364


def randomize_target(np_random, eval_mode: bool) -> Tuple[float, float, float]:
    """env_locomotion.py:67-74 -> (dist, angle, stop_frames); draw order preserved."""
    if eval_mode:
        dist, angle = 4, 0
    else:
        dist = np_random.uniform(3, 5)
        angle = np_random.uniform(-np.pi / 2, np.pi / 2)
    stop_frames = np_random.choice([30.0, 60.0])
    return dist, angle, float(stop_frames)


def reset_pose(np_random, mdl: M.MoccaModel, random_pose: bool = True) -> Tuple[np.ndarray, bool]:
    """robots.py:179-194: mirror coin flip, +-0.1 rad noise, clip to +-0.95 of the range -> (q[21], mirrored)."""
    nj = mdl.n_joints
    base = np.array([mdl.init_q[b] for b in range(1, nj + 1)], dtype=np.float64)
    right = np.array(list(mdl.mirror_right)[: mdl.n_mirror_side], dtype=np.int64)
    left = np.array(list(mdl.mirror_left)[: mdl.n_mirror_side], dtype=np.int64)
    neg = np.array(list(mdl.mirror_neg)[: mdl.n_mirror_neg], dtype=np.int64)
    mirrored = bool(np_random.rand() < 0.5)
    if mirrored:
        rl, lr = np.concatenate((right, left)), np.concatenate((left, right))
        base[rl] = base[lr]
        base[neg] *= -1
    if random_pose:
        lo, hi = M.joint_limits(mdl)
        weight, bias = (hi - lo).astype(np.float32), lo.astype(np.float32)  # robots.py:125-130 (float32)
        ds = np_random.uniform(low=-0.1, high=0.1, size=nj)
        ps = 2 * (base + ds - bias) / weight - 1
        base = weight * (np.clip(ps, -0.95, 0.95) + 1) / 2 + bias
    return base, mirrored


def generate_step_placements(np_random, curriculum: int) -> np.ndarray:
    """env_locomotion.py:395-441 -> [20, 6] table (x, y, z, phi, x_tilt, y_tilt)."""
    curriculum = min(curriculum, MAX_CURRICULUM)
    ratio = curriculum / MAX_CURRICULUM
    dist_range = np.array([0.65, 1.25])
    dist_upper = np.linspace(*dist_range, MAX_CURRICULUM + 1)
    d_range = np.array([dist_range[0], dist_upper[curriculum]])
    yaw_range = np.array([-20, 20]) * ratio * DEG2RAD
    pitch_range = np.array([-30, 30]) * ratio * DEG2RAD + np.pi / 2
    tilt_range = np.array([-15, 15]) * ratio * DEG2RAD
    n = N_STEPS
    dr = np_random.uniform(*d_range, size=n)
    dphi = np_random.uniform(*yaw_range, size=n)
    dtheta = np_random.uniform(*pitch_range, size=n)
    x_tilt = np_random.uniform(*tilt_range, size=n)
    y_tilt = np_random.uniform(*tilt_range, size=n)
    dr[0], dphi[0], dtheta[0] = 0.0, 0.0, np.pi / 2
    dr[1:3], dphi[1:3], dtheta[1:3] = INIT_STEP_SEPARATION, 0.0, np.pi / 2
    x_tilt[0:3] = 0
    y_tilt[0:3] = 0
    dphi = np.cumsum(dphi)
    dx = dr * np.sin(dtheta) * np.cos(dphi)
    dy = dr * np.sin(dtheta) * np.sin(dphi)
    dz = dr * np.cos(dtheta)
    dx_max = np.maximum(np.abs(dx[2:]), STEP_RADIUS * 2.5)
    dx[2:] = np.sign(dx[2:]) * np.minimum(dx_max, dist_range[1])
    return np.stack((np.cumsum(dx), np.cumsum(dy), np.cumsum(dz), dphi, x_tilt, y_tilt), axis=1)


def applied_gain(curriculum: int) -> float:
    return float(np.linspace(1.0, 1.2, MAX_CURRICULUM + 1)[min(curriculum, MAX_CURRICULUM)])  # :369


def terminal_height(curriculum: int) -> float:
    return float(np.linspace(0.75, 0.45, MAX_CURRICULUM + 1)[min(curriculum, MAX_CURRICULUM)])  # :368


def initial_state(mdl: M.MoccaModel, q: np.ndarray) -> np.ndarray:
    """robots.py:196-204: base at init_pos, identity orientation, at rest -> dynamic state record."""
    st = np.zeros(mdl.state_dim, dtype=np.float32)
    st[0:3] = list(mdl.init_pos)
    st[6] = 1.0
    st[13:13 + mdl.n_joints] = q
    return st


def task_record(**kw) -> np.ndarray:
    """float64 task record in the layout of include/mocca_model.h (see vec_env.task_from_float64)."""
    t = np.zeros(M.TASK_WORDS, dtype=np.float64)
    names = {"walk_target": 0, "linear_potential": 3, "angular_potential": 4, "close_count": 5, "stop_frames": 6,
             "done": 7, "t": 8, "episode": 9, "draw": 10, "mirrored": 11, "feet_contact": 12, "dist": 14, "angle": 15,
             "next_step_index": 16, "target_reached_count": 17, "stop_on_next_step": 18, "set_stop_on_next_step": 19,
             "curriculum": 20, "applied_gain": 21, "prev_body_x": 22}
    t[21] = 1.0
    for k, v in kw.items():
        i = names[k]
        v = np.atleast_1d(np.asarray(v, dtype=np.float64))
        t[i:i + len(v)] = v
    return t


# ---- mirror indices (static; SymmetricRL consumes them) -----------------------------------------------
def mirror_indices(mdl: M.MoccaModel, stepper: bool):
    """env_locomotion.py:224-282 (Custom) / :761-840 (Stepper)."""
    nj, nf = mdl.n_joints, mdl.n_feet
    right = np.array(list(mdl.mirror_right)[: mdl.n_mirror_side], dtype=np.int64)
    left = np.array(list(mdl.mirror_left)[: mdl.n_mirror_side], dtype=np.int64)
    neg = np.array(list(mdl.mirror_neg)[: mdl.n_mirror_neg], dtype=np.int64)
    right_obs = np.concatenate((right + 6, right + 6 + nj, [6 + 2 * nj + 2 * i for i in range(nf // 2)]))
    left_obs = np.concatenate((left + 6, left + 6 + nj, [6 + 2 * nj + 2 * i + 1 for i in range(nf // 2)]))
    robot_neg = np.concatenate(([2, 4], 6 + neg, 6 + neg + nj))
    if not stepper:
        neg_obs = np.concatenate((robot_neg, [6 + 2 * nj + nf]))
    else:
        robot_obs_dim = 6 + 2 * nj + nf
        steps_neg = np.array([(i * 5 + 0, i * 5 + 3) for i in range(3)], dtype=np.int64).flatten()
        neg_obs = np.concatenate((robot_neg, steps_neg + robot_obs_dim))
    return (neg_obs.astype(np.int64), right_obs.astype(np.int64), left_obs.astype(np.int64), neg.copy(), right.copy(), left.copy())
