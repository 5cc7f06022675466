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


def _dec(x) -> float:
    """The short decimal an fp32 blob constant renders (0.65, 0.45, 1.2 ...): the reference computes with the decimal."""
    return round(float(x), 6)


def generate_step_placements(np_random, curriculum: int, mdl: M.MoccaModel = None) -> np.ndarray:
    """Stepping-stone table [20, 6] = (x, y, z, heading, x_tilt, y_tilt), the distribution of
    env_locomotion.py:395-441: per step a radial distance, a heading increment, a polar angle and two tilts,
    each uniform in a range that widens with the curriculum; draw order = five blocks of 20.  The ranges are the
    env class's attributes (Walker3DStepperEnv :380-385 / LaikagoStepperEnv :919-922), carried by the model blob."""
    if mdl is None:
        mdl = M.set_stepper_params(M.MoccaModel())
    d0, d1 = _dec(mdl.dist_range[0]), _dec(mdl.dist_range[1])
    yaw, pitch, tilt = _dec(mdl.yaw_range_deg), _dec(mdl.pitch_range_deg), _dec(mdl.tilt_range_deg)
    level = min(int(curriculum), MAX_CURRICULUM)
    frac = level / MAX_CURRICULUM
    half_pi = np.pi / 2
    ranges = (
        (d0, np.linspace(d0, d1, MAX_CURRICULUM + 1)[level]),                   # radial distance [m]
        (-yaw * frac * DEG2RAD, yaw * frac * DEG2RAD),                          # heading increment
        (-pitch * frac * DEG2RAD + half_pi, pitch * frac * DEG2RAD + half_pi),  # polar angle (pi/2 = level ground)
        (-tilt * frac * DEG2RAD, tilt * frac * DEG2RAD),                        # tilt about x
        (-tilt * frac * DEG2RAD, tilt * frac * DEG2RAD),                        # tilt about y
    )
    radial, turn, polar, tilt_x, tilt_y = (np_random.uniform(lo, hi, size=N_STEPS) for lo, hi in ranges)
    # the robot starts on step 0; steps 1 and 2 lie flat, straight ahead, init_step_separation apart
    sep = _dec(mdl.init_step_separation)
    radial[:3] = (0.0, sep, sep)
    turn[:3], polar[:3], tilt_x[:3], tilt_y[:3] = 0.0, half_pi, 0.0, 0.0
    heading = np.cumsum(turn)
    ground = radial * np.sin(polar)
    hop = np.stack((ground * np.cos(heading), ground * np.sin(heading), radial * np.cos(polar)), axis=1)
    # from the third step on keep consecutive planks from overlapping or drifting apart along x
    fwd = hop[2:, 0]
    hop[2:, 0] = np.sign(fwd) * np.clip(np.abs(fwd), 2.5 * _dec(mdl.step_radius), d1)
    return np.column_stack((np.cumsum(hop, axis=0), heading, tilt_x, tilt_y))


def applied_gain(curriculum: int, mdl: M.MoccaModel = None) -> float:
    a, b = (1.0, 1.2) if mdl is None else (_dec(mdl.gain_cur[0]), _dec(mdl.gain_cur[1]))
    return float(np.linspace(a, b, MAX_CURRICULUM + 1)[min(curriculum, MAX_CURRICULUM)])  # :369 / :918


def terminal_height(curriculum: int, mdl: M.MoccaModel = None) -> float:
    a, b = (0.75, 0.45) if mdl is None else (_dec(mdl.term_height_cur[0]), _dec(mdl.term_height_cur[1]))
    return float(np.linspace(a, b, MAX_CURRICULUM + 1)[min(curriculum, MAX_CURRICULUM)])  # :368 / :917


def yaw_from_quat(x: float, y: float, z: float, w: float) -> float:
    """Yaw of getEulerFromQuaternion (robots.py:57), including Bullet's clamp at the pitch = +-90 degree singularity,
    which Child3D's "crawl" pose starts on."""
    sarg = -2.0 * (x * z - w * y)
    if sarg <= -0.99999:
        return float(2 * np.arctan2(x, -y))
    if sarg >= 0.99999:
        return float(2 * np.arctan2(-x, y))
    return float(np.arctan2(2 * (x * y + w * z), w * w + x * x - y * y - z * z))


def initial_state(mdl: M.MoccaModel, q: np.ndarray) -> np.ndarray:
    """robots.py:196-204: base at init_pos / init_quat, at rest -> dynamic state record."""
    st = np.zeros(mdl.state_dim, dtype=np.float32)
    st[0:3] = list(mdl.init_pos)
    st[3:7] = list(mdl.init_quat)
    st[7:10] = list(mdl.init_vel)     # robot_init_velocity (env_locomotion.py:92,493; LaikagoStepperEnv :901)
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
        n_targets = 2 + (mdl.lookbehind or 1)      # lookahead + lookbehind rows of 5 (:806-812)
        steps_neg = np.array([(i * 5 + 0, i * 5 + 3) for i in range(n_targets)], dtype=np.int64).flatten()
        neg_obs = np.concatenate((robot_neg, steps_neg + robot_obs_dim))
    return (neg_obs.astype(np.int64), right_obs.astype(np.int64), left_obs.astype(np.int64), neg.copy(), right.copy(), left.copy())


# ---------------------------------------------------------------------------------------------------------------------
# Height-field terrain of the planner envs (bullet_objects.py:338-441, misc_utils.py:4-58); pinned by tests/test_golden_planner.py
# ---------------------------------------------------------------------------------------------------------------------
HEIGHT_FIELD_FILE = "height_field_map_0.npy"   # Walker3DPlannerEnv.create_terrain, env_locomotion.py:1015-1021
HEIGHT_FIELD_SIZE, HEIGHT_FIELD_SCALE = (128, 128), 4


def height_at(data2d: np.ndarray, scale: float, x: float, y: float) -> float:
    """HeightField.get_height_at (bullet_objects.py:348-353): data2d[int((y + oy) * scale), int((x + ox) * scale)] with
    ox, oy = data_size / scale / 2 -- the grid point below-left of (x, y) on a grid whose point 0 sits at -size / scale / 2
    (half a cell off Bullet's centred grid; the reference uses it for the z of the walk target only)."""
    rows, cols = data2d.shape
    ox, oy = rows / scale / 2, cols / scale / 2
    return float(data2d[int((y + oy) * scale), int((x + ox) * scale)])


def _fade(t):
    return t * t * t * (t * (6 * t - 15) + 10)


def perlin_noise_2d(shape, res, rng) -> np.ndarray:
    """Perlin noise on a shape[0] x shape[1] grid with res[0] x res[1] lattice cells (misc_utils.generate_perlin_noise_2d):
    (res + 1)^2 unit gradients from rng.rand, corner ramps blended with the quintic fade, scaled by sqrt(2).  Same draws, same values."""
    n0, n1 = shape
    r0, r1 = res
    theta = 2 * np.pi * rng.rand(r0 + 1, r1 + 1)
    gx, gy = np.cos(theta), np.sin(theta)
    per0, per1 = n0 // r0, n1 // r1                  # pixels per lattice cell
    i, j = np.arange(n0), np.arange(n1)
    ci, cj = (i // per0)[:, None], (j // per1)[None, :]                      # lattice cell of each pixel
    u = ((i * (r0 / n0)) % 1)[:, None] * np.ones((1, n1))                    # position inside the cell
    v = np.ones((n0, 1)) * ((j * (r1 / n1)) % 1)[None, :]

    def ramp(di, dj):
        return (u - di) * gx[ci + di, cj + dj] + (v - dj) * gy[ci + di, cj + dj]

    fu, fv = _fade(u), _fade(v)
    low = ramp(0, 0) * (1 - fu) + fu * ramp(1, 0)
    high = ramp(0, 1) * (1 - fu) + fu * ramp(1, 1)
    return np.sqrt(2) * ((1 - fv) * low + fv * high)


def fractal_noise_2d(shape, res, octaves: int = 1, persistence: float = 0.5, rng=None) -> np.ndarray:
    """Sum of `octaves` Perlin layers, lattice doubling and amplitude x persistence per layer (misc_utils.generate_fractal_noise_2d)."""
    rng = rng or np.random
    total, freq, amp = np.zeros(shape), 1, 1.0
    for _ in range(octaves):
        total += amp * perlin_noise_2d(shape, (freq * res[0], freq * res[1]), rng)
        freq, amp = 2 * freq, amp * persistence
    return total


def random_height_field(rng, size=HEIGHT_FIELD_SIZE, scale: float = HEIGHT_FIELD_SCALE, n_peaks: int = 256) -> np.ndarray:
    """HeightField.get_random_height_field (bullet_objects.py:395-441), flat array of size[0] * size[1] heights: 256 super-Gaussian
    peaks (height U(0.1, 3), centre uniform on the field, spreads U(1, 16), even exponents 2 .. 10) divided by the scale, plus two
    octaves of fractal noise, minus the mean of the 5 x 5 corner platform, which is flattened first.  Draw order preserved."""
    height = rng.uniform(0.1, 3, size=n_peaks)
    half_x, half_y = size[0] / scale / 2, size[1] / scale / 2
    cx, cy = rng.uniform(-half_x, half_x, size=n_peaks), rng.uniform(-half_y, half_y, size=n_peaks)
    sx, sy = rng.uniform(1, 16, size=n_peaks), rng.uniform(1, 16, size=n_peaks)
    ex, ey = rng.randint(1, 6, size=n_peaks) * 2, rng.randint(1, 6, size=n_peaks) * 2
    gx, gy = np.linspace(-half_x, half_x, size[0]), np.linspace(-half_y, half_y, size[1])
    field = np.zeros(size)
    for k in range(n_peaks):       # one peak at a time keeps the temporary at the size of the field, not 256 x the field
        field += height[k] * np.exp(-(((gx - cx[k]) ** ex[k]) / sx[k])[:, None] - (((gy - cy[k]) ** ey[k]) / sy[k])[None, :])
    flat = field.flatten() / scale + fractal_noise_2d(size, (4, 4), 2, 1, rng).flatten()
    corner = flat.reshape(size)[0:5, 0:5]          # a view: the platform the robot starts on, at the corner it starts in
    level = corner.mean()
    corner[:] = level
    return flat - level
