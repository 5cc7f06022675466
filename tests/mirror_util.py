"""Left/right mirror maps of the dynamic state, the action, the observation and the model blob (test infrastructure).

Two different statements are tested with them (tests/test_mirror_oracle.py on the CPU oracle, tests/test_gpu_mirror.py on HIP):

 * the REFERENCE'S claim -- swapping the index sets of `get_mirror_indices()` / `_right_joint_indices`, `_left_joint_indices`,
   `_negation_joint_indices` is a symmetry of the robot (/root/reference/mocca_envs/robots.py:182-188,282-290,
   env_locomotion.py:224-282; SymmetricRL trains on it): `IndexMirror`, built from the blob's copies of those sets, never by hand;
 * a law of mechanics -- the mirror image of a world evolves like the mirror image of the world's evolution: `reflect_model` builds
   the mirror-image ROBOT (every body seen through its own x-z plane, same joint order, same joint angles), `reflect_state` the
   mirror-image state.  Rows and contacts keep their order, so this holds to rounding in EVERY contact configuration.

Conventions: the world is mirrored in its x-z plane, S = diag(1, -1, 1).  A body-fixed point p becomes S p, a rotation R becomes
S R S (quaternion (x, y, z, w) -> (-x, y, -z, w)), an angular velocity w becomes -S w, a hinge axis a becomes -S a with the angle kept.
"""
from __future__ import annotations

import numpy as np

S = np.array([1.0, -1.0, 1.0])


def reflect_state(st: np.ndarray, nj: int, plane: str = "xz") -> np.ndarray:
    """Mirror image of dynamic state records [N][>= 13 + 2 nj] in the world's x-z plane (or, plane="yz", its y-z plane: the crab, whose
    two legs stand at x = +-0.25 in the plane it moves in, crab2d.xml:16-37), joint coordinates untouched."""
    o = np.array(st, copy=True)
    k = 1 if plane == "xz" else 0       # the axis that changes sign
    o[:, k] *= -1                       # position
    o[:, 7 + k] *= -1                   # velocity
    for i in range(3):                  # quaternion vector part and angular velocity: the OTHER two components
        if i != k:
            o[:, 3 + i] *= -1
            o[:, 10 + i] *= -1
    return o


class IndexMirror:
    """The reference's mirror: reflect the base, swap the right / left joint sets, negate the `neg` set.  Built from the model blob's
    mirror_right / mirror_left / mirror_neg (robots.py:282-288 as compile_model copied them); `extra_neg` are joints that the
    reference lists in a per-side negation set of their own (Cassie's hip abduction / yaw, env_cassie.py:554-571 sideneg_*)."""

    def __init__(self, m, right=None, left=None, neg=None, extra_neg=(), plane="xz"):
        nj = m.n_joints
        self.plane = plane
        right = list(m.mirror_right)[: m.n_mirror_side] if right is None else list(right)
        left = list(m.mirror_left)[: m.n_mirror_side] if left is None else list(left)
        neg = list(m.mirror_neg)[: m.n_mirror_neg] if neg is None else list(neg)
        self.nj = nj
        self.perm = np.arange(nj)
        self.perm[right], self.perm[left] = left, right
        self.sign = np.ones(nj)
        self.sign[neg] = -1.0
        self.sign[list(extra_neg)] = -1.0

    def joints(self, q):
        return q[..., self.perm] * self.sign

    def state(self, st):
        nj = self.nj
        o = reflect_state(st, nj, self.plane)
        o[:, 13:13 + nj] = self.joints(st[:, 13:13 + nj])
        o[:, 13 + nj:13 + 2 * nj] = self.joints(st[:, 13 + nj:13 + 2 * nj])
        return o

    def action(self, a):
        return (a[..., self.perm] * self.sign).astype(a.dtype)

    def task(self, tk):
        """Task record (oracle layout, float64): walk target y, the pending re-target angle, and the two feet-contact flags."""
        o = np.array(tk, copy=True)
        if self.plane == "xz":
            o[:, 1] *= -1
            o[:, 15] *= -1
        else:
            o[:, 0] *= -1
        o[:, 12], o[:, 13] = tk[:, 13], tk[:, 12]
        # Stepper: which planks' covers each foot touches, 4 bits per foot (word 26, mocca_device.h cover_targets): the feet exchange their nibbles
        cov = tk[:, 26].astype(np.int64)
        o[:, 26] = ((cov & 0xF) << 4) | ((cov >> 4) & 0xF) | (cov & ~0xFF)
        return o


def obs_mirror(mirror_indices, dim):
    """(perm, sign) of the observation transform SymmetricRL applies with get_mirror_indices()'s first three sets."""
    neg, right, left = (np.asarray(x, dtype=np.int64) for x in mirror_indices[:3])
    perm, sign = np.arange(dim), np.ones(dim)
    perm[right], perm[left] = left, right
    sign[neg] = -1.0
    return perm, sign


_POLAR3 = ("jpos", "com", "g_p1", "g_p2", "foot_point", "cl_point_a", "cl_point_b")


def reflect_model(m, plane: str = "xz"):
    """The mirror-image robot: every body seen through the x-z plane of its own frame (plane="yz": its y-z plane -- the in-plane mirror of
    the planar robots); frames stay right-handed: F' = S F S.  Body-fixed points S p; hinge axes -S a (the same angle then describes the
    mirrored rotation); rotations S R S; the inertia products that involve the mirrored axis change sign.  Joint order, limits, gains,
    masses are untouched: the SAME q, qd and torques drive it."""
    k = 1 if plane == "xz" else 0                       # the axis that changes sign
    r = type(m).from_bytes(m.to_bytes())
    for name in _POLAR3:
        arr = getattr(r, name)
        for i in range(len(arr)):
            arr[i][k] = -arr[i][k]
    rot_idx = [3 * i + j for i in range(3) for j in range(3) if (i == k) != (j == k)]   # S R S: entries with exactly one index on the axis
    prod_idx = {1: (3, 5), 0: (3, 4)}[k]                 # xx yy zz xy xz yz
    for b in range(len(r.jaxis)):
        for i in range(3):
            if i != k:
                r.jaxis[b][i] = -r.jaxis[b][i]
        for x in rot_idx:
            r.jrot[b][x] = -r.jrot[b][x]
        for x in prod_idx:
            r.inertia[b][x] = -r.inertia[b][x]
    r.init_pos[k] = -r.init_pos[k]
    r.init_vel[k] = -r.init_vel[k]
    r.cassie_target[k] = -r.cassie_target[k]
    for i in range(3):
        if i != k:
            r.init_quat[i] = -r.init_quat[i]
    r.finalize_tables()
    return r


def relabel_model(m, mir: IndexMirror):
    """The blob with its right and left bodies exchanged (joint j -> perm[j]) and the `neg` joints' axes flipped, WITHOUT touching
    geometry: together with reflect_model this states the asset's symmetry, reflect(relabel(m)) == m."""
    r = type(m).from_bytes(m.to_bytes())
    nb = m.n_bodies
    bperm = np.concatenate(([0], 1 + mir.perm))          # body of joint j is j + 1
    for name in ("jpos", "jrot", "jaxis", "com", "inertia"):
        src, dst = getattr(m, name), getattr(r, name)
        for b in range(nb):
            for k in range(len(src[b])):
                dst[b][k] = src[int(bperm[b])][k]
    for name in ("jlo", "jhi", "jdamp", "jarm", "gain", "mass", "init_q", "torque_limit"):
        src, dst = getattr(m, name), getattr(r, name)
        for b in range(nb):
            dst[b] = src[int(bperm[b])]
    for j in range(m.n_joints):                          # a negated joint: axis flipped, limits and start angle mirrored
        if mir.sign[j] < 0:
            b = j + 1
            for k in range(3):
                r.jaxis[b][k] = -r.jaxis[b][k]
            r.jlo[b], r.jhi[b] = -r.jhi[b], -r.jlo[b]
            r.init_q[b] = -r.init_q[b]
    return r


def reflect_terrain(ter: np.ndarray) -> np.ndarray:
    """Stepping-stone tables [N][>= 120] of (x, y, z, heading, x_tilt, y_tilt) rows mirrored in the x-z plane."""
    o = np.array(ter, copy=True)
    t = o[:, :120].reshape(len(o), 20, 6)
    t[:, :, 1] *= -1
    t[:, :, 3] *= -1
    t[:, :, 4] *= -1
    return o
