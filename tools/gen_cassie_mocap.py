#!/usr/bin/env python3
"""Build mocca_envs_amd/data/cassie_mocap.npz from the reference's motion-capture assets (build container only).

The Cassie mocap / phase envs (env_cassie.py:481-660) read their reference motion through `loadstep.CassieTrajectory`, a
module that is NOT in the reference tree.  What the tree does hold is the data that class read:

  data/robots/cassie/mocap/stepdata.bin        1682 x 98 float64: time (1), qpos (35), qvel (32), torque (10), mpos (10),
                                               mvel (10) of one 0.8405 s walking cycle of the MuJoCo Cassie model, every 0.5 ms
  data/robots/cassie/mocap/cassie_step_data.pkl a pickled (1682, 14) float64 Fortran-order ndarray: the cycle's joint angles in
                                               the order of `Cassie.ordered_joints` (left 7, right 7), fitted to the URDF robot

This script turns them into one data-only table for the re-created trajectory class (mocca_envs_amd/trajectory.py):

  time[1682]            stepdata.bin column 0
  joint_angles[1682,14] the pkl array (read from the pickle's byte string WITHOUT unpickling: no code from the file runs)
  joint_speeds[1682,14] stepdata.bin qvel columns of the same 14 hinges (MuJoCo dof order: 6 base, then per leg hip roll / yaw /
                        pitch, 3 achilles dofs, knee, shin, tarsus, heel spring, foot crank, plantar rod, foot).  Check done
                        here: the time derivative of the pkl angles agrees with these columns to ~1 % rms.
  rod_angles[1682,4]    angles of fixed_{right,left}_achilles_rod_joint_{z,y} (the order of CassieMoccaEnv.resetJoints,
                        env_cassie.py:589-599) that close the two four-bar loops for the frame's joint angles: least-squares
                        solution of |pivot(tarsus) - pivot(rod)| on THIS project's Cassie model (the file's own achilles
                        "quaternion" columns are not unit quaternions and cannot be used).

No source text of the reference is copied; the output holds numbers only.
"""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
MOCAP = "/root/reference/mocca_envs/data/robots/cassie/mocap"

QVEL_COLS = [6, 7, 8, 12, 13, 14, 18, 19, 20, 21, 25, 26, 27, 31]   # the 14 ordered hinges within MuJoCo's 32 dofs
QPOS_COLS = [7, 8, 9, 14, 15, 16, 20, 21, 22, 23, 28, 29, 30, 34]   # the same hinges within qpos (35); cross-check only


def read_pkl_array(path, shape):
    """The ndarray's raw bytes sit in one BINBYTES record of the pickle; locate it by its length prefix."""
    data = open(path, "rb").read()
    n = int(np.prod(shape)) * 8
    at = data.find(b"B" + struct.pack("<I", n))
    assert at >= 0, "pickle layout changed"
    return np.frombuffer(data[at + 5: at + 5 + n], dtype="<f8").reshape(shape[::-1]).T.copy()   # Fortran order


def rot(axis, q):
    a = axis / np.linalg.norm(axis)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(q) * K + (1 - np.cos(q)) * (K @ K)


def fk(m, q):
    """Body frames (R, p) relative to the base for joint angles q[body]."""
    nb = m.n_bodies
    R, p = [np.eye(3)] * nb, [np.zeros(3)] * nb
    for b in range(1, nb):
        pa = m.parent[b]
        jr = np.array(m.jrot[b][:]).reshape(3, 3)
        R[b] = R[pa] @ jr @ rot(np.array(m.jaxis[b][:]), q[b])
        p[b] = p[pa] + R[pa] @ np.array(m.jpos[b][:])
    return R, p


def closure_gap(m, q, c):
    R, p = fk(m, q)
    a, b = m.cl_body_a[c], m.cl_body_b[c]
    return (p[a] + R[a] @ np.array(m.cl_point_a[c][:])) - (p[b] + R[b] @ np.array(m.cl_point_b[c][:]))


def main():
    from mocca_envs_amd import model as M
    sd = np.fromfile(os.path.join(MOCAP, "stepdata.bin"), dtype="<f8").reshape(-1, 98)
    time, qpos, qvel = sd[:, 0], sd[:, 1:36], sd[:, 36:68]
    ang = read_pkl_array(os.path.join(MOCAP, "cassie_step_data.pkl"), (len(time), 14))
    spd = qvel[:, QVEL_COLS]
    # sanity: same hinges, same signs
    assert np.abs(ang[:, [0, 6, 7, 13]] - qpos[:, [QPOS_COLS[k] for k in (0, 6, 7, 13)]]).max() < 2e-4
    fd = np.gradient(ang, time, axis=0)
    rel = np.sqrt(((fd - spd) ** 2).mean(0)) / np.sqrt((spd ** 2).mean(0))
    assert rel.max() < 0.1, rel

    m = M.compile_cassie()
    names = {b: None for b in range(m.n_bodies)}
    rod_bodies = []   # right z, right y, left z, left y (env_cassie.py:591-596)
    # body order of the blob: recover names through the joint table used by compile_cassie
    from mocca_envs_amd import cassie_table as CT
    order = []

    def walk(link):
        for j in [j for j in CT.JOINTS if j["parent"] == link]:
            if j["type"] != "fixed":
                order.append(j["name"])
            walk(j["child"])
    walk("pelvis")
    body_of = {n: i + 1 for i, n in enumerate(order)}
    for side in ("right", "left"):
        for ax in ("z", "y"):
            rod_bodies.append(body_of["fixed_%s_achilles_rod_joint_%s" % (side, ax)])
    rods = np.zeros((len(time), 4))
    worst = 0.0
    x = {0: np.array([M.CASSIE_ROD_ANGLES["fixed_right_achilles_rod_joint_z"], M.CASSIE_ROD_ANGLES["fixed_right_achilles_rod_joint_y"]]),
         1: np.array([M.CASSIE_ROD_ANGLES["fixed_left_achilles_rod_joint_z"], M.CASSIE_ROD_ANGLES["fixed_left_achilles_rod_joint_y"]])}
    for f in range(len(time)):
        q = np.zeros(m.n_bodies)
        for k in range(14):
            q[m.ordered_body[k]] = ang[f, k]
        for si, side in enumerate(("right", "left")):
            c = 1 if side == "right" else 0   # closures are compiled left, right
            bz, by = rod_bodies[2 * si], rod_bodies[2 * si + 1]
            xs = x[si].copy()                 # warm start from the previous frame
            for _ in range(12):               # Gauss-Newton on the 3-vector gap, 2 unknowns
                q[bz], q[by] = xs
                g0 = closure_gap(m, q, c)
                J = np.zeros((3, 2))
                for u, bb in enumerate((bz, by)):
                    qq = q.copy(); qq[bb] += 1e-6
                    J[:, u] = (closure_gap(m, qq, c) - g0) / 1e-6
                dx = np.linalg.lstsq(J, -g0, rcond=None)[0]
                xs = xs + dx
                if np.abs(dx).max() < 1e-10:
                    break
            q[bz], q[by] = xs
            worst = max(worst, float(np.linalg.norm(closure_gap(m, q, c))))
            x[si] = xs
            rods[f, 2 * si: 2 * si + 2] = xs
    out = os.path.join(REPO, "mocca_envs_amd", "data")
    os.makedirs(out, exist_ok=True)
    np.savez_compressed(os.path.join(out, "cassie_mocap.npz"), time=time, joint_angles=ang.astype(np.float32),
                        joint_speeds=spd.astype(np.float32), rod_angles=rods.astype(np.float32),
                        rod_bodies=np.array(rod_bodies, dtype=np.int32))
    print("frames", len(time), "period", time[-1], "worst closure residual [m]", worst, "rod bodies", rod_bodies)
    print("rod angle ranges", rods.min(0), rods.max(0))


if __name__ == "__main__":
    main()
