#!/usr/bin/env python3
"""Run WHERE PYBULLET EXISTS (it does not in the build image): dump what is needed to pin physics parity.

Writes pybullet_walker3d.npz with
  * the multibody Bullet actually builds from walker3d.xml: per link getDynamicsInfo (mass, local inertia diagonal,
    inertial frame), getJointInfo (names, axes, limits, parent frames, damping), getCollisionShapeData;
  * a teacher-forcing trace of N steps: state before (base pose/velocity, q, qd), the 21 torques applied, state after
    one stepSimulation with the reference's parameters (fixedTimeStep 1/60, 4 substeps, 5 iterations, contact ERP 0.9),
    foot contact flags.
The loader side is mocca_envs_amd/pybullet_dump.py (from_pybullet_dump: model blob from this record, no importer
assumptions left) and tests/test_pybullet_trace.py (skipped while the file is absent): it feeds every "before" state
through the f64 oracle and through the HIP stepper and bounds the one-step error against Bullet's "after" by the north
star's 1e-4 -- the number that cannot be produced in the build image.  Copy the file to
tests/golden/pybullet_walker3d.npz.
Usage: python tools/dump_pybullet_trace.py /path/to/mocca_envs/data 1000
"""
import sys

import numpy as np


def main(data_dir, n_steps):
    import pybullet as p
    p.connect(p.DIRECT)
    p.setGravity(0, 0, -9.8)
    p.setDefaultContactERP(0.9)
    p.setPhysicsEngineParameter(fixedTimeStep=1 / 60, numSolverIterations=5, numSubSteps=4)
    plane = p.loadSDF(f"{data_dir}/objects/misc/plane_stadium.sdf")[0]
    p.changeDynamics(plane, -1, lateralFriction=0.8, restitution=0.5)
    flags = p.MJCF_COLORS_FROM_FILE | p.URDF_USE_SELF_COLLISION | p.URDF_USE_SELF_COLLISION_EXCLUDE_ALL_PARENTS
    robot = p.loadMJCF(f"{data_dir}/robots/walker3d.xml", flags=flags)[0]
    nj = p.getNumJoints(robot)
    out = {"n_links": np.array(nj)}
    jinfo = [p.getJointInfo(robot, j) for j in range(nj)]
    out["joint_names"] = np.array([ji[1].decode() for ji in jinfo])
    out["link_names"] = np.array([ji[12].decode() for ji in jinfo])
    out["joint_type"] = np.array([ji[2] for ji in jinfo])
    out["joint_damping"] = np.array([ji[6] for ji in jinfo])
    out["joint_limits"] = np.array([[ji[8], ji[9]] for ji in jinfo])
    out["joint_axis"] = np.array([ji[13] for ji in jinfo])
    out["parent_frame_pos"] = np.array([ji[14] for ji in jinfo])
    out["parent_frame_orn"] = np.array([ji[15] for ji in jinfo])
    out["parent_index"] = np.array([ji[16] for ji in jinfo])
    dyn = [p.getDynamicsInfo(robot, l) for l in range(-1, nj)]
    out["mass"] = np.array([d[0] for d in dyn])
    out["lateral_friction"] = np.array([d[1] for d in dyn])
    out["local_inertia_diag"] = np.array([d[2] for d in dyn])
    out["inertial_pos"] = np.array([d[3] for d in dyn])
    out["inertial_orn"] = np.array([d[4] for d in dyn])
    shapes = []
    for l in range(-1, nj):
        for s in p.getCollisionShapeData(robot, l):
            shapes.append([l, s[2], *s[3], *s[5], *s[6]])
    out["collision_shapes"] = np.array(shapes, dtype=np.float64)
    act = [j for j in range(nj) if not jinfo[j][1].decode().startswith(("jointfix", "ignore"))]
    for j in range(nj):
        p.setJointMotorControl2(robot, j, p.POSITION_CONTROL, positionGain=0.1, velocityGain=0.1, force=0)
    gains = np.array([60, 80, 60, 80, 60, 100, 90, 60, 80, 60, 100, 90, 60, 60, 60, 50, 60, 60, 60, 50, 60], float)
    rng = np.random.default_rng(0)
    feet = [list(out["link_names"]).index(n) for n in ("right_foot", "left_foot")]

    def snap():
        pos, orn = p.getBasePositionAndOrientation(robot)
        lin, ang = p.getBaseVelocity(robot)
        js = p.getJointStates(robot, act)
        return np.concatenate([pos, orn, lin, ang, [s[0] for s in js], [s[1] for s in js]])

    p.resetBasePositionAndOrientation(robot, [0, 0, 1.32], [0, 0, 0, 1])
    before, after, torques, contacts, feet_pos = [], [], [], [], []
    for t in range(n_steps):
        a = rng.uniform(-1, 1, 21)
        before.append(snap())
        p.setJointMotorControlArray(robot, act, p.TORQUE_CONTROL, forces=list(gains * a))
        p.stepSimulation()
        after.append(snap()); torques.append(gains * a)
        contacts.append([int(any(c[2] == plane for c in p.getContactPoints(bodyA=robot, linkIndexA=f))) for f in feet])
        feet_pos.append([p.getLinkState(robot, f)[0] for f in feet])
        if after[-1][2] < 0.5:  # fallen: restart from the initial pose
            p.resetBasePositionAndOrientation(robot, [0, 0, 1.32], [0, 0, 0, 1])
            p.resetBaseVelocity(robot, [0, 0, 0], [0, 0, 0])
            for j in act:
                p.resetJointState(robot, j, 0.0, 0.0)
    out.update(before=np.array(before), after=np.array(after), torques=np.array(torques), feet_contact=np.array(contacts),
               feet_pos=np.array(feet_pos))
    np.savez_compressed("pybullet_walker3d.npz", **out)
    print("wrote pybullet_walker3d.npz")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1000)
