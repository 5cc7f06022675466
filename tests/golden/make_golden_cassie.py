#!/usr/bin/env python3
"""Golden vectors for CassieEnv's task logic, captured from the reference's own code (build container only).

`mocca_envs/env_cassie.py` cannot be imported or constructed as shipped (missing `loadstep.py`, missing imports,
constructor argument mismatch: SURVEY.md section 0.5).  This script supplies exactly what is missing -- a stub
`loadstep` module, the two names the file forgot to import, and an instance built without the broken __init__ --
and then runs the reference's real `Cassie.calc_state`, `CassieEnv.reset/step/pd_control/get_obs/compute_rewards`
against a fake pybullet client whose `stepSimulation` is this project's f64 CPU oracle.  The physics is therefore the
oracle's on both sides; what the vectors pin is everything else: the 50-iteration PD loop, the joint-speed filter,
torque clipping, residual targets, the observation layout and the reward / termination rule.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

from make_golden import install_stubs  # noqa: E402


class FakeBulletCassie:
    POSITION_CONTROL, VELOCITY_CONTROL, TORQUE_CONTROL = 2, 0, 1
    URDF_USE_SELF_COLLISION, URDF_USE_SELF_COLLISION_EXCLUDE_ALL_PARENTS, URDF_USE_INERTIA_FROM_FILE = 8, 16, 2
    JOINT_POINT2POINT = 5
    ROBOT = 1

    def __init__(self, model, oracle):
        from mocca_envs_amd import cassie_table as CT
        self.m, self.o = model, oracle
        self._client = 0
        kids = {}
        for j in CT.JOINTS:
            kids.setdefault(j["parent"], []).append(j)
        self.joints = []  # Bullet link order: DFS over the URDF tree

        def rec(link):
            for j in kids.get(link, []):
                self.joints.append(j)
                rec(j["child"])
        rec("pelvis")
        # body index (this project's tree) of every movable joint, same DFS
        self.body_of = {}
        b = 0
        for j in self.joints:
            if j["type"] != "fixed":
                b += 1
                self.body_of[j["name"]] = b
        self.tau = np.zeros(model.n_joints)
        self.created_constraints = []
        self.dynamics = []

    # ---- setup (no-ops that record what the reference asked for)
    def setGravity(self, *a): pass
    def setDefaultContactERP(self, v): pass
    def setPhysicsEngineParameter(self, **k): self.physics = k
    def loadURDF(self, *a, **k): return self.ROBOT
    def getNumJoints(self, body): return len(self.joints)
    def changeDynamics(self, body, link, **k): self.dynamics.append((link, k))
    def createConstraint(self, *a, **k): self.created_constraints.append((a, k)); return len(self.created_constraints)
    def setCollisionFilterGroupMask(self, *a): pass
    def restoreState(self, sid): pass
    def saveState(self): return 0

    def getJointInfo(self, body, j):
        jj = self.joints[j]
        info = [None] * 17
        lo = jj["lower"] if jj["lower"] is not None else 0.0
        hi = jj["upper"] if jj["upper"] is not None else -1.0
        if self.m.planar:                            # cassie_collide_2d.urdf: the four hip roll / yaw limits are +-0.01 rad
            from mocca_envs_amd.model import CASSIE_2D_LIMITS
            lo, hi = CASSIE_2D_LIMITS.get(jj["name"], (lo, hi))
        info[0], info[1], info[8], info[9], info[12] = j, jj["name"].encode(), lo, hi, jj["child"].encode()
        return tuple(info)

    # ---- state
    def _st(self):
        return self.o.get_state()

    def getJointState(self, body, j):
        jj = self.joints[j]
        if jj["type"] == "fixed":
            return (0.0, 0.0, (0,) * 6, 0.0)
        b = self.body_of[jj["name"]]
        st = self._st()[0]
        nj = self.m.n_joints
        return (st[13 + b - 1], st[13 + nj + b - 1], (0,) * 6, 0.0)

    def resetJointState(self, body, j, targetValue=0.0, targetVelocity=0.0):
        jj = self.joints[j]
        if jj["type"] == "fixed":
            return
        b = self.body_of[jj["name"]]
        st = self._st()
        nj = self.m.n_joints
        st[0, 13 + b - 1], st[0, 13 + nj + b - 1] = targetValue, targetVelocity
        self.o.set_state(st)

    def getBasePositionAndOrientation(self, body):
        st = self._st()[0]
        return tuple(st[0:3]), tuple(st[3:7])

    def resetBasePositionAndOrientation(self, body, posObj=None, ornObj=None):
        st = self._st()
        st[0, 0:3], st[0, 3:7] = posObj, ornObj
        self.o.set_state(st)

    def getBaseVelocity(self, body):
        st = self._st()[0]
        return tuple(st[7:10]), tuple(st[10:13])

    def resetBaseVelocity(self, body, lin, ang):
        st = self._st()
        st[0, 7:10], st[0, 10:13] = lin, ang
        self.o.set_state(st)

    def getLinkState(self, body, link, computeLinkVelocity=0):
        jj = self.joints[link]
        if jj["name"] not in self.body_of:  # fixed link: only queried for BodyPart.initialPosition, never used
            return ((0.0, 0.0, 0.0), (0, 0, 0, 1), None, None, None, None)
        b = self.body_of[jj["name"]]
        fr = self.o.link_frames(0, self.m.n_bodies)
        return (tuple(fr[b, 12:15]), (0, 0, 0, 1), None, None, None, None)

    def setJointMotorControl2(self, bodyIndex=None, jointIndex=None, controlMode=None, force=None, *a, **k):
        if a and bodyIndex is not None and jointIndex is not None and controlMode is None:
            controlMode = a[0]
        if controlMode == self.TORQUE_CONTROL:
            self.tau[self.body_of[self.joints[jointIndex]["name"]] - 1] = force

    def stepSimulation(self):
        self.o.physics_substeps(0, self.tau, 1)
        self.tau[:] = 0.0


def main():
    install_stubs()
    sys.path.insert(0, REF)
    sys.modules.setdefault("torch", types.ModuleType("torch"))
    ls = types.ModuleType("mocca_envs.loadstep")
    ls.CassieTrajectory = type("CassieTrajectory", (), {})
    sys.modules["mocca_envs.loadstep"] = ls
    import scipy.ndimage
    if "scipy.ndimage.filters" not in sys.modules:
        f = types.ModuleType("scipy.ndimage.filters")
        f.gaussian_filter = scipy.ndimage.gaussian_filter
        sys.modules["scipy.ndimage.filters"] = f
    import mocca_envs  # noqa
    import mocca_envs.env_cassie as ec
    from mocca_envs.bullet_utils import BodyPart, Joint, SinglePlayerStadiumScene
    ec.BodyPart, ec.Joint = BodyPart, Joint  # the two names env_cassie.py uses without importing

    from mocca_envs_amd import model as M
    from oracle.oracle import Oracle

    mdl = M.compile_cassie()
    phys = Oracle(mdl.to_bytes(), M.TASK_CASSIE, 1, "f64")
    phys.reset(seed=0)
    p = FakeBulletCassie(mdl, phys)

    robot = ec.Cassie(p)
    robot.initialize()
    out = {
        "ordered_joint_names": np.array([j.joint_name for j in robot.ordered_joints]),
        "torque_limits": np.array([j.torque_limit for j in robot.ordered_joints]),
        "joint_lo": np.array([j.lowerLimit for j in robot.ordered_joints]),
        "joint_hi": np.array([j.upperLimit for j in robot.ordered_joints]),
        "joint_damping": np.array([d[1]["jointDamping"] for d in p.dynamics if "jointDamping" in d[1]]),
        "kp": ec.CassieEnv.kp.copy(), "kd": ec.CassieEnv.kd.copy(), "jvel_alpha": np.array(ec.CassieEnv.jvel_alpha),
        "control_step": np.array(ec.CassieEnv.control_step), "llc_frame_skip": np.array(ec.CassieEnv.llc_frame_skip),
        "n_constraints": np.array(len(p.created_constraints)),
        "constraint_parent_pos": np.array([c[1]["parentFramePosition"] for c in p.created_constraints]),
        "constraint_child_pos": np.array([c[1]["childFramePosition"] for c in p.created_constraints]),
        "base_position": np.array(ec.Cassie.base_position), "base_joint_angles": np.array(ec.Cassie.base_joint_angles),
        "rod_joint_angles": np.array(ec.Cassie.rod_joint_angles),
        "powered": np.array(ec.Cassie.powered_joint_inds), "springs": np.array(ec.Cassie.spring_joint_inds),
    }
    # CassieEnv without its broken constructor (env_cassie.py:342 passes `render` into EnvBase's robot_kwargs slot)
    env = object.__new__(ec.CassieEnv)
    env._p, env.robot, env.state_id = p, robot, 0
    env.is_rendered, env.planar, env.residual_control, env.rsi = False, False, True, True
    env.scene = SinglePlayerStadiumScene(p, gravity=9.8, timestep=0.03 / 50 / 1, frame_skip=1)
    out["scene_fixedTimeStep"] = np.array(p.physics["fixedTimeStep"])

    rng = np.random.default_rng(0)
    for ep in range(2):
        phys.reset(seed=0)
        obs0 = env.reset()
        acts, obs_l, rew_l, done_l = [], [obs0], [], []
        pre_state, pre_jvel, pre_pot = [], [], []   # what a teacher-forced replay (tests/test_gpu_golden.py) restarts every step from
        for t in range(8 if ep == 0 else 30):
            a = (0.25 if ep == 0 else 0.6) * rng.uniform(-1, 1, 10)
            pre_state.append(phys.get_state()[0].copy())
            pre_jvel.append(np.array(env.jvel, dtype=np.float64) * np.ones(14))
            pre_pot.append(float(env.potential))
            o, r, d, info = env.step(a)
            acts.append(a); obs_l.append(o); rew_l.append(r); done_l.append(d)
            if d:
                break
        out[f"ep{ep}_actions"] = np.array(acts)
        out[f"ep{ep}_obs"] = np.array(obs_l)
        out[f"ep{ep}_rew"] = np.array(rew_l)
        out[f"ep{ep}_done"] = np.array(done_l).astype(np.int32)
        out[f"ep{ep}_final_state"] = phys.get_state()[0].copy()
        out[f"ep{ep}_pre_state"] = np.array(pre_state)
        out[f"ep{ep}_pre_jvel"] = np.array(pre_jvel)
        out[f"ep{ep}_pre_potential"] = np.array(pre_pot)
    np.savez_compressed(os.path.join(HERE, "cassie_reference.npz"), **out)
    print("wrote cassie_reference.npz", {k: v.shape for k, v in out.items() if k.startswith("ep")})


if __name__ == "__main__":
    main()
