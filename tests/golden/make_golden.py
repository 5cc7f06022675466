#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference's own Python (build container only).

The reference (/root/reference/mocca_envs) is pure Python over `pybullet` + `gym`, neither of
which is installed here.  Everything *except* rigid-body physics can still
be executed: this script puts a throw-away stub `gym` and a scripted fake `pybullet` client in
front of the reference's real classes (`Walker3D`, `Walker3DCustomEnv`, `Walker3DStepperEnv`) and
records what they compute -- observation packing, normalisation, reward terms, termination,
target / stepping-stone bookkeeping, terrain generation, reset pose logic, mirror indices, torques.

Inputs and expected outputs are written to tests/golden/*.npz (data only).  The reference's source
never enters the repository.  Re-run:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import math
import os
import sys
import tempfile
import types
import xml.etree.ElementTree as ET

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)


# ----------------------------------------------------------------------------
# stub gym + pybullet modules
# ----------------------------------------------------------------------------
class TapeRNG:
    """Stands in for numpy RandomState; every draw pops uniforms from a recorded tape."""

    def __init__(self, seed=0):
        self.tape = np.random.default_rng(1234 + (seed or 0)).random(200000)
        self.pos = 0

    def _pop(self, n=None):
        if n is None:
            v = self.tape[self.pos]
            self.pos += 1
            return float(v)
        v = self.tape[self.pos:self.pos + n].copy()
        self.pos += n
        return v

    def rand(self):
        return self._pop()

    def uniform(self, low=0.0, high=1.0, size=None):
        if size is None:
            return low + (high - low) * self._pop()
        return low + (high - low) * self._pop(int(size))

    def choice(self, seq):
        assert len(seq) == 2
        return seq[0] if self._pop() < 0.5 else seq[1]


def install_stubs():
    gym = types.ModuleType("gym")

    class Env:
        metadata = {}

    class Box:
        def __init__(self, low, high, dtype=np.float32):
            self.low, self.high, self.dtype = np.asarray(low), np.asarray(high), dtype
            self.shape = self.low.shape

    spaces = types.ModuleType("gym.spaces")
    spaces.Box = Box
    utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")
    seeding.np_random = lambda seed=None: (TapeRNG(seed), seed)
    utils.seeding = seeding
    envs = types.ModuleType("gym.envs")
    registration = types.ModuleType("gym.envs.registration")

    class _Reg:
        env_specs = {}

    registration.registry = _Reg()
    registered = {}

    def register(id, **kw):
        registered[id] = kw
        _Reg.env_specs[id] = kw

    registration.register = register
    envs.registration = registration
    gym.Env, gym.spaces, gym.utils, gym.envs = Env, spaces, utils, envs
    for n, mod in [("gym", gym), ("gym.spaces", spaces), ("gym.utils", utils), ("gym.utils.seeding", seeding),
                   ("gym.envs", envs), ("gym.envs.registration", registration)]:
        sys.modules[n] = mod

    pb = types.ModuleType("pybullet")
    pb.GUI, pb.DIRECT, pb.SHARED_MEMORY = 1, 2, 3
    pb.error = RuntimeError
    pb.POSITION_CONTROL, pb.VELOCITY_CONTROL, pb.TORQUE_CONTROL = 2, 0, 1

    def getEulerFromQuaternion(q):
        # standard ZYX extraction; Bullet's singularity clamp is [UNVERIFIED-BULLET]
        x, y, z, w = q
        sarg = -2.0 * (x * z - w * y)
        if sarg <= -0.99999:
            return (0.0, -0.5 * math.pi, 2 * math.atan2(x, -y))
        if sarg >= 0.99999:
            return (0.0, 0.5 * math.pi, 2 * math.atan2(-x, y))
        return (math.atan2(2 * (y * z + w * x), w * w - x * x - y * y + z * z), math.asin(sarg),
                math.atan2(2 * (x * y + w * z), w * w + x * x - y * y - z * z))

    pb.getEulerFromQuaternion = getEulerFromQuaternion
    sys.modules["pybullet"] = pb
    torch_stub = None
    return registered


def quat_from_euler(r, p, y):
    cr, sr, cp, sp, cy, sy = math.cos(r / 2), math.sin(r / 2), math.cos(p / 2), math.sin(p / 2), math.cos(y / 2), math.sin(y / 2)
    return (sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy)


# ----------------------------------------------------------------------------
# scripted fake Bullet client
# ----------------------------------------------------------------------------
class FakeBullet:
    """Implements exactly the pybullet calls of SURVEY.md section 2.3 over a scripted state."""

    POSITION_CONTROL, VELOCITY_CONTROL, TORQUE_CONTROL = 2, 0, 1
    COV_ENABLE_RENDERING = COV_ENABLE_GUI = COV_ENABLE_KEYBOARD_SHORTCUTS = 0
    COV_ENABLE_SEGMENTATION_MARK_PREVIEW = COV_ENABLE_DEPTH_BUFFER_PREVIEW = COV_ENABLE_RGB_BUFFER_PREVIEW = 0
    MJCF_COLORS_FROM_FILE, URDF_USE_SELF_COLLISION, URDF_USE_SELF_COLLISION_EXCLUDE_ALL_PARENTS = 512, 8, 16
    ROBOT, PLANE = 1, 0

    def __init__(self, *a, **k):
        self._client = 0
        self.joints = []  # (joint_name, link_name, lo, hi)
        self._parse_walker(os.path.join(REF, "mocca_envs", "data", "robots", "walker3d.xml"))
        n = len(self.joints)
        self.q = np.zeros(n)
        self.qd = np.zeros(n)
        self.base_pos = np.array([0.0, 0.0, 1.32])
        self.base_quat = np.array([0.0, 0.0, 0.0, 1.0])
        self.base_vel = np.zeros(3)
        self.link_pos = {}
        self.contacts = {}  # link index -> list of (bodyB, linkB)
        self.torques = None
        self.next_body = 10
        self.plank_pose = {}
        self.on_step = None
        self.physics = {}
        self.dynamics = []

    def _parse_walker(self, path):
        root = ET.parse(path).getroot()
        nfix = [0]

        def rec(body):
            js = body.findall("joint")
            for k, j in enumerate(js):
                lo, hi = [float(v) * math.pi / 180 for v in j.get("range").split()]
                link = body.get("name") if k == len(js) - 1 else "link_dummy_%s" % j.get("name")
                self.joints.append((j.get("name"), link, lo, hi))
            if not js and body.get("name") != "walker3d":
                self.joints.append(("jointfix_%d" % nfix[0], body.get("name"), 0.0, -1.0))
                nfix[0] += 1
            for ch in body.findall("body"):
                rec(ch)

        rec(root.find("worldbody").find("body"))

    # ---- setup calls
    def configureDebugVisualizer(self, *a, **k): pass
    def setGravity(self, *a): self.physics["gravity"] = a
    def setDefaultContactERP(self, v): self.physics["erp"] = v
    def setPhysicsEngineParameter(self, **k): self.physics.update(k)
    def loadSDF(self, f): return (self.PLANE,)
    def changeDynamics(self, *a, **k): self.dynamics.append((a, k))
    def loadMJCF(self, path, flags=0): return (self.ROBOT,)
    def saveState(self): return 0
    def disconnect(self): pass

    def loadURDF(self, f, basePosition=None, baseOrientation=None, useFixedBase=False, globalScaling=1.0):
        bid = self.next_body
        self.next_body += 1
        # plank_large.urdf:8 inertial origin z=-0.275 scaled
        self.plank_pose[bid] = (np.array([0, 0, -0.275 * globalScaling]), np.array([0, 0, 0, 1.0]))
        return bid

    def getNumJoints(self, body): return len(self.joints) if body == self.ROBOT else 1
    def setJointMotorControl2(self, *a, **k): pass

    def getJointInfo(self, body, j):
        name, link, lo, hi = self.joints[j]
        info = [None] * 17
        info[0], info[1], info[8], info[9], info[12] = j, name.encode(), lo, hi, link.encode()
        return tuple(info)

    # ---- state queries
    def getBasePositionAndOrientation(self, body):
        if body == self.ROBOT:
            return tuple(self.base_pos), tuple(self.base_quat)
        p, q = self.plank_pose[body]
        return tuple(p), tuple(q)

    def getBaseVelocity(self, body): return tuple(self.base_vel), (0.0, 0.0, 0.0)

    def getLinkState(self, body, link, computeLinkVelocity=0):
        p = self.link_pos.get(link, np.zeros(3))
        return (tuple(p), (0, 0, 0, 1), None, None, None, None)

    def getJointStates(self, body, ids): return [(self.q[j], self.qd[j], (0,) * 6, 0.0) for j in ids]

    def getContactPoints(self, bodyA=None, linkIndexA=None):
        return [(0, bodyA, bB, linkIndexA, lB) for (bB, lB) in self.contacts.get(linkIndexA, [])]

    def getQuaternionFromEuler(self, e): return quat_from_euler(*e)

    # ---- writes
    def resetJointState(self, body, j, targetValue=0.0, targetVelocity=0.0):
        self.q[j], self.qd[j] = targetValue, targetVelocity

    def setJointMotorControlArray(self, bodyIndex=None, jointIndices=None, controlMode=None, forces=None, **k):
        if controlMode == self.TORQUE_CONTROL:
            self.torques = np.array(forces, dtype=np.float64)

    def resetBasePositionAndOrientation(self, body, posObj=None, ornObj=None, *a):
        if posObj is None and a:
            posObj, ornObj = a[0], a[1]
        if body == self.ROBOT:
            self.base_pos, self.base_quat = np.array(posObj, float), np.array(ornObj, float)
        else:
            self.plank_pose[body] = (np.array(posObj, float), np.array(ornObj, float))

    def resetBaseVelocity(self, body, lin, ang): self.base_vel = np.array(lin, float)

    def stepSimulation(self):
        if self.on_step:
            self.on_step()


def make_env(cls_name, **kw):
    import mocca_envs.env_base as env_base
    import mocca_envs.env_locomotion as loco
    holder = {}

    def factory(*a, **k):
        holder["p"] = FakeBullet()
        return holder["p"]

    env_base.BulletClient = factory
    env = getattr(loco, cls_name)(**kw)
    return env, holder["p"]


# ----------------------------------------------------------------------------
def main():
    registered = install_stubs()
    sys.path.insert(0, REF)
    sys.modules.setdefault("torch", types.ModuleType("torch"))  # env_locomotion.py:6 imports it, Walker3D envs never use it
    import scipy.ndimage  # bullet_objects.py:6 uses the deprecated scipy.ndimage.filters path
    if "scipy.ndimage.filters" not in sys.modules:
        f = types.ModuleType("scipy.ndimage.filters")
        f.gaussian_filter = scipy.ndimage.gaussian_filter
        sys.modules["scipy.ndimage.filters"] = f
    import mocca_envs  # noqa: registers ids

    from mocca_envs_amd import model as M
    from oracle.oracle import Oracle

    out = {}
    # ---------------- registration surface (__init__.py:18-116)
    ids = sorted(registered)
    out["registered_ids"] = np.array(ids)
    out["registered_max_steps"] = np.array([registered[i].get("max_episode_steps", -1) for i in ids])
    out["registered_entry"] = np.array([registered[i]["entry_point"] for i in ids])

    # ---------------- Walker3DCustomEnv
    env, p = make_env("Walker3DCustomEnv")
    rob = env.robot
    nj = 21
    out["joint_names"] = np.array([j.joint_name for j in rob.ordered_joints])
    out["joint_lo"] = np.array([j.lowerLimit for j in rob.ordered_joints])
    out["joint_hi"] = np.array([j.upperLimit for j in rob.ordered_joints])
    out["gains"] = np.array(rob.ordered_joint_base_gains, dtype=np.float64)
    out["running_start"] = rob.base_joint_angles.copy()
    out["base_position"] = np.array(rob.base_position, dtype=np.float64)
    out["physics_fixedTimeStep"] = np.array(p.physics["fixedTimeStep"])
    out["physics_numSubSteps"] = np.array(p.physics["numSubSteps"])
    out["physics_numSolverIterations"] = np.array(p.physics["numSolverIterations"])
    out["physics_erp"] = np.array(p.physics["erp"])
    out["physics_gravity"] = np.array(p.physics["gravity"])
    out["scene_dt"] = np.array(env.scene.dt)
    out["obs_dim_custom"] = np.array(env.observation_space.shape[0])
    out["act_dim"] = np.array(env.action_space.shape[0])
    for k, v in zip(["neg_obs", "right_obs", "left_obs", "neg_act", "right_act", "left_act"], env.get_mirror_indices()):
        out["mirror_custom_" + k] = np.asarray(v, dtype=np.int64)

    # to_normalized / to_radians (robots.py:122-132)
    rng = np.random.default_rng(7)
    ang = rng.uniform(-2.5, 2.5, (16, nj))
    out["norm_in"] = ang
    out["norm_out"] = np.stack([rob.to_normalized(a) for a in ang])
    out["rad_out"] = np.stack([rob.to_radians(a) for a in rng.uniform(-1, 1, (16, nj))])
    out["rad_in"] = np.random.default_rng(7).uniform(-2.5, 2.5, (16, nj)) * 0  # placeholder keeps key order stable
    rr = np.random.default_rng(8).uniform(-1, 1, (16, nj))
    out["rad_in"] = rr
    out["rad_out"] = np.stack([rob.to_radians(a) for a in rr])

    # apply_action torques (robots.py:31-40)
    acts = rng.uniform(-2, 2, (8, nj))
    tq = []
    for g in (1.0, 1.2):
        rob.applied_gain = g
        for a in acts:
            rob.apply_action(a)
            tq.append(p.torques.copy())
    rob.applied_gain = 1.0
    out["torque_act"] = acts
    out["torque_out"] = np.array(tq).reshape(2, 8, nj)

    # oracle twin used only to produce kinematically consistent foot positions for the script
    mdl = M.compile_walker3d(M.TASK_WALKER3D_CUSTOM)
    orc = Oracle(mdl.to_bytes(), M.TASK_WALKER3D_CUSTOM, 1, "f64")
    foot_links = [rob.parts[f].bodyPartIndex for f in rob.foot_names]
    joint_ids = rob.ordered_joint_ids
    lo, hi = out["joint_lo"], out["joint_hi"]

    def push_state(p, st, touch, env=None, target=None, plank_ids=None):
        """write a 55-float dynamic state + contact flags into the fake client"""
        full = np.zeros((1, orc.state_dim))
        full[0, :55] = st
        orc.set_state(full)
        fr = orc.link_frames(0, mdl.n_bodies)
        p.base_pos, p.base_quat, p.base_vel = st[0:3].copy(), st[3:7].copy(), st[7:10].copy()
        for k, jid in enumerate(joint_ids):
            p.q[jid], p.qd[jid] = st[13 + k], st[13 + nj + k]
        for k, fl in enumerate(foot_links):
            p.link_pos[fl] = fr[mdl.foot_body[k], 12:15].copy()
        p.contacts = {}
        for k, fl in enumerate(foot_links):
            lst = []
            if touch[k]:
                if plank_ids is None:
                    lst.append((FakeBullet.PLANE, -1))
                else:
                    if target is not None and target[k]:
                        lst.append((plank_ids[env.next_step_index % 3], 0))
                    else:
                        lst.append((plank_ids[(env.next_step_index + 1) % 3], -1))
            p.contacts[fl] = lst

    def random_state(rng, pos, yaw, fall=False, wild=False, ground=0.0):
        st = np.zeros(55)
        st[0:3] = pos
        roll = rng.normal(0, 0.45 if wild else 0.15)
        pitch = rng.normal(0.1, 0.5 if wild else 0.15)
        st[3:7] = quat_from_euler(roll, pitch, yaw)
        st[7:10] = rng.normal(0, 1.0, 3)
        st[10:13] = rng.normal(0, 1.0, 3)
        q = lo + (hi - lo) * rng.uniform(-0.02, 1.02, nj)  # sometimes past the limits -> joints_at_limit
        st[13:13 + nj] = q
        st[13 + nj:13 + 2 * nj] = rng.normal(0, 40.0 if wild else 3.0, nj)  # wild: |0.1 qd| > 5 exercises the clip
        # height = base z - lowest foot z depends only on the leg pose: keep the legs fairly straight
        # unless this frame is meant to terminate the episode (hip_y = joints 5/10, knee = 6/11)
        if not fall:
            st[13 + 1] = np.deg2rad(rng.uniform(-25, 16))  # abdomen_y: folding the torso also lowers it
        for hip, knee in ((5, 6), (10, 11)):
            if fall:
                st[13 + hip], st[13 + knee] = np.deg2rad(-95.0), np.deg2rad(-140.0)
            else:
                st[13 + hip] = np.deg2rad(rng.uniform(-30, 21))
                st[13 + knee] = np.deg2rad(rng.uniform(-40, 1))
        st[2] = ground + rng.uniform(1.0, 1.3)
        return st

    # ---- scripted episodes, Custom env
    episodes = []
    for ep in range(4):
        env.seed(ep)
        env.robot.np_random = env.np_random  # the reference leaves this stale (env_base.py:93); be explicit
        tape = env.np_random.tape.copy()
        if ep == 3:
            env.eval_mode = True
        obs0 = env.reset()
        rec = dict(tape=tape[:640], eval_mode=int(env.eval_mode), reset_obs=obs0,
                   reset_q=np.array([p.q[j] for j in joint_ids]), reset_mirrored=int(rob.mirrored),
                   reset_walk_target=env.walk_target.copy(), reset_stop_frames=float(env.stop_frames),
                   reset_dist=float(env.dist), reset_angle=float(env.angle))
        rng = np.random.default_rng(100 + ep)
        T = 260 if ep < 3 else 40
        states, touches, actions, obs_l, rew_l, done_l, wt_l, cc_l, terms = [], [], [], [], [], [], [], [], []
        pos = np.array([0.0, 0.0, 1.25])
        yaw = 0.0
        for t in range(T):
            # walk toward the current target at 0.08 m/step, hover inside the 0.15 m radius when there
            d = env.walk_target[:2] - pos[:2]
            dist = np.linalg.norm(d)
            if dist > 0.1:
                pos[:2] += d / dist * min(0.08, dist)
            else:
                pos[:2] += rng.normal(0, 0.01, 2)
            yaw += rng.normal(0, 0.05)
            fall = (ep == 1 and t >= T - 3) or (ep == 0 and t == 150)
            st = random_state(rng, pos, yaw, fall=fall, wild=(ep == 2 and t % 7 == 0))
            touch = (rng.random(2) < 0.6).astype(np.int32)
            a = rng.uniform(-1.5, 1.5, nj)
            if ep == 2 and t == 200:
                st[7] = np.nan  # non-finite observation -> done (env_locomotion.py:205-207)
            p.on_step = (lambda st=st, touch=touch: push_state(p, st, touch))
            o, r, dn, info = env.step(a)
            states.append(st); touches.append(touch); actions.append(a)
            obs_l.append(o); rew_l.append(r); done_l.append(dn)
            wt_l.append(env.walk_target.copy()); cc_l.append(env.close_count)
            terms.append([env.progress, env.target_bonus, env.energy_penalty, env.tall_bonus, env.posture_penalty,
                          env.joints_penalty, env.linear_potential, env.distance_to_target, env.angle_to_target])
        rec.update(states=np.array(states), touch=np.array(touches), actions=np.array(actions), obs=np.array(obs_l),
                   rew=np.array(rew_l), done=np.array(done_l).astype(np.int32), walk_target=np.array(wt_l),
                   close_count=np.array(cc_l), terms=np.array(terms))
        episodes.append(rec)
    for i, rec in enumerate(episodes):
        for k, v in rec.items():
            out[f"custom_ep{i}_{k}"] = np.asarray(v)
    out["custom_n_episodes"] = np.array(len(episodes))

    # ---------------- Walker3DStepperEnv
    env, p = make_env("Walker3DStepperEnv")
    rob = env.robot
    plank_ids = [s.id for s in env.steps]
    out["obs_dim_stepper"] = np.array(env.observation_space.shape[0])
    for k, v in zip(["neg_obs", "right_obs", "left_obs", "neg_act", "right_act", "left_act"], env.get_mirror_indices()):
        out["mirror_stepper_" + k] = np.asarray(v, dtype=np.int64)
    out["stepper_init_position"] = np.array(env.robot_init_position, dtype=np.float64)
    out["plank_pos_offset"] = np.array(env.steps[0]._pos_offset)
    out["terminal_height_curriculum"] = env.terminal_height_curriculum.copy()
    out["applied_gain_curriculum"] = env.applied_gain_curriculum.copy()

    # generate_step_placements for several curricula (env_locomotion.py:395-441)
    for cur in (0, 5, 9):
        env.seed(50 + cur)
        env.curriculum = cur
        tape = env.np_random.tape[:100].copy()
        out[f"terrain_c{cur}_tape"] = tape
        out[f"terrain_c{cur}_table"] = env.generate_step_placements()

    mdl_s = M.compile_walker3d(M.TASK_WALKER3D_STEPPER)
    orc = Oracle(mdl_s.to_bytes(), M.TASK_WALKER3D_STEPPER, 1, "f64")
    mdl = mdl_s
    sepisodes = []
    for ep, cur in enumerate((0, 9, 4)):
        env.seed(200 + ep)
        env.robot.np_random = env.np_random
        env.curriculum = cur
        tape = env.np_random.tape.copy()
        p.contacts = {}
        obs0 = env.reset()
        rec = dict(tape=tape[:640], curriculum=cur, reset_obs=obs0, reset_q=np.array([p.q[j] for j in joint_ids]),
                   reset_mirrored=int(rob.mirrored), terrain=env.terrain_info.copy(),
                   applied_gain=float(rob.applied_gain), reset_base=np.array(p.base_pos))
        rng = np.random.default_rng(300 + ep)
        T = 700 if ep == 0 else 420
        states, touches, targets, actions, obs_l, rew_l, done_l, info_l, nsi_l, plank_l, terms = ([] for _ in range(11))
        pos = np.array([0.3, 0.0, 1.25])
        yaw = 0.0
        hold = 0
        for t in range(T):
            tgt = env.terrain_info[env.next_step_index]
            d = tgt[:2] - pos[:2]
            dist = np.linalg.norm(d)
            if dist > 0.05:
                pos[:2] += d / dist * min(0.07, dist)
            yaw += rng.normal(0, 0.04)
            fall = (ep == 2 and t >= T - 2) or (ep == 1 and t == 300)
            st = random_state(rng, pos, yaw, fall=fall, wild=(t % 11 == 0), ground=tgt[2])
            touch = (rng.random(2) < 0.7).astype(np.int32)
            target = np.zeros(2, np.int32)
            if dist < 0.3:
                hold += 1
                # touch the target cover for a few consecutive frames, with gaps, to walk the state machine
                if hold % 9 in (3, 4, 5) or env.stop_on_next_step:
                    k = int(rng.integers(0, 2))
                    target[k] = 1
                    touch[k] = 1
            else:
                hold = 0
            a = rng.uniform(-1.5, 1.5, nj)
            p.on_step = (lambda st=st, touch=touch, target=target: push_state(p, st, touch, env, target, plank_ids))
            o, r, dn, info = env.step(a)
            states.append(st); touches.append(touch); targets.append(target); actions.append(a)
            obs_l.append(o); rew_l.append(r); done_l.append(dn); info_l.append(info.get("steps_reached", -1))
            nsi_l.append(env.next_step_index)
            plank_l.append([p.plank_pose[b][0] for b in plank_ids])
            terms.append([env.progress, env.energy_penalty, env.step_bonus, env.target_bonus, env.tall_bonus,
                          env.posture_penalty, env.joints_penalty, env.target_reached_count, int(env.stop_on_next_step)])
        rec.update(states=np.array(states), touch=np.array(touches), target=np.array(targets), actions=np.array(actions),
                   obs=np.array(obs_l), rew=np.array(rew_l), done=np.array(done_l).astype(np.int32),
                   info=np.array(info_l), next_step_index=np.array(nsi_l), plank_pos=np.array(plank_l),
                   terms=np.array(terms))
        sepisodes.append(rec)
    for i, rec in enumerate(sepisodes):
        for k, v in rec.items():
            out[f"stepper_ep{i}_{k}"] = np.asarray(v)
    out["stepper_n_episodes"] = np.array(len(sepisodes))

    np.savez_compressed(os.path.join(HERE, "walker3d_reference.npz"), **out)
    print("wrote", os.path.join(HERE, "walker3d_reference.npz"), len(out), "arrays")


if __name__ == "__main__":
    main()
