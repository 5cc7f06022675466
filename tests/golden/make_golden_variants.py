#!/usr/bin/env python3
"""Golden vectors for the robots that share Walker3D's tree: Child3DCustomEnv and MikeStepperEnv
(build container only).  Same method as make_golden.py -- the reference's real classes driven over a stub gym and a
scripted fake pybullet client -- recording what differs from the Walker3D envs: gains, limits, reset pose (joint
angles, base position and orientation), termination height, the changeDynamics call, and a short scripted episode
around the 0.1 m termination height.  Output: tests/golden/variants_reference.npz (data only).
Re-run:  python tests/golden/make_golden_variants.py
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402


class FakeBulletAnyMJCF(G.FakeBullet):
    root_link = None  # name of the link the robot reports as its body (planar robots: "pelvis")

    def loadMJCF(self, path, flags=0):
        self.joints = []
        self.mjcf_path = path
        self.mjcf_flags = flags
        self._parse_any(path)
        self.q, self.qd = np.zeros(len(self.joints)), np.zeros(len(self.joints))
        return (self.ROBOT,)

    def restoreState(self, *a, **k): pass

    def _parse_any(self, path):
        """Links in file order; the root body's "ignore*" joints are real joints (without a range) of the fake client."""
        import math
        import xml.etree.ElementTree as ET
        nfix = [0]

        def rec(body, is_root):
            js = body.findall("joint")
            for k, j in enumerate(js):
                rng = j.get("range")
                lo, hi = [float(v) * math.pi / 180 for v in rng.split()] if rng else (0.0, -1.0)
                link = body.get("name") if k == len(js) - 1 else "link_dummy_%s" % j.get("name")
                self.joints.append((j.get("name"), link, lo, hi))
            if not js and not is_root:
                self.joints.append(("jointfix_%d" % nfix[0], body.get("name"), 0.0, -1.0))
                nfix[0] += 1
            for ch in body.findall("body"):
                rec(ch, False)

        root = ET.parse(path).getroot().find("worldbody").find("body")
        rec(root, not root.findall("joint"))

    def getLinkState(self, body, link, computeLinkVelocity=0):
        if self.root_link is not None and self.joints[link][1] == self.root_link:
            p, q = tuple(self.base_pos), tuple(self.base_quat)
            v, w = tuple(self.base_vel), (0.0, 0.0, 0.0)
        else:
            p, q, v, w = tuple(self.link_pos.get(link, np.zeros(3))), (0, 0, 0, 1), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0)
        return (p, q, None, None, None, None, v, w) if computeLinkVelocity else (p, q, None, None, None, None)


class FakeBulletURDFRobot(FakeBulletAnyMJCF):
    """The robot comes from loadURDF (Laikago, robots.py:584-600): joints in file order with their types and limits."""
    JOINT_REVOLUTE, JOINT_PRISMATIC, JOINT_FIXED = 0, 1, 4
    robot_urdf = None

    def loadURDF(self, f, basePosition=None, baseOrientation=None, useFixedBase=False, globalScaling=1.0, flags=0):
        if not f.endswith(self.robot_urdf):
            return super().loadURDF(f, basePosition, baseOrientation, useFixedBase, globalScaling)
        import xml.etree.ElementTree as ET
        self.mjcf_path, self.mjcf_flags, self.joints, self.jtypes = f, flags, [], []
        for j in ET.parse(f).getroot().findall("joint"):
            lim = j.find("limit")
            lo, hi = (float(lim.get("lower")), float(lim.get("upper"))) if lim is not None else (0.0, -1.0)
            self.joints.append((j.get("name"), j.find("child").get("link"), lo, hi))
            self.jtypes.append(self.JOINT_FIXED if j.get("type") == "fixed" else self.JOINT_REVOLUTE)
        self.q, self.qd = np.zeros(len(self.joints)), np.zeros(len(self.joints))
        return self.ROBOT

    def getJointInfo(self, body, j):
        info = list(super().getJointInfo(body, j))
        if body == self.ROBOT:
            info[2] = self.jtypes[j]
        return tuple(info)

    def getContactPoints(self, bodyA=None, linkIndexA=None):
        if linkIndexA is not None:
            return super().getContactPoints(bodyA, linkIndexA)
        return [(0, bodyA, bB, lA, lB) for lA, lst in self.contacts.items() for (bB, lB) in lst]   # every link of the body


def make_env(cls_name, root_link=None, **kw):
    import mocca_envs.env_base as env_base
    import mocca_envs.env_locomotion as loco
    holder = {}

    def factory(*a, **k):
        holder["p"] = FakeBulletURDFRobot() if kw.get("_urdf") else FakeBulletAnyMJCF()
        holder["p"].root_link = root_link
        holder["p"].robot_urdf = kw.get("_urdf")
        return holder["p"]

    env_base.BulletClient = factory
    return getattr(loco, cls_name)(**{k: v for k, v in kw.items() if not k.startswith("_")}), holder["p"]


def robot_constants(out, tag, env, p):
    rob = env.robot
    out[f"{tag}_mjcf"] = np.array(os.path.basename(p.mjcf_path))
    if hasattr(rob, "ordered_joints"):
        out[f"{tag}_joint_names"] = np.array([j.joint_name for j in rob.ordered_joints])
        out[f"{tag}_joint_lo"] = np.array([j.lowerLimit for j in rob.ordered_joints])
        out[f"{tag}_joint_hi"] = np.array([j.upperLimit for j in rob.ordered_joints])
    else:   # Laikago keeps ids only (robots.py:609-626); limits as its to_radians sees them: theta = -1 / +1
        n = len(rob.ordered_joint_ids)
        out[f"{tag}_joint_names"] = np.array([p.joints[j][0] for j in rob.ordered_joint_ids])
        out[f"{tag}_joint_lo"] = np.asarray(rob.to_radians(-np.ones(n)), dtype=np.float64)
        out[f"{tag}_joint_hi"] = np.asarray(rob.to_radians(np.ones(n)), dtype=np.float64)
    out[f"{tag}_gains"] = np.array(rob.ordered_joint_base_gains, dtype=np.float64)
    out[f"{tag}_base_joint_angles"] = np.array(rob.base_joint_angles, dtype=np.float64)
    out[f"{tag}_base_position"] = np.array(rob.base_position, dtype=np.float64)
    out[f"{tag}_base_orientation"] = np.array(rob.base_orientation, dtype=np.float64)
    out[f"{tag}_obs_dim"] = np.array(env.observation_space.shape[0])
    for k, v in zip(["neg_obs", "right_obs", "left_obs", "neg_act", "right_act", "left_act"], env.get_mirror_indices()):
        out[f"{tag}_mirror_{k}"] = np.asarray(v, dtype=np.int64)
    # changeDynamics calls that name a mass (Mike's waist, robots.py:507-510): (link name, mass)
    masses = [(p.joints[a[1]][1], k["mass"]) for a, k in p.dynamics if "mass" in k and a[0] == p.ROBOT]
    out[f"{tag}_mass_links"] = np.array([m[0] for m in masses])
    out[f"{tag}_mass_values"] = np.array([m[1] for m in masses], dtype=np.float64)


def main():
    G.install_stubs()
    sys.path.insert(0, G.REF)
    sys.modules.setdefault("torch", types.ModuleType("torch"))
    import scipy.ndimage
    if "scipy.ndimage.filters" not in sys.modules:
        f = types.ModuleType("scipy.ndimage.filters")
        f.gaussian_filter = scipy.ndimage.gaussian_filter
        sys.modules["scipy.ndimage.filters"] = f
    import mocca_envs  # noqa

    from mocca_envs_amd import model as M
    from oracle.oracle import Oracle

    out = {}
    nj = 21

    # ---------------- Child3DCustomEnv (env_locomotion.py:317-327)
    env, p = make_env("Child3DCustomEnv")
    rob = env.robot
    robot_constants(out, "child", env, p)
    out["child_termination_height"] = np.array(env.termination_height)
    joint_ids = rob.ordered_joint_ids
    foot_links = [rob.parts[f].bodyPartIndex for f in rob.foot_names]
    mdl = M.compile_child3d()
    orc = Oracle(mdl.to_bytes(), M.TASK_WALKER3D_CUSTOM, 1, "f64")
    lo, hi = out["child_joint_lo"], out["child_joint_hi"]

    def push_state(st, touch):
        full = np.zeros((1, orc.state_dim))
        full[0, :55] = st
        orc.set_state(full)
        fr = orc.link_frames(0, mdl.n_bodies)
        p.base_pos, p.base_quat, p.base_vel = st[0:3].copy(), st[3:7].copy(), st[7:10].copy()
        for k, jid in enumerate(joint_ids):
            p.q[jid], p.qd[jid] = st[13 + k], st[13 + nj + k]
        for k, fl in enumerate(foot_links):
            p.link_pos[fl] = fr[mdl.foot_body[k], 12:15].copy()
        p.contacts = {fl: ([(G.FakeBullet.PLANE, -1)] if touch[k] else []) for k, fl in enumerate(foot_links)}

    for ep in range(2):
        env.seed(40 + ep)
        env.robot.np_random = env.np_random
        tape = env.np_random.tape.copy()
        obs0 = env.reset()
        rec = dict(tape=tape[:640], reset_obs=obs0, reset_q=np.array([p.q[j] for j in joint_ids]),
                   reset_mirrored=int(rob.mirrored), reset_base_pos=np.array(p.base_pos), reset_base_quat=np.array(p.base_quat),
                   reset_walk_target=env.walk_target.copy(), reset_stop_frames=float(env.stop_frames))
        rng = np.random.default_rng(400 + ep)
        states, touches, actions, obs_l, rew_l, done_l, terms = [], [], [], [], [], [], []
        T = 60
        for t in range(T):
            st = np.zeros(55)
            # on all fours: base pitched about 90 degrees, 0.15 .. 0.5 m up; the last frames drop below 0.1 m
            low = t >= T - 2
            st[0:3] = [0.02 * t, rng.normal(0, 0.02), rng.uniform(0.02, 0.06) if low else rng.uniform(0.15, 0.5)]
            st[3:7] = G.quat_from_euler(rng.normal(0, 0.2), np.pi / 2 + rng.normal(0, 0.3), rng.normal(0, 0.3))
            st[7:10] = rng.normal(0, 0.5, 3)
            st[10:13] = rng.normal(0, 0.5, 3)
            st[13:13 + nj] = np.clip(out["child_base_joint_angles"] + rng.normal(0, 0.2, nj), lo, hi)
            if low:   # limbs flat: feet level with the base
                st[13 + 5] = st[13 + 10] = st[13 + 6] = st[13 + 11] = 0.0
            st[13 + nj:13 + 2 * nj] = rng.normal(0, 3.0, nj)
            touch = (rng.random(2) < 0.6).astype(np.int32)
            a = rng.uniform(-1.5, 1.5, nj)
            p.on_step = (lambda st=st, touch=touch: push_state(st, touch))
            o, r, dn, _ = env.step(a)
            states.append(st); touches.append(touch); actions.append(a); obs_l.append(o); rew_l.append(r); done_l.append(dn)
            terms.append([env.progress, env.target_bonus, env.energy_penalty, env.tall_bonus, env.posture_penalty,
                          env.joints_penalty])
        rec.update(states=np.array(states), touch=np.array(touches), actions=np.array(actions), obs=np.array(obs_l),
                   rew=np.array(rew_l), done=np.array(done_l).astype(np.int32), terms=np.array(terms))
        for k, v in rec.items():
            out[f"child_ep{ep}_{k}"] = np.asarray(v)
    out["child_n_episodes"] = np.array(2)

    # ---------------- MikeStepperEnv (env_locomotion.py:843-851)
    env, p = make_env("MikeStepperEnv")
    rob = env.robot
    robot_constants(out, "mike", env, p)
    out["mike_init_position"] = np.array(env.robot_init_position, dtype=np.float64)
    joint_ids = rob.ordered_joint_ids
    for ep, cur in enumerate((0, 9)):
        env.seed(60 + ep)
        env.robot.np_random = env.np_random
        env.curriculum = cur
        tape = env.np_random.tape.copy()
        p.contacts = {}
        obs0 = env.reset()
        for k, v in dict(tape=tape[:640], curriculum=cur, reset_obs=obs0, reset_q=np.array([p.q[j] for j in joint_ids]),
                         reset_mirrored=int(rob.mirrored), terrain=env.terrain_info.copy(),
                         applied_gain=float(rob.applied_gain), reset_base_pos=np.array(p.base_pos),
                         reset_base_quat=np.array(p.base_quat)).items():
            out[f"mike_ep{ep}_{k}"] = np.asarray(v)
        # torques for one action at the reset pose: gains x applied_gain (robots.py:31-40)
        a = np.random.default_rng(70 + ep).uniform(-1.2, 1.2, nj)
        rob.apply_action(a)
        out[f"mike_ep{ep}_torque_act"], out[f"mike_ep{ep}_torque_out"] = a, p.torques.copy()
    out["mike_n_episodes"] = np.array(2)

    # ---------------- Walker2DCustomEnv / Crab2DCustomEnv (env_locomotion.py:285-314)
    for tag, cls, compile_fn in (("walker2d", "Walker2DCustomEnv", M.compile_walker2d), ("crab2d", "Crab2DCustomEnv", M.compile_crab2d)):
        env, p = make_env(cls, root_link="pelvis")
        rob = env.robot
        robot_constants(out, tag, env, p)
        n2 = len(rob.ordered_joints)
        out[f"{tag}_load_flags"] = np.array(p.mjcf_flags)
        out[f"{tag}_self_collision"] = np.array(int(bool(p.mjcf_flags & p.URDF_USE_SELF_COLLISION)))
        out[f"{tag}_init_position"] = np.array(env.robot_init_position, dtype=np.float64)
        out[f"{tag}_termination_height"] = np.array(env.termination_height)
        joint_ids = rob.ordered_joint_ids
        foot_links = [rob.parts[f].bodyPartIndex for f in rob.foot_names]
        mdl = compile_fn()
        orc = Oracle(mdl.to_bytes(), M.TASK_WALKER3D_CUSTOM, 1, "f64")
        lo, hi = out[f"{tag}_joint_lo"], out[f"{tag}_joint_hi"]
        sd = 13 + 2 * n2

        def push_state2(st, touch, p=p, orc=orc, mdl=mdl, joint_ids=joint_ids, foot_links=foot_links, n2=n2, sd=sd):
            full = np.zeros((1, orc.state_dim))
            full[0, :sd] = st
            orc.set_state(full)
            fr = orc.link_frames(0, mdl.n_bodies)
            p.base_pos, p.base_quat, p.base_vel = st[0:3].copy(), st[3:7].copy(), st[7:10].copy()
            for k, jid in enumerate(joint_ids):
                p.q[jid], p.qd[jid] = st[13 + k], st[13 + n2 + k]
            for k, fl in enumerate(foot_links):
                p.link_pos[fl] = fr[mdl.foot_body[k], 12:15].copy()
            p.contacts = {fl: ([(G.FakeBullet.PLANE, -1)] if touch[k] else []) for k, fl in enumerate(foot_links)}

        env.seed(80)
        env.robot.np_random = env.np_random
        tape = env.np_random.tape.copy()
        p.base_pos = np.array(mdl.init_pos, dtype=np.float64)   # the fake's link pose is whatever the script says
        obs0 = env.reset()
        rec = dict(tape=tape[:640], reset_obs=obs0, reset_q=np.array([p.q[j] for j in joint_ids]), reset_mirrored=int(rob.mirrored),
                   reset_walk_target=env.walk_target.copy())
        rng = np.random.default_rng(500)
        states, touches, actions, obs_l, rew_l, done_l, terms = [], [], [], [], [], [], []
        for t in range(40):
            st = np.zeros(sd)
            pitch = rng.normal(0.1, 0.4)
            st[0:3] = [0.03 * t, 0.0, rng.uniform(0.3, 1.2)]      # heights on both sides of 0.7: done must stay False
            st[3:7] = G.quat_from_euler(0.0, pitch, 0.0)
            st[7:10] = [rng.normal(0, 1), 0.0, rng.normal(0, 1)]
            st[11] = rng.normal(0, 1)
            st[13:13 + n2] = lo + (hi - lo) * rng.uniform(-0.02, 1.02, n2)
            st[13 + n2:sd] = rng.normal(0, 3.0, n2)
            touch = (rng.random(2) < 0.6).astype(np.int32)
            a = rng.uniform(-1.5, 1.5, n2)
            p.on_step = (lambda st=st, touch=touch: push_state2(st, touch))
            o, r, dn, _ = env.step(a)
            states.append(st); touches.append(touch); actions.append(a); obs_l.append(o); rew_l.append(r); done_l.append(dn)
            terms.append([env.progress, env.target_bonus, env.energy_penalty, env.tall_bonus, env.posture_penalty, env.joints_penalty])
        rec.update(states=np.array(states), touch=np.array(touches), actions=np.array(actions), obs=np.array(obs_l),
                   rew=np.array(rew_l), done=np.array(done_l).astype(np.int32), terms=np.array(terms))
        for k, v in rec.items():
            out[f"{tag}_ep0_{k}"] = np.asarray(v)

    # ---------------- LaikagoCustomEnv (env_locomotion.py:854-890)
    env, p = make_env("LaikagoCustomEnv", _urdf="laikago_toes_limits.urdf")
    rob = env.robot
    robot_constants(out, "laikago", env, p)
    n2 = len(rob.ordered_joint_ids)
    out["laikago_init_position"] = np.array(env.robot_init_position, dtype=np.float64)
    out["laikago_termination_height"] = np.array(env.termination_height)
    out["laikago_random_start"] = np.array(int(env.robot_random_start))
    out["laikago_physics_fixedTimeStep"] = np.array(p.physics["fixedTimeStep"])
    out["laikago_physics_numSubSteps"] = np.array(p.physics["numSubSteps"])
    out["laikago_foot_names"] = np.array(rob.foot_names)
    joint_ids = rob.ordered_joint_ids
    foot_links = [rob.parts[f].bodyPartIndex for f in rob.foot_names]
    chassis_link, knee_link = -1, rob.parts["FR_lower_leg"].bodyPartIndex
    mdl = M.compile_laikago()
    orc = Oracle(mdl.to_bytes(), M.TASK_WALKER3D_CUSTOM, 1, "f64")
    lo, hi = out["laikago_joint_lo"], out["laikago_joint_hi"]
    sd = 13 + 2 * n2

    def push_state4(st, touch, body):
        full = np.zeros((1, orc.state_dim))
        full[0, :sd] = st
        orc.set_state(full)
        fr = orc.link_frames(0, mdl.n_bodies)
        p.base_pos, p.base_quat, p.base_vel = st[0:3].copy(), st[3:7].copy(), st[7:10].copy()
        for k, jid in enumerate(joint_ids):
            p.q[jid], p.qd[jid] = st[13 + k], st[13 + n2 + k]
        # the foot LINK is the toe: a fixed child of the lower leg whose origin is the toe sphere's centre
        for k, fl in enumerate(foot_links):
            gi = [g for g in range(mdl.n_geoms) if mdl.g_foot[g] == k][0]
            R = fr[mdl.foot_body[k], 0:9].reshape(3, 3)
            p.link_pos[fl] = R @ np.array([mdl.g_p1[gi][i] for i in range(3)]) + fr[mdl.foot_body[k], 9:12]   # (world: includes the base position)
        p.contacts = {fl: ([(G.FakeBullet.PLANE, -1)] if touch[k] else []) for k, fl in enumerate(foot_links)}
        if body == 1:
            p.contacts[chassis_link] = [(G.FakeBullet.PLANE, -1)]
        elif body == 2:
            p.contacts[knee_link] = [(G.FakeBullet.PLANE, -1)]

    for ep in range(2):
        env.seed(90 + ep)
        env.robot.np_random = env.np_random
        tape = env.np_random.tape.copy()
        p.contacts = {}
        p.link_pos = {}
        obs0 = env.reset()
        rec = dict(tape=tape[:640], reset_obs=obs0, reset_q=np.array([p.q[j] for j in joint_ids]), reset_mirrored=int(rob.mirrored),
                   reset_base_pos=np.array(p.base_pos), reset_base_quat=np.array(p.base_quat), reset_walk_target=env.walk_target.copy())
        rng = np.random.default_rng(600 + ep)
        states, touches, bodies, actions, obs_l, rew_l, done_l, terms = [], [], [], [], [], [], [], []
        T = 50
        for t in range(T):
            st = np.zeros(sd)
            st[0:3] = [0.02 * t, rng.normal(0, 0.02), rng.uniform(0.3, 0.55)]
            st[3:7] = G.quat_from_euler(rng.normal(0, 0.2), rng.normal(0, 0.25), rng.normal(0, 0.3))
            st[7:10] = rng.normal(0, 0.5, 3)
            st[10:13] = rng.normal(0, 0.5, 3)
            st[13:13 + n2] = lo + (hi - lo) * rng.uniform(-0.02, 1.02, n2)
            st[13 + n2:sd] = rng.normal(0, 3.0, n2)
            touch = (rng.random(4) < 0.6).astype(np.int32)
            body = 0 if t < T - 1 else (1 + ep)          # the last frame: chassis (ep 0) / a lower leg (ep 1) on the ground
            a = rng.uniform(-1.5, 1.5, n2)
            p.on_step = (lambda st=st, touch=touch, body=body: push_state4(st, touch, body))
            o, r, dn, _ = env.step(a)
            states.append(st); touches.append(touch); bodies.append(int(body > 0)); actions.append(a)
            obs_l.append(o); rew_l.append(r); done_l.append(dn)
            terms.append([env.progress, env.target_bonus, env.energy_penalty, env.tall_bonus, env.posture_penalty, env.joints_penalty])
        rec.update(states=np.array(states), touch=np.array(touches), body=np.array(bodies), actions=np.array(actions),
                   obs=np.array(obs_l), rew=np.array(rew_l), done=np.array(done_l).astype(np.int32), terms=np.array(terms))
        for k, v in rec.items():
            out[f"laikago_ep{ep}_{k}"] = np.asarray(v)
    out["laikago_n_episodes"] = np.array(2)

    path = os.path.join(HERE, "variants_reference.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays")


if __name__ == "__main__":
    main()
