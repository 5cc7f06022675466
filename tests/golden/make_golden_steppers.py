#!/usr/bin/env python3
"""Golden vectors for the remaining Stepper variants (build container only), same method as make_golden.py -- the
reference's real classes over a stub gym and a scripted fake pybullet client:

  * LaikagoStepperEnv (env_locomotion.py:893-979): class constants, terrain generator with its own ranges, mirror indices,
    and scripted episodes through its four-plank / four-foot state machine, its own posture penalty, doubled progress,
    time-based early termination and body-contact termination;
  * Walker3DStepperEnv(random_reward=True) (:533-547): one episode whose reward is weighted by eight np_random draws a step;
  * Walker3DStepperEnv(plank_class="Plank" / "Pillar") (:342,356-357; bullet_objects.py:86-97): what the env does with the
    other step objects (scale, position offset).

Output: tests/golden/steppers_reference.npz (data only).  Re-run:  python tests/golden/make_golden_steppers.py
"""
from __future__ import annotations

import os
import sys
import types
import xml.etree.ElementTree as ET

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402
import make_golden_variants as V  # noqa: E402


class StepObjectsMixin:
    """loadURDF of a step object: report the scaled z of the first link's inertial origin, as Bullet's base pose does."""

    def loadURDF(self, f, basePosition=None, baseOrientation=None, useFixedBase=False, globalScaling=1.0, flags=0):
        if "objects" not in f:
            return super().loadURDF(f, basePosition, baseOrientation, useFixedBase, globalScaling, flags)
        bid = self.next_body
        self.next_body += 1
        z = float(ET.parse(f).getroot().find("link").find("inertial").find("origin").get("xyz").split()[2])
        self.plank_pose[bid] = (np.array([0, 0, z * globalScaling]), np.array([0, 0, 0, 1.0]))
        self.step_files = getattr(self, "step_files", []) + [(os.path.basename(f), globalScaling)]
        return bid


class FakeLaikago(StepObjectsMixin, V.FakeBulletURDFRobot):
    pass


class FakeWalker(StepObjectsMixin, G.FakeBullet):
    def loadURDF(self, f, basePosition=None, baseOrientation=None, useFixedBase=False, globalScaling=1.0, flags=0):
        return StepObjectsMixin.loadURDF(self, f, basePosition, baseOrientation, useFixedBase, globalScaling, flags)


def make_env(cls_name, fake_cls, urdf=None, **kw):
    import mocca_envs.env_base as env_base
    import mocca_envs.env_locomotion as loco
    holder = {}

    def factory(*a, **k):
        holder["p"] = fake_cls()
        holder["p"].root_link = None
        holder["p"].robot_urdf = urdf
        return holder["p"]

    env_base.BulletClient = factory
    return getattr(loco, cls_name)(**kw), holder["p"]


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

    # ---------------- LaikagoStepperEnv
    env, p = make_env("LaikagoStepperEnv", FakeLaikago, urdf="laikago_toes_limits.urdf")
    rob = env.robot
    n2 = len(rob.ordered_joint_ids)
    sd = 13 + 2 * n2
    plank_ids = [s.id for s in env.steps]
    out["lstep_obs_dim"] = np.array(env.observation_space.shape[0])
    for k, v in zip(["neg_obs", "right_obs", "left_obs", "neg_act", "right_act", "left_act"], env.get_mirror_indices()):
        out["lstep_mirror_" + k] = np.asarray(v, dtype=np.int64)
    out["lstep_init_position"] = np.array(env.robot_init_position, dtype=np.float64)
    out["lstep_init_velocity"] = np.array(env.robot_init_velocity, dtype=np.float64)
    out["lstep_random_start"] = np.array(int(env.robot_random_start))
    out["lstep_consts"] = np.array([env.step_radius, env.rendered_step_count, env.init_step_separation, env.lookahead, env.lookbehind,
                                    env.step_bonus_smoothness, env.n_steps], dtype=np.float64)
    out["lstep_ranges"] = np.concatenate([env.dist_range, env.pitch_range, env.yaw_range, env.tilt_range]).astype(np.float64)
    out["lstep_terminal_height_curriculum"] = env.terminal_height_curriculum.copy()
    out["lstep_applied_gain_curriculum"] = env.applied_gain_curriculum.copy()
    out["lstep_plank_pos_offset"] = np.array(env.steps[0]._pos_offset)
    out["lstep_step_file"] = np.array(p.step_files[0][0]); out["lstep_step_scale"] = np.array(p.step_files[0][1])
    out["lstep_physics_fixedTimeStep"] = np.array(p.physics["fixedTimeStep"])
    out["lstep_physics_numSubSteps"] = np.array(p.physics["numSubSteps"])
    out["lstep_n_planks"] = np.array(len(plank_ids))
    for cur in (0, 5, 9):
        env.seed(70 + cur)
        env.curriculum = cur
        out[f"lstep_terrain_c{cur}_tape"] = env.np_random.tape[:100].copy()
        out[f"lstep_terrain_c{cur}_table"] = env.generate_step_placements()

    mdl = M.compile_laikago(stepper=True)
    orc = Oracle(mdl.to_bytes(), M.TASK_WALKER3D_STEPPER, 1, "f64")
    joint_ids = rob.ordered_joint_ids
    foot_links = [rob.parts[f].bodyPartIndex for f in rob.foot_names]
    chassis_link, knee_link = -1, rob.parts["FR_lower_leg"].bodyPartIndex
    lo = np.asarray(rob.to_radians(-np.ones(n2)), dtype=np.float64)
    hi = np.asarray(rob.to_radians(np.ones(n2)), dtype=np.float64)

    def push(st, touch, target, body):
        full = np.zeros((1, orc.state_dim))
        full[0, :sd] = st
        orc.set_state(full)
        fr = orc.link_frames(0, mdl.n_bodies)
        p.base_pos, p.base_quat, p.base_vel = st[0:3].copy(), st[3:7].copy(), st[7:10].copy()
        for k, jid in enumerate(joint_ids):
            p.q[jid], p.qd[jid] = st[13 + k], st[13 + n2 + k]
        for k, fl in enumerate(foot_links):
            gi = [g for g in range(mdl.n_geoms) if mdl.g_foot[g] == k][0]
            R = fr[mdl.foot_body[k], 0:9].reshape(3, 3)
            p.link_pos[fl] = R @ np.array([mdl.g_p1[gi][i] for i in range(3)]) + fr[mdl.foot_body[k], 9:12]
        p.contacts = {}
        nplk = len(plank_ids)
        for k, fl in enumerate(foot_links):
            lst = []
            if touch[k]:
                lst.append((plank_ids[env.next_step_index % nplk], 0) if target[k] else (plank_ids[(env.next_step_index + 1) % nplk], -1))
            p.contacts[fl] = lst
        if body == 1:
            p.contacts[chassis_link] = [(plank_ids[0], -1)]       # the chassis on a plank's base link
        elif body == 2:
            p.contacts[knee_link] = [(plank_ids[1], 0)]           # a lower leg on a plank's cover

    for ep, cur in enumerate((3, 9)):
        env.seed(400 + ep)
        env.robot.np_random = env.np_random
        env.curriculum = cur
        tape = env.np_random.tape.copy()
        p.contacts, p.link_pos = {}, {}
        obs0 = env.reset()
        rec = dict(tape=tape[:640], curriculum=cur, reset_obs=obs0, reset_q=np.array([p.q[j] for j in joint_ids]),
                   reset_mirrored=int(rob.mirrored), terrain=env.terrain_info.copy(), applied_gain=float(rob.applied_gain),
                   reset_base_pos=np.array(p.base_pos), reset_base_quat=np.array(p.base_quat), reset_base_vel=np.array(p.base_vel))
        rng = np.random.default_rng(500 + ep)
        T = 560 if ep == 0 else 260
        states, touches, targets, bodies, actions, obs_l, rew_l, done_l, nsi_l, plank_l, terms = ([] for _ in range(11))
        pos = np.array([0.25, 0.0, 0.45])
        hold = 0
        for t in range(T):
            tgt = env.terrain_info[env.next_step_index]
            d = tgt[:2] - pos[:2]
            dist = np.linalg.norm(d)
            if dist > 0.05 and ep == 0:
                pos[:2] += d / dist * min(0.05, dist)
            st = np.zeros(sd)
            st[0:3] = [pos[0], pos[1], tgt[2] + rng.uniform(0.3, 0.5)]
            wild = t % 9 == 0
            st[3:7] = G.quat_from_euler(rng.normal(0, 0.2), rng.normal(0, 0.6 if wild else 0.2), rng.normal(0, 0.3))
            st[7:10] = rng.normal(0, 0.5, 3)
            st[10:13] = rng.normal(0, 0.5, 3)
            st[13:13 + n2] = lo + (hi - lo) * rng.uniform(-0.02, 1.02, n2)
            if not wild:   # most frames: a posture inside the reward's "good" bands for some joints, outside for others
                st[13:13 + n2:3] = np.deg2rad(rng.uniform(-30, 30, 4))
                st[14:13 + n2:3] = np.deg2rad(rng.uniform(-40, 40, 4))
                st[15:13 + n2:3] = np.deg2rad(rng.uniform(-80, -10, 4))
            st[13 + n2:sd] = rng.normal(0, 3.0, n2)
            touch = (rng.random(4) < 0.6).astype(np.int32)
            target = np.zeros(4, np.int32)
            if ep == 0 and dist < 0.2:
                hold += 1
                if hold % 7 in (2, 3, 4) or env.stop_on_next_step:
                    k = int(rng.integers(0, 4))
                    target[k] = 1
                    touch[k] = 1
            else:
                hold = 0
            body = 0
            if t == T - 1:
                body = 1 + ep           # last frame: chassis (ep 0) / a lower leg (ep 1) touches a plank
            a = rng.uniform(-1.5, 1.5, n2)
            p.on_step = (lambda st=st, touch=touch, target=target, body=body: push(st, touch, target, body))
            o, r, dn, info = env.step(a)
            states.append(st); touches.append(touch); targets.append(target); bodies.append(int(body > 0)); actions.append(a)
            obs_l.append(o); rew_l.append(r); done_l.append(dn); nsi_l.append(env.next_step_index)
            plank_l.append([p.plank_pose[b][0] for b in plank_ids])
            terms.append([env.progress, env.energy_penalty, env.step_bonus, env.target_bonus, env.tall_bonus,
                          env.posture_penalty, env.joints_penalty, env.target_reached_count, int(env.stop_on_next_step)])
        rec.update(states=np.array(states), touch=np.array(touches), target=np.array(targets), body=np.array(bodies),
                   actions=np.array(actions), obs=np.array(obs_l), rew=np.array(rew_l), done=np.array(done_l).astype(np.int32),
                   next_step_index=np.array(nsi_l), plank_pos=np.array(plank_l), terms=np.array(terms))
        for k, v in rec.items():
            out[f"lstep_ep{ep}_{k}"] = np.asarray(v)
    out["lstep_n_episodes"] = np.array(2)

    # ---------------- Walker3DStepperEnv(random_reward=True)
    env, p = make_env("Walker3DStepperEnv", FakeWalker, random_reward=True)
    rob = env.robot
    nj = 21
    plank_ids = [s.id for s in env.steps]
    mdl = M.compile_walker3d(M.TASK_WALKER3D_STEPPER)
    orc = Oracle(mdl.to_bytes(), M.TASK_WALKER3D_STEPPER, 1, "f64")
    joint_ids = rob.ordered_joint_ids
    foot_links = [rob.parts[f].bodyPartIndex for f in rob.foot_names]
    lo = np.array([j.lowerLimit for j in rob.ordered_joints]); hi = np.array([j.upperLimit for j in rob.ordered_joints])

    def push_w(st, touch, target):
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
                lst.append((plank_ids[env.next_step_index % 3], 0) if target[k] else (plank_ids[(env.next_step_index + 1) % 3], -1))
            p.contacts[fl] = lst

    env.seed(900)
    env.robot.np_random = env.np_random
    env.curriculum = 6
    tape = env.np_random.tape.copy()
    p.contacts = {}
    obs0 = env.reset()
    T = 150
    rec = dict(tape=tape[:122 + 8 * T + 16], curriculum=6, reset_obs=obs0, reset_q=np.array([p.q[j] for j in joint_ids]),
               terrain=env.terrain_info.copy(), random_reward=1)
    rng = np.random.default_rng(901)
    states, touches, targets, actions, obs_l, rew_l, done_l, nsi_l, terms = ([] for _ in range(9))
    pos = np.array([0.3, 0.0, 1.25])
    yaw, hold = 0.0, 0
    for t in range(T):
        tgt = env.terrain_info[env.next_step_index]
        d = tgt[:2] - pos[:2]
        dist = np.linalg.norm(d)
        if dist > 0.05:
            pos[:2] += d / dist * min(0.07, dist)
        yaw += rng.normal(0, 0.04)
        st = np.zeros(55)
        st[0:3] = [pos[0], pos[1], tgt[2] + rng.uniform(1.0, 1.3)]
        st[3:7] = G.quat_from_euler(rng.normal(0, 0.3), rng.normal(0.1, 0.3), yaw)
        st[7:10] = rng.normal(0, 1.0, 3); st[10:13] = rng.normal(0, 1.0, 3)
        st[13:13 + nj] = lo + (hi - lo) * rng.uniform(-0.02, 1.02, nj)
        for hip, knee in ((5, 6), (10, 11)):
            st[13 + hip], st[13 + knee] = np.deg2rad(rng.uniform(-30, 21)), np.deg2rad(rng.uniform(-40, 1))
        st[13 + 1] = np.deg2rad(rng.uniform(-25, 16))
        st[13 + nj:55] = rng.normal(0, 3.0, nj)
        touch = (rng.random(2) < 0.7).astype(np.int32)
        target = np.zeros(2, np.int32)
        if dist < 0.3:
            hold += 1
            if hold % 9 in (3, 4, 5):
                k = int(rng.integers(0, 2)); target[k] = 1; touch[k] = 1
        else:
            hold = 0
        a = rng.uniform(-1.5, 1.5, nj)
        p.on_step = (lambda st=st, touch=touch, target=target: push_w(st, touch, target))
        o, r, dn, info = env.step(a)
        states.append(st); touches.append(touch); targets.append(target); actions.append(a)
        obs_l.append(o); rew_l.append(r); done_l.append(dn); nsi_l.append(env.next_step_index)
        terms.append([env.progress, -env.energy_penalty, env.step_bonus, env.target_bonus, 0.0, env.tall_bonus, -env.posture_penalty, -env.joints_penalty])
    rec.update(states=np.array(states), touch=np.array(touches), target=np.array(targets), actions=np.array(actions), obs=np.array(obs_l),
               rew=np.array(rew_l), done=np.array(done_l).astype(np.int32), next_step_index=np.array(nsi_l), terms=np.array(terms))
    for k, v in rec.items():
        out[f"rr_ep0_{k}"] = np.asarray(v)

    # ---------------- reset() on the contact manifolds of the episode before (env_locomotion.py:484-499)
    # Walker3DStepperEnv.reset calls calc_feet_state() right after robot.reset() -- before the terrain is redrawn and next_step_index is
    # rewound, and with no stepSimulation in between, so getContactPoints still answers with the last frame of the OLD episode.  The fake
    # client does what Bullet does there: it keeps returning the contacts of its last step.  Four endings: a foot on the target cover
    # (count 0 -> 1 in the last step); the same with the count at 1 (the last step advances next_step_index: the cover touched is no
    # longer the target's); one foot on the target cover and the other on the cover of the plank AFTER it while the index advances
    # (that plank is the target by the time reset() looks); no contact at all.
    env, p = make_env("Walker3DStepperEnv", FakeWalker)
    rob = env.robot
    plank_ids = [s.id for s in env.steps]
    joint_ids = rob.ordered_joint_ids
    foot_links = [rob.parts[f].bodyPartIndex for f in rob.foot_names]

    def push_k(st, kinds):   # per foot: 0 no contact, 1 cover of the target plank, 2 cover of the plank after it, 3 base link of the plank after it
        full = np.zeros((1, orc.state_dim))
        full[0, :55] = st
        orc.set_state(full)
        fr = orc.link_frames(0, mdl.n_bodies)
        p.base_pos, p.base_quat, p.base_vel = st[0:3].copy(), st[3:7].copy(), st[7:10].copy()
        for k, jid in enumerate(joint_ids):
            p.q[jid], p.qd[jid] = st[13 + k], st[13 + nj + k]
        for k, fl in enumerate(foot_links):
            p.link_pos[fl] = fr[mdl.foot_body[k], 12:15].copy()
        n0 = env.next_step_index
        p.contacts = {fl: {0: [], 1: [(plank_ids[n0 % 3], 0)], 2: [(plank_ids[(n0 + 1) % 3], 0)], 3: [(plank_ids[(n0 + 1) % 3], -1)]}[int(kinds[k])]
                      for k, fl in enumerate(foot_links)}

    endings = {"count0": [(1, 3)], "count1": [(1, 0), (1, 3)], "advance": [(1, 0), (1, 2)], "none": [(3, 3), (0, 0)]}
    for si, (name, tail) in enumerate(endings.items()):
        env.seed(950 + si)
        env.robot.np_random = env.np_random
        env.curriculum = 5
        p.contacts = {}
        pos0 = env.np_random.pos
        env.reset()
        tape_a = env.np_random.tape[pos0:env.np_random.pos].copy()
        rng = np.random.default_rng(960 + si)
        kinds_seq = [(3, 0), (0, 3)] + list(tail)            # two plain frames, then the ending
        states, kinds_l, actions, obs_l, rew_l, nsi_l, trc_l = ([] for _ in range(7))

        def frame(kinds):
            tgt = env.terrain_info[env.next_step_index]
            st = np.zeros(55)
            st[0:3] = [tgt[0] + rng.normal(0, 0.1), tgt[1] + rng.normal(0, 0.1), tgt[2] + rng.uniform(1.0, 1.3)]
            st[3:7] = G.quat_from_euler(rng.normal(0, 0.2), rng.normal(0.1, 0.2), rng.normal(0, 0.3))
            st[7:10] = rng.normal(0, 1.0, 3); st[10:13] = rng.normal(0, 1.0, 3)
            st[13:13 + nj] = lo + (hi - lo) * rng.uniform(0.1, 0.9, nj)
            st[13 + nj:55] = rng.normal(0, 3.0, nj)
            a = rng.uniform(-1.5, 1.5, nj)
            p.on_step = (lambda st=st, kinds=kinds: push_k(st, kinds))
            o, r, dn, info = env.step(a)
            states.append(st); kinds_l.append(kinds); actions.append(a); obs_l.append(o); rew_l.append(r)
            nsi_l.append(env.next_step_index); trc_l.append(env.target_reached_count)

        for kinds in kinds_seq:
            frame(kinds)
        n_before = len(states)
        p.on_step = None
        pos1 = env.np_random.pos
        obs_r = env.reset()                                   # NO p.contacts = {} here: the client still holds the last frame's contacts
        tape_b = env.np_random.tape[pos1:env.np_random.pos].copy()
        rec = dict(tape_a=tape_a, tape_b=tape_b, curriculum=5, n_before=n_before, reset_obs=obs_r, reset_trc=env.target_reached_count,
                   reset_nsi=env.next_step_index, reset_feet_contact=np.array(rob.feet_contact, dtype=np.float64), terrain_b=env.terrain_info.copy())
        for kinds in [(1, 0), (1, 3), (0, 0)]:                # the new episode: a foot on the target cover in its very first frames
            frame(kinds)
        rec.update(states=np.array(states), kinds=np.array(kinds_l), actions=np.array(actions), obs=np.array(obs_l), rew=np.array(rew_l),
                   next_step_index=np.array(nsi_l), trc=np.array(trc_l))
        for k, v in rec.items():
            out[f"stale_{name}_{k}"] = np.asarray(v)
    out["stale_names"] = np.array(list(endings))

    # ---------------- the other step objects
    for pc in ("Plank", "Pillar", "LargePlank", "NoSuchPlank"):
        env, p = make_env("Walker3DStepperEnv", FakeWalker, plank_class=pc)
        out[f"plank_{pc}_file"] = np.array(p.step_files[0][0])
        out[f"plank_{pc}_scale"] = np.array(p.step_files[0][1])
        out[f"plank_{pc}_pos_offset"] = np.array(env.steps[0]._pos_offset)
        out[f"plank_{pc}_count"] = np.array(len(env.steps))
        path = os.path.join(G.REF, "mocca_envs", "data", "objects", "steps", p.step_files[0][0])
        shapes = []
        for link in ET.parse(path).getroot().findall("link"):
            g = link.find("collision").find("geometry")
            z = float(link.find("collision").find("origin").get("xyz").split()[2])
            if g.find("box") is not None:
                shapes.append([0] + [float(v) for v in g.find("box").get("size").split()] + [z])
            else:
                c = g.find("cylinder")
                shapes.append([1, float(c.get("radius")), float(c.get("radius")), float(c.get("length")), z])
        out[f"plank_{pc}_shapes"] = np.array(shapes)        # per link: kind (0 box, 1 cylinder), size x y z (or r r length), centre z -- unscaled

    path = os.path.join(HERE, "steppers_reference.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays")


if __name__ == "__main__":
    main()
