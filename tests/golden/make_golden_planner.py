#!/usr/bin/env python3
"""Golden vectors for the height-field terrain and the planner envs (build container only): bullet_objects.HeightField
(get_height_at, reload, get_random_height_field over misc_utils' Perlin / fractal noise) and Walker3DPlannerEnv / MikePlannerEnv
(env_locomotion.py:982-1133) -- the reference's real classes driven over a stub gym and a scripted fake pybullet client, same
method as make_golden.py.  The base controller the env unpickles (`MikePlannerBase.pt`, a torch policy class that is not in the
tree) is replaced by an injected callable, which is what the class calls through anyway (`self.query_base_controller`).
Output: tests/golden/planner_reference.npz (numbers and short name strings only).
Re-run:  python tests/golden/make_golden_planner.py
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402
import make_golden_variants as GV  # noqa: E402


class FakeBulletTerrain(GV.FakeBulletAnyMJCF):
    """+ the calls HeightField.reload and VSphere make."""
    GEOM_SPHERE, GEOM_HEIGHTFIELD = 2, 9

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.shapes, self.bodies, self.removed = [], {}, []

    def createCollisionShape(self, **kw):
        self.shapes.append(kw)
        return len(self.shapes) - 1

    def createVisualShape(self, *a, **kw):
        return -1

    def createMultiBody(self, *a, **kw):
        bid = self.next_body
        self.next_body += 1
        self.bodies[bid] = (a, kw)
        return bid

    def removeBody(self, bid):
        self.removed.append(bid)

    def changeVisualShape(self, *a, **k):
        pass

    def resetBasePositionAndOrientation(self, body, posObj=None, ornObj=None, *a):
        if body in self.bodies:
            self.bodies[body] = (self.bodies[body][0], dict(self.bodies[body][1], basePosition=tuple(posObj)))
            return
        return super().resetBasePositionAndOrientation(body, posObj, ornObj, *a)

    def getContactPoints(self, bodyA=None, linkIndexA=None):
        return [(0, bodyA, bB, linkIndexA, lB) for (bB, lB) in self.contacts.get(linkIndexA, [])]


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
    import mocca_envs.bullet_objects as BO
    import mocca_envs.env_base as env_base
    import mocca_envs.env_locomotion as loco
    import mocca_envs.misc_utils as MU

    from mocca_envs_amd import model as M
    from oracle.oracle import Oracle

    out = {}
    nj = 21

    # ---------------- noise generators (misc_utils.py:4-58), real numpy RandomState
    out["perlin_16x16_res4_seed3"] = MU.generate_perlin_noise_2d((16, 16), (4, 4), np.random.RandomState(3))
    out["fractal_32x32_res4_oct2_p1_seed4"] = MU.generate_fractal_noise_2d((32, 32), (4, 4), 2, 1, np.random.RandomState(4))

    # ---------------- HeightField (bullet_objects.py:338-441)
    p = FakeBulletTerrain()
    hf = BO.HeightField(p, (128, 128), 4)
    hf.reload(data="height_field_map_0.npy")
    sh = p.shapes[-1]
    out["hf_rows"], out["hf_cols"] = np.array(sh["numHeightfieldRows"]), np.array(sh["numHeightfieldColumns"])
    out["hf_mesh_scale"] = np.array(sh["meshScale"], float)
    (a, kw) = p.bodies[hf.id]
    out["hf_body_position"] = np.array(a[3] if len(a) > 3 else kw["basePosition"], float)      # (0, 0, (max + min) / 2)
    out["hf_data_minmax"] = np.array([hf.data.min(), hf.data.max()])
    dyn = [k for (aa, k) in p.dynamics if aa and aa[0] == hf.id][-1]
    out["hf_dynamics"] = np.array([dyn["lateralFriction"], dyn["restitution"], dyn["contactStiffness"], dyn["contactDamping"]], float)
    rng = np.random.default_rng(11)
    xy = np.concatenate([rng.uniform(-15.99, 15.99, (400, 2)), [[-16.0, -16.0], [15.99, 15.99], [0.0, 0.0], [-15.5, -15.5], [0.124, -0.126]]])
    out["hf_probe_xy"] = xy
    out["hf_probe_z"] = np.array([hf.get_height_at(x, y) for x, y in xy])
    # a small random field (the env itself always loads the file; reload(data=None) draws one): same call, 32 x 32 grid at scale 2
    for seed in (0, 5):
        small = BO.HeightField(FakeBulletTerrain(), (32, 32), 2)
        out[f"hf_random_32_s{seed}"] = small.get_random_height_field(np.random.RandomState(seed))

    # ---------------- Walker3DPlannerEnv / MikePlannerEnv
    ctrl = {}

    def controller(o):
        ctrl["obs"] = np.array(o, copy=True)
        return np.float32(ctrl["value"]), np.asarray(ctrl["action"], np.float32)

    for tag, cls, compile_fn in (("planner", "Walker3DPlannerEnv", lambda: M.compile_walker3d(M.TASK_WALKER3D_PLANNER)),
                                 ("mikeplanner", "MikePlannerEnv", lambda: M.compile_mike(planner=True))):
        getattr(loco, cls).load_base_controller = lambda self, fn: controller
        holder = {}

        def factory(*a, **k):
            holder["p"] = FakeBulletTerrain()
            return holder["p"]

        env_base.BulletClient = factory
        env = getattr(loco, cls)()
        p = holder["p"]
        rob = env.robot
        out[f"{tag}_mjcf"] = np.array(os.path.basename(p.mjcf_path))
        out[f"{tag}_obs_dim"], out[f"{tag}_act_dim"] = np.array(env.observation_space.shape[0]), np.array(env.action_space.shape[0])
        out[f"{tag}_init_position"] = np.array(env.robot_init_position, float)
        out[f"{tag}_termination_height"] = np.array(env.termination_height)
        out[f"{tag}_action_scale"] = np.array(env.action_scale)
        out[f"{tag}_torso_link"] = np.array(p.joints[env.robot_torso_id][1])
        out[f"{tag}_torso_joint"] = np.array(p.joints[env.robot_torso_id][0])
        out[f"{tag}_has_ground_ids"] = np.array(int(hasattr(env, "ground_ids")))
        out[f"{tag}_terrain_shape_rows_cols_scale"] = np.array([env.terrain.data_size[0], env.terrain.data_size[1], env.terrain.scale])
        joint_ids = rob.ordered_joint_ids
        foot_links = [rob.parts[f].bodyPartIndex for f in rob.foot_names]
        mdl = compile_fn()
        orc = Oracle(mdl.to_bytes(), M.TASK_WALKER3D_PLANNER, 1, "f64")

        def push_state(st, torso_touch):
            full = np.zeros((1, orc.state_dim))
            full[0, :55] = st
            orc.set_state(full)
            fr = orc.link_frames(0, mdl.n_bodies)
            p.base_pos, p.base_quat, p.base_vel = st[0:3].copy(), st[3:7].copy(), st[7:10].copy()
            for k, jid in enumerate(joint_ids):
                p.q[jid], p.qd[jid] = st[13 + k], st[13 + nj + k]
            for k, fl in enumerate(foot_links):
                p.link_pos[fl] = fr[mdl.foot_body[k], 12:15].copy()
            # feet on the terrain all the time: calc_state() is called WITHOUT contact ids in this env, the flags must stay 0
            p.contacts = {fl: [(env.terrain.id, -1)] for fl in foot_links}
            if torso_touch:
                p.contacts[env.robot_torso_id] = [(env.terrain.id, -1)]

        for ep in range(3):
            env.seed(80 + ep)
            env.robot.np_random = env.np_random
            tape = env.np_random.tape.copy()
            p.contacts = {}
            obs0 = env.reset()
            rec = dict(tape=tape[:64], reset_obs=obs0, reset_q=np.array([p.q[j] for j in joint_ids]), reset_mirrored=int(rob.mirrored),
                       reset_base_pos=np.array(p.base_pos), reset_walk_target=np.asarray(env.walk_target, np.float64),
                       reset_target_marker=np.array(p.bodies[env.target.id][1]["basePosition"], float))
            rng = np.random.default_rng(800 + ep)
            lo, hi = np.array([j.lowerLimit for j in rob.ordered_joints]), np.array([j.upperLimit for j in rob.ordered_joints])
            states, torso, plans, values, base_actions, base_obs, torques, obs_l, rew_l, done_l, prog = [], [], [], [], [], [], [], [], [], [], []
            T = 40
            for t in range(T):
                st = np.zeros(55)
                # walking towards the target from the start corner; episode 0 ends by height, 1 by falling off the terrain, 2 by torso contact
                z = rng.uniform(1.0, 1.4)
                if ep == 0 and t >= T - 2:
                    z = rng.uniform(0.3, 0.45)
                if ep == 1 and t >= T - 2:
                    z = -5.5
                st[0:3] = [-15.5 + 0.03 * t + rng.normal(0, 0.01), -15.5 + 0.02 * t + rng.normal(0, 0.01), z]
                st[3:7] = G.quat_from_euler(rng.normal(0, 0.15), rng.normal(0, 0.2), rng.normal(0.6, 0.4))
                st[7:10] = rng.normal(0, 0.5, 3)
                st[10:13] = rng.normal(0, 0.5, 3)
                st[13:13 + nj] = np.clip(rng.normal(0, 0.3, nj), lo, hi)
                if ep == 0 and t >= T - 2:   # pitched flat: the feet level with the base, relative height below 0.5
                    st[3:7] = G.quat_from_euler(rng.normal(0, 0.05), 1.5, rng.normal(0.6, 0.1))
                st[13 + nj:13 + 2 * nj] = rng.normal(0, 3.0, nj)
                tt = int(ep == 2 and t >= T - 3)
                plan = rng.normal(0, 1.0, 15)
                ctrl["value"] = float(rng.uniform(-2.0, 30.0))          # below 1: the log term vanishes (max(1, value))
                ctrl["action"] = rng.uniform(-1.3, 1.3, nj)             # beyond +-1: apply_action clips
                p.on_step = (lambda st=st, tt=tt: push_state(st, tt))
                o, r, dn, _ = env.step(plan)
                states.append(st); torso.append(tt); plans.append(plan); values.append(ctrl["value"]); base_actions.append(ctrl["action"])
                base_obs.append(ctrl["obs"]); torques.append(p.torques.copy()); obs_l.append(o); rew_l.append(r); done_l.append(bool(dn)); prog.append(env.progress)
            rec.update(states=np.array(states), torso_touch=np.array(torso), plans=np.array(plans), values=np.array(values),
                       base_actions=np.array(base_actions), base_obs=np.array(base_obs), torques=np.array(torques), obs=np.array(obs_l),
                       rew=np.array(rew_l, float), done=np.array(done_l).astype(np.int32), progress=np.array(prog, float))
            if ep > 0:
                del rec["base_obs"]      # concat(robot_state, plan * action_scale), :1093: one episode pins it
            for k, v in rec.items():
                out[f"{tag}_ep{ep}_{k}"] = np.asarray(v)
        out[f"{tag}_n_episodes"] = np.array(3)

    path = os.path.join(HERE, "planner_reference.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, f"({os.path.getsize(path) / 1024:.0f} KiB, {len(out)} arrays)")


if __name__ == "__main__":
    main()
