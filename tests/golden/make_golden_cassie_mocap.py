#!/usr/bin/env python3
"""Golden vectors for the Cassie mocap / phase envs, captured from the reference's own classes (build container only).

`CassiePhaseMoccaEnv` and `CassiePhaseMirrorEnv` (env_cassie.py:481-660; ids CassiePhaseMocca2DEnv-v0 / CassiePhaseMirror2DEnv-v0,
__init__.py:31-43) cannot be constructed as shipped: `loadstep.CassieTrajectory` is missing from the tree, besides the defects
make_golden_cassie.py already works around.  This script supplies the missing module with THIS project's re-creation of the class
(mocca_envs_amd/trajectory.py, built from the reference's own mocap data files), builds the env objects without the broken
constructors, and runs the reference's real `reset / step / pd_control / base_angles / compute_rewards / get_obs` over a fake
pybullet client whose stepSimulation is this project's f64 CPU oracle (planar Cassie).  The physics is the oracle's on both sides;
the vectors pin everything else: time-varying PD targets, random-state initialisation from the motion (joint angles, speeds, rod
angles, forward speed), the six mocap reward terms and their weights, the 42-float observation, the gait phases, the mirrored
observation of CassiePhaseMirrorEnv.
"""
import copy
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

from make_golden import install_stubs, quat_from_euler  # noqa: E402
from make_golden_cassie import FakeBulletCassie  # noqa: E402


class FakeBulletMocap(FakeBulletCassie):
    def getQuaternionFromEuler(self, e):
        return quat_from_euler(*e)


def main():
    install_stubs()
    sys.path.insert(0, REF)
    sys.modules.setdefault("torch", types.ModuleType("torch"))
    from mocca_envs_amd.trajectory import CassieTrajectory
    ls = types.ModuleType("mocca_envs.loadstep")
    ls.CassieTrajectory = CassieTrajectory          # the re-created class stands in for the missing module
    sys.modules["mocca_envs.loadstep"] = ls
    import scipy.ndimage
    if "scipy.ndimage.filters" not in sys.modules:
        f = types.ModuleType("scipy.ndimage.filters")
        f.gaussian_filter = scipy.ndimage.gaussian_filter
        sys.modules["scipy.ndimage.filters"] = f
    import mocca_envs  # noqa
    import mocca_envs.env_cassie as ec
    from mocca_envs.bullet_utils import BodyPart, Joint, SinglePlayerStadiumScene
    ec.BodyPart, ec.Joint = BodyPart, Joint
    if not hasattr(ec, "copy"):
        ec.copy = copy                               # CassiePhaseMoccaEnv.__init__ uses copy.deepcopy (:633)

    from mocca_envs_amd import model as M
    from oracle.oracle import Oracle

    mdl = M.compile_cassie(planar=True)              # physics only: the task layer is the reference's
    phys = Oracle(mdl.to_bytes(), M.TASK_CASSIE, 1, "f64")
    phys.reset(seed=0)
    out = {}
    rng = np.random.default_rng(7)
    # (class, tag, initial isteps): 3971 is the value the reference's comment names (:587); 700 starts just below phase_l = 0.5 and crosses it in the first step
    for cls, tag in ((ec.CassiePhaseMoccaEnv, "mocca"), (ec.CassiePhaseMirrorEnv, "mirror")):
        p = FakeBulletMocap(mdl, phys)
        robot = ec.Cassie(p)
        robot.initialize()
        env = object.__new__(cls)
        env._p, env.robot, env.state_id = p, robot, 0
        env.is_rendered, env.planar, env.residual_control, env.rsi = False, True, True, True
        env.scene = SinglePlayerStadiumScene(p, gravity=9.8, timestep=0.03 / 50 / 1, frame_skip=1)
        env.traj = CassieTrajectory()
        # CassieMocapRewEnv.__init__ (:483-493), evaluated by the reference's own arithmetic
        env.weights = {"SpeedRew": 0.1, "CoMRew": 0.02 if env.planar else 0.05, "OrientationRew": 0 if env.planar else 0.05,
                       "AngularSpeedRew": 0.1}
        wleft = 1 - sum(env.weights.values())
        env.weights["JPosRew"] = wleft / 5 * 4
        env.weights["JVelRew"] = wleft / 5
        env.mirror_indices = copy.deepcopy(cls.mirror_indices)
        env.mirror_indices["left_obs_inds"] += [40]
        env.mirror_indices["right_obs_inds"] += [41]
        if cls is ec.CassiePhaseMirrorEnv:            # :646-655
            env.neg_inds = env.mirror_indices["neg_obs_inds"] + env.mirror_indices["sideneg_obs_inds"]
            env.lr_inds = env.mirror_indices["left_obs_inds"] + env.mirror_indices["right_obs_inds"]
            env.rl_inds = env.mirror_indices["right_obs_inds"] + env.mirror_indices["left_obs_inds"]
        out[f"{tag}_weights"] = np.array([env.weights[k] for k in ("SpeedRew", "JPosRew", "JVelRew", "OrientationRew",
                                                                  "AngularSpeedRew", "CoMRew")])
        out[f"{tag}_initial_velocity"] = np.array(env.initial_velocity, dtype=np.float64)
        for k, v in env.mirror_indices.items():
            out[f"{tag}_mi_{k}"] = np.array(v, dtype=np.int64)
        for ep, istep0 in enumerate((3971, 700, 0)):
            phys.reset(seed=0)                        # restoreState(state_id): nominal pose, zero velocities, no warm starts
            obs0 = env.reset(istep=istep0)
            acts, obs_l, rew_l, done_l, terms = [], [obs0], [], [], []
            pre_state, pre_jvel, pre_istep = [phys.get_state()[0].copy()], [np.array(env.jvel, dtype=np.float64)], [env.istep]
            for t in range(14):
                a = 0.15 * rng.uniform(-1, 1, 10)
                o, r, d, info = env.step(a)
                acts.append(a); obs_l.append(o); rew_l.append(r); done_l.append(d)
                terms.append([info[k] for k in ("SpeedRew", "JPosRew", "JVelRew", "OrientationRew", "AngularSpeedRew", "CoMRew")])
                pre_state.append(phys.get_state()[0].copy())
                pre_jvel.append(np.array(env.jvel, dtype=np.float64))
                pre_istep.append(env.istep)
                if d:
                    break
            out[f"{tag}_ep{ep}_istep0"] = np.array(istep0)
            out[f"{tag}_ep{ep}_actions"] = np.array(acts)
            out[f"{tag}_ep{ep}_obs"] = np.array(obs_l)
            out[f"{tag}_ep{ep}_rew"] = np.array(rew_l)
            out[f"{tag}_ep{ep}_rew_terms"] = np.array(terms)            # already weighted (:529)
            out[f"{tag}_ep{ep}_done"] = np.array(done_l).astype(np.int32)
            out[f"{tag}_ep{ep}_state"] = np.array(pre_state)           # state[t] = before step t (state[0] = after reset)
            out[f"{tag}_ep{ep}_jvel"] = np.array(pre_jvel)
            out[f"{tag}_ep{ep}_istep"] = np.array(pre_istep)
    np.savez_compressed(os.path.join(HERE, "cassie_mocap_reference.npz"), **out)
    print("wrote cassie_mocap_reference.npz", {k: v.shape for k, v in out.items() if "_ep" in k and k.endswith("obs")})
    for tag in ("mocca", "mirror"):
        for ep in range(3):
            print(tag, ep, "rewards", np.round(out[f"{tag}_ep{ep}_rew"], 3), "done", out[f"{tag}_ep{ep}_done"])


if __name__ == "__main__":
    main()
