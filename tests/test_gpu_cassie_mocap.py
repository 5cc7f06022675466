"""CassiePhaseMocca2DEnv-v0 / CassiePhaseMirror2DEnv-v0 (env_cassie.py:481-660) on the GPU.

(i) the episodes the reference's own classes produced (tests/golden/make_golden_cassie_mocap.py) replayed through
    libmocca_hip.so: taped reset (istep of the recorded np_random.randint), then every env.step restarted from the recorded
    state (teacher forcing) with the kernel's own fp32 50-iteration PD + physics loop;
(ii) HIP vs the f32 oracle on random actions with in-kernel auto-resets (random state initialisation from the motion);
(iii) the gym classes.  -m gpu."""
import os

import numpy as np
import pytest

from mocca_envs_amd import model as M

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "cassie_mocap_reference.npz"))
IDS = {"mocca": "CassiePhaseMocca2DEnv-v0", "mirror": "CassiePhaseMirror2DEnv-v0"}
REPL = 3


def _oracle_for(env, n, precision="f32"):
    from oracle.oracle import Oracle
    o = Oracle(env.model.to_bytes(), M.TASK_CASSIE, n, precision)
    o.set_trajectory(env.trajectory.table(), env.trajectory.max_time(), 0.03)
    return o


@pytest.mark.parametrize("tag", ["mocca", "mirror"])
def test_recorded_episodes_replay_on_the_gpu(tag):
    import torch
    from mocca_envs_amd.vec_env import VecEnv, task_to_float64, task_from_float64
    env = VecEnv(IDS[tag], REPL, auto_reset=False, seed=0)
    assert (env.obs_dim, env.act_dim) == (42, 10)
    o32 = _oracle_for(env, 1)                # the yardstick: the SAME teacher-forced step in fp32 on the CPU, against the f64 recording
    o32.reset(seed=0)
    errs, yard = [], []
    for ep in range(3):
        istep0 = int(G[f"{tag}_ep{ep}_istep0"])
        env.set_draw_tape(np.full((REPL, 1), (istep0 + 0.5) / 10000.0, np.float32))   # np_random.randint(0, 10000) of the recording
        obs0 = env.reset().cpu().numpy()
        env.set_draw_tape(None)
        gobs = G[f"{tag}_ep{ep}_obs"]
        assert (obs0 == obs0[0]).all()
        np.testing.assert_allclose(obs0[0], gobs[0], atol=3e-6, rtol=2e-6)             # a table lookup and one kinematics pass
        st = env.get_state().cpu().numpy()[0]
        nd = 13 + 2 * env.model.n_joints
        np.testing.assert_allclose(st[:nd], G[f"{tag}_ep{ep}_state"][0][:nd], atol=1e-6)
        assert int(task_to_float64(env.get_task())[0, 39]) == istep0
        for t, a in enumerate(G[f"{tag}_ep{ep}_actions"]):
            s = np.zeros((REPL, env.state_dim), np.float32)
            s[:, : len(G[f"{tag}_ep{ep}_state"][t])] = G[f"{tag}_ep{ep}_state"][t]
            env.set_state(torch.from_numpy(s))
            tk = task_to_float64(env.get_task())
            tk[:, 24:38] = G[f"{tag}_ep{ep}_jvel"][t]
            tk[:, 39] = G[f"{tag}_ep{ep}_istep"][t]
            tk[:, 7] = 0
            env.set_task(task_from_float64(tk))
            o, r, d, _ = env.step(torch.from_numpy(np.tile(a[None].astype(np.float32), (REPL, 1))).cuda())
            o, r, d = o.cpu().numpy(), r.cpu().numpy(), d.cpu().numpy()
            assert (o == o[0]).all() and (r == r[0]).all() and (d == d[0]).all()       # every wave produces the same bits
            want = gobs[t + 1]
            errs.append(np.abs(o[0] - want) / (1e-3 + 1e-3 * np.abs(want)))
            so = o32.get_state(); so[:] = 0; so[0, : len(G[f"{tag}_ep{ep}_state"][t])] = G[f"{tag}_ep{ep}_state"][t]; o32.set_state(so)
            to = o32.get_task(); to[0, 24:38] = G[f"{tag}_ep{ep}_jvel"][t]; to[0, 39] = G[f"{tag}_ep{ep}_istep"][t]; to[0, 7] = 0
            o32.set_task(to)
            yard.append(np.abs(o32.step(a[None].astype(np.float32))[0][0] - want) / (1e-3 + 1e-3 * np.abs(want)))
            # exact entries: the two phases (f64 time arithmetic in the kernel) -- also proves the mirrored layout was chosen alike
            np.testing.assert_allclose(o[0, 40:42], want[40:42], atol=1e-6)
            # 50 stiff fp32 iterations from an f64 state; the finite-difference joint speeds (26..39) divide by 0.03: bounded below, against
            # what the fp32 oracle loses on the same steps
            assert abs(float(r[0]) - G[f"{tag}_ep{ep}_rew"][t]) < 1e-3, (ep, t, float(r[0]), G[f"{tag}_ep{ep}_rew"][t])
            assert bool(int(d[0]) & 1) == bool(G[f"{tag}_ep{ep}_done"][t]), (ep, t)
            assert int(task_to_float64(env.get_task())[0, 39]) == G[f"{tag}_ep{ep}_istep"][t + 1]
    e = np.concatenate(errs)
    print(f"\n{tag}: GPU vs reference-code-over-f64-oracle, one env.step: median {np.median(e):.3g} p99 {np.percentile(e, 99):.3g} "
          f"max {e.max():.3g} units of 1e-3 (1 + |x|)")
    # The typical step is 3e-5 absolute (0.03 units).  In a rare step fp32 rounding flips one clamp of one of the 50 solves, on EITHER
    # implementation (profiles/archive/r03_mocap_step_probe.txt: one step of 42 where the f32 oracle AND the kernel are 3e-2 from the f64 recording
    # and 2e-5 from each other; earlier blobs showed such a step on one side only).  So: median and count of outlier steps bounded
    # absolutely, the worst step against 3 x the fp32 yardstick's worst step.
    per_step, y = np.array([x.max() for x in errs]), np.array([x.max() for x in yard])
    print(f"   f32 oracle on the same steps: median {np.median(np.concatenate(yard)):.3g} worst step {y.max():.3g}; kernel worst step {per_step.max():.3g}")
    assert np.median(e) < 0.05 and np.median(e) < 3 * max(np.median(np.concatenate(yard)), 0.01)
    assert per_step.max() < max(30.0, 3 * y.max()), (per_step.max(), y.max())      # a flip costs up to 1e-2 (the solver's on / off switches), not more
    assert (per_step > 0.5).sum() <= max(2, 2 * int((y > 0.5).sum())), per_step   # of ~42 steps: the flips, nothing systematic
    env.close()


def test_mocap_env_matches_the_oracle_with_auto_resets():
    """Free-running teacher forcing against the f32 oracle, 64 envs, in-kernel resets from random frames of the motion."""
    import torch
    from mocca_envs_amd.vec_env import VecEnv, task_from_float64
    from oracle.oracle import PARAM_AUTO_RESET
    n = 64
    env = VecEnv("CassiePhaseMirror2DEnv-v0", n, auto_reset=True, seed=5)
    orc = _oracle_for(env, n)
    orc.set_param(PARAM_AUTO_RESET, 1)
    og = env.reset().cpu().numpy()
    oc = orc.reset(seed=5)
    np.testing.assert_allclose(og, oc, atol=3e-6, rtol=2e-6)                  # same Philox draw -> same istep -> same frame
    np.testing.assert_allclose(env.get_state().cpu().numpy(), orc.get_state(), atol=1e-6)
    isteps = orc.get_task()[:, 39]
    assert len(set(isteps.tolist())) > 40 and isteps.min() >= 0 and isteps.max() < 10000
    rng = np.random.default_rng(0)
    n_reset, errs = 0, []
    for t in range(25):
        env.set_state(torch.from_numpy(orc.get_state().astype(np.float32)))
        env.set_task(task_from_float64(orc.get_task()))
        a = (0.2 * rng.uniform(-1, 1, (n, 10))).astype(np.float32)
        og, rg, dg, _ = env.step(torch.from_numpy(a).cuda())
        oc, rc, dc, _ = orc.step(a)
        og, rg, dg = og.cpu().numpy(), rg.cpu().numpy(), dg.cpu().numpy()
        same = dg == dc
        assert same.mean() > 0.95                                               # a height within fp32 noise of 0.6 m may flip
        np.testing.assert_allclose(rg[same], rc[same], atol=5e-3)
        # reset envs: exact table lookups; running envs: fp32 physics noise
        rs = same & (dc != 0)
        n_reset += int(rs.sum())
        np.testing.assert_allclose(og[rs], oc[rs], atol=3e-6, rtol=2e-6)
        run = same & (dc == 0)
        errs.append((np.abs(og[run] - oc[run]) / (1e-3 + 1e-3 * np.abs(oc[run]))).max(axis=1))
        np.testing.assert_allclose(og[run][:, 40:42], oc[run][:, 40:42], atol=1e-6)
    e = np.concatenate(errs)
    print(f"\nmocap Cassie, GPU vs f32 oracle per env.step: median {np.median(e):.3g} p90 {np.percentile(e, 90):.3g} units; {n_reset} in-kernel resets")
    assert n_reset > 10 and np.median(e) < 0.5 and np.percentile(e, 90) < 3.0
    env.close()


def test_missing_trajectory_is_an_error_not_a_crash():
    import ctypes as C
    import torch
    from mocca_envs_amd import lib as L
    lib = L.load()
    blob = M.compile_cassie(planar=True, mode=M.CASSIE_PHASE_MOCCA).to_bytes()
    buf = C.create_string_buffer(blob, len(blob))
    h = C.c_void_p()
    assert lib.mocca_create(buf, len(blob), M.TASK_CASSIE, 2, 0, C.byref(h)) == 0
    assert lib.mocca_obs_dim(h) == 42
    obs = torch.zeros(2, 42, device="cuda")
    assert lib.mocca_reset(h, None, 0, C.c_void_p(obs.data_ptr()), None) == -1
    assert b"mocca_set_trajectory" in lib.mocca_last_error(h)
    assert lib.mocca_set_trajectory(h, None, 10, 1.0, 0.03) == -1
    lib.mocca_destroy(h)


def ph_of(base):
    return (base.mocap_time() / base.traj.max_time()) % 1


def test_phase_gym_classes():
    import mocca_envs_amd
    for env_id, mirrored in (("CassiePhaseMocca2DEnv-v0", False), ("CassiePhaseMirror2DEnv-v0", True)):
        env = mocca_envs_amd.make(env_id)
        base = env.unwrapped
        assert base.observation_space.shape == (42,) and base.action_space.shape == (10,)
        assert base.planar and base.initial_velocity == [0.8, 0, 0]
        assert base.mirror_indices["left_obs_inds"][-1] == 40 and base.mirror_indices["right_obs_inds"][-1] == 41
        base.seed(3)
        o1 = env.reset()
        assert o1.shape == (42,) and 0 <= base.istep < 10000
        ph = (base.mocap_time() / base.traj.max_time()) % 1
        want = (ph, (ph + 0.5) % 1)
        if mirrored and ph > 0.5:
            want = want[::-1]
        np.testing.assert_allclose(o1[40:42], want, atol=1e-6)
        o2 = base.reset(istep=3971)     # the reference's reset signature (:585)
        assert base.istep == 3971
        ang = base.traj.joint_angles(base.mocap_time())
        if mirrored and ph_of(base) > 0.5:
            ang = np.concatenate([-ang[7:9], ang[9:14], ang[0:7]])   # left <- right (abduction, yaw negated), right <- left
        np.testing.assert_allclose(o2[6:20], ang, atol=2e-6)
        i0 = base.istep
        ob, r, d, info = env.step(np.zeros(10))
        assert base.istep == i0 + 50 and ob.shape == (42,) and 0.0 < r <= 1.0 and isinstance(d, bool)
        np.testing.assert_allclose(base.base_angles(), base.traj.joint_angles(base.mocap_time()))
        env.close()
