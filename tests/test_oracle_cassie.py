"""Cassie on the CPU oracle: model facts from the reference's env_cassie.py, loop closure geometry, PD loop."""
import numpy as np

from mocca_envs_amd import model as M
from oracle.oracle import Oracle


def _closure_err(o, m, e=0):
    fr = o.link_frames(e, m.n_bodies)
    out = []
    for c in range(m.n_closures):
        a, b = m.cl_body_a[c], m.cl_body_b[c]
        pa = fr[a, 9:12] + fr[a, :9].reshape(3, 3) @ np.array(list(m.cl_point_a[c]))
        pb = fr[b, 9:12] + fr[b, :9].reshape(3, 3) @ np.array(list(m.cl_point_b[c]))
        out.append(np.linalg.norm(pa - pb))
    return out


def test_cassie_model_facts():
    m = M.compile_cassie()
    assert (m.n_bodies, m.n_joints, m.n_closures, m.n_ctrl, m.n_ordered, m.n_llc) == (19, 18, 2, 12, 14, 50)
    assert abs(sum(m.mass[b] for b in range(m.n_bodies)) - 32.1974) < 1e-3   # sum of the URDF link masses
    assert abs(m.dt - 0.0006) < 1e-9 and abs(m.control_dt - 0.03) < 1e-9     # env_cassie.py:287-289
    kp = np.array([100, 100, 88, 96, 50, 100, 100, 88, 96, 50, 400, 400]) / 1.9  # :292-317
    np.testing.assert_allclose(list(m.ctrl_kp)[:12], kp, rtol=1e-6)
    np.testing.assert_allclose(list(m.ctrl_kd)[:12], kp / 10, rtol=1e-6)
    assert list(m.ctrl_oidx)[:12] == [0, 1, 2, 3, 6, 7, 8, 9, 10, 13, 4, 11]     # :59-60
    damp = [m.jdamp[m.ordered_body[k]] for k in range(14)]
    np.testing.assert_allclose(damp, [1, 1, 1, 1, 0.1, 0, 1] * 2, rtol=1e-6)     # :57
    lim = [m.torque_limit[m.ordered_body[k]] for k in range(14)]
    np.testing.assert_allclose(lim, [112.5, 112.5, 195.2, 195.2, 200, 200, 45] * 2, rtol=1e-6)  # :41-56


def test_nominal_pose_closes_the_four_bar_loops():
    """createConstraint pivots are COM-frame relative: with the nominal angles (env_cassie.py:20-39) the tarsus and
    achilles-rod pivots coincide to 3 mm -- an end-to-end check of the URDF frame conventions of the compiler."""
    m = M.compile_cassie()
    o = Oracle(m.to_bytes(), M.TASK_CASSIE, 1, "f64")
    obs = o.reset(seed=0)
    assert max(_closure_err(o, m)) < 3.5e-3
    assert obs.shape == (1, 36) and obs[0, 34] == 1000.0


def test_pd_loop_and_closures_over_an_episode():
    m = M.compile_cassie()
    o = Oracle(m.to_bytes(), M.TASK_CASSIE, 1, "f64")
    o.reset(seed=0)
    for t in range(12):
        obs, rew, done, _ = o.step(np.zeros((1, 10), np.float32))
        assert np.isfinite(obs).all()
        assert max(_closure_err(o, m)) < 1e-3       # erp 0.9 pulls the 3 mm initial gap shut and keeps it shut
        if done[0]:
            break
    tk = o.get_task()[0]
    assert tk[39] == 50 * (t + 1)                    # istep counts PD iterations (env_cassie.py:381)
    assert o.last_rows() >= 6                        # 2 closures x 3 rows always present


def test_planar_cassie_takes_the_2d_urdfs_hip_limits():
    """Cassie2D (env_cassie.py:279-282) means data/robots/cassie/urdf/cassie_collide_2d.urdf: its planar root joints are this blob's three
    planar rows, and the limits of hip abduction / rotation are +-0.01 rad on both sides (:479,486,535,542 of that file) -- everything else
    equals the 3-D file.  The nominal pose (0.0356, -0.0135) starts outside them; the limit rows bring the joints back and hold them."""
    m3, m2 = M.compile_cassie(), M.compile_cassie(planar=True)
    names = M.CASSIE_ORDERED_JOINTS
    for k, n in enumerate(names):
        b = m2.ordered_body[k]
        if n in M.CASSIE_2D_LIMITS:
            assert (round(m2.jlo[b], 6), round(m2.jhi[b], 6)) == (-0.01, 0.01)
            assert m3.jhi[b] - m3.jlo[b] > 0.6
        else:
            assert (m2.jlo[b], m2.jhi[b]) == (m3.jlo[b], m3.jhi[b])
    assert sum(n in M.CASSIE_2D_LIMITS for n in names) == 4
    o = Oracle(m2.to_bytes(), M.TASK_CASSIE, 1, "f64")
    o.reset(seed=0)
    roll_yaw = [k for k, n in enumerate(names) if n in M.CASSIE_2D_LIMITS]
    for t in range(6):
        obs, _, done, _ = o.step(np.zeros((1, 10), np.float32))
        st = o.get_state()[0]
        q = np.array([st[13 + m2.ordered_body[k] - 1] for k in roll_yaw])
        assert np.abs(q).max() < 0.012, (t, q)       # one control step (50 substeps at erp 0.9) is enough to be inside
        assert abs(st[1]) < 1e-6                     # and the pelvis stays in the y = 0 plane
        if done[0]:
            break


def test_cassie_task_logic_matches_the_reference_code():
    """tests/golden/make_golden_cassie.py ran the reference's real Cassie / CassieEnv methods (with the missing
    imports supplied) over THIS oracle's physics; the oracle's own step must then reproduce them: PD loop, joint
    speed filter, torque clipping, residual targets, observation layout, reward, termination."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cassie_reference.npz"))
    m = M.compile_cassie()
    # static facts read from the reference objects
    assert list(g["ordered_joint_names"]) == M.CASSIE_ORDERED_JOINTS
    np.testing.assert_allclose([m.torque_limit[m.ordered_body[k]] for k in range(14)], g["torque_limits"], rtol=1e-6)
    np.testing.assert_allclose([m.jlo[m.ordered_body[k]] for k in range(14)], g["joint_lo"], rtol=1e-6)
    np.testing.assert_allclose([m.jhi[m.ordered_body[k]] for k in range(14)], g["joint_hi"], rtol=1e-6)
    np.testing.assert_allclose([m.jdamp[m.ordered_body[k]] for k in range(14)], g["joint_damping"], rtol=1e-6)
    np.testing.assert_allclose(list(m.ctrl_kp)[:12], g["kp"], rtol=1e-6)
    np.testing.assert_allclose(list(m.ctrl_kd)[:12], g["kd"], rtol=1e-6)
    assert abs(m.jvel_alpha - float(g["jvel_alpha"])) < 1e-7 and m.n_llc == int(g["llc_frame_skip"])
    assert abs(m.dt - float(g["scene_fixedTimeStep"])) < 1e-9 and int(g["n_constraints"]) == m.n_closures
    assert list(g["powered"]) + list(g["springs"]) == list(m.ctrl_oidx)[:12]
    np.testing.assert_allclose([m.init_q[m.ordered_body[k]] for k in range(14)], g["base_joint_angles"], rtol=1e-6)
    for ep in range(2):
        o = Oracle(m.to_bytes(), M.TASK_CASSIE, 1, "f64")
        obs = o.reset(seed=0)
        np.testing.assert_allclose(obs[0], g[f"ep{ep}_obs"][0], atol=1e-5)
        for t, a in enumerate(g[f"ep{ep}_actions"]):
            obs, rew, done, _ = o.step(a[None].astype(np.float32))
            # actions pass through float32 on this side only; 50 PD iterations amplify that to ~1e-4
            np.testing.assert_allclose(obs[0], g[f"ep{ep}_obs"][t + 1], atol=2e-3, rtol=2e-3, err_msg=f"ep{ep} t{t}")
            assert abs(rew[0] - g[f"ep{ep}_rew"][t]) < 2e-2, (ep, t, rew[0], g[f"ep{ep}_rew"][t])
            assert bool(done[0] & 1) == bool(g[f"ep{ep}_done"][t])
        np.testing.assert_allclose(o.get_state()[0][:49], g[f"ep{ep}_final_state"][:49], atol=2e-3, rtol=2e-3)


def test_exact_position_gaps_buy_less_than_half_of_cassies_fp32_error():
    """Round 6's conditioning A/B as a test (tools/fp32_yardstick.py at full size: 9.3 -> 5.5 units): the f32 oracle's distance from the f64 oracle
    over single teacher-forced substeps, with and without the oracle's experiment switch that forms the closure gaps and the toe depths in DOUBLE
    precision (orc_set_precise_gaps; everything else stays fp32).  The exact gaps help -- and leave more than a third of the error: the rest is the
    fp32 solve of stiff rows at dt = 0.6 ms, which is why the kernel was not given a second, double-precision kinematics walk."""
    import ctypes
    from mocca_envs_amd.vec_env import compile_model_for
    m = compile_model_for("CassieEnv-v0")
    m.n_substeps, m.n_llc = 1, 1
    blob, n = m.to_bytes(), 96
    nd = 13 + 2 * m.n_joints
    med = {}
    for bits in (0, 3):
        o32, o64 = Oracle(blob, M.TASK_CASSIE, n, "f32"), Oracle(blob, M.TASK_CASSIE, n, "f64")
        o32.lib.orc_set_precise_gaps.argtypes, o32.lib.orc_set_precise_gaps.restype = [ctypes.c_void_p, ctypes.c_int], None
        o32.lib.orc_set_precise_gaps(o32.h, bits)
        o32.reset(seed=4); o64.reset(seed=4)
        rng = np.random.default_rng(2)
        errs = []
        for t in range(60):
            o64.set_state(o32.get_state()); o64.set_task(o32.get_task())
            a = ((1.0 if t % 3 else 0.3) * rng.uniform(-1, 1, (n, o32.act_dim))).astype(np.float32)
            _, _, done, _ = o32.step(a); o64.step(a)
            s32, s64 = o32.get_state(), o64.get_state()
            same = (o32.get_debug()[:, :12] == o64.get_debug()[:, :12]).all(1) & np.isfinite(s32).all(1) & np.isfinite(s64).all(1)
            errs.append((np.abs(s32[same][:, :nd] - s64[same][:, :nd]) / (1e-5 * (1 + np.abs(s64[same][:, :nd])))).max(1))
            if t % 8 == 7 and (done != 0).any():
                o32.reset(seed=4, mask=(done != 0).astype(np.uint8))
        med[bits] = float(np.median(np.concatenate(errs)))
    print(f"\nCassie, one substep, f32 vs f64 oracle, units of 1e-5 (1 + |x|): median {med[0]:.2f}, with exact closure gaps and toe depths {med[3]:.2f}")
    assert med[3] < 0.95 * med[0] and med[3] > 0.35 * med[0] and med[0] > 3.0
