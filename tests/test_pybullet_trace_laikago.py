"""Laikago through the PyBullet dump chain: tools/dump_pybullet_trace.py's Laikago section loads laikago_toes_limits.urdf the
way Laikago.initialize does (robots.py:584-600) and steps it with LaikagoCustomEnv's parameters (env_locomotion.py:856-864: 8 substeps of
1/480 s, start height 0.56, lower legs at -pi/6); it records the multibody -- the one place where the inertia Bullet derives for the mesh
links would show (the URDF's tensors are zero; compile_laikago uses bounding boxes, [UNVERIFIED-BULLET]) --, teacher-forced steps with
contact points, the feet flags and the env's termination test, and two free-running rollouts.  from_pybullet_dump merges the fixed toe
links into the lower legs; this harness replays the file on the f64 oracle and on libmocca_hip.so.  No PyBullet here: the tool runs against
tests/fake_pybullet.py; the branches on the REAL file are skipped until tests/golden/pybullet_laikago.npz exists."""
import os

import numpy as np
import pytest

from mocca_envs_amd import model as M

TRACE = os.path.join(os.path.dirname(__file__), "golden", "pybullet_laikago.npz")
needs_trace = pytest.mark.skipif(not os.path.exists(TRACE), reason="no PyBullet Laikago trace: run tools/dump_pybullet_trace.py <data> 1000 laikago where pybullet is installed")
NJ = 12
ND = 13 + 2 * NJ
TOL = 1e-4     # BASELINE.json north star: joint state within 1e-4 of PyBullet


def _blob(g):
    from mocca_envs_amd.pybullet_dump import from_pybullet_dump
    return from_pybullet_dump(g, M.compile_laikago(), M.LAIKAGO_JOINTS)


def _rows(g, m, key):
    a = np.asarray(g[key], float)
    out = np.zeros((len(a), ND + m.n_slots))     # (+ the warm-start words of the state record: zero, the blob does not warm-start)
    out[:, :ND] = a
    return out


def _joint_err(a, b):
    return np.abs(np.asarray(a)[..., 13:ND] - np.asarray(b)[..., 13:ND]).max(axis=-1)


def one_step_errors_oracle(g, m, precision="f64"):
    """Every recorded (state before, torques) through one stepSimulation = 8 substeps of the oracle: joint-state error against the recorded
    state after it, and whether the feet flags / the termination test of the last substep agree with the record."""
    from oracle.oracle import Oracle
    o = Oracle(m.to_bytes(), M.TASK_WALKER3D_CUSTOM, 1, precision)
    o.reset(seed=0)
    before, after, torques = _rows(g, m, "lk_before"), np.asarray(g["lk_after"]), np.asarray(g["lk_torques"])
    errs, flags_ok = [], []
    for k in range(len(before)):
        o.set_state(before[k][None].copy())
        o.physics_substeps(0, torques[k], int(m.n_substeps))
        errs.append(_joint_err(o.get_state()[0], after[k]))
        slots = [int(c[2]) for c in o.last_contacts() if int(c[1]) < 0]
        feet = [int(any(m.g_foot[s] == f for s in slots)) for f in range(4)]
        body = int(any(m.g_foot[s] < 0 for s in slots))
        flags_ok.append(feet == [int(v) for v in g["lk_feet_contact"][k]] and body == int(g["lk_body_contact"][k]))
    return np.array(errs), np.array(flags_ok)


def one_step_errors_hip(g, m):
    import torch
    from mocca_envs_amd.vec_env import VecEnv
    before, after, torques = _rows(g, m, "lk_before"), np.asarray(g["lk_after"]), np.asarray(g["lk_torques"])
    gains = np.array([m.gain[b] for b in range(1, NJ + 1)])
    env = VecEnv("LaikagoCustomEnv-v0", len(before), auto_reset=False, seed=0, model_blob=m.to_bytes())
    env.reset()
    env.set_state(before.astype(np.float32))
    env.step(torch.from_numpy((torques / gains).astype(np.float32)).cuda())
    got = env.get_state().cpu().numpy()
    env.close()
    return _joint_err(got, after)


def free_run_errors(g, m, tag="lkfree", hip=False, precision="f64"):
    states, actions = _rows(g, m, tag + "_states"), np.asarray(g[tag + "_actions"], np.float64)   # (the HIP path rounds them to float32, as env.step does)
    if hip:
        import torch
        from mocca_envs_amd.vec_env import VecEnv
        env = VecEnv("LaikagoCustomEnv-v0", 1, auto_reset=False, seed=0, model_blob=m.to_bytes())
        env.reset(); env.set_state(states[:1].astype(np.float32))
        step, get = (lambda a: env.step(torch.from_numpy(a[None].astype(np.float32)).cuda())), (lambda: env.get_state().cpu().numpy()[0])
    else:
        from oracle.oracle import Oracle
        o = Oracle(m.to_bytes(), M.TASK_WALKER3D_CUSTOM, 1, precision)
        o.reset(seed=0); o.set_state(states[:1].copy())
        gains = np.array([m.gain[b] for b in range(1, NJ + 1)])
        step, get = (lambda a: o.physics_substeps(0, gains * a, int(m.n_substeps))), (lambda: o.get_state()[0])
    errs = []
    for t, a in enumerate(actions):
        step(a)
        errs.append(_joint_err(get(), states[t + 1]))
    return np.array(errs)


# ---------------------------------------------------------------------------------------------- the TOOL's Laikago section, run here
@pytest.fixture(scope="module")
def tool_file(tmp_path_factory):
    import importlib.util
    import sys
    from fake_pybullet import make_laikago_module
    fake = make_laikago_module()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("dump_pybullet_trace", os.path.join(root, "tools", "dump_pybullet_trace.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    out = tmp_path_factory.mktemp("laikago_dump")
    old, cwd = sys.modules.get("pybullet"), os.getcwd()
    sys.modules["pybullet"] = fake
    os.chdir(out)
    try:
        tool.main_laikago("/nonexistent/data", 40)
    finally:
        os.chdir(cwd)
        if old is None:
            del sys.modules["pybullet"]
        else:
            sys.modules["pybullet"] = old
    return np.load(os.path.join(out, "pybullet_laikago.npz")), fake


def test_the_tool_loads_and_steps_laikago_like_the_reference(tool_file):
    """What the tool asked of the client (the URDF with URDF_USE_SELF_COLLISION only, 1/60 s in 8 substeps, contact ERP 0.9), what it wrote
    (twelve ordered joints in URDF order, the four toe links as feet), and that the file loads into a blob -- toes merged into the lower legs,
    total mass kept -- whose oracle replay reproduces the (oracle-backed) fake exactly, flags included."""
    g, fake = tool_file
    calls = dict((c[0], c[1]) for c in fake.fake_calls)
    assert calls["loadURDF"] == fake.URDF_USE_SELF_COLLISION and calls["contactERP"] == 0.9
    assert abs(calls["engine"]["fixedTimeStep"] - 1 / 60) < 1e-15 and calls["engine"]["numSubSteps"] == 8 and calls["engine"]["numSolverIterations"] == 5
    assert [str(n) for n in g["ordered_joint_names"]] == M.LAIKAGO_JOINTS and str(g["robot"]) == "laikago"
    links = [str(n) for n in g["link_names"]]
    assert [links[int(l)] for l in g["foot_links"]] == M.LAIKAGO_FEET
    assert all(int(g["joint_type"][int(l)]) == fake.JOINT_FIXED for l in g["foot_links"])
    np.testing.assert_allclose(g["lk_before"][0, :3], [0, 0, 0.56])
    np.testing.assert_allclose(g["lk_before"][0, 13:13 + NJ][[2, 5, 8, 11]], -np.pi / 6)
    np.testing.assert_allclose(np.abs(g["lk_torques"]).max(), 40.0, rtol=0.05)
    m, tm = _blob(g), M.compile_laikago()
    assert abs(sum(m.mass[b] for b in range(m.n_bodies)) - sum(tm.mass[b] for b in range(tm.n_bodies))) < 1e-9
    assert m.n_substeps == 8 and m.n_bodies == 13
    e, ok = one_step_errors_oracle(g, m)
    print(f"tool -> file -> blob -> oracle, Laikago: one-step joint-state error {e.max():.2e}; {int(g['lk_feet_contact'].sum())} foot contacts, "
          f"{int(g['lk_body_contact'].sum())} terminations in {len(e)} steps")
    assert e.max() < 1e-9 and ok.all()
    assert g["lk_feet_contact"].sum() > 0 and (np.asarray(g["lk_contact_points"])[:, :, 0] >= -1).any()
    for tag in ("lkfree", "lkfree03"):
        fr = free_run_errors(g, m, tag)
        assert len(fr) == 40 and fr.max() < 1e-9, (tag, fr.max())
    e32, _ = one_step_errors_oracle(g, m, "f32")
    assert 0 < e32.max() < 2e-3


@pytest.mark.gpu
def test_the_laikago_tool_file_on_the_hip_path(tool_file):
    g, _ = tool_file
    m = _blob(g)
    e_hip = one_step_errors_hip(g, m)
    e32, _ = one_step_errors_oracle(g, m, "f32")
    print(f"HIP, Laikago, one step vs the tool's f64 trace: median {np.median(e_hip):.3e} max {e_hip.max():.3e}; f32 oracle {np.median(e32):.3e} / {e32.max():.3e}")
    assert np.median(e_hip) < max(2e-5, 3 * np.median(e32)) and e_hip.max() < max(1e-3, 3 * e32.max())
    fr = free_run_errors(g, m, "lkfree03", hip=True)
    assert np.isfinite(fr).all() and fr[0] < max(1e-3, 3 * e32.max())


# ---------------------------------------------------------------------------------------------- branches on a real PyBullet file
@needs_trace
def test_laikago_link_inertias_against_bullet():
    """The record's own content before any stepping: masses and principal inertias Bullet derived for the mesh links against compile_laikago's
    bounding-box rule (reported, bounded loosely: it is the assumption this file exists to replace)."""
    g = np.load(TRACE)
    m, tm = _blob(g), M.compile_laikago()
    dm = max(abs(m.mass[b] - tm.mass[b]) for b in range(m.n_bodies))
    di = max(abs(m.inertia[b][k] - tm.inertia[b][k]) / max(1e-6, abs(tm.inertia[b][k])) for b in range(m.n_bodies) for k in range(3))
    print(f"Laikago, Bullet's multibody vs compile_laikago: largest mass difference {dm:.3e} kg, largest relative difference of a diagonal inertia term {di:.3f}")
    assert dm < 1e-6


@needs_trace
def test_laikago_one_step_of_the_oracle_against_bullet():
    g = np.load(TRACE)
    e, ok = one_step_errors_oracle(g, _blob(g))
    print(f"oracle (f64), Laikago: one-step joint-state error vs PyBullet: median {np.median(e):.3e} p99 {np.percentile(e, 99):.3e}; "
          f"feet flags and termination test agree in {ok.mean():.3f} of the steps")
    assert np.percentile(e, 99) < TOL


@needs_trace
def test_laikago_free_run_of_the_oracle_against_bullet():
    g = np.load(TRACE)
    fr = free_run_errors(g, _blob(g), "lkfree03")
    print("oracle (f64), Laikago free-running at action scale 0.3, error at steps 1 / 10 / 100 / last:", fr[0], fr[min(9, len(fr) - 1)], fr[min(99, len(fr) - 1)], fr[-1])
    assert fr.max() < TOL


@needs_trace
@pytest.mark.gpu
def test_laikago_one_step_of_the_hip_path_against_bullet():
    g = np.load(TRACE)
    e = one_step_errors_hip(g, _blob(g))
    assert np.percentile(e, 99) < TOL
