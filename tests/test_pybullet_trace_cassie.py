"""BASELINE config 4 (Cassie) through the PyBullet dump chain: tools/dump_pybullet_trace.py's Cassie
section loads cassie_collide.urdf the way env_cassie.py does (flags, per-joint damping, the two point-to-point constraints, the rods'
collision filter) and records the multibody, teacher-forced env steps through CassieEnv.step's own 50-iteration PD loop -- with the
contact POSITIONS Bullet's 4-point manifolds kept and the constraint forces -- and a free-running rollout at action scale 0.1.
pybullet_dump.from_pybullet_dump turns the record into the Cassie blob (closure pivots from the createConstraint rows); this harness
replays it on the f64 oracle and on libmocca_hip.so.  No PyBullet here: the chain runs on a synthetic record (tests/pybullet_synth.py) and
on the file the tool itself writes against tests/fake_pybullet.py; the branches on the REAL file are skipped until
tests/golden/pybullet_cassie.npz exists."""
import os

import numpy as np
import pytest

from mocca_envs_amd import model as M

TRACE = os.path.join(os.path.dirname(__file__), "golden", "pybullet_cassie.npz")
needs_trace = pytest.mark.skipif(not os.path.exists(TRACE), reason="no PyBullet Cassie trace: run tools/dump_pybullet_trace.py <data> 300 cassie where pybullet is installed")
TOL = 1e-4     # BASELINE.json north star: joint state within 1e-4 of PyBullet


def _blob(g):
    from mocca_envs_amd.pybullet_dump import from_pybullet_dump
    return from_pybullet_dump(g, M.compile_cassie(), M.cassie_joint_names()[0])


def _task_rows(g, key, n):
    from pybullet_synth import cassie_jvel
    tk = np.zeros((n, M.TASK_WORDS))
    tk[:, 21] = 1.0
    tk[:, 24:38] = cassie_jvel(g, key)
    tk[:, 38] = 1.085       # initial_z: the pelvis height the episode started at (only the observation reads it)
    return tk


def _qerr(a, b, nj):
    return np.abs(np.asarray(a)[..., 13:13 + nj] - np.asarray(b)[..., 13:13 + nj]).max(axis=-1)


def env_step_errors_oracle(g, m, precision="f64"):
    """One CassieEnv.step from every recorded (state, jvel, action): joint-angle error against the recorded state after it, [N]."""
    from oracle.oracle import Oracle
    from pybullet_synth import cassie_rows, cassie_jvel
    before, after = cassie_rows(g, m, "cas_before"), cassie_rows(g, m, "cas_after")
    tk, jv_after = _task_rows(g, "cas_jvel_before", len(before)), cassie_jvel(g, "cas_jvel_after")
    o = Oracle(m.to_bytes(), M.TASK_CASSIE, 1, precision)
    o.reset(seed=0)
    errs, jerr = [], []
    for k in range(len(before)):
        o.set_state(before[k][None]); o.set_task(tk[k][None])
        o.step(np.asarray(g["cas_action"][k], np.float32)[None])
        errs.append(_qerr(o.get_state()[0], after[k], m.n_joints)); jerr.append(np.abs(o.get_task()[0, 24:38] - jv_after[k]).max())
    return np.array(errs), np.array(jerr)


def env_step_errors_hip(g, m):
    import torch
    from mocca_envs_amd.vec_env import VecEnv, task_from_float64
    from pybullet_synth import cassie_rows
    before, after = cassie_rows(g, m, "cas_before"), cassie_rows(g, m, "cas_after")
    n = len(before)
    env = VecEnv("CassieEnv-v0", n, auto_reset=False, seed=0, model_blob=m.to_bytes())
    env.reset()
    env.set_state(before.astype(np.float32)); env.set_task(task_from_float64(_task_rows(g, "cas_jvel_before", n)))
    env.step(torch.from_numpy(np.asarray(g["cas_action"], np.float32)).cuda())
    got = env.get_state().cpu().numpy()
    env.close()
    return _qerr(got, after, m.n_joints)


def free_run_errors(g, m, hip=False, precision="f64"):
    from pybullet_synth import cassie_rows
    states, actions = cassie_rows(g, m, "casfree_states"), np.asarray(g["casfree_actions"], np.float32)
    tk0 = _task_rows(g, "casfree_jvel", len(states))[:1]
    if hip:
        import torch
        from mocca_envs_amd.vec_env import VecEnv, task_from_float64
        env = VecEnv("CassieEnv-v0", 1, auto_reset=False, seed=0, model_blob=m.to_bytes())
        env.reset(); env.set_state(states[:1].astype(np.float32)); env.set_task(task_from_float64(tk0))
        step, get = (lambda a: env.step(torch.from_numpy(a[None]).cuda())), (lambda: env.get_state().cpu().numpy()[0])
    else:
        from oracle.oracle import Oracle
        o = Oracle(m.to_bytes(), M.TASK_CASSIE, 1, precision)
        o.reset(seed=0); o.set_state(states[:1]); o.set_task(tk0)
        step, get = (lambda a: o.step(a[None])), (lambda: o.get_state()[0])
    errs = []
    for t, a in enumerate(actions):
        step(a)
        errs.append(_qerr(get(), states[t + 1], m.n_joints))
    return np.array(errs)


# ---------------------------------------------------------------------------------------------- the chain on a synthetic record
@pytest.fixture(scope="module")
def synth():
    from pybullet_synth import synthetic_record_cassie
    return synthetic_record_cassie(n_trace=10, n_free=8)


def test_cassie_blob_round_trip_through_the_dump_format(synth):
    """compile_cassie -> PyBullet-convention record (inertial frames at the COMs, a fixed link split off, createConstraint pivots in the links'
    inertial frames) -> from_pybullet_dump: masses, closures and joint data survive; a record without its constraints is refused."""
    from mocca_envs_amd.pybullet_dump import from_pybullet_dump
    g, m = synth
    tm = M.compile_cassie()
    assert m.n_closures == 2 and abs(sum(m.mass[b] for b in range(m.n_bodies)) - sum(tm.mass[b] for b in range(tm.n_bodies))) < 1e-9
    for k in range(2):
        assert (m.cl_body_a[k], m.cl_body_b[k]) == (tm.cl_body_a[k], tm.cl_body_b[k])
    np.testing.assert_allclose([m.jdamp[b] for b in range(1, m.n_bodies)], [tm.jdamp[b] for b in range(1, tm.n_bodies)], atol=1e-9)
    bad = {k: v for k, v in g.items() if k != "constraints"}
    with pytest.raises(ValueError, match="constraints"):
        from_pybullet_dump(bad, tm, M.cassie_joint_names()[0])
    swapped = dict(g, constraints=np.asarray(g["constraints"])[:, [1, 0, 2, 6, 7, 8, 3, 4, 5]])     # child and parent exchanged: same closure
    m2 = from_pybullet_dump(swapped, tm, M.cassie_joint_names()[0])
    np.testing.assert_allclose(np.array(m2.cl_point_a), np.array(from_pybullet_dump(g, tm, M.cassie_joint_names()[0]).cl_point_a), atol=1e-12)


def test_the_cassie_harness_on_a_synthetic_record(synth):
    """The traces are the f64 oracle's on the loaded blob: the oracle branch reproduces them exactly, teacher-forced and free-running."""
    g, m = synth
    e, je = env_step_errors_oracle(g, m)
    assert e.max() < 1e-9 and je.max() < 1e-9, (e.max(), je.max())
    fr = free_run_errors(g, m)
    assert len(fr) == 8 and fr.max() < 1e-9, fr.max()
    e32, _ = env_step_errors_oracle(g, m, "f32")
    assert 0 < e32.max() < 5e-3            # fp32 through 50 stiff low-level iterations: the yardstick of the HIP branch


@pytest.mark.gpu
def test_the_cassie_hip_harness_on_a_synthetic_record(synth):
    g, m = synth
    e_hip = env_step_errors_hip(g, m)
    e32, _ = env_step_errors_oracle(g, m, "f32")
    print(f"HIP, Cassie, one env step (50 low-level iterations) vs the synthetic f64 trace: median {np.median(e_hip):.3e} max {e_hip.max():.3e}; f32 oracle {np.median(e32):.3e} / {e32.max():.3e}")
    assert np.median(e_hip) < max(2e-5, 3 * np.median(e32)) and e_hip.max() < max(1e-3, 3 * e32.max())
    fr = free_run_errors(g, m, hip=True)
    assert np.isfinite(fr).all() and fr[0] < max(1e-3, 3 * e32.max())


# ---------------------------------------------------------------------------------------------- the TOOL's Cassie section, run here
@pytest.fixture(scope="module")
def tool_file(tmp_path_factory):
    import importlib.util
    import sys
    from fake_pybullet import make_cassie_module
    fake = make_cassie_module()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("dump_pybullet_trace", os.path.join(root, "tools", "dump_pybullet_trace.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    out = tmp_path_factory.mktemp("cassie_dump")
    old, cwd = sys.modules.get("pybullet"), os.getcwd()
    sys.modules["pybullet"] = fake
    os.chdir(out)
    try:
        tool.main_cassie("/nonexistent/data", 6)
    finally:
        os.chdir(cwd)
        if old is None:
            del sys.modules["pybullet"]
        else:
            sys.modules["pybullet"] = old
    return np.load(os.path.join(out, "pybullet_cassie.npz")), fake


def test_the_tool_wires_cassie_like_the_reference(tool_file):
    """What the tool did to the client: 14 damped ordered joints in the reference's order, 4 rods reset to the nominal rod angles, two
    point-to-point constraints tarsus <-> achilles rod with env_cassie.py:114-137's pivots, the rods' collision filter -- and the file it
    wrote loads into a blob whose oracle replay reproduces the (oracle-backed) fake exactly."""
    import warnings
    g, fake = tool_file
    assert [str(n) for n in g["ordered_joint_names"]] == M.CASSIE_ORDERED_JOINTS and len(g["rod_joint_names"]) == 4
    links = [str(n) for n in g["link_names"]]
    cons = np.asarray(g["constraints"])
    assert [links[int(r[0])] for r in cons] == ["left_tarsus", "right_tarsus"] and [links[int(r[1])] for r in cons] == ["left_achilles_rod", "right_achilles_rod"]
    np.testing.assert_allclose(cons[:, 3:6], [[-0.22735404, 0.05761813, 0.00711836], [-0.22735404, 0.05761813, -0.00711836]])
    np.testing.assert_allclose(g["constraint_info_pivots"][:, 3:6], [[0.254001, 0, 0]] * 2)
    assert sorted(links[int(l)] for l in g["collision_filter_off"]) == ["left_achilles_rod", "left_achilles_rod_y", "right_achilles_rod", "right_achilles_rod_y"]
    dmp = {str(n): float(d) for n, d in zip(g["joint_names"], g["joint_damping"])}
    assert [dmp[n] for n in M.CASSIE_ORDERED_JOINTS] == [1, 1, 1, 1, 0.1, 0, 1, 1, 1, 1, 1, 0.1, 0, 1]
    assert float(g["engine_contactERP"]) == 0.9 and "engine_erp" not in g            # the fake answers like an OLD pybullet: few engine keys
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m = _blob(g)
    assert any("not recorded" in str(x.message) for x in w), "the loader must say which solver parameters the record does not pin"
    e, je = env_step_errors_oracle(g, m)
    print(f"tool -> file -> blob -> oracle: one env step joint error {e.max():.2e}, jvel error {je.max():.2e}")
    assert e.max() < 1e-6 and je.max() < 1e-5        # (the tool's float32 normalised angles enter the PD law exactly as the env's do)
    assert free_run_errors(g, m)[:3].max() < 1e-5
    assert (np.asarray(g["cas_contact_points"])[:, :, 0] >= -1).any() and np.abs(g["cas_constraint_forces"]).max() > 0


@pytest.mark.gpu
def test_the_tool_file_on_the_hip_path(tool_file):
    g, _ = tool_file
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = _blob(g)
    e_hip = env_step_errors_hip(g, m)
    e32, _ = env_step_errors_oracle(g, m, "f32")
    assert e_hip.max() < max(1e-3, 3 * e32.max()), (e_hip.max(), e32.max())


# ---------------------------------------------------------------------------------------------- branches on a real PyBullet file
@needs_trace
def test_cassie_env_step_of_the_oracle_against_bullet():
    g = np.load(TRACE)
    e, je = env_step_errors_oracle(g, _blob(g))
    print(f"oracle (f64), Cassie: one env step (30 ms) joint error vs PyBullet: median {np.median(e):.3e} p99 {np.percentile(e, 99):.3e}; jvel {np.median(je):.3e}")
    assert np.percentile(e, 99) < TOL


@needs_trace
def test_cassie_free_run_of_the_oracle_against_bullet():
    g = np.load(TRACE)
    fr = free_run_errors(g, _blob(g))
    print("oracle (f64), Cassie free-running at action scale", float(g["cas_action_scale"]), "error at steps 1 / 10 / last:", fr[0], fr[min(9, len(fr) - 1)], fr[-1])
    assert fr.max() < TOL


@needs_trace
@pytest.mark.gpu
def test_cassie_env_step_of_the_hip_path_against_bullet():
    g = np.load(TRACE)
    e = env_step_errors_hip(g, _blob(g))
    assert np.percentile(e, 99) < TOL


# ---------------------------------------------------------------------------------------------- one height-field frame (planner envs)
HF_TRACE = os.path.join(os.path.dirname(__file__), "golden", "pybullet_heightfield.npz")


def heightfield_probe_errors(g):
    """Every probe of the file (radius, centre, Bullet's normal and distance) against the oracle's sphere_heightfield on the shipped map."""
    from mocca_envs_amd.terrain import load_height_field
    from oracle.oracle import Oracle
    data, scale = load_height_field()
    assert int(g["scale"]) == int(scale) and tuple(int(v) for v in g["shape"]) == data.shape
    o = Oracle(M.compile_walker3d(M.TASK_WALKER3D_PLANNER).to_bytes(), M.TASK_WALKER3D_PLANNER, 1, "f64")
    o.set_heightfield(data, scale)
    errs = []
    for row in np.asarray(g["probes"], float):
        rad, C, n_b, d_b = row[0], row[1:4], row[4:7], row[7]
        gap, n = o.heightfield_probe(C, rad, 0.02)
        if d_b < 0.02 and gap < 0.02:
            errs.append(max(abs(gap - d_b), np.abs(n - n_b).max()))
    return np.array(errs)


def test_the_tool_writes_a_height_field_frame(tmp_path):
    """tools/dump_pybullet_trace.py's height-field frame against a client that answers getClosestPoints with the dense reference's exhaustive
    search: the file's probes agree with the oracle's windowed search (incl. the 14 cm and 23 cm spheres)."""
    import importlib.util
    import sys
    import shutil
    from fake_pybullet import make_heightfield_module
    from mocca_envs_amd.terrain import load_height_field
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("dump_pybullet_trace", os.path.join(root, "tools", "dump_pybullet_trace.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    data_dir = tmp_path / "data"
    (data_dir / "objects" / "misc").mkdir(parents=True)
    np.save(data_dir / "objects" / "misc" / "height_field_map_0.npy", load_height_field()[0])   # (this project's copy of the reference's data asset)
    old, cwd = sys.modules.get("pybullet"), os.getcwd()
    sys.modules["pybullet"] = make_heightfield_module()
    os.chdir(tmp_path)
    try:
        tool.main_heightfield(str(data_dir))
    finally:
        os.chdir(cwd)
        if old is None:
            del sys.modules["pybullet"]
        else:
            sys.modules["pybullet"] = old
    g = np.load(tmp_path / "pybullet_heightfield.npz")
    e = heightfield_probe_errors(g)
    assert len(e) > 300 and e.max() < 1e-6, (len(e), e.max() if len(e) else None)


@pytest.mark.skipif(not os.path.exists(HF_TRACE), reason="no PyBullet height-field frame: run tools/dump_pybullet_trace.py <data> 0 heightfield where pybullet is installed")
def test_height_field_contacts_against_bullet():
    e = heightfield_probe_errors(np.load(HF_TRACE))
    print(f"sphere vs height field, oracle against Bullet's closest points: {len(e)} probes in contact, worst gap / normal difference {e.max():.3e}")
    assert e.max() < 1e-4
