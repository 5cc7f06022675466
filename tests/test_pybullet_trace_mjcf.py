"""Child3D and Mike through the PyBullet dump chain (SURVEY 8 f1 x f3): the two MJCF walkers that share Walker3D's tree and joint names.
tools/dump_pybullet_trace.py <data> N child3d | mike loads child3d.xml / mike.xml with the walkers' flags, applies what the robot class does
to the multibody before anything is recorded (Mike.load_robot_model: changeDynamics(waist, mass=8), robots.py:507-510), starts from the
env's own pose (Child3DCustomEnv: "crawl", base pitched 90 degrees, 0.38 m up, gains x 0.4, fallen below 0.1 m; MikeStepperEnv: running
start at (0.3, 0, 1.0) on planks, no ground plane) and writes the same keys as the Walker3D file -- Child3D the flat sections, Mike the
stepping-stone section.  The harness is tests/test_pybullet_trace.py's (same tree, same state layout) with the robot's own template blob.
No PyBullet here: the tool runs against tests/fake_pybullet.py; the real-file branches wait for tests/golden/pybullet_{child3d,mike}.npz."""
import os

import numpy as np
import pytest

import test_pybullet_trace as W
from mocca_envs_amd import model as M

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-4


def _blob(g, robot):
    from mocca_envs_amd.pybullet_dump import from_pybullet_dump
    return from_pybullet_dump(g, {"child3d": M.compile_child3d, "mike": M.compile_mike}[robot](), M.WALKER3D_JOINT_NAMES)


def _run_tool(tmp, robot, n):
    import importlib.util
    import sys
    from fake_pybullet import make_module
    fake = make_module(robot=robot)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("dump_pybullet_trace", os.path.join(root, "tools", "dump_pybullet_trace.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    old, cwd = sys.modules.get("pybullet"), os.getcwd()
    sys.modules["pybullet"] = fake
    os.chdir(tmp)
    try:
        tool.main("/nonexistent/data", n, robot)
    finally:
        os.chdir(cwd)
        if old is None:
            del sys.modules["pybullet"]
        else:
            sys.modules["pybullet"] = old
    return np.load(os.path.join(tmp, f"pybullet_{robot}.npz")), fake


def hip_one_step_errors(g, m, env_id, prefix=""):
    """Every recorded (state before, torques) of the `prefix` section in one launch of `env_id` built from blob m."""
    import torch
    from mocca_envs_amd.vec_env import VecEnv, task_to_float64, task_from_float64
    before, after, torques = (W._stepper_rows(g, m) if prefix else W._rows(g, m))
    gains = np.array([m.gain[b] for b in range(1, W.NJ + 1)])
    env = VecEnv(env_id, len(before), auto_reset=False, seed=0, model_blob=m.to_bytes())
    env.reset()
    if prefix:
        ter = np.zeros((len(before), 128), np.float32)
        ter[:, :124] = W._stepper_terrain(g)
        env.set_terrain(ter)
        tk = task_to_float64(env.get_task())
        tk[:, 21] = 1.0                                   # applied_gain 1 (the reset set the curriculum's)
        env.set_task(task_from_float64(tk))
    env.set_state(before.astype(np.float32))
    act = np.divide(torques, gains, out=np.zeros_like(torques), where=gains > 0)     # Mike's abdomen is passive: gain 0, torque 0
    env.step(torch.from_numpy(act.astype(np.float32)).cuda())
    got = env.get_state().cpu().numpy()
    env.close()
    return W._joint_err(got, after)


# ---------------------------------------------------------------------------------------------- Child3D
@pytest.fixture(scope="module")
def child_file(tmp_path_factory):
    return _run_tool(tmp_path_factory.mktemp("child3d_dump"), "child3d", 30)


def test_the_tool_records_child3d_from_its_crawl_pose(child_file):
    g, fake = child_file
    h = np.sqrt(0.5)
    assert str(g["robot"]) == "child3d" and "stp_before" not in g
    np.testing.assert_allclose(g["free_states"][0, :7], [0, 0, 0.38, 0, h, 0, h], atol=1e-12)          # robots.py:316-318,335
    q0 = g["free_states"][0, 13:34]
    np.testing.assert_allclose(q0[[13, 17, 14, 18]], np.pi / 2)
    np.testing.assert_allclose(q0[[6, 11]], -120 * np.pi / 180)
    assert np.abs(g["torques"]).max() <= 0.4 * 100 + 1e-9 and np.abs(g["torques"]).max() > 0.3 * 100       # gains x 0.4, robots.py:328
    m, tm = _blob(g, "child3d"), M.compile_child3d()
    assert abs(sum(m.mass[b] for b in range(m.n_bodies)) - sum(tm.mass[b] for b in range(tm.n_bodies))) < 1e-9
    e = W.one_step_errors_oracle(g, m)
    assert e.max() < 1e-9, e.max()
    for tag in ("free", "free03"):
        fr = W.free_run_errors_oracle(g, m, tag)
        assert len(fr) == 30 and fr.max() < 1e-9, (tag, fr.max())
    e32 = W.one_step_errors_oracle(g, m, "f32")
    assert 0 < e32.max() < 5e-3


@pytest.mark.gpu
def test_the_child3d_tool_file_on_the_hip_path(child_file):
    g, _ = child_file
    m = _blob(g, "child3d")
    e_hip, e32 = hip_one_step_errors(g, m, "Child3DCustomEnv-v0"), W.one_step_errors_oracle(g, m, "f32")
    print(f"HIP, Child3D, one step vs the tool's f64 trace: median {np.median(e_hip):.3e} max {e_hip.max():.3e}; f32 oracle {np.median(e32):.3e} / {e32.max():.3e}")
    assert np.median(e_hip) < max(2e-5, 3 * np.median(e32)) and e_hip.max() < max(1e-3, 3 * e32.max())


# ---------------------------------------------------------------------------------------------- Mike
@pytest.fixture(scope="module")
def mike_file(tmp_path_factory):
    return _run_tool(tmp_path_factory.mktemp("mike_dump"), "mike", 30)


def test_the_tool_records_mike_on_planks_with_the_8_kg_waist(mike_file):
    g, fake = mike_file
    assert str(g["robot"]) == "mike" and "before" not in g and "free_states" not in g            # MikeStepperEnv has no ground plane
    links = [str(n) for n in g["link_names"]]
    cd = [c for c in fake.fake_calls if c[0] == "changeDynamics" and c[2].get("mass") is not None]
    assert len(cd) == 1 and links[cd[0][1][1]] == "waist" and cd[0][2]["mass"] == 8                # robots.py:507-510, before the record is taken
    np.testing.assert_allclose(g["stp_before"][0, :3], [0.3, 0.0, 1.0])                           # env_locomotion.py:845
    assert np.abs(g["stp_torques"][:, :3]).max() == 0.0 and np.abs(g["stp_torques"][:, 3:]).max() > 10      # the passive abdomen (power_coef 0)
    m, tm = _blob(g, "mike"), M.compile_mike()
    assert abs(m.mass[2] - 8.0) < 1e-9 and abs(tm.mass[2] - 8.0) < 1e-9                           # body 2 = the link that carries MJCF body "waist"
    assert abs(float(g["stp_pos_offset"][2]) - m.plank_com_z) < 1e-6
    e = W.stepper_one_step_errors_oracle(g, m)
    assert e.max() < 1e-9, e.max()
    e32 = W.stepper_one_step_errors_oracle(g, m, "f32")
    assert 0 < e32.max() < 5e-3


@pytest.mark.gpu
def test_the_mike_tool_file_on_the_hip_path(mike_file):
    g, _ = mike_file
    m = _blob(g, "mike")
    e_hip, e32 = hip_one_step_errors(g, m, "MikeStepperEnv-v0", "stp_"), W.stepper_one_step_errors_oracle(g, m, "f32")
    print(f"HIP, Mike on planks, one step vs the tool's f64 trace: median {np.median(e_hip):.3e} max {e_hip.max():.3e}; f32 oracle {np.median(e32):.3e} / {e32.max():.3e}")
    # (re)starts excluded from the strict bound: at rest in the reset pose the three abdomen hinges of mike.xml (locked by a range of
    # +-0.001 degrees) and the left knee (0 = its stop) sit ON joint stops, so which limit rows exist in which substep is decided by the last
    # bit -- the f32 oracle already differs from the f64 one there (3 / 4 rows in substeps 2 and 3) -- and joint speeds come out rad/s apart on
    # the HIP path; limit-row flips are counted and bounded on full-size batches in tests/test_gpu_substep.py
    start = np.abs(np.asarray(g["stp_before"])[:, 7:13]).max(axis=1) + np.abs(np.asarray(g["stp_before"])[:, 34:55]).max(axis=1) == 0.0
    assert 1 <= start.sum() <= 3 and np.isfinite(e_hip).all()
    assert np.median(e_hip) < max(2e-5, 3 * np.median(e32)) and e_hip[~start].max() < max(1e-3, 3 * e32.max()), np.sort(e_hip[~start])[-3:]


# ---------------------------------------------------------------------------------------------- branches on real PyBullet files
@pytest.mark.skipif(not os.path.exists(os.path.join(GOLDEN, "pybullet_child3d.npz")), reason="no PyBullet Child3D trace: run tools/dump_pybullet_trace.py <data> 1000 child3d where pybullet is installed")
def test_child3d_one_step_of_the_oracle_against_bullet():
    g = np.load(os.path.join(GOLDEN, "pybullet_child3d.npz"))
    e = W.one_step_errors_oracle(g, _blob(g, "child3d"))
    print(f"oracle (f64), Child3D: one-step joint-state error vs PyBullet: median {np.median(e):.3e} p99 {np.percentile(e, 99):.3e}")
    assert np.percentile(e, 99) < TOL


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLDEN, "pybullet_mike.npz")), reason="no PyBullet Mike trace: run tools/dump_pybullet_trace.py <data> 1000 mike where pybullet is installed")
def test_mike_on_planks_of_the_oracle_against_bullet():
    g = np.load(os.path.join(GOLDEN, "pybullet_mike.npz"))
    m = _blob(g, "mike")
    print("Mike's waist link after changeDynamics(mass=8): mass", m.mass[2], "inertia", list(m.inertia[2])[:3], "(compile_mike scales the file's inertia with the mass)")
    e = W.stepper_one_step_errors_oracle(g, m)
    print(f"oracle (f64), Mike on planks: one-step joint-state error vs PyBullet: median {np.median(e):.3e} p99 {np.percentile(e, 99):.3e}")
    assert np.percentile(e, 99) < TOL
