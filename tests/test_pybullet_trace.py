"""Physics parity against a real PyBullet trace, if one is supplied (tools/dump_pybullet_trace.py).

PyBullet cannot be installed in the build image, so tests/golden/pybullet_walker3d.npz does not exist and these
tests are skipped: rigid-body physics parity stays *unpinned* (DESIGN.md section 4).  The harness is here so that a
trace produced on any machine with pybullet turns that statement into a measured number without new code."""
import os

import numpy as np
import pytest

TRACE = os.path.join(os.path.dirname(__file__), "golden", "pybullet_walker3d.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(TRACE), reason="no PyBullet trace supplied (parity unpinned)")


def test_model_blob_against_bullet_multibody():
    from mocca_envs_amd import model as M
    g = np.load(TRACE)
    m = M.compile_walker3d()
    print("Bullet link count", int(g["n_links"]), "masses", g["mass"])
    assert abs(g["mass"].sum() - sum(m.mass[b] for b in range(m.n_bodies))) < 0.5


def test_one_step_error_of_the_oracle_against_bullet():
    from mocca_envs_amd import model as M
    from oracle.oracle import Oracle
    g = np.load(TRACE)
    m = M.compile_walker3d()
    gains = np.array([m.gain[b] for b in range(1, 22)])
    o = Oracle(m.to_bytes(), 0, 1, "f64")
    o.reset(seed=0)
    errs = []
    for b, a, tq in zip(g["before"], g["after"], g["torques"]):
        st = np.zeros((1, o.state_dim)); st[0, :55] = b
        o.set_state(st)
        o.step((tq / gains)[None].astype(np.float32))
        errs.append(np.abs(o.get_state()[0, 13:34] - a[13:34]).max())
    errs = np.array(errs)
    print(f"one-step joint-angle error vs PyBullet: median {np.median(errs):.3e} p99 {np.percentile(errs, 99):.3e}")
