"""Height-field terrain and planner envs (SURVEY 8 f4 tail) against golden vectors captured from the reference's own classes
(tests/golden/make_golden_planner.py: bullet_objects.HeightField, misc_utils' noise, Walker3DPlannerEnv / MikePlannerEnv over a
scripted fake pybullet, the unpicklable base controller replaced by an injected callable).  CPU only: host logic + oracle."""
import os

import numpy as np
import pytest

from mocca_envs_amd import host_logic as H
from mocca_envs_amd import model as M
from oracle.oracle import Oracle

NJ = 21


@pytest.fixture(scope="module")
def pg():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "planner_reference.npz"), allow_pickle=False)


def _field():
    from mocca_envs_amd.terrain import load_height_field
    return load_height_field()


def test_noise_generators_match_the_reference(pg):
    np.testing.assert_allclose(H.perlin_noise_2d((16, 16), (4, 4), np.random.RandomState(3)), pg["perlin_16x16_res4_seed3"], atol=1e-12)
    np.testing.assert_allclose(H.fractal_noise_2d((32, 32), (4, 4), 2, 1, np.random.RandomState(4)), pg["fractal_32x32_res4_oct2_p1_seed4"], atol=1e-12)
    for seed in (0, 5):
        got = H.random_height_field(np.random.RandomState(seed), (32, 32), 2)
        np.testing.assert_allclose(got, pg[f"hf_random_32_s{seed}"], atol=1e-10)
        assert np.allclose(got.reshape(32, 32)[:5, :5], 0.0, atol=1e-12)      # the start platform


def test_height_field_constants_and_lookup(pg):
    data, scale = _field()
    assert data.shape == (int(pg["hf_rows"]), int(pg["hf_cols"])) == H.HEIGHT_FIELD_SIZE and scale == H.HEIGHT_FIELD_SCALE
    np.testing.assert_allclose(pg["hf_mesh_scale"], [1 / scale, 1 / scale, 1])
    np.testing.assert_allclose([data.min(), data.max()], pg["hf_data_minmax"], atol=1e-6)
    # Bullet centres the height range on the shape's origin; the body is created at (max + min) / 2: world heights = the data
    np.testing.assert_allclose(pg["hf_body_position"], [0, 0, (data.max() + data.min()) / 2], atol=1e-6)
    m = M.compile_walker3d(M.TASK_WALKER3D_PLANNER)
    np.testing.assert_allclose([m.plank_friction, m.plank_stiffness, m.plank_damping], pg["hf_dynamics"][[0, 2, 3]])
    o = Oracle(m.to_bytes(), M.TASK_WALKER3D_PLANNER, 1, "f64")
    o.set_heightfield(data, scale)
    for (x, y), z in zip(pg["hf_probe_xy"], pg["hf_probe_z"]):
        assert abs(H.height_at(data, scale, x, y) - z) < 1e-6          # the shipped grid is float32
        assert abs(o.height_at(x, y) - z) < 1e-6


@pytest.mark.parametrize("tag", ["planner", "mikeplanner"])
def test_planner_env_constants(pg, tag):
    m = M.compile_walker3d(M.TASK_WALKER3D_PLANNER) if tag == "planner" else M.compile_mike(planner=True)
    np.testing.assert_allclose(list(m.init_pos), pg[f"{tag}_init_position"], atol=1e-6)
    assert abs(m.termination_height - float(pg[f"{tag}_termination_height"])) < 1e-7
    assert int(pg[f"{tag}_obs_dim"]) == 52 and int(pg[f"{tag}_act_dim"]) == 15 and float(pg[f"{tag}_action_scale"]) == 2
    assert int(pg[f"{tag}_has_ground_ids"]) == 0                          # remove_ground: calc_state() never sees contact ids
    # the torso link is the one that carries MJCF body "waist": its last hinge is abdomen_y = blob body 2
    assert str(pg[f"{tag}_torso_link"]) == "waist" and str(pg[f"{tag}_torso_joint"]) == "abdomen_y"
    torso_geoms = [g for g in range(m.n_geoms) if m.g_torso[g]]
    assert torso_geoms and all(m.g_body[g] == 1 + M.WALKER3D_JOINT_NAMES.index("abdomen_y") for g in torso_geoms)
    o = Oracle(m.to_bytes(), M.TASK_WALKER3D_PLANNER, 1, "f64")
    assert o.obs_dim == 52 and o.act_dim == 21                            # the kernel takes the base controller's 21 joint actions


@pytest.mark.parametrize("prec,tol", [("f64", 2e-6), ("f32", 2e-5)])
@pytest.mark.parametrize("tag", ["planner", "mikeplanner"])
def test_planner_env_episodes(pg, tag, prec, tol):
    """reset (robot.reset draws, then the target: xy ~ U(-16, 16)^2, z from get_height_at, float32) and 3 scripted episodes ending by
    relative height < 0.5, by z < -5 and by torso contact; reward = progress + log(max(1, value)) / 3 with the controller's value."""
    m = M.compile_walker3d(M.TASK_WALKER3D_PLANNER) if tag == "planner" else M.compile_mike(planner=True)
    data, scale = _field()
    gains = np.array([m.gain[b] for b in range(1, NJ + 1)], np.float64)
    for ep in range(int(pg[f"{tag}_n_episodes"])):
        g = lambda k: pg[f"{tag}_ep{ep}_{k}"]
        o = Oracle(m.to_bytes(), M.TASK_WALKER3D_PLANNER, 1, prec)
        o.set_heightfield(data, scale)
        o.set_tape(g("tape"))
        obs0 = o.reset(seed=0)
        st, tk = o.get_state()[0], o.get_task()[0]
        np.testing.assert_allclose(st[13:13 + NJ], g("reset_q"), atol=tol)
        np.testing.assert_allclose(st[0:3], g("reset_base_pos"), atol=tol)
        assert int(tk[11]) == int(g("reset_mirrored"))
        np.testing.assert_allclose(tk[0:3], g("reset_walk_target"), atol=1e-6)       # float32 in the reference
        np.testing.assert_allclose(g("reset_target_marker"), g("reset_walk_target"), atol=1e-6)
        np.testing.assert_allclose(obs0[0, 1:], g("reset_obs")[1:], atol=5 * tol)   # [0] = height: the fake client has no feet yet at reset
        if ep == 0:   # base_obs = concat(robot_state, plan * action_scale), :1093
            np.testing.assert_allclose(g("base_obs")[1][50:], 2 * g("plans")[1], atol=1e-6)
            np.testing.assert_allclose(g("base_obs")[1][:50], g("obs")[0][:50], atol=1e-6)
        done_seen = False
        for t in range(len(g("states"))):
            full = np.zeros((1, o.state_dim)); full[0, :55] = g("states")[t]
            o.set_state(full)
            a = g("base_actions")[t].astype(np.float32)
            np.testing.assert_allclose(gains * np.clip(a, -1, 1), g("torques")[t], rtol=1e-6)      # apply_action(base_action), :1096
            ob, r, d, _ = o.task_step(a[None], np.ones((1, 2), np.int32), None, np.array([g("torso_touch")[t]], np.int32))
            np.testing.assert_allclose(ob[0], g("obs")[t], atol=5 * tol, err_msg=f"ep{ep} t{t} obs")
            assert ob[0, 48] == 0 and ob[0, 49] == 0                                               # feet_contact stays 0 in this env
            np.testing.assert_allclose(r[0], g("progress")[t], atol=5e-6 if prec == "f64" else 5e-4, err_msg=f"ep{ep} t{t} progress")
            value_term = np.log(max(1.0, float(np.float32(g("values")[t])))) / 3                    # the caller's share of the reward
            np.testing.assert_allclose(r[0] + value_term, g("rew")[t], atol=5e-6 if prec == "f64" else 5e-4)
            assert bool(d[0] & 1) == bool(g("done")[t]), (ep, t)
            done_seen |= bool(d[0] & 1)
        assert done_seen
