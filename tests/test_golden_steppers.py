"""LaikagoStepperEnv, Walker3DStepperEnv(random_reward=True) and the other step objects: model blobs and oracle task logic vs
golden vectors captured from the reference's own classes (tests/golden/make_golden_steppers.py).  CPU only."""
import os

import numpy as np
import pytest

from mocca_envs_amd import host_logic as H
from mocca_envs_amd import model as M
from oracle.oracle import PARAM_CURRICULUM, PARAM_RANDOM_POSE, PARAM_RANDOM_REWARD, Oracle

REW_TOL = {"f64": 5e-6, "f32": 1e-3}     # LaikagoStepper doubles the progress term (a difference of O(300) potentials)


@pytest.fixture(scope="module")
def sg():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "steppers_reference.npz"), allow_pickle=False)


def test_laikago_stepper_constants(sg):
    m = M.compile_laikago(stepper=True)
    c = sg["lstep_consts"]
    assert (abs(m.step_radius - c[0]) < 1e-7 and m.n_planks == int(c[1]) == int(sg["lstep_n_planks"]) and abs(m.init_step_separation - c[2]) < 1e-7
            and int(c[3]) == 2 and m.lookbehind == int(c[4]) and m.step_bonus_smoothness == c[5] and int(c[6]) == 20)
    r = sg["lstep_ranges"]
    np.testing.assert_allclose([m.dist_range[0], m.dist_range[1], -m.pitch_range_deg, m.pitch_range_deg, -m.yaw_range_deg, m.yaw_range_deg,
                                -m.tilt_range_deg, m.tilt_range_deg], r, atol=1e-6)
    np.testing.assert_allclose([H.terminal_height(k, m) for k in range(10)], sg["lstep_terminal_height_curriculum"], atol=1e-12)
    np.testing.assert_allclose([H.applied_gain(k, m) for k in range(10)], sg["lstep_applied_gain_curriculum"], atol=1e-12)
    np.testing.assert_allclose(list(m.init_pos), sg["lstep_init_position"], atol=1e-7)
    np.testing.assert_allclose(list(m.init_vel), sg["lstep_init_velocity"], atol=1e-7)
    assert int(sg["lstep_random_start"]) == 0
    assert abs(m.plank_com_z - sg["lstep_plank_pos_offset"][2]) < 1e-7 and str(sg["lstep_step_file"]) == "plank_large.urdf"
    np.testing.assert_allclose(list(m.plank_half), np.array([1.0, 20.0, 0.5]) * float(sg["lstep_step_scale"]) / 2, atol=1e-6)
    assert m.n_substeps == int(sg["lstep_physics_numSubSteps"]) and abs(m.dt * m.n_substeps - float(sg["lstep_physics_fixedTimeStep"])) < 1e-9
    assert int(sg["lstep_obs_dim"]) == 6 + 2 * 12 + 4 + 5 * (m.lookbehind + 2) == 54
    got = H.mirror_indices(m, stepper=True)
    for k, g in zip(["neg_obs", "right_obs", "left_obs", "neg_act", "right_act", "left_act"], got):
        np.testing.assert_array_equal(np.asarray(g), sg[f"lstep_mirror_{k}"], err_msg=k)


def test_laikago_stepper_terrain_generator(sg):
    m = M.compile_laikago(stepper=True)
    for cur in (0, 5, 9):
        orc = Oracle(m.to_bytes(), M.TASK_WALKER3D_STEPPER, 1, "f64")
        orc.set_param(PARAM_CURRICULUM, cur)
        orc.set_param(PARAM_RANDOM_POSE, 0)
        tape = np.concatenate([np.full(1, 0.5), sg[f"lstep_terrain_c{cur}_tape"]])        # robot.reset draws the mirror coin only (no random pose)
        orc.set_tape(tape)
        orc.reset(seed=0)
        np.testing.assert_allclose(orc.get_terrain()[0][:120].reshape(20, 6), sg[f"lstep_terrain_c{cur}_table"], atol=1e-9)

        class Tape:      # the host generator with the same numbers
            def __init__(self, t): self.t, self.i = t, 0
            def uniform(self, lo, hi, size):
                v = lo + (hi - lo) * self.t[self.i:self.i + size]; self.i += size; return v
        np.testing.assert_allclose(H.generate_step_placements(Tape(sg[f"lstep_terrain_c{cur}_tape"]), cur, m), sg[f"lstep_terrain_c{cur}_table"], atol=1e-12)


@pytest.mark.parametrize("prec,tol", [("f64", 2e-6), ("f32", 3e-5)])
def test_laikago_stepper_episodes(sg, prec, tol):
    m = M.compile_laikago(stepper=True)
    nj, sd = 12, 13 + 24
    seen = dict(max_nsi=0, stops=set(), early=0, body=0, recycled=False)
    for ep in range(int(sg["lstep_n_episodes"])):
        g = lambda k: sg[f"lstep_ep{ep}_{k}"]
        orc = Oracle(m.to_bytes(), M.TASK_WALKER3D_STEPPER, 1, prec)
        orc.set_param(PARAM_CURRICULUM, int(g("curriculum")))
        orc.set_param(PARAM_RANDOM_POSE, 0)
        orc.set_tape(g("tape"))
        obs0 = orc.reset(seed=0)
        st = orc.get_state()[0]
        np.testing.assert_allclose(st[13:13 + nj], g("reset_q"), atol=tol)
        np.testing.assert_allclose(st[0:3], g("reset_base_pos"), atol=tol)
        np.testing.assert_allclose(st[7:10], g("reset_base_vel"), atol=tol)               # robot_init_velocity
        table = orc.get_terrain()[0][:120].reshape(20, 6)
        np.testing.assert_allclose(table, g("terrain"), atol=10 * tol)
        tk = orc.get_task()[0]
        assert abs(tk[21] - float(g("applied_gain"))) < 1e-6 and int(tk[16]) == 2        # next_step_index starts at lookbehind
        # the fake client reports the feet at the origin at reset: height (entry 0) is not comparable, the four target rows are
        np.testing.assert_allclose(obs0[0, 1:], g("reset_obs")[1:], atol=5 * tol)
        states, touch, target, body, actions = g("states"), g("touch"), g("target"), g("body"), g("actions")
        for t in range(len(states)):
            full = np.zeros((1, orc.state_dim)); full[0, :sd] = states[t]
            orc.set_state(full)
            o, r, d, info = orc.task_step(actions[t][None], touch[t][None], target[t][None], body[t:t + 1])
            np.testing.assert_allclose(o[0], g("obs")[t], atol=5 * tol, err_msg=f"ep{ep} t{t} obs")
            assert (d[0] & 1) == int(g("done")[t]), f"ep{ep} t{t} done"
            np.testing.assert_allclose(r[0], g("rew")[t], atol=REW_TOL[prec], err_msg=f"ep{ep} t{t} reward")
            assert int(info[0]) == int(g("next_step_index")[t]), f"ep{ep} t{t} next_step_index"
            ter = orc.get_terrain()[0]
            pinfo = ter[120:124].astype(int)
            want = g("plank_pos")[t] - np.array([0, 0, m.plank_com_z])
            np.testing.assert_allclose(table[pinfo, 0:3], want, atol=10 * tol, err_msg=f"ep{ep} t{t} planks")
            seen["max_nsi"] = max(seen["max_nsi"], int(info[0]))
            if int(orc.get_task()[0][18]):
                seen["stops"].add(int(info[0]))
            seen["recycled"] |= sorted(pinfo) != [0, 1, 2, 3]
            seen["early"] += int(g("done")[t]) and t > 239 and not body[t]
            seen["body"] += int(body[t])
    assert seen["max_nsi"] == 19 and {6, 7} & seen["stops"] and {13, 14} & seen["stops"] and seen["recycled"], seen
    assert seen["early"] >= 5 and seen["body"] == 2, seen            # time-based early termination (:968) and body contacts (:970-974)


@pytest.mark.parametrize("prec,tol", [("f64", 2e-6), ("f32", 3e-5)])
def test_random_reward_episode(sg, prec, tol):
    """Walker3DStepperEnv(random_reward=True), env_locomotion.py:533-547: eight U(0.8, 1.2) weights drawn EVERY step."""
    m = M.compile_walker3d(M.TASK_WALKER3D_STEPPER)
    g = lambda k: sg[f"rr_ep0_{k}"]
    orc = Oracle(m.to_bytes(), M.TASK_WALKER3D_STEPPER, 1, prec)
    orc.set_param(PARAM_CURRICULUM, int(g("curriculum")))
    orc.set_param(PARAM_RANDOM_REWARD, 1)
    orc.set_tape(g("tape"))
    orc.reset(seed=0)
    np.testing.assert_allclose(orc.get_terrain()[0][:120].reshape(20, 6), g("terrain"), atol=10 * tol)
    states, touch, target, actions, terms = g("states"), g("touch"), g("target"), g("actions"), g("terms")
    used = []
    for t in range(len(states)):
        full = np.zeros((1, orc.state_dim)); full[0, :55] = states[t]
        orc.set_state(full)
        o, r, d, info = orc.task_step(actions[t][None], touch[t][None], target[t][None])
        np.testing.assert_allclose(o[0], g("obs")[t], atol=5 * tol)
        np.testing.assert_allclose(r[0], g("rew")[t], atol={"f64": 5e-6, "f32": 5e-4}[prec], err_msg=f"t{t} reward")
        w = orc.get_task()[0][30:38]
        assert (w >= 0.8).all() and (w < 1.2).all()
        np.testing.assert_allclose(w @ terms[t], g("rew")[t], atol=1e-9 if prec == "f64" else 1e-4)    # the weights ARE the reference's draws
        used.append(w.copy())
        assert int(info[0]) == int(g("next_step_index")[t])
    used = np.array(used)
    assert np.abs(np.diff(used, axis=0)).min() > 0             # new weights every step
    assert int(orc.get_task()[0][10]) == 122 + 8 * len(states)  # 1 + 21 + 100 draws at reset, 8 per step


@pytest.mark.parametrize("pc", ["Plank", "Pillar", "LargePlank"])
def test_step_objects(sg, pc):
    """plank_class (env_locomotion.py:342,356-357): shape, scaled extents and the un-rotated position offset of each step object."""
    m = M.compile_walker3d(M.TASK_WALKER3D_STEPPER, plank_class=pc)
    shapes, scale = sg[f"plank_{pc}_shapes"], float(sg[f"plank_{pc}_scale"])
    assert int(sg[f"plank_{pc}_count"]) == m.n_planks == 3
    assert abs(m.plank_com_z - sg[f"plank_{pc}_pos_offset"][2]) < 1e-7
    kind = int(shapes[0, 0])
    assert kind == m.plank_shape
    height = shapes[:, 3].sum()                               # base + cover, stacked: top face at local z = 0
    assert abs((shapes[1, 4] + shapes[1, 3] / 2)) < 1e-12 and abs(shapes[0, 4] + shapes[0, 3] / 2 - (shapes[1, 4] - shapes[1, 3] / 2)) < 1e-12
    half = np.array([shapes[0, 1], shapes[0, 2], height]) * scale * (0.5 if kind == 0 else np.array([1.0, 1.0, 0.5]))
    np.testing.assert_allclose(list(m.plank_half), half, atol=1e-6)
    assert abs(shapes[1, 3] / height - 0.1) < 1e-12           # the cover is the top tenth: the kernel's target test (lz >= 0.8 h)
    assert str(sg["plank_NoSuchPlank_file"]) == "plank_large.urdf"     # unknown names fall back to the default


def _stale_replay(g, name, step_fn, reset_fn, tol):
    """Replays one `stale_*` record: the frames before the reset, the reset (on the contacts of the last frame), the frames after it."""
    k = lambda key: g[f"stale_{name}_{key}"]
    states, kinds, actions, nb = k("states"), k("kinds"), k("actions"), int(k("n_before"))
    for t in range(len(states)):
        if t == nb:
            obs_r, trc, nsi, fc, terrain = reset_fn(k("tape_b"))
            np.testing.assert_allclose(obs_r[1:], k("reset_obs")[1:], atol=5 * tol)  # robot.reset()'s own observation: feet_contact 0 (robots.py:197-200); [0], the height, needs the link poses the fake client does not compute
            assert (trc, nsi) == (int(k("reset_trc")), int(k("reset_nsi"))), (name, trc, nsi)
            np.testing.assert_array_equal(fc, k("reset_feet_contact"))               # robot.feet_contact[:] = the OLD episode's last contacts (:656)
            np.testing.assert_allclose(terrain, k("terrain_b"), atol=10 * tol)
        touch = (kinds[t] != 0).astype(np.int32)
        target = np.where(kinds[t] == 1, 1, np.where(kinds[t] == 2, 2, 0)).astype(np.int32)
        o, r, info, trc = step_fn(states[t], actions[t], touch, target)
        np.testing.assert_allclose(o, k("obs")[t], atol=5 * tol, err_msg=f"{name} t{t}")
        np.testing.assert_allclose(r, k("rew")[t], atol=max(5e-6, 20 * tol), err_msg=f"{name} t{t} reward")
        assert (info, trc) == (int(k("next_step_index")[t]), int(k("trc")[t])), (name, t, info, trc)


@pytest.mark.parametrize("prec,tol", [("f64", 2e-6), ("f32", 3e-5)])
def test_reset_reads_the_contacts_of_the_episode_before(sg, prec, tol):
    """Walker3DStepperEnv.reset -> calc_feet_state() on Bullet's stale manifolds (env_locomotion.py:484-499, MOCCA_TASKF_STALE_RESET_CONTACTS):
    the reference's own class over a client that keeps answering with its last frame's contacts, replayed through the oracle -- the new
    episode starts with the old feet_contact flags (first step's observation) and target_reached_count = 1 where a foot was on the cover of
    the plank that was the target by then; with the flag cleared the episode starts clean."""
    m = M.compile_walker3d(M.TASK_WALKER3D_STEPPER)
    assert m.task_flags & M.TASKF_STALE_RESET_CONTACTS
    for name in [str(n) for n in sg["stale_names"]]:
        orc = Oracle(m.to_bytes(), M.TASK_WALKER3D_STEPPER, 1, prec)
        orc.set_param(PARAM_CURRICULUM, int(sg[f"stale_{name}_curriculum"]))
        orc.set_tape(sg[f"stale_{name}_tape_a"])
        orc.reset(seed=0)

        def step_fn(st, a, touch, target):
            full = np.zeros((1, orc.state_dim)); full[0, :55] = st
            orc.set_state(full)
            o, r, d, info = orc.task_step(a[None], touch[None], target[None])
            return o[0], r[0], int(info[0]), int(orc.get_task()[0][17])

        def reset_fn(tape):
            orc.set_tape(tape)
            obs = orc.reset(seed=0)
            tk = orc.get_task()[0]
            return obs[0], int(tk[17]), int(tk[16]), tk[12:14], orc.get_terrain()[0][:120].reshape(20, 6)

        _stale_replay(sg, name, step_fn, reset_fn, tol)
    # flag cleared: a clean start whatever the old episode ended on
    m2 = M.compile_walker3d(M.TASK_WALKER3D_STEPPER)
    m2.task_flags &= ~M.TASKF_STALE_RESET_CONTACTS
    orc = Oracle(m2.to_bytes(), M.TASK_WALKER3D_STEPPER, 1, prec)
    orc.set_tape(sg["stale_count0_tape_a"]); orc.reset(seed=0)
    full = np.zeros((1, orc.state_dim)); full[0, :55] = sg["stale_count0_states"][0]
    orc.set_state(full)
    orc.task_step(sg["stale_count0_actions"][0][None], np.array([[1, 1]], np.int32), np.array([[1, 0]], np.int32))
    orc.set_tape(sg["stale_count0_tape_b"]); orc.reset(seed=0)
    tk = orc.get_task()[0]
    assert int(tk[17]) == 0 and (tk[12:14] == 0).all()
