"""The reference's own scripted episodes replayed through the HIP task layer (mocca_task_step, mocca_set_draw_tape).

tests/golden/*.npz hold what the reference's real classes computed (tests/golden/make_golden*.py import them from
/root/reference): states handed to `calc_state`, contact query results, actions, and the observations / rewards / done
flags / targets / plank moves that came out.  tests/test_golden_*.py check the CPU oracle against them; here the very same
numbers go through libmocca_hip.so on the GPU: the dynamic state of frame t is written with mocca_set_state, the foot /
target / body contact flags are injected, np_random's uniforms are fed through the draw tape, and the kernel's task
layer (same source as mocca_step, zero substeps) must reproduce the reference within fp32 rounding.

Covers on the GPU what random flailing never reaches: Stepper steps up to index 19, both stop windows, the > 120-frame
release, plank recycling, last-step bonus (env_locomotion.py:632-693, 472-479), the in-kernel re-target (:214-222),
eval mode, the planar envs' forced done = False, Laikago's body-contact termination, Child3D's 0.1 m line.
Needs a real MI355X: -m gpu.  f32 tolerances = those of the f32 oracle in tests/test_golden_task.py.
"""
import os

import numpy as np
import pytest

from mocca_envs_amd import model as M

pytestmark = pytest.mark.gpu

TOL = 3e-5          # fp32 task arithmetic (the oracle's f32 build is held to 2e-5 .. 3e-5)
OBS_TOL = 1e-4      # observations are O(1) (clipped to +-5): a few ulp of fp32 through atan2 / asin / the heading rotation
# reward = progress + ...: progress is the difference of two potentials of O(300) (distance * 60 Hz), one fp32 ulp of which is 3e-5;
# measured worst case of the f32 oracle over the same goldens: 6.4e-5 (Custom), 2.0e-4 (Stepper).  At this tolerance a wrong weight
# fails the replay: joints_at_limit_cost 0.1 -> 0.09 moves the reward by 0.01 per joint at its limit, stall_torque_cost by ~0.03.
REW_TOL = 5e-4
REPL = 3            # replicas of the episode in the batch: every wave must produce the same bits


@pytest.fixture(scope="module")
def vg():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "variants_reference.npz"), allow_pickle=False)


def _env(env_id, tape, params=()):
    import torch
    from mocca_envs_amd.vec_env import VecEnv
    env = VecEnv(env_id, REPL, auto_reset=False, seed=0)
    for pid, val in params:
        env.set_param(pid, val)
    env.set_draw_tape(np.tile(np.asarray(tape, np.float32)[None], (REPL, 1)))
    return env


def _set_state(env, st):
    full = np.zeros((REPL, env.state_dim), np.float32)
    full[:, :len(st)] = st
    env.set_state(full)


def _same_in_every_replica(*tensors):
    for x in tensors:
        x = x.cpu().numpy()
        assert all(np.array_equal(x[0], x[k], equal_nan=True) for k in range(1, REPL)), "replicas differ"


def _replay_custom(env, g, sd, n_feet=2, body=None, check_done=True):
    """Teacher-forced replay of one scripted Custom-task episode; returns the per-step task records."""
    import torch
    from mocca_envs_amd.vec_env import task_to_float64
    states, touch, actions = g("states"), g("touch"), g("actions")
    recs = []
    for t in range(len(states)):
        _set_state(env, states[t][:sd])
        a = torch.from_numpy(np.tile(actions[t][None].astype(np.float32), (REPL, 1)))
        tc = np.tile(np.asarray(touch[t], np.int32).reshape(1, n_feet), (REPL, 1))
        bd = None if body is None else np.full(REPL, int(body[t]), np.int32)
        o, r, d, _ = env.task_step(a, tc, None, bd)
        _same_in_every_replica(o, r, d)
        o, r, d = o.cpu().numpy()[0], float(r[0]), int(d[0])
        want_o = g("obs")[t]
        fin = np.isfinite(want_o)
        np.testing.assert_allclose(o[fin], want_o[fin], atol=OBS_TOL, err_msg=f"t{t} obs")
        if check_done:
            assert (d & 1) == int(g("done")[t]), f"t{t} done"
        if np.isfinite(g("rew")[t]):
            np.testing.assert_allclose(r, g("rew")[t], atol=REW_TOL, err_msg=f"t{t} reward")
        recs.append(task_to_float64(env.get_task())[0])
    return recs


def test_custom_env_episodes_on_the_gpu(golden):
    """4 scripted Walker3DCustomEnv episodes (incl. evaluation mode and target re-randomisation) -- env_locomotion.py:79-222."""
    from mocca_envs_amd import lib as L
    from mocca_envs_amd.vec_env import task_to_float64
    retargets = 0
    for ep in range(int(golden["custom_n_episodes"])):
        g = lambda k: golden[f"custom_ep{ep}_{k}"]
        env = _env("Walker3DCustomEnv-v0", g("tape"), [(L.PARAM_EVAL_MODE, int(g("eval_mode")))])
        obs0 = env.reset().cpu().numpy()
        st = env.get_state().cpu().numpy()
        tk = task_to_float64(env.get_task())
        np.testing.assert_allclose(st[0, 13:34], g("reset_q"), atol=TOL, err_msg=f"ep{ep} reset pose")
        assert int(tk[0, 11]) == int(g("reset_mirrored"))
        np.testing.assert_allclose(tk[0, 0:3], g("reset_walk_target"), atol=TOL)
        assert tk[0, 6] == float(g("reset_stop_frames"))
        np.testing.assert_allclose(obs0[0, 1:], g("reset_obs")[1:], atol=10 * TOL, err_msg=f"ep{ep} reset obs")
        recs = _replay_custom(env, g, 55)
        for t, tk in enumerate(recs):
            np.testing.assert_allclose(tk[0:3], g("walk_target")[t], atol=10 * TOL, err_msg=f"ep{ep} t{t} target")
            assert int(tk[5]) == int(g("close_count")[t]), f"ep{ep} t{t} close_count"
        wt = g("walk_target")
        if not int(g("eval_mode")):
            retargets += int((np.abs(np.diff(wt[:, :2], axis=0)).max(axis=1) > 1e-9).sum())
        env.close()
    assert retargets >= 1, "the scripts must drive the in-kernel re-target (env_locomotion.py:214-222) at least once"


@pytest.mark.parametrize("field,factor", [("joints_at_limit_cost", 0.9), ("stall_torque_cost", 0.87), ("electricity_cost", 0.99)])
def test_the_reward_tolerance_catches_a_wrong_weight(golden, field, factor):
    """Mutation check of REW_TOL: the same replay with ONE reward weight off by 10 % (joints_at_limit_cost 0.1 -> 0.09), by 0.03
    (stall_torque_cost 0.225 -> 0.196) or by 1 % (electricity_cost) must FAIL on the GPU.  At the former tolerance of 4e-2 the first
    two passed."""
    import torch
    from mocca_envs_amd import lib as L
    from mocca_envs_amd.vec_env import VecEnv
    g = lambda k: golden[f"custom_ep0_{k}"]
    m = M.compile_walker3d()
    setattr(m, field, getattr(m, field) * factor)
    env = VecEnv("Walker3DCustomEnv-v0", REPL, auto_reset=False, seed=0, model_blob=m.to_bytes())
    env.set_param(L.PARAM_EVAL_MODE, int(g("eval_mode")))
    env.set_draw_tape(np.tile(np.asarray(g("tape"), np.float32)[None], (REPL, 1)))
    env.reset()
    states, touch, actions = g("states"), g("touch"), g("actions")
    worst = 0.0
    for t in range(len(states)):
        _set_state(env, states[t][:55])
        a = torch.from_numpy(np.tile(actions[t][None].astype(np.float32), (REPL, 1)))
        _, r, _, _ = env.task_step(a, np.tile(np.asarray(touch[t], np.int32).reshape(1, 2), (REPL, 1)), None, None)
        if np.isfinite(g("rew")[t]):
            worst = max(worst, abs(float(r[0]) - float(g("rew")[t])))
    print(f"{field} x {factor}: worst reward error {worst:.3e} (tolerance {REW_TOL:.0e})")
    assert worst > 2 * REW_TOL, f"a wrong {field} would pass the golden replay"
    env.close()


def test_stepper_env_episodes_on_the_gpu(golden):
    """3 scripted Walker3DStepperEnv episodes: the foot / target state machine up to the last step, stop windows,
    >120-frame release, plank recycling, step and target bonuses -- env_locomotion.py:472-479, 515-568, 632-693."""
    import torch
    from mocca_envs_amd import lib as L
    from mocca_envs_amd.vec_env import task_to_float64
    m = M.compile_walker3d(M.TASK_WALKER3D_STEPPER)
    seen = dict(max_nsi=0, stops=set(), released=False, recycled=False)
    for ep in range(int(golden["stepper_n_episodes"])):
        g = lambda k: golden[f"stepper_ep{ep}_{k}"]
        env = _env("Walker3DStepperEnv-v0", g("tape"), [(L.PARAM_CURRICULUM, int(g("curriculum")))])
        obs0 = env.reset().cpu().numpy()
        st = env.get_state().cpu().numpy()
        np.testing.assert_allclose(st[0, 13:34], g("reset_q"), atol=TOL)
        np.testing.assert_allclose(st[0, 0:3], g("reset_base"), atol=TOL)
        table = env.get_terrain().cpu().numpy()[0, :120].reshape(20, 6)
        np.testing.assert_allclose(table, g("terrain"), atol=10 * TOL)      # generate_step_placements, :395-441
        tk = task_to_float64(env.get_task())[0]
        assert abs(tk[21] - float(g("applied_gain"))) < 1e-6
        np.testing.assert_allclose(obs0[0, 1:], g("reset_obs")[1:], atol=OBS_TOL)
        states, touch, target, actions = g("states"), g("touch"), g("target"), g("actions")
        prev_trc = 0
        for t in range(len(states)):
            _set_state(env, states[t])
            a = torch.from_numpy(np.tile(actions[t][None].astype(np.float32), (REPL, 1)))
            o, r, d, info = env.task_step(a, np.tile(touch[t].astype(np.int32)[None], (REPL, 1)),
                                          np.tile(target[t].astype(np.int32)[None], (REPL, 1)))
            _same_in_every_replica(o, r, d, info)
            np.testing.assert_allclose(o.cpu().numpy()[0], g("obs")[t], atol=OBS_TOL, err_msg=f"ep{ep} t{t} obs")
            assert (int(d[0]) & 1) == int(g("done")[t]), f"ep{ep} t{t} done"
            np.testing.assert_allclose(float(r[0]), g("rew")[t], atol=REW_TOL, err_msg=f"ep{ep} t{t} reward")
            nsi = int(info[0])
            assert nsi == int(g("next_step_index")[t]), f"ep{ep} t{t} next_step_index"
            ter = env.get_terrain().cpu().numpy()[0]
            pinfo = ter[120:123].astype(int)
            want = g("plank_pos")[t] - np.array([0, 0, m.plank_com_z])
            np.testing.assert_allclose(table[pinfo, 0:3], want, atol=10 * TOL, err_msg=f"ep{ep} t{t} planks")
            # bookkeeping of what the replay exercised on the GPU
            tk = task_to_float64(env.get_task())[0]
            seen["max_nsi"] = max(seen["max_nsi"], nsi)
            if int(tk[18]):
                seen["stops"].add(nsi)
            if prev_trc >= 120 and int(tk[17]) == 0:      # the count passed 120 on a stop step: released (:655-657), index advanced
                seen["released"] = True
            prev_trc = int(tk[17])
            if sorted(pinfo) != [0, 1, 2]:
                seen["recycled"] = True
        env.close()
    assert seen["max_nsi"] == 19, seen                       # walked the whole staircase
    assert {6, 7} & seen["stops"] and {13, 14} & seen["stops"], seen   # both stop windows (:522)
    assert seen["released"] and seen["recycled"], seen       # > 120 frames on a stop step (:655-657); update_steps (:472-479)


def test_child3d_episodes_on_the_gpu(vg):
    for ep in range(int(vg["child_n_episodes"])):
        g = lambda k: vg[f"child_ep{ep}_{k}"]
        env = _env("Child3DCustomEnv-v0", g("tape"))
        env.reset()
        st = env.get_state().cpu().numpy()
        np.testing.assert_allclose(st[0, 13:34], g("reset_q"), atol=TOL, err_msg="crawl pose")
        np.testing.assert_allclose(st[0, 3:7], g("reset_base_quat"), atol=TOL)
        _replay_custom(env, g, 55)
        env.close()


@pytest.mark.parametrize("tag,env_id", [("walker2d", "Walker2DCustomEnv-v0"), ("crab2d", "Crab2DCustomEnv-v0")])
def test_planar_episode_on_the_gpu(vg, tag, env_id):
    g = lambda k: vg[f"{tag}_ep0_{k}"]
    env = _env(env_id, g("tape"))
    obs0 = env.reset().cpu().numpy()
    np.testing.assert_array_equal(obs0[0, -2:], [0.0, 0.0])          # env_locomotion.py:299-300
    nj = env.act_dim
    np.testing.assert_allclose(env.get_state().cpu().numpy()[0, 13:13 + nj], g("reset_q"), atol=TOL)
    recs = _replay_custom(env, g, 13 + 2 * nj, check_done=False)
    assert all(int(tk[7]) == 0 for tk in recs) and (g("done") == 0).all()    # :302-309: never done
    env.close()


def test_laikago_episodes_on_the_gpu(vg):
    from mocca_envs_amd import lib as L
    for ep in range(int(vg["laikago_n_episodes"])):
        g = lambda k: vg[f"laikago_ep{ep}_{k}"]
        env = _env("LaikagoCustomEnv-v0", g("tape"), [(L.PARAM_RANDOM_POSE, 0)])
        env.reset()
        st = env.get_state().cpu().numpy()
        np.testing.assert_allclose(st[0, 13:25], g("reset_q"), atol=TOL)
        np.testing.assert_allclose(st[0, 0:3], g("reset_base_pos"), atol=TOL)
        _replay_custom(env, g, 13 + 24, n_feet=4, body=g("body"))
        assert g("done")[-1] == 1 and g("terms")[-1, 3] == -1      # the script ends on the body contact (:880-890)
        env.close()


def test_mike_reset_on_the_gpu(vg):
    from mocca_envs_amd import lib as L
    from mocca_envs_amd.vec_env import task_to_float64
    for ep in range(int(vg["mike_n_episodes"])):
        g = lambda k: vg[f"mike_ep{ep}_{k}"]
        env = _env("MikeStepperEnv-v0", g("tape"), [(L.PARAM_CURRICULUM, int(g("curriculum")))])
        obs0 = env.reset().cpu().numpy()
        st = env.get_state().cpu().numpy()
        np.testing.assert_allclose(st[0, 13:34], g("reset_q"), atol=TOL)
        np.testing.assert_allclose(st[0, 0:3], g("reset_base_pos"), atol=TOL)
        np.testing.assert_allclose(env.get_terrain().cpu().numpy()[0, :120].reshape(20, 6), g("terrain"), atol=10 * TOL)
        tk = task_to_float64(env.get_task())[0]
        assert abs(tk[21] - float(g("applied_gain"))) < 1e-6 and int(tk[11]) == int(g("reset_mirrored"))
        np.testing.assert_allclose(obs0[0, 1:], g("reset_obs")[1:], atol=OBS_TOL)
        env.close()


def test_cassie_steps_against_the_reference_code_on_the_gpu():
    """tests/golden/make_golden_cassie.py ran the reference's real Cassie / CassieEnv methods over the f64 oracle's physics
    and recorded the state before every env.step.  The GPU restarts each step from that state (teacher forcing) and
    runs its own 50-iteration PD + physics loop in fp32: observation, reward and done must match the reference's."""
    import torch
    from mocca_envs_amd.vec_env import VecEnv, task_to_float64, task_from_float64
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cassie_reference.npz"))
    env = VecEnv("CassieEnv-v0", REPL, auto_reset=False, seed=0)
    obs0 = env.reset().cpu().numpy()
    np.testing.assert_allclose(obs0[0], g["ep0_obs"][0], atol=1e-5)
    from oracle.oracle import Oracle
    orc = Oracle(env.model.to_bytes(), M.TASK_CASSIE, 1, "f32")
    orc.reset(seed=0)
    errs, errs32, rerr, rerr32 = [], [], [], []
    for ep in range(2):
        env.reset()
        for t, a in enumerate(g[f"ep{ep}_actions"]):
            _set_state(env, g[f"ep{ep}_pre_state"][t])
            tk = task_to_float64(env.get_task())
            tk[:, 24:38] = g[f"ep{ep}_pre_jvel"][t]
            tk[:, 3] = g[f"ep{ep}_pre_potential"][t]
            tk[:, 7] = 0
            env.set_task(task_from_float64(tk))
            o, r, d, _ = env.step(torch.from_numpy(np.tile(a[None].astype(np.float32), (REPL, 1))).cuda())
            _same_in_every_replica(o, r, d)
            o = o.cpu().numpy()[0]
            want = g[f"ep{ep}_obs"][t + 1]
            errs.append(np.abs(o - want) / (1e-3 + 1e-3 * np.abs(want)))
            rerr.append(abs(float(r[0]) - g[f"ep{ep}_rew"][t]))
            assert bool(int(d[0]) & 1) == bool(g[f"ep{ep}_done"][t]), (ep, t)
            # the yardstick: the scalar f32 oracle from the same state, same task record -- what fp32 arithmetic itself costs over
            # the 50 stiff PD + physics iterations of one env.step (the golden comes from f64 physics)
            st = np.zeros((1, orc.state_dim)); st[0, :len(g[f"ep{ep}_pre_state"][t])] = g[f"ep{ep}_pre_state"][t]
            orc.set_state(st); orc.set_task(tk[:1])
            oc, rc, dc, _ = orc.step(a[None].astype(np.float32))
            errs32.append(np.abs(oc[0] - want) / (1e-3 + 1e-3 * np.abs(want)))
            rerr32.append(abs(float(rc[0]) - g[f"ep{ep}_rew"][t]))
    e, e32 = np.concatenate(errs), np.concatenate(errs32)
    q = lambda x, p: float(np.percentile(x, p))
    print(f"\nCassie vs reference-code-over-f64-oracle, one env.step, units of 1e-3 (1 + |x|): GPU median {np.median(e):.3g} p99 {q(e, 99):.3g} "
          f"max {e.max():.3g} | f32 oracle median {np.median(e32):.3g} p99 {q(e32, 99):.3g} max {e32.max():.3g}; reward error GPU max "
          f"{max(rerr):.3g}, f32 oracle max {max(rerr32):.3g}")
    # no further from the reference than 3 x the f32 oracle is (floors: 0.02 / 0.2 / 1 units; reward 2e-3)
    assert np.median(e) < 3 * np.median(e32) + 0.02, (np.median(e), np.median(e32))
    assert q(e, 99) < 3 * q(e32, 99) + 0.2, (q(e, 99), q(e32, 99))
    assert e.max() < 3 * e32.max() + 1.0, (e.max(), e32.max())
    assert max(rerr) < 3 * max(rerr32) + 2e-3, (max(rerr), max(rerr32))
    env.close()


@pytest.fixture(scope="module")
def sg():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "steppers_reference.npz"), allow_pickle=False)


def test_laikago_stepper_episodes_on_the_gpu(sg):
    """LaikagoStepperEnv (env_locomotion.py:893-979) through the HIP task layer: four feet on four live planks, two planks of
    look-behind, joint-angle posture penalty, doubled progress, time-based early termination, body contact on a plank."""
    import torch
    from mocca_envs_amd import lib as L
    from mocca_envs_amd.vec_env import task_to_float64
    m = M.compile_laikago(stepper=True)
    seen = dict(max_nsi=0, stops=set(), early=0, recycled=False)
    for ep in range(int(sg["lstep_n_episodes"])):
        g = lambda k: sg[f"lstep_ep{ep}_{k}"]
        env = _env("LaikagoStepperEnv-v0", g("tape"), [(L.PARAM_CURRICULUM, int(g("curriculum"))), (L.PARAM_RANDOM_POSE, 0)])
        assert env.obs_dim == 54
        obs0 = env.reset().cpu().numpy()
        st = env.get_state().cpu().numpy()
        np.testing.assert_allclose(st[0, 13:25], g("reset_q"), atol=TOL)
        np.testing.assert_allclose(st[0, 7:10], g("reset_base_vel"), atol=TOL)
        table = env.get_terrain().cpu().numpy()[0, :120].reshape(20, 6)
        np.testing.assert_allclose(table, g("terrain"), atol=10 * TOL)
        np.testing.assert_allclose(obs0[0, 1:], g("reset_obs")[1:], atol=OBS_TOL)
        states, touch, target, body, actions = g("states"), g("touch"), g("target"), g("body"), g("actions")
        for t in range(len(states)):
            _set_state(env, states[t])
            a = torch.from_numpy(np.tile(actions[t][None].astype(np.float32), (REPL, 1)))
            tile = lambda x: np.tile(np.asarray(x, np.int32).reshape(1, -1), (REPL, 1))
            o, r, d, info = env.task_step(a, tile(touch[t]), tile(target[t]), np.full(REPL, int(body[t]), np.int32))
            _same_in_every_replica(o, r, d, info)
            np.testing.assert_allclose(o.cpu().numpy()[0], g("obs")[t], atol=OBS_TOL, err_msg=f"ep{ep} t{t} obs")
            assert (int(d[0]) & 1) == int(g("done")[t]), f"ep{ep} t{t} done"
            np.testing.assert_allclose(float(r[0]), g("rew")[t], atol=2 * REW_TOL, err_msg=f"ep{ep} t{t} reward")   # progress x 2
            nsi = int(info[0])
            assert nsi == int(g("next_step_index")[t]), f"ep{ep} t{t} next_step_index"
            pinfo = env.get_terrain().cpu().numpy()[0, 120:124].astype(int)
            want = g("plank_pos")[t] - np.array([0, 0, m.plank_com_z])
            np.testing.assert_allclose(table[pinfo, 0:3], want, atol=10 * TOL, err_msg=f"ep{ep} t{t} planks")
            seen["max_nsi"] = max(seen["max_nsi"], nsi)
            if int(task_to_float64(env.get_task())[0][18]):
                seen["stops"].add(nsi)
            seen["recycled"] |= sorted(pinfo) != [0, 1, 2, 3]
            seen["early"] += int(g("done")[t]) and t > 239 and not body[t]
        env.close()
    assert seen["max_nsi"] == 19 and {6, 7} & seen["stops"] and {13, 14} & seen["stops"] and seen["recycled"] and seen["early"] >= 5, seen


def test_random_reward_episode_on_the_gpu(sg):
    """Walker3DStepperEnv(random_reward=True), env_locomotion.py:533-547: the kernel draws eight weights per step; fed the
    reference's np_random numbers through the tape it must reproduce the reference's rewards."""
    import torch
    from mocca_envs_amd import lib as L
    from mocca_envs_amd.vec_env import task_to_float64
    g = lambda k: sg[f"rr_ep0_{k}"]
    env = _env("Walker3DStepperEnv-v0", g("tape"), [(L.PARAM_CURRICULUM, int(g("curriculum"))), (L.PARAM_RANDOM_REWARD, 1)])
    env.reset()
    states, touch, target, actions, terms = g("states"), g("touch"), g("target"), g("actions"), g("terms")
    for t in range(len(states)):
        _set_state(env, states[t])
        a = torch.from_numpy(np.tile(actions[t][None].astype(np.float32), (REPL, 1)))
        tile = lambda x: np.tile(np.asarray(x, np.int32).reshape(1, -1), (REPL, 1))
        o, r, d, info = env.task_step(a, tile(touch[t]), tile(target[t]))
        _same_in_every_replica(o, r, d, info)
        np.testing.assert_allclose(float(r[0]), g("rew")[t], atol=REW_TOL, err_msg=f"t{t} reward")
        w = task_to_float64(env.get_task())[0][30:38]
        np.testing.assert_allclose(w @ terms[t], g("rew")[t], atol=2e-4)       # the weights in the task record are the reference's draws
        assert int(info[0]) == int(g("next_step_index")[t])
    assert int(task_to_float64(env.get_task())[0][10]) == 122 + 8 * len(states)
    # mode 2: the host writes the weights (the single-env class keeps np_random on the host): all ones = the plain reward, all twos = twice it
    from mocca_envs_amd.vec_env import task_from_float64
    tk0 = task_to_float64(env.get_task())
    rews = {}
    for mode, w in ((0, 1.0), (2, 1.0), (2, 2.0)):
        env.set_param(L.PARAM_RANDOM_REWARD, mode)
        tk = tk0.copy(); tk[:, 30:38] = w
        env.set_task(task_from_float64(tk))
        _set_state(env, states[-1])
        o, r, d, info = env.task_step(a, tile(touch[-1]), tile(target[-1]))
        rews[(mode, w)] = float(r[0])
    assert abs(rews[(2, 1.0)] - rews[(0, 1.0)]) < 1e-4 and abs(rews[(2, 2.0)] - 2 * rews[(0, 1.0)]) < 2e-4, rews
    env.close()


def test_reset_reads_the_contacts_of_the_episode_before_on_the_gpu(sg):
    """Walker3DStepperEnv.reset -> calc_feet_state() on Bullet's stale manifolds (env_locomotion.py:484-499, MOCCA_TASKF_STALE_RESET_CONTACTS):
    the reference's own class over a client that keeps answering with its last frame's contacts, replayed through the HIP task layer and the
    reset kernel (tests/test_golden_steppers.py holds the oracle's replay of the same records)."""
    import torch
    from mocca_envs_amd import lib as L
    from mocca_envs_amd.vec_env import task_to_float64
    from test_golden_steppers import _stale_replay
    tile = lambda x: np.tile(np.asarray(x, np.int32).reshape(1, -1), (REPL, 1))
    for name in [str(n) for n in sg["stale_names"]]:
        env = _env("Walker3DStepperEnv-v0", sg[f"stale_{name}_tape_a"], [(L.PARAM_CURRICULUM, int(sg[f"stale_{name}_curriculum"]))])
        env.reset()

        def step_fn(st, a, touch, target):
            _set_state(env, st)
            o, r, d, info = env.task_step(torch.from_numpy(np.tile(a[None].astype(np.float32), (REPL, 1))), tile(touch), tile(target))
            _same_in_every_replica(o, r, d, info)
            return o.cpu().numpy()[0], float(r[0]), int(info[0]), int(task_to_float64(env.get_task())[0][17])

        def reset_fn(tape):
            env.set_draw_tape(np.tile(np.asarray(tape, np.float32)[None], (REPL, 1)))
            obs = env.reset().cpu().numpy()
            tk = task_to_float64(env.get_task())[0]
            return obs[0], int(tk[17]), int(tk[16]), tk[12:14], env.get_terrain().cpu().numpy()[0][:120].reshape(20, 6)

        _stale_replay(sg, name, step_fn, reset_fn, TOL)
        env.close()


@pytest.mark.parametrize("tag,env_id", [("planner", "Walker3DPlannerEnv-v0"), ("mikeplanner", "MikePlannerEnv-v0")])
def test_planner_env_episodes_on_the_gpu(tag, env_id):
    """Walker3DPlannerEnv / MikePlannerEnv (env_locomotion.py:982-1133) through the HIP task layer: reset (robot.reset draws, then the
    target on the height field), progress reward, feet_contact pinned to 0, and the three ways an episode ends -- relative torso height
    below 0.5, z < -5, the torso link touching something.  The controller's value term of the reward is the caller's (envs.py)."""
    import torch
    from mocca_envs_amd.vec_env import task_to_float64
    pg = np.load(os.path.join(os.path.dirname(__file__), "golden", "planner_reference.npz"), allow_pickle=False)
    ends = 0
    for ep in range(int(pg[f"{tag}_n_episodes"])):
        g = lambda k: pg[f"{tag}_ep{ep}_{k}"]
        env = _env(env_id, g("tape"))
        obs0 = env.reset().cpu().numpy()
        st, tk = env.get_state().cpu().numpy(), task_to_float64(env.get_task())
        np.testing.assert_allclose(st[0, 13:34], g("reset_q"), atol=TOL)
        np.testing.assert_allclose(st[0, 0:3], g("reset_base_pos"), atol=TOL)
        assert int(tk[0, 11]) == int(g("reset_mirrored"))
        np.testing.assert_allclose(tk[0, 0:3], g("reset_walk_target"), atol=TOL)          # z = get_height_at on the device copy of the grid
        np.testing.assert_allclose(obs0[0, 1:], g("reset_obs")[1:], atol=OBS_TOL)
        for t in range(len(g("states"))):
            _set_state(env, g("states")[t])
            a = torch.from_numpy(np.tile(g("base_actions")[t][None].astype(np.float32), (REPL, 1)))
            o, r, d, _ = env.task_step(a, np.ones((REPL, 2), np.int32), None, np.full(REPL, int(g("torso_touch")[t]), np.int32))
            _same_in_every_replica(o, r, d)
            np.testing.assert_allclose(o.cpu().numpy()[0], g("obs")[t], atol=OBS_TOL, err_msg=f"ep{ep} t{t} obs")
            np.testing.assert_allclose(float(r[0]), g("progress")[t], atol=REW_TOL, err_msg=f"ep{ep} t{t} progress")
            assert bool(int(d[0]) & 1) == bool(g("done")[t]), (ep, t)
            ends += int(d[0]) & 1
        env.close()
    assert ends >= 3
