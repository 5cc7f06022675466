"""Edge cases of the HIP path against the oracle: row/contact caps, non-finite state, eval mode, TimeLimit,
odd batch sizes, argument checking.  Needs a real MI355X: -m gpu."""
import numpy as np
import pytest

from mocca_envs_amd import model as M

pytestmark = pytest.mark.gpu
NJ = 21


def _pair(env_id, task, n, seed=2, **kw):
    from mocca_envs_amd.vec_env import VecEnv
    from oracle.oracle import Oracle
    env = VecEnv(env_id, n, auto_reset=False, seed=seed, **kw)
    env.set_param(10, 1)   # MOCCA_PARAM_PERSIST_IMPULSES: the last substep's normal impulses stay readable through get_state (the oracle always keeps them)
    orc = Oracle(env.model.to_bytes(), task, n, "f32")
    env.reset(); orc.reset(seed=seed)
    return env, orc


def _sync(env, orc, task=0):
    from mocca_envs_amd.vec_env import task_from_float64
    env.set_state(orc.get_state().astype(np.float32))
    env.set_task(task_from_float64(orc.get_task()))
    if task:
        ter = np.zeros((env.n_envs, 128), np.float32); ter[:, :124] = orc.get_terrain(); env.set_terrain(ter)


def _close(a, b, tol=5.0):
    e = np.abs(a - b) / (1e-3 + 1e-3 * np.abs(b))
    return np.nanmax(e) < tol, np.nanmax(e)


def test_contact_and_row_caps_are_applied_identically():
    """A robot lying 2-4 cm above the plane touches with > 12 points: both sides keep the same max_contacts = 12
    (terrain slot order) and stay inside the 48-row budget (12 contacts + limit rows) in the first substeps."""
    import torch
    env, orc = _pair("Walker3DCustomEnv-v0", 0, 16)
    rng = np.random.default_rng(0)
    st = np.zeros_like(orc.get_state())
    for e in range(16):
        st[e, 2] = 0.02 + 0.02 * rng.random()      # (contacts open within millimetres: the body has to be IN the ground to touch everywhere)
        pitch = np.pi / 2 * (1 if e % 2 else -1) + rng.normal(0, 0.02)  # face down / face up
        st[e, 3:7] = [0, np.sin(pitch / 2), 0, np.cos(pitch / 2)]
        st[e, 13:13 + NJ] = rng.uniform(-0.05, 0.05, NJ)
    # the caps are really exercised: first substep of the last env has 12 contacts
    orc.set_state(st)
    orc.physics_substeps(15, np.zeros(NJ), 1)
    assert len(orc.last_contacts()) == 12 and orc.last_rows() >= 36
    orc.set_state(st)
    a = rng.uniform(-1, 1, (16, NJ)).astype(np.float32)
    _sync(env, orc)
    og, rg, dg, _ = env.step(torch.from_numpy(a).cuda())
    oc, rc, dc, _ = orc.step(a)
    sg, sc = env.get_state().cpu().numpy(), orc.get_state()
    assert np.isfinite(sg).all()
    e = np.abs(sg[:, :55] - sc[:, :55]) / (1e-3 + 1e-3 * np.abs(sc[:, :55]))
    # violent 5 cm push-out at erp 0.9: fp32 noise is amplified, but a wrong row set would be off by O(1000)
    assert np.median(e.max(axis=1)) < 2.0 and e.max() < 200.0, (np.median(e.max(axis=1)), e.max())
    np.testing.assert_array_equal(dg.cpu().numpy(), dc)
    # warm-start impulses of the kept slots agree, dropped slots are zero on both sides
    np.testing.assert_array_equal(sg[:, 55:] != 0, sc[:, 55:] != 0)


def test_non_finite_state_sets_done_and_auto_reset_recovers():
    """env_locomotion.py:205-207: a non-finite observation is not an error, it ends the episode."""
    import torch
    from mocca_envs_amd.vec_env import VecEnv
    env = VecEnv("Walker3DCustomEnv-v0", 8, auto_reset=False, seed=1)
    env.reset()
    st = env.get_state().cpu().numpy()
    st[3, 13 + NJ + 4] = np.nan      # one joint speed of env 3
    st[5, 6] = np.nan                # base orientation of env 5 (an infinite height or speed is clipped to 5: finite)
    env.set_state(st)
    obs, rew, done, _ = env.step(torch.zeros(8, NJ, device="cuda"))
    d = done.cpu().numpy()
    assert d[3] & 1 and d[5] & 1 and not (d[[0, 1, 2, 4, 6, 7]] & 1).any()
    assert not torch.isfinite(obs[3]).all()
    env.set_param(0, 1)              # auto-reset on: the poisoned envs come back finite
    obs, rew, done, _ = env.step(torch.zeros(8, NJ, device="cuda"))
    assert torch.isfinite(obs).all() and torch.isfinite(env.get_state()).all()
    env.close()


def test_eval_mode_and_time_limit():
    import torch
    from mocca_envs_amd.vec_env import task_to_float64, task_from_float64
    from oracle.oracle import PARAM_EVAL_MODE
    env, orc = _pair("Walker3DCustomEnv-v0", 0, 4)
    env.set_param(1, 1); orc.set_param(PARAM_EVAL_MODE, 1)
    env.reset(); orc.reset(seed=2)
    np.testing.assert_allclose(task_to_float64(env.get_task())[:, 14:16], [[4, 0]] * 4)   # dist 4, angle 0
    a = np.zeros((4, NJ), np.float32)
    _sync(env, orc)
    og, _, _, _ = env.step(torch.from_numpy(a).cuda()); oc, _, _, _ = orc.step(a)
    tg, tc = task_to_float64(env.get_task()), orc.get_task()
    np.testing.assert_allclose(tg[:, 0:3], tc[:, 0:3], atol=1e-5)                         # target = prev x + 4
    assert np.allclose(tg[:, 0], 4.0, atol=0.05) and np.allclose(tg[:, 1], 0.0)
    # TimeLimit: max_episode_steps = 1000 (reference __init__.py:55) -> done bit1
    tk = orc.get_task(); tk[:, 8] = 999; orc.set_task(tk)
    _sync(env, orc)
    _, _, dg, _ = env.step(torch.from_numpy(a).cuda()); _, _, dc, _ = orc.step(a)
    assert (dg.cpu().numpy() & 2).all() and (dc & 2).all()


@pytest.mark.parametrize("cur", [0, 4])
def test_stepper_standing_on_planks(cur):
    """Feet on the first plank (soft-contact rows with stiffness/damping erp/cfm, friction 1.2)."""
    import torch
    from oracle.oracle import PARAM_CURRICULUM
    from mocca_envs_amd.vec_env import VecEnv
    from oracle.oracle import Oracle
    env = VecEnv("Walker3DStepperEnv-v0", 32, auto_reset=False, seed=4)
    orc = Oracle(env.model.to_bytes(), 1, 32, "f32")
    env.set_param(2, cur); orc.set_param(PARAM_CURRICULUM, cur)
    env.reset(); orc.reset(seed=4)
    st = orc.get_state(); st[:, 13:13 + NJ] = 0; st[:, 2] = 1.27                      # T-pose, feet on the plank ...
    lo, hi = M.joint_limits(env.model)
    st[:, 13:13 + NJ] += 0.05 * (lo > -1e-6) - 0.05 * (hi < 1e-6)   # ... with knees and elbows 0.05 rad inside their stops: a joint exactly AT its
    orc.set_state(st)                                              # limit switches its row on or off with the last bit (limit_at_violation)
    rng = np.random.default_rng(3)
    touched = 0
    for t in range(12):
        _sync(env, orc, 1)
        a = (0.2 * rng.uniform(-1, 1, (32, NJ))).astype(np.float32)
        og, rg, dg, ig = env.step(torch.from_numpy(a).cuda())
        oc, rc, dc, ic = orc.step(a)
        touched += int(oc[:, 48:50].sum())
        ok, e = _close(env.get_state().cpu().numpy()[:, :55], orc.get_state()[:, :55], tol=30.0)
        assert ok, (t, e)
        np.testing.assert_array_equal(ig.cpu().numpy(), ic)
    assert touched > 100   # the feet really were in contact with the planks


@pytest.mark.parametrize("n", [1, 3, 65])
def test_odd_batch_sizes(n):
    import torch
    env, orc = _pair("Walker3DCustomEnv-v0", 0, n)
    a = np.random.default_rng(n).uniform(-1, 1, (n, NJ)).astype(np.float32)
    _sync(env, orc)
    og, rg, dg, _ = env.step(torch.from_numpy(a).cuda()); oc, rc, dc, _ = orc.step(a)
    np.testing.assert_allclose(og.cpu().numpy(), oc, atol=5e-3, rtol=5e-3)
    np.testing.assert_array_equal(dg.cpu().numpy(), dc)


def test_argument_checking_and_action_handling():
    import torch
    from mocca_envs_amd.vec_env import VecEnv
    env = VecEnv("Walker3DCustomEnv-v0", 4, auto_reset=False, seed=0)
    env.reset()
    with pytest.raises(ValueError):
        env.step(torch.zeros(4, 20, device="cuda"))
    with pytest.raises(KeyError):
        VecEnv("Monkey3DCustomEnv-v0", 4)   # a reference id without a GPU stepper
    # out-of-range actions are clipped for the torque (robots.py:33), a float64 / CPU / strided tensor is accepted
    st0 = env.get_state().clone()
    big = torch.full((4, NJ), 7.0, dtype=torch.float64)
    o1 = env.step(big)[0].clone()
    env.set_state(st0)
    o2 = env.step(torch.ones(8, NJ, device="cuda")[::2])[0].clone()
    # same physics (clipped torque); the energy penalty differs but not the observation
    assert torch.allclose(o1, o2, atol=1e-6)
    env.close()


def test_seed_is_a_full_64_bit_key():
    """mocca_set_seed / mocca_reset take the Philox key as uint64: seeds that differ only above bit 53 (where a double cannot tell
    them apart, the weakness of MOCCA_PARAM_SEED) give different episodes, equal seeds give equal ones; VecEnv.seed() re-keys the
    in-kernel draws of step() without touching the state."""
    import torch
    from mocca_envs_amd.vec_env import VecEnv
    big = (1 << 62) + 12345
    obs = {}
    for s in (big, big + 1, big):
        env = VecEnv("Walker3DCustomEnv-v0", 64, auto_reset=True, seed=s)
        obs.setdefault(s, []).append(env.reset().clone())
        env.close()
    assert torch.equal(obs[big][0], obs[big][1]) and not torch.equal(obs[big][0], obs[big + 1][0])
    a = VecEnv("Walker3DCustomEnv-v0", 64, auto_reset=True, seed=7)
    b = VecEnv("Walker3DCustomEnv-v0", 64, auto_reset=True, seed=7)
    a.reset(); b.reset()
    st = a.get_state().clone()
    b.seed(big, rewind=False)                     # same states, other key from now on
    assert torch.equal(b.get_state(), st)
    act = torch.zeros(64, 21, device="cuda")
    differ = False
    for k in range(60):                           # passive ragdolls fall and auto-reset within 60 steps: the reset draws differ
        oa, _, da, _ = a.step(act)
        ob, _, db, _ = b.step(act)
        differ |= not torch.equal(oa, ob)
    assert differ
    # seed(s) rewinds the per-env episode counters (the gym contract: seed, reset -> a reproducible run): after any history,
    # seed(s) + reset() + steps replay what a fresh VecEnv(seed=s) produces, bit for bit, across in-kernel auto-resets
    fresh = VecEnv("Walker3DCustomEnv-v0", 64, auto_reset=True, seed=11)
    a.seed(11)
    o1, o2 = fresh.reset().clone(), a.reset().clone()
    assert torch.equal(o1, o2)
    for k in range(60):
        of, rf, df, _ = fresh.step(act)
        oa, ra, da, _ = a.step(act)
        assert torch.equal(of, oa) and torch.equal(rf, ra) and torch.equal(df, da)
    assert torch.equal(fresh.get_task()[:, 9], a.get_task()[:, 9]) and int(a.get_task()[:, 9].max()) >= 1
    a.close(); b.close(); fresh.close()


@pytest.mark.gpu
def test_issue_priority_thresholds_change_timing_only():
    """MOCCA_PARAM_ISSUE_PRIORITY (include/mocca.h) moves waves between the hardware's issue priorities: same results bit for bit."""
    import torch
    from mocca_envs_amd import lib as _lib
    from mocca_envs_amd.vec_env import VecEnv
    outs = []
    for prio in (None, 2 + 64 * 4 + 4096 * 6, 40 + 64 * 50 + 4096 * 60):
        env = VecEnv("Walker3DStepperEnv-v0", 256, auto_reset=True, seed=5)
        if prio is not None:
            env.set_param(_lib.PARAM_ISSUE_PRIORITY, prio)
        env.reset()
        g = torch.Generator(device="cuda").manual_seed(3)
        for _ in range(40):
            o, r, d, _i = env.step(torch.rand(256, env.act_dim, device="cuda", generator=g) * 2 - 1)
        outs.append((o.clone(), r.clone(), d.clone(), env.get_state().clone()))
        with pytest.raises(Exception):
            env.set_param(_lib.PARAM_ISSUE_PRIORITY, 1 << 18)
        env.close()
    for k in (1, 2):
        for a, b in zip(outs[0], outs[k]):
            assert torch.equal(a, b)


@pytest.mark.parametrize("env_id,steps", [("Walker3DCustomEnv-v0", 80), ("Walker3DStepperEnv-v0", 80), ("LaikagoCustomEnv-v0", 40),
                                          ("CassieEnv-v0", 25), ("CassiePhaseMocca2DEnv-v0", 12)])
def test_terminal_observation_under_auto_reset(env_id, steps):
    """With auto-reset the obs row of a finished env holds the NEXT episode's first observation; the optional terminal-observation
    buffer (mocca_set_terminal_obs_buffer) receives what the reference's step() returns together with done -- the observation of
    the final state (env_locomotion.py:128-141; gym's TimeLimit returns it on truncation too, __init__.py:55).  Bit-for-bit: for
    envs that finish, it equals the observation the same step returns with auto-reset OFF; rows of the other envs are untouched."""
    import torch
    from mocca_envs_amd.vec_env import VecEnv
    n = 256
    a_env = VecEnv(env_id, n, auto_reset=True, seed=9, terminal_obs=True)
    b_env = VecEnv(env_id, n, auto_reset=False, seed=9)
    a_env.reset(); b_env.reset()
    # a TimeLimit truncation (done bit 1) in the sample: a third of the envs start at t = 997
    tk = a_env.get_task()
    tk[::3, 8] = 997
    a_env.set_task(tk)
    g = torch.Generator(device="cuda").manual_seed(3)
    scale = 0.2 if "Cassie" in env_id else 1.0
    sentinel = -123.0
    n_term = n_trunc = 0
    for t in range(steps):
        b_env.set_state(a_env.get_state()); b_env.set_task(a_env.get_task())
        if "Stepper" in env_id:
            b_env.set_terrain(a_env.get_terrain())
        a_env.terminal_obs.fill_(sentinel)
        act = scale * (torch.rand(n, a_env.act_dim, device="cuda", generator=g) * 2 - 1)
        oa, ra, da, _ = a_env.step(act)
        ob, rb, db, _ = b_env.step(act)
        assert torch.equal(da, db) and torch.equal(ra, rb)
        fin = da != 0
        assert torch.equal(oa[~fin], ob[~fin])                                 # running envs: the same observation
        assert torch.equal(a_env.terminal_obs[fin], ob[fin])                   # finished envs: the FINAL state's observation, bit for bit
        assert (a_env.terminal_obs[~fin] == sentinel).all()                    # nobody else's row is written
        if fin.any():
            assert not torch.equal(oa[fin], ob[fin])                           # ... while obs already shows the next episode
        n_term += int((da & 1).ne(0).sum()); n_trunc += int((da == 2).sum())
    assert n_trunc > 0 and (n_term > 0 or "Cassie" in env_id), (n_term, n_trunc)
    # detached: the step runs as before and leaves the old buffer alone
    buf = a_env.terminal_obs
    a_env.keep_terminal_obs(False)
    buf.fill_(sentinel)
    a_env.step(act)
    torch.cuda.synchronize()
    assert (buf == sentinel).all()
    a_env.close(); b_env.close()


def test_scalar_applied_gain_is_ordered_on_the_callers_stream():
    """mocca_set_param(APPLIED_GAIN) has no stream argument: the value reaches the task records through the next call that takes
    a stream, ON that stream (it used to be written on the NULL stream, unordered against steps in flight on a non-blocking
    user stream, which also write the word).  set_robot_params({"applied_gain": g}) acts on the next apply_action, robots.py:33."""
    import torch
    from mocca_envs_amd import lib as L
    from mocca_envs_amd.vec_env import VecEnv, task_to_float64
    n = 64
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        env = VecEnv("Walker3DCustomEnv-v0", n, auto_reset=False, seed=5)
        ref = VecEnv("Walker3DCustomEnv-v0", n, auto_reset=False, seed=5)
        env.reset(); ref.reset()
        act = torch.rand(n, 21, device="cuda") * 2 - 1
        for k in range(20):                                   # steps in flight on the user stream
            env.step(act); ref.step(act)
        env.set_param(L.PARAM_APPLIED_GAIN, 0.5)              # host-side only: nothing is enqueued yet
        ref.set_param_v(L.PARAM_APPLIED_GAIN, np.full(n, 0.5, np.float32))   # the per-env form always took the stream
        env.step(act); ref.step(act)
        s.synchronize()
        assert torch.equal(env.get_state(), ref.get_state())
        np.testing.assert_array_equal(task_to_float64(env.get_task())[:, 21], 0.5)
        # a restored snapshot wins over a pending scalar
        snap = env.get_task()
        env.set_param(L.PARAM_APPLIED_GAIN, 0.9)
        env.set_task(snap)
        np.testing.assert_array_equal(task_to_float64(env.get_task())[:, 21], 0.5)
        # call order wins: scalar, THEN one value per env (VecEnv.set_robot_params reaches this) -- the scalar was still pending when
        # the per-env write went out and the next step's flush overwrote every env with it
        vec = np.linspace(0.7, 1.3, n).astype(np.float32)
        env.set_param(L.PARAM_APPLIED_GAIN, 0.25)
        env.set_param_v(L.PARAM_APPLIED_GAIN, vec)
        env.step(act)
        s.synchronize()
        np.testing.assert_array_equal(task_to_float64(env.get_task())[:, 21].astype(np.float32), vec)
        # ... and per-env, THEN scalar: the scalar wins
        env.set_param(L.PARAM_APPLIED_GAIN, 0.75)
        env.step(act)
        s.synchronize()
        np.testing.assert_array_equal(task_to_float64(env.get_task())[:, 21], 0.75)
        env.close(); ref.close()


def test_planner_env_on_a_random_height_field():
    """mocca_set_heightfield with another grid: a field drawn by host_logic.random_height_field (HeightField.reload(data=None),
    bullet_objects.py:395-441; 64 x 64 points, 2 per metre) replaces the shipped one; HIP and oracle agree on it, robots scattered over it
    stand on THEIR ground (the contact normals follow the slopes), and a grid of the wrong shape is refused."""
    import torch
    from mocca_envs_amd import host_logic as H
    from mocca_envs_amd import lib as L
    from mocca_envs_amd.vec_env import VecEnv, task_from_float64
    from oracle.oracle import Oracle
    n = 128
    field = H.random_height_field(np.random.RandomState(5), (64, 64), 2).reshape(64, 64).astype(np.float32)
    env = VecEnv("Walker3DPlannerEnv-v0", n, auto_reset=False, seed=3)
    env.set_heightfield(field, 2)
    orc = Oracle(env.model.to_bytes(), M.TASK_WALKER3D_PLANNER, n, "f32")
    orc.set_heightfield(field, 2)
    env.reset(); orc.reset(seed=3)
    rng = np.random.default_rng(1)
    st = orc.get_state()
    for e in range(n):
        xy = rng.uniform(-13, 13, 2)
        st[e, 0:2], st[e, 2] = xy, orc.height_at(*xy) + 1.34
    orc.set_state(st)
    worst = 0.0
    for t in range(40):
        env.set_state(orc.get_state().astype(np.float32)); env.set_task(task_from_float64(orc.get_task()))
        a = rng.uniform(-0.3, 0.3, (n, 21)).astype(np.float32)
        og, rg, dg, _ = env.step(torch.from_numpy(a).cuda())
        oc, rc, dc, _ = orc.step(a)
        e = np.abs(env.get_state().cpu().numpy()[:, :55] - orc.get_state()[:, :55]) / (1e-3 + 1e-3 * np.abs(orc.get_state()[:, :55]))
        worst = max(worst, float(np.median(e.max(axis=1))))
    assert worst < 1.0, worst                                   # teacher-forced: fp32 rounding only
    z = orc.get_state()[:, 2] - np.array([orc.height_at(x, y) for x, y in orc.get_state()[:, 0:2]])
    assert (z > 0.5).mean() > 0.5                               # most robots still stand on the slopes they were put on (40 steps of flailing)
    with pytest.raises(L.MoccaError):
        env.set_heightfield(np.zeros((1, 64), np.float32), 2)
    with pytest.raises(L.MoccaError, match="too fine"):       # 40 points per metre: the 14 cm pelvis sphere would span 7 cells each way (window > 4)
        env.set_heightfield(np.zeros((64, 64), np.float32), 40)
    env.set_heightfield(field, 2)                             # a refused grid leaves the handle as it was
    env.step(torch.zeros(n, 21, device="cuda"))
    assert torch.isfinite(env.get_state()).all()
    env.close()


def test_reference_named_setters_on_the_batched_env(golden):
    """VecEnv.set_env_params / set_robot_params / evaluation_mode / get_mirror_indices: the reference's env-level calls (env_base.py:103-118,
    env_locomotion.py:76-77,224-282), scalar or one value per env."""
    from mocca_envs_amd.vec_env import VecEnv, task_to_float64
    n = 32
    env = VecEnv("Walker3DStepperEnv-v0", n, auto_reset=False, seed=2)
    env.set_env_params({"curriculum": 7, "not_an_attribute": 1})
    env.reset()
    assert (task_to_float64(env.get_task())[:, 20] == 7).all()
    cur = (np.arange(n) % 10).astype(np.float32)
    env.set_env_params({"curriculum": cur})
    env.reset()
    np.testing.assert_array_equal(task_to_float64(env.get_task())[:, 20], cur)
    for got, key in zip(env.get_mirror_indices(), ("neg_obs", "right_obs", "left_obs", "neg_act", "right_act", "left_act")):
        np.testing.assert_array_equal(got, golden[f"mirror_stepper_{key}"])
    env.close()
    env = VecEnv("Walker3DCustomEnv-v0", n, auto_reset=False, seed=2)
    env.set_robot_params({"applied_gain": np.linspace(0.8, 1.2, n).astype(np.float32)})
    env.evaluation_mode((np.arange(n) % 2).astype(np.float32))
    env.reset()
    tk = task_to_float64(env.get_task())
    np.testing.assert_allclose(tk[:, 21], np.linspace(0.8, 1.2, n), atol=1e-6)
    assert (tk[1::2, 14] == 4.0).all() and (tk[0::2, 14] != 4.0).all()          # eval mode: dist 4, angle 0 (env_locomotion.py:69-70)
    env.evaluation_mode(False)
    env.set_robot_params({"applied_gain": 1.0})
    env.reset()
    tk = task_to_float64(env.get_task())
    assert (tk[:, 21] == 1.0).all() and (tk[:, 14] != 4.0).all()
    for got, key in zip(env.get_mirror_indices(), ("neg_obs", "right_obs", "left_obs", "neg_act", "right_act", "left_act")):
        np.testing.assert_array_equal(got, golden[f"mirror_custom_{key}"])
    env.close()


def test_abi_errors_of_the_planner_task_and_the_massive_instances():
    """Through the raw C ABI: the planner task refuses to run without its height field and on another tree; a blob that gives mass to the
    intermediate links is ACCEPTED (round 2 returned MOCCA_E_TOPOLOGY) and steps on the ...Massive kernel instance."""
    import ctypes as C
    import torch
    from mocca_envs_amd import lib as L
    from mocca_envs_amd.vec_env import compile_model_for
    lib = L.load()
    obs = torch.zeros(4, 64, device="cuda")

    def create(blob, task):
        buf, h = C.create_string_buffer(blob, len(blob)), C.c_void_p()
        return lib.mocca_create(buf, len(blob), task, 4, torch.cuda.current_device(), C.byref(h)), h

    rc, h = create(compile_model_for("Walker3DPlannerEnv-v0").to_bytes(), M.TASK_WALKER3D_PLANNER)
    assert rc == 0
    assert lib.mocca_reset(h, None, 0, C.c_void_p(obs.data_ptr()), None) == -1                      # MOCCA_E_ARG
    assert b"mocca_set_heightfield" in lib.mocca_last_error(h)
    grid = np.zeros((8, 8), np.float32)
    assert lib.mocca_set_heightfield(h, grid.ctypes.data_as(C.c_void_p), 8, 8, 1.0) == 0
    assert lib.mocca_reset(h, None, 0, C.c_void_p(obs.data_ptr()), None) == 0
    torch.cuda.synchronize()
    lib.mocca_destroy(h)
    rc, h = create(compile_model_for("LaikagoCustomEnv-v0").to_bytes(), M.TASK_WALKER3D_PLANNER)
    assert rc == -3 and not h                                                                       # MOCCA_E_TOPOLOGY
    m = compile_model_for("Walker3DCustomEnv-v0")
    m.mass[1] = 0.1                                                                                  # an intermediate link of the 2-hinge abdomen
    rc, h = create(m.finalize_tables().to_bytes(), M.TASK_WALKER3D_CUSTOM)
    assert rc == 0
    assert lib.mocca_reset(h, None, 0, C.c_void_p(obs.data_ptr()), None) == 0
    act, rew, done = torch.zeros(4, 21, device="cuda"), torch.zeros(4, device="cuda"), torch.zeros(4, dtype=torch.uint8, device="cuda")
    assert lib.mocca_step(h, C.c_void_p(act.data_ptr()), C.c_void_p(obs.data_ptr()), C.c_void_p(rew.data_ptr()), C.c_void_p(done.data_ptr()), None, None) == 0
    torch.cuda.synchronize()
    first = obs.flatten()[:4 * 52].reshape(4, 52)             # the library writes rows of obs_dim = 52 floats
    assert torch.isfinite(first).all() and torch.isfinite(rew).all() and (first[:, 0] > 1.0).all()     # standing: relative height ~1.2
    lib.mocca_destroy(h)


@pytest.mark.parametrize("env_id,n", [("Walker3DStepperEnv-v0", 37), ("CassieEnv-v0", 5)])
def test_host_image_equals_the_separate_reads(env_id, n):
    """VecEnv.step_host / reset_host / observe_host (one upload, one download, one synchronize per call -- what the single-env gym classes
    and a host-side trainer use): the pinned host image holds the same bits as step() + get_state() + get_task() read one by one."""
    import torch
    from mocca_envs_amd.vec_env import VecEnv
    a_env, b_env = VecEnv(env_id, n, auto_reset=True, seed=11), VecEnv(env_id, n, auto_reset=True, seed=11)
    img = b_env.reset_host()
    np.testing.assert_array_equal(a_env.reset().cpu().numpy(), img["obs"])
    np.testing.assert_array_equal(a_env.get_state().cpu().numpy(), img["state"])
    rng = np.random.default_rng(3)
    for t in range(30):
        a = rng.uniform(-1, 1, (n, a_env.act_dim)).astype(np.float32)
        o, r, d, i = a_env.step(torch.from_numpy(a).cuda())
        img = b_env.step_host(a)
        for want, key in ((o, "obs"), (r, "rew"), (d, "done"), (i, "info"), (a_env.get_state(), "state"), (a_env.get_task(), "task")):
            np.testing.assert_array_equal(want.cpu().numpy(), img[key], err_msg=f"{key} t{t}")
        assert b_env.obs.data_ptr() == b_env._dev_views["obs"].data_ptr()          # the public tensors are views of the packed buffer
    np.testing.assert_array_equal(a_env.observe().cpu().numpy(), b_env.observe_host()["obs"])
    a_env.close(); b_env.close()
