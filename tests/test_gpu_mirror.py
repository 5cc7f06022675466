"""Mirror symmetry of the dynamics THROUGH libmocca_hip.so (the CPU oracle's side: tests/test_mirror_oracle.py, which explains T1 / T2).

4096 envs, one physics substep and one full env.step, states taken from a free-running batch (standing, stepping, falling, lying robots):
  T1  step(reflect(model), S s, a) vs S step(model, s, a): the mirror-image ROBOT in the mirror-image state -- rows keep their order, so the
      two launches run the same instruction stream on sign-flipped data: every contact configuration must agree to fp32 rounding;
  T2  step(model, M s, M a) vs M step(model, s, a) with M from the reference's index sets (robots.py:282-288 via the blob): free flight and
      single-row states to fp32 rounding; multi-row states (Gauss-Seidel visits the mirrored rows in another order) statistically, against
      the f64 oracle's own mirror residual on the same states.
Needs a real MI355X: -m gpu.
"""
import numpy as np
import pytest

from mocca_envs_amd import model as M
from mocca_envs_amd.vec_env import TASKS, compile_model_for

from mirror_util import IndexMirror, obs_mirror, reflect_model, reflect_state, reflect_terrain

pytestmark = pytest.mark.gpu

UNIT = 1e-5     # errors are quoted in units of 1e-5 (1 + |x|), the fp32 yardstick of tests/test_gpu_substep.py


def _model(env_id, substeps_one):
    m = compile_model_for(env_id)
    if substeps_one:
        m.n_substeps = 1
        if TASKS[env_id] == M.TASK_CASSIE:
            m.n_llc = 1
    return m


def _envs(env_id, blob_a, blob_b, n):
    from mocca_envs_amd.vec_env import VecEnv, _DEFAULT_PARAMS
    gen = VecEnv(env_id, n, auto_reset=True, seed=21, model_blob=blob_a)
    a = VecEnv(env_id, n, auto_reset=False, seed=21, model_blob=blob_a)
    b = VecEnv(env_id, n, auto_reset=False, seed=21, model_blob=blob_b)
    for e in (gen, a, b):
        if TASKS[env_id] == M.TASK_WALKER3D_STEPPER:
            e.set_param(2, 9)
        e.reset()
    return gen, a, b


def _flip_task_words(tk, words):
    """Negate float words of the int32 task record (sign bit)."""
    t = tk.clone()
    for w in words:
        t[:, w] = t[:, w] ^ torch_sign_bit()
    return t


def torch_sign_bit():
    return -2147483648


def _units(x, ref):
    return np.abs(x - ref) / (UNIT * (1.0 + np.abs(ref)))


@pytest.mark.parametrize("one_substep", [True, False])
@pytest.mark.parametrize("env_id", ["Walker3DCustomEnv-v0", "Walker3DStepperEnv-v0", "Child3DCustomEnv-v0", "MikeStepperEnv-v0",
                                    "LaikagoCustomEnv-v0", "CassieEnv-v0", "Walker2DCustomEnv-v0:yz", "Crab2DCustomEnv-v0:yz", "Walker3DCustomEnv-v0:yz"])
def test_reflected_world_on_hip(env_id, one_substep):
    """T1 at 4096 envs (Cassie 2048): the kernel is covariant under the reflection of the world, in every contact configuration.  ":yz" = the
    world mirrored in its y-z plane (x -> -x): the planar robots' geometry lies IN the x-z plane, this is their mirror that is not the identity
    (their whole-step parity with the oracle rests on looser bounds than the 3-D robots': two coplanar end spheres share a load the solver
    cannot split uniquely -- here the kernel is held to ITSELF, bit for bit, on exactly those states)."""
    import torch
    env_id, _, plane = env_id.partition(":")
    plane = plane or "xz"
    task = TASKS[env_id]
    n = 2048 if task == M.TASK_CASSIE else 4096
    m = _model(env_id, one_substep)
    nj, nd = m.n_joints, 13 + 2 * m.n_joints
    gen, A, B = _envs(env_id, m.to_bytes(), reflect_model(m, plane).to_bytes(), n)
    dA, dB = A.set_debug(True), B.set_debug(True)
    g = torch.Generator(device="cuda").manual_seed(3)
    rounds = 6 if task == M.TASK_CASSIE else 12
    worst, n_cmp, n_sig_diff, rows_max, n_multi = 0.0, 0, 0, 0, 0
    errs = []
    for r in range(rounds):
        for _ in range(3 if task == M.TASK_CASSIE else 25):
            gen.step(torch.rand(n, gen.act_dim, device="cuda", generator=g) * 2 - 1)
        s = gen.get_state().cpu().numpy()
        tk = gen.get_task()
        A.set_state(s); A.set_task(tk)
        B.set_state(reflect_state(s, nj, plane)); B.set_task(_flip_task_words(tk, [1] if plane == "xz" else [0, 22]))
        if task == M.TASK_WALKER3D_STEPPER:
            ter = gen.get_terrain().cpu().numpy()
            A.set_terrain(ter); B.set_terrain(reflect_terrain(ter))
        a = torch.rand(n, gen.act_dim, device="cuda", generator=g) * 2 - 1
        _, ra, da, _ = A.step(a)
        _, rb, db, _ = B.step(a)
        sa, sb = A.get_state().cpu().numpy(), B.get_state().cpu().numpy()
        ga, gb = dA.cpu().numpy(), dB.cpu().numpy()
        fin = np.isfinite(sa).all(axis=1) & np.isfinite(sb).all(axis=1)
        sig_words = slice(0, 12) if one_substep else slice(16, 19)       # last substep's active set / the whole step's decision signature
        same = (ga[:, sig_words] == gb[:, sig_words]).all(axis=1) & fin
        n_sig_diff += int((~same & fin).sum()); n_cmp += int(fin.sum())
        e = _units(reflect_state(sa, nj, plane)[:, :nd], sb[:, :nd]).max(axis=1)
        errs.append(e[same])
        assert torch.equal(da, db)
        if plane == "xz":
            assert float((ra - rb).abs()[torch.from_numpy(same).cuda()].max()) <= 1e-4
        rows_max = max(rows_max, int(ga[:, 0].max())); n_multi += int(((ga[:, 1] + ga[:, 2]) >= 2).sum())
    errs = np.concatenate(errs)
    print(f"\n{env_id} ({'one substep' if one_substep else 'full step'}): {n_cmp} samples, {n_multi} with >= 2 limit rows / contacts (max {rows_max} rows); "
          f"decision signatures differ in {n_sig_diff}; mirror residual in units of 1e-5 (1 + |x|): median {np.median(errs):.3g} p99 {np.percentile(errs, 99):.3g} max {errs.max():.3g}; "
          f"bit-identical in {100 * float((errs == 0).mean()):.1f} %")
    assert n_multi > n_cmp // 10, "the sample must contain contact-rich states"
    # same instruction stream on sign-flipped data: nothing but the odd asymmetric rounding (libm range reduction, a compare against zero)
    # can differ -- no decision flips beyond a handful, residual far below the fp32 yardstick of the substep parity test
    assert n_sig_diff <= max(2, n_cmp // 5000)
    assert np.percentile(errs, 99) < 1.0 and errs.max() < 30.0
    for e in (gen, A, B):
        e.close()


@pytest.mark.parametrize("one_substep", [True, False])
@pytest.mark.parametrize("env_id", ["Walker3DCustomEnv-v0", "Child3DCustomEnv-v0", "MikeStepperEnv-v0", "Walker3DStepperEnv-v0"])
def test_the_reference_mirror_sets_on_hip(env_id, one_substep):
    """T2 at 4096 envs: step(M s, M a) == M step(s, a) on the compiled blob with the reference's index sets; observation / reward / done through
    get_mirror_indices()'s sets (what SymmetricRL relies on)."""
    import torch
    from mocca_envs_amd import host_logic as H
    from mocca_envs_amd.symmetry import MirrorTransform
    from oracle.oracle import Oracle, PARAM_CURRICULUM
    task = TASKS[env_id]
    n, n_orc = 4096, 256
    m = _model(env_id, one_substep)
    mir = IndexMirror(m)
    nj, nd = m.n_joints, 13 + 2 * m.n_joints
    gen, A, B = _envs(env_id, m.to_bytes(), m.to_bytes(), n)
    dA = A.set_debug(True)
    o1, o2 = Oracle(m.to_bytes(), task, n_orc, "f64"), Oracle(m.to_bytes(), task, n_orc, "f64")
    for o in (o1, o2):
        if task == M.TASK_WALKER3D_STEPPER:
            o.set_param(PARAM_CURRICULUM, 9)
        o.reset(seed=1)
    mt = MirrorTransform(H.mirror_indices(m, stepper=task == M.TASK_WALKER3D_STEPPER), A.obs_dim, A.act_dim, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(5)
    from mocca_envs_amd.vec_env import task_to_float64, task_from_float64
    exact, multi, multi_orc, obs_err, rew_err, done_diff = [], [], [], 0.0, 0.0, 0
    for r in range(12):
        for _ in range(25):
            gen.step(torch.rand(n, gen.act_dim, device="cuda", generator=g) * 2 - 1)
        s = gen.get_state().cpu().numpy()
        tk64 = task_to_float64(gen.get_task())
        A.set_state(s); A.set_task(task_from_float64(tk64))
        B.set_state(mir.state(s)); B.set_task(task_from_float64(mir.task(tk64)))
        if task == M.TASK_WALKER3D_STEPPER:
            ter = gen.get_terrain().cpu().numpy()
            A.set_terrain(ter); B.set_terrain(reflect_terrain(ter))
        a = torch.rand(n, gen.act_dim, device="cuda", generator=g) * 2 - 1
        am = mt.act(a).contiguous()
        dA.zero_()
        oa, ra, da, _ = A.step(a)
        ob, rb, db, _ = B.step(am)
        sa, sb = A.get_state().cpu().numpy(), B.get_state().cpu().numpy()
        ga = dA.cpu().numpy()
        fin = np.isfinite(sa).all(axis=1) & np.isfinite(sb).all(axis=1)
        e = _units(mir.state(sa)[:, :nd], sb[:, :nd]).max(axis=1)
        few = (ga[:, 1] + ga[:, 2]) <= 1
        if not one_substep:                 # a full step: "few" must hold in EVERY substep -- word 15 is the largest row count since the record was zeroed
            few &= (ga[:, 15] <= 1) | ((ga[:, 15] == 3) & (ga[:, 2] == 1) & (ga[:, 1] == 0))
        exact.append(e[fin & few]); multi.append(e[fin & ~few])
        if True:       # Custom AND Stepper: the Stepper's task record mirrors too (feet contact flags, cover masks, terrain: mirror_util.IndexMirror.task)
            sel = torch.from_numpy(fin & few).cuda()
            d = (mt.obs(oa) - ob).abs()[sel]
            obs_err = max(obs_err, float(d.max()) if d.numel() else 0.0)
            rew_err = max(rew_err, float((ra - rb).abs()[sel].max()) if sel.any() else 0.0)
            done_diff += int((da != db)[sel].sum())
        # the f64 oracle's own mirror residual on (a subset of) the same states
        ss = s[:n_orc].astype(np.float64)
        o1.set_state(ss); o1.set_task(tk64[:n_orc]); o2.set_state(mir.state(ss)); o2.set_task(mir.task(tk64[:n_orc]))
        if task == M.TASK_WALKER3D_STEPPER:
            o1.set_terrain(ter[:n_orc, :124]); o2.set_terrain(reflect_terrain(ter[:n_orc])[:, :124])
        an = a[:n_orc].cpu().numpy()
        o1.step(an); o2.step(am[:n_orc].cpu().numpy())
        s1, s2 = o1.get_state(), o2.get_state()
        eo = _units(mir.state(s1)[:, :nd], s2[:, :nd]).max(axis=1)
        multi_orc.append(eo[np.isfinite(eo) & ~few[:n_orc]])
    exact, multi, multi_orc = np.concatenate(exact), np.concatenate(multi), np.concatenate(multi_orc)
    q = lambda x, p: float(np.percentile(x, p)) if len(x) else 0.0
    frac = lambda x: float((x < 10).mean())
    print(f"\n{env_id} ({'one substep' if one_substep else 'full step'}): at most one row (n={len(exact)}): mirror residual median {q(exact, 50):.3g} p99 {q(exact, 99):.3g} "
          f"max {exact.max(initial=0.0):.3g} units of 1e-5 (1 + |x|); more rows (n={len(multi)}): {100 * frac(multi):.1f} % below 10 units, p90 {q(multi, 90):.3g} | f64 oracle on {len(multi_orc)} of "
          f"the same states: {100 * frac(multi_orc):.1f} % below 10 units, p90 {q(multi_orc, 90):.3g}; obs {obs_err:.2e} reward {rew_err:.2e} done flips {done_diff}")
    # (Mike on the planks is rarely on fewer than two rows in a substep, and never through a whole env.step)
    assert len(multi) > 10000 and (len(exact) > (150 if "Mike" in env_id else (2000 if one_substep else 500)) or ("Mike" in env_id and not one_substep))
    # free flight / one row: the mirrored state runs the same arithmetic on other lanes -- fp32 rounding, the yardstick of the substep test
    if len(exact):
        assert q(exact, 50) < 1.0 and q(exact, 99) < 10.0 and exact.max() < (30.0 if one_substep else 300.0)
    assert obs_err < 5e-3 and rew_err < 5e-2 and done_diff <= 2
    # more rows: Gauss-Seidel's visiting order is not mirrored; HIP must be as (a)symmetric as the exact algorithm is
    assert abs(frac(multi) - frac(multi_orc)) < 0.08
    assert q(multi, 90) < 3 * q(multi_orc, 90) + 10
    for e in (gen, A, B):
        e.close()
