"""HIP path vs f32 oracle at the granularity of ONE physics substep, with the active sets of both sides compared.

An env.step() is 4 substeps (Cassie: 50); every substep takes discrete decisions -- which contact slots are within the
margin, which joint-limit rows are built, which rows are clamped -- and a 1-ulp difference that flips one of them is
amplified by the following substeps.  Here every model is compiled with n_substeps = 1 (Cassie: n_llc = 1), every
step is teacher-forced from the oracle's state, and both sides export the active set of the substep (row count, limit
rows, contacts, contact-slot bitmask, limit-candidate bitmask, self-contact count, and the solver's own discrete
decisions: which rows each of the five PGS iterations left ON a bound, folded into a 64-bit signature:
include/mocca.h MOCCA_DBG_*).
  * where the active sets agree -- same rows AND the same clamp pattern in every iteration -- the two sides ran the same
    piecewise-linear map and the new state must agree like arithmetic does: percentiles tied to the f32-vs-f64 yardstick
    and a hard cap of 30 units of 1e-5 (1 + |x|) (Cassie: 3 x the yardstick's own worst sample);
  * the fraction of (env, substep) samples whose ROW sets differ must stay below 1 %;
  * samples with the same rows but another clamp pattern are a normal impulse that is 0 on one side and 1e-9 on the other (two end
    spheres of one flat foot share a load the solver cannot split uniquely: up to 16 % of the planar walkers' substeps, 0.02 % of
    Walker3D's, 42 % of Crab2D's): they are compared too, under the looser bound such a flip can cost (median 100, worst 1000 units).
Needs a real MI355X: -m gpu.
"""
import numpy as np
import pytest

from mocca_envs_amd import model as M

pytestmark = pytest.mark.gpu

CASES = [("Walker3DCustomEnv-v0", M.TASK_WALKER3D_CUSTOM, {}), ("Walker3DStepperEnv-v0", M.TASK_WALKER3D_STEPPER, {}),
         ("Child3DCustomEnv-v0", M.TASK_WALKER3D_CUSTOM, {}), ("MikeStepperEnv-v0", M.TASK_WALKER3D_STEPPER, {}),
         ("Walker2DCustomEnv-v0", M.TASK_WALKER3D_CUSTOM, {}), ("Crab2DCustomEnv-v0", M.TASK_WALKER3D_CUSTOM, {}),
         ("LaikagoCustomEnv-v0", M.TASK_WALKER3D_CUSTOM, {}), ("CassieEnv-v0", M.TASK_CASSIE, {}),
         ("LaikagoStepperEnv-v0", M.TASK_WALKER3D_STEPPER, {}), ("Cassie2DEnv-v0", M.TASK_CASSIE, {}),
         # the other step objects of Walker3DStepperEnv (plank_class, bullet_objects.py:86-97): short box, upright cylinder
         ("Walker3DStepperEnv-v0", M.TASK_WALKER3D_STEPPER, {"plank_class": "Plank"}),
         ("Walker3DStepperEnv-v0", M.TASK_WALKER3D_STEPPER, {"plank_class": "Pillar"}),
         # SURVEY 8 f1 on the HIP path: blobs built by from_pybullet_dump from a record in PyBullet's conventions (other frames: base at
         # the root link's COM, principal-axes inertial frames, fixed links merged) -- the dump -> blob -> kernel chain, before a real
         # file arrives; "massive": the record gives mass and inertia to the nine intermediate links of the multi-hinge joints, which
         # the compiled model keeps massless -- mocca_create selects the TopoWalker3DMassive kernel instance for it
         ("Walker3DCustomEnv-v0", M.TASK_WALKER3D_CUSTOM, {"_dump": "plain"}),
         ("Walker3DCustomEnv-v0", M.TASK_WALKER3D_CUSTOM, {"_dump": "massive"}),
         ("Walker3DStepperEnv-v0", M.TASK_WALKER3D_STEPPER, {"_dump": "massive"}),
         # the planner envs (env_locomotion.py:982-1133): spheres / capsule ends against the triangles of the height field; the envs are
         # scattered over the field after reset (the episodes start on its flat corner platform)
         ("Walker3DPlannerEnv-v0", M.TASK_WALKER3D_PLANNER, {}), ("MikePlannerEnv-v0", M.TASK_WALKER3D_PLANNER, {}),
         # ... and on a STEEP random field (3 x HeightField.reload(data=None)): the wide spheres' 4 x 4-cell search windows (Mike's 23 cm waist,
         # the walker's 14 cm pelvis) meet triangles outside the central 2 x 2 cells
         ("MikePlannerEnv-v0", M.TASK_WALKER3D_PLANNER, {"_steep": True}), ("Walker3DPlannerEnv-v0", M.TASK_WALKER3D_PLANNER, {"_steep": True, "_caps": (32, 10)}),
         # Cassie with mass on the two links its URDF leaves without inertia: the TopoCassieMassive kernel instance
         ("CassieEnv-v0", M.TASK_CASSIE, {"_massive": True}),
         # the solver's warm-start path (the compiled blobs start from zero, as Bullet's multibody contacts do; a record may say otherwise)
         ("Walker3DCustomEnv-v0", M.TASK_WALKER3D_CUSTOM, {"_warm": 0.85}), ("CassieEnv-v0", M.TASK_CASSIE, {"_warm": 0.85}),
         # the pyramid friction path (pybullet's enableConeFriction = 0; the compiled blobs use Bullet's implicit cone)
         ("Walker3DCustomEnv-v0", M.TASK_WALKER3D_CUSTOM, {"_cone": 0}), ("Walker3DStepperEnv-v0", M.TASK_WALKER3D_STEPPER, {"_cone": 0}),
         # limit rows from a predicted gap on (limit_at_violation = 0; the compiled blobs build them at / past the limit only)
         ("Walker3DCustomEnv-v0", M.TASK_WALKER3D_CUSTOM, {"_predict": True}), ("CassieEnv-v0", M.TASK_CASSIE, {"_predict": True}),
         # the COMPACT instance of the step kernel (mocca_r32.hip: 32 rows / 10 contacts per env, the articulated-body view aliased under a
         # 32 x 32 Delassus matrix; mocca_create picks it from the blob's caps) against the oracle with the same caps
         ("Walker3DCustomEnv-v0", M.TASK_WALKER3D_CUSTOM, {"_caps": (32, 10)}), ("Walker3DStepperEnv-v0", M.TASK_WALKER3D_STEPPER, {"_caps": (32, 10)}),
         ("LaikagoStepperEnv-v0", M.TASK_WALKER3D_STEPPER, {"_caps": (32, 10)}), ("Crab2DCustomEnv-v0", M.TASK_WALKER3D_CUSTOM, {"_caps": (32, 10)}),
         ("Walker3DPlannerEnv-v0", M.TASK_WALKER3D_PLANNER, {"_caps": (32, 10)}), ("Walker3DCustomEnv-v0", M.TASK_WALKER3D_CUSTOM, {"_caps": (24, 6), "_warm": 0.85}),
         # the ACCURACY instance (mocca_r64.hip: 64 rows / 20 contacts per env -- Bullet has no cap) against the oracle with the same caps; the 2 cm
         # margins put more contacts on the lying robots, so that row counts beyond 48 are in the sample
         ("Walker3DStepperEnv-v0", M.TASK_WALKER3D_STEPPER, {"_caps": (64, 20)}), ("Walker3DStepperEnv-v0", M.TASK_WALKER3D_STEPPER, {"_caps": (64, 20), "_abs_margin": True}),
         ("Walker3DCustomEnv-v0", M.TASK_WALKER3D_CUSTOM, {"_caps": (64, 20), "_abs_margin": True}), ("CassieEnv-v0", M.TASK_CASSIE, {"_caps": (64, 20)}),
         # Bullet's m_linearSlop (MoccaModel.linear_slop; pybullet contactSlop 1e-5 m, here 20 x that)
         ("Walker3DCustomEnv-v0", M.TASK_WALKER3D_CUSTOM, {"_slop": 2e-4}), ("Walker3DStepperEnv-v0", M.TASK_WALKER3D_STEPPER, {"_slop": 2e-4}),
         # Bullet's alternating sweep direction of the non-contact rows (MoccaModel.sweep_alternate): limit rows (many of them with the predicted-gap
         # law), Cassie's closures + limits, the planar Cassie's closures + planar rows + its hips on their limits
         ("Walker3DCustomEnv-v0", M.TASK_WALKER3D_CUSTOM, {"_alt": True}), ("Walker3DCustomEnv-v0", M.TASK_WALKER3D_CUSTOM, {"_alt": True, "_predict": True}),
         ("CassieEnv-v0", M.TASK_CASSIE, {"_alt": True}), ("Cassie2DEnv-v0", M.TASK_CASSIE, {"_alt": True}),
         ("Walker3DStepperEnv-v0", M.TASK_WALKER3D_STEPPER, {"_alt": True, "_caps": (32, 10)}), ("Walker3DStepperEnv-v0", M.TASK_WALKER3D_STEPPER, {"_alt": True, "_caps": (64, 20)}),
         # model.bullet_fidelity(): 64 / 20 caps + alternating sweeps + 1e-5 m slop, what a real PyBullet trace is to be read with first
         ("Walker3DStepperEnv-v0", M.TASK_WALKER3D_STEPPER, {"_fidelity": True}), ("CassieEnv-v0", M.TASK_CASSIE, {"_fidelity": True}),
         # one absolute contact margin of 2 cm for every pair (g_margin <= 0; the compiled blobs carry Bullet's relative thresholds, millimetres)
         ("Walker3DCustomEnv-v0", M.TASK_WALKER3D_CUSTOM, {"_abs_margin": True}), ("Walker3DStepperEnv-v0", M.TASK_WALKER3D_STEPPER, {"_abs_margin": True})]


def _one_substep_blob(env_id, **kw):
    from mocca_envs_amd.vec_env import compile_model_for
    dump, massive, warm, cone = kw.pop("_dump", None), kw.pop("_massive", False), kw.pop("_warm", None), kw.pop("_cone", None)
    predict, abs_margin, caps = kw.pop("_predict", False), kw.pop("_abs_margin", False), kw.pop("_caps", None)
    slop, alt, fidelity = kw.pop("_slop", None), kw.pop("_alt", False), kw.pop("_fidelity", False)
    m = compile_model_for(env_id, **kw)
    if caps:
        m.max_rows, m.max_contacts = caps
    assert m.warmstart == 0.0 and m.friction_cone == 1 and m.limit_at_violation == 1
    if predict:
        m.limit_at_violation = 0
    assert 0.001 < m.slot_margin[0] < 0.01
    if abs_margin:
        for g in range(m.n_geoms):
            m.g_margin[g] = 0.0
        m.finalize_tables()
        assert abs(m.slot_margin[0] - 0.02) < 1e-4 and (m.n_pairs == 0 or abs(m.pair_margin[0] - 0.02) < 1e-4)
    if slop is not None:
        m.linear_slop = slop
    if alt:
        m.sweep_alternate = 1
    if fidelity:
        M.bullet_fidelity(m)
    if warm is not None:
        m.warmstart = warm
    if cone is not None:
        m.friction_cone = cone
    if massive:
        n = 0
        for b in range(1, m.n_bodies):
            if m.mass[b] == 0.0:
                m.mass[b] = 0.05
                for k in range(3):
                    m.inertia[b][k] = 1e-4 * (1 + 0.5 * k)
                n += 1
        assert n >= 1
        m.finalize_tables()
    if dump:
        from mocca_envs_amd.pybullet_dump import from_pybullet_dump, synthetic_dump
        src = m
        if dump == "massive":
            n = 0
            for b in range(1, src.n_bodies):
                if src.mass[b] == 0.0:
                    src.mass[b] = 0.08
                    for k in range(3):
                        src.inertia[b][k] = 2e-4 * (1 + 0.3 * k)
                        src.com[b][k] = 0.01 * (k - 1)
                    n += 1
            assert n == 9
            src.finalize_tables()
        rec = synthetic_dump(src, M.WALKER3D_JOINT_NAMES, fixed_children={4: 0.3, 17: 0.5})
        from pybullet_synth import upright_reset
        tmpl = compile_model_for(env_id, **kw)
        m = upright_reset(from_pybullet_dump(rec, tmpl, M.WALKER3D_JOINT_NAMES), rec, tmpl)
        assert abs(list(m.com[0])[0]) < 1e-6 and (m.mass[1] > 0) == (dump == "massive")   # Bullet's frames; the intermediate links' mass
    m.n_substeps = 1
    if env_id.startswith("Cassie"):
        m.n_llc = 1
    return m


@pytest.mark.parametrize("env_id,task,kw", CASES)
def test_single_substep_parity_with_matching_active_sets(env_id, task, kw):
    import torch
    from mocca_envs_amd.vec_env import VecEnv, task_from_float64, _DEFAULT_PARAMS
    from oracle.oracle import Oracle, PARAM_CURRICULUM
    n, steps = 256, 160
    kw = dict(kw)
    steep = kw.pop("_steep", False)
    m = _one_substep_blob(env_id, **kw)
    blob = m.to_bytes()
    env = VecEnv(env_id, n, auto_reset=False, seed=4, model_blob=blob)
    dbg = env.set_debug(True)
    orc = Oracle(blob, task, n, "f32")
    o64 = Oracle(blob, task, n, "f64")          # the yardstick: how far fp32 arithmetic itself is from the exact substep
    for pid, val in _DEFAULT_PARAMS.get(env_id, {}).items():
        env.set_param(pid, val); orc.set_param(pid, val); o64.set_param(pid, val)
    if task == M.TASK_WALKER3D_STEPPER:
        env.set_param(2, 9); orc.set_param(PARAM_CURRICULUM, 9); o64.set_param(PARAM_CURRICULUM, 9)
    if steep:
        from mocca_envs_amd import host_logic as H
        env.set_heightfield(3.0 * H.random_height_field(np.random.RandomState(11), (128, 128), 4).reshape(128, 128).astype(np.float32), 4)
    if task == M.TASK_WALKER3D_PLANNER:
        orc.set_heightfield(*env.height_field); o64.set_heightfield(*env.height_field)
    env.reset(); orc.reset(seed=4); o64.reset(seed=4)
    rng = np.random.default_rng(2)

    def scatter(mask=None):   # planner envs: move (the selected) robots from the flat start platform to random spots of the field
        st = orc.get_state()
        for e in range(n):
            if mask is None or mask[e]:
                xy = rng.uniform(-14, 14, 2)
                st[e, 0:2] = xy
                st[e, 2] = orc.height_at(*xy) + m.init_pos[2] + 0.02
        orc.set_state(st)

    if task == M.TASK_WALKER3D_PLANNER:
        scatter()
    nd = 13 + 2 * m.n_joints
    n_same = n_diff = n_clamp_diff = 0
    e_gpu, e_f32, e_flip, e_f32_flip, rows_seen = [], [], [], [], []
    units = lambda a, b: np.abs(a - b) / (1e-5 * (1.0 + np.abs(b)))
    for t in range(steps):
        env.set_state(orc.get_state().astype(np.float32))
        env.set_task(task_from_float64(orc.get_task()))
        o64.set_state(orc.get_state()); o64.set_task(orc.get_task())
        if task == M.TASK_WALKER3D_STEPPER:
            ter = np.zeros((n, 128), np.float32); ter[:, :124] = orc.get_terrain(); env.set_terrain(ter)
            o64.set_terrain(orc.get_terrain())
        scale = 1.0 if t % 3 else 0.3
        a = (scale * rng.uniform(-1, 1, (n, env.act_dim))).astype(np.float32)
        env.step(torch.from_numpy(a).cuda())
        _, _, dc, _ = orc.step(a)
        o64.step(a)
        sg, sc, s6 = env.get_state().cpu().numpy(), orc.get_state(), o64.get_state()
        dg_, dc_, d6_ = dbg.cpu().numpy(), orc.get_debug(), o64.get_debug()
        ok = np.isfinite(sc).all(axis=1) & np.isfinite(s6).all(axis=1)
        rows_same = (dg_[:, :8] == dc_[:, :8]).all(axis=1) & ok            # rows, contacts, slot / limit masks
        same = rows_same & (dg_[:, 8:12] == dc_[:, 8:12]).all(axis=1)      # ... and every clamp decision of the solver
        n_same += int(same.sum()); n_diff += int((~rows_same & ok).sum()); n_clamp_diff += int((rows_same & ~same).sum())
        rows_seen.append(dc_[ok, 0])
        if same.any():
            e_gpu.append(units(sg[same][:, :nd], sc[same][:, :nd]).max(axis=1))
        flip = rows_same & ~same
        if flip.any():
            e_flip.append(units(sg[flip][:, :nd], sc[flip][:, :nd]).max(axis=1))
        same64 = (d6_[:, :12] == dc_[:, :12]).all(axis=1) & ok
        if same64.any():
            e_f32.append(units(sc[same64][:, :nd], s6[same64][:, :nd]).max(axis=1))
        flip64 = (d6_[:, :8] == dc_[:, :8]).all(axis=1) & ok & ~same64      # the yardstick's own flips: f32 vs f64 oracle, same rows, another pattern
        if flip64.any():
            e_f32_flip.append(units(sc[flip64][:, :nd], s6[flip64][:, :nd]).max(axis=1))
        # restart fallen envs so the sample keeps standing / stepping / falling robots
        if t % 8 == 7:
            fallen = (dc != 0).astype(np.uint8)
            if fallen.any():
                orc.reset(seed=4, mask=fallen)
                if task == M.TASK_WALKER3D_PLANNER:
                    scatter(fallen)
    e_gpu, e_f32 = np.concatenate(e_gpu), np.concatenate(e_f32)
    rows = np.concatenate(rows_seen)
    total = max(1, n_same + n_diff + n_clamp_diff)
    frac, frac_clamp = n_diff / total, n_clamp_diff / total
    q = lambda x, p: float(np.percentile(x, p))
    print(f"\n{env_id}: {total} substeps, rows/substep median {np.median(rows):.0f} max {rows.max()}, row sets differ in "
          f"{100 * frac:.3f} %, clamp patterns (same rows) in {100 * frac_clamp:.3f} %; same active set, state error in units of 1e-5 (1+|x|): GPU vs f32 oracle median {q(e_gpu, 50):.3g} "
          f"p90 {q(e_gpu, 90):.3g} p99 {q(e_gpu, 99):.3g} max {e_gpu.max():.3g} | f32 oracle vs f64 oracle median {q(e_f32, 50):.3g} "
          f"p90 {q(e_f32, 90):.3g} p99 {q(e_f32, 99):.3g} max {e_f32.max():.3g}")
    assert rows.max() >= (6 if task == M.TASK_CASSIE else 12), "the sample must contain contact-rich substeps"
    if kw.get("_caps") == (64, 20) and kw.get("_abs_margin") and task == M.TASK_WALKER3D_STEPPER:
        assert rows.max() > 48, "the accuracy instance's sample must contain substeps beyond the product's 48-row cap"
    assert frac < 0.01, f"active sets differ in {100 * frac:.2f} % of the substeps"
    if e_flip:
        e_flip = np.concatenate(e_flip)
        print(f"  same rows, another clamp pattern ({len(e_flip)} samples): state error median {q(e_flip, 50):.3g} p99 {q(e_flip, 99):.3g} max {e_flip.max():.3g}")
        # A flipped clamp is ANOTHER linear system -- and since the friction rows follow Bullet's `if (normal impulse > 0)` switch, a normal
        # impulse that is 0 on one side and 1e-9 on the other turns a whole friction pair on or off for a sweep: what a flip costs is a
        # property of the solver, not of the arithmetic that triggered it (1e-2 relative on single samples).  What is bounded is how OFTEN it
        # happens -- against the fp32 oracle's own flip rate relative to the f64 oracle on the same substeps (same rows, another pattern) --
        # and, for buckets large enough to have one, the median; single samples only get a sanity bound.
        yf = np.concatenate(e_f32_flip) if e_f32_flip else np.zeros(0)
        rate_y = len(yf) / total
        print(f"  the f32 oracle's own flips against the f64 oracle: {len(yf)} samples ({100 * rate_y:.3f} %)" + (f", median {q(yf, 50):.3g} max {yf.max():.3g}" if len(yf) else ""))
        # rate: at most twice the fp32 oracle's own flip rate against the f64 oracle on the same substeps (+ 20 samples: buckets of a handful)
        assert frac_clamp <= 2 * rate_y + 5e-4, (frac_clamp, rate_y)
        # size: no worse than 10 x the yardstick's worst flip -- with a floor, because a bucket of < 10 samples has no stable maximum (the
        # stepping stones' soft contacts: 1.4e4 units on one side, 13 on the other, and the reverse in the next case of this file)
        assert e_flip.max() < max(10 * (yf.max() if len(yf) else 0.0), 2e4), (e_flip.max(), yf.max() if len(yf) else None)
        if len(e_flip) >= 200 and len(yf) >= 200:   # the planar walkers (a fifth to a half of their substeps): the distributions themselves
            assert q(e_flip, 99) <= max(30.0, 3 * q(yf, 99)), (q(e_flip, 99), q(yf, 99))
            assert q(e_flip, 90) <= max(10.0, 3 * q(yf, 90)), (q(e_flip, 90), q(yf, 90))
        if len(e_flip) >= 20:
            assert q(e_flip, 50) < max(100.0, 5 * q(e_f32, 50), 3 * (q(yf, 50) if len(yf) else 0.0)), (q(e_flip, 50), len(yf))
    # Same rows, same arithmetic, another association order.  The fp32 tolerance of ONE substep is what fp32 arithmetic itself
    # costs on this substep: the f32 oracle's distance from the f64 oracle (stiff rows divide position errors of 1e-7 by dt:
    # Cassie's closure rows at dt = 0.6 ms turn them into 1e-4 of velocity).  The kernel may be no further from the f32 oracle
    # than 3x that, with 1e-5 relative as the floor.  The PGS clamps (friction bounds, unilateral normals and limits)
    # are part of the compared set (MOCCA_DBG_CLAMPSIG_*): on the samples kept, both sides solved the SAME linear system, so the
    # tail is bounded too -- 30 units (3e-4 relative, one substep) or 3x the yardstick's own worst sample, whichever is larger.
    assert q(e_gpu, 50) < max(1.0, 3 * q(e_f32, 50)), (q(e_gpu, 50), q(e_f32, 50))
    assert q(e_gpu, 99) < max(10.0, 3 * q(e_f32, 99)), (q(e_gpu, 99), q(e_f32, 99))
    assert e_gpu.max() < max(30.0, 3 * e_f32.max()), (e_gpu.max(), e_f32.max())
    env.close()


def test_per_env_parameters_through_the_abi():
    """mocca_set_param_v: two halves of one batch run different curricula (terrain spread, applied gain, terminal height)
    and each env matches the oracle run with ITS curriculum -- env_base.py:103-106, env_locomotion.py:368-369,489,628."""
    import torch
    from mocca_envs_amd import lib as L
    from mocca_envs_amd.vec_env import VecEnv, task_to_float64
    from oracle.oracle import Oracle, PARAM_CURRICULUM
    n = 64
    env = VecEnv("Walker3DStepperEnv-v0", n, auto_reset=True, seed=6)
    cur = np.where(np.arange(n) % 2 == 0, 2, 9).astype(np.float32)
    env.set_param_v(L.PARAM_CURRICULUM, cur)
    og = env.reset().cpu().numpy()
    tk = task_to_float64(env.get_task())
    np.testing.assert_array_equal(tk[:, 20], cur)                                      # task word 20 = curriculum
    np.testing.assert_allclose(tk[:, 21], 1.0 + 0.2 * cur / 9, atol=1e-6)              # applied gain, :369,489
    ter_g = env.get_terrain().cpu().numpy()
    for c in (2, 9):
        orc = Oracle(env.model.to_bytes(), M.TASK_WALKER3D_STEPPER, n, "f32")
        orc.set_param(PARAM_CURRICULUM, c)
        oc = orc.reset(seed=6)
        sel = cur == c
        np.testing.assert_allclose(og[sel], oc[sel], atol=2e-6)
        np.testing.assert_allclose(ter_g[sel][:, :124], orc.get_terrain()[sel], atol=5e-6)
    spread = np.abs(ter_g[:, 1:120:6]).max(axis=1)                                     # lateral spread grows with the curriculum
    assert spread[cur == 9].mean() > 2 * spread[cur == 2].mean()
    # terminal height follows the env's own curriculum (:368,628): a robot pitched 1.3 rad stands ~0.58 m above its lower
    # foot -- alive at curriculum 9 (0.45 m), dead at curriculum 2 (0.683 m)
    st = env.get_state().cpu().numpy()
    st[:, 3:7] = [0, np.sin(0.65), 0, np.cos(0.65)]
    st[:, 13 + 21:13 + 42] = 0
    env.set_param(L.PARAM_AUTO_RESET, 0)
    env.set_state(st)
    z = np.zeros((n, 2), np.int32)
    o, _, d, _ = env.task_step(torch.zeros(n, 21), z, z)
    h = o[:, 0].cpu().numpy()                                  # differs a little from env to env (random start poses)
    d = d.cpu().numpy() & 1
    term_h = 0.75 + (0.45 - 0.75) * cur / 9
    margin = np.abs(h - term_h) > 1e-4
    np.testing.assert_array_equal(d[margin], (h <= term_h)[margin].astype(d.dtype))
    both = (h > 0.46) & (h < 0.68)
    assert both.sum() > n // 2 and (d[both & (cur == 2)] == 1).all() and (d[both & (cur == 9)] == 0).all()
    # broadcast form + scalar form drop back to one value for all
    env.set_param_v(L.PARAM_CURRICULUM, np.array([5.0], np.float32), broadcast=True)
    env.reset()
    assert (task_to_float64(env.get_task())[:, 20] == 5).all()
    env.set_param(L.PARAM_CURRICULUM, 1)
    env.reset()
    assert (task_to_float64(env.get_task())[:, 20] == 1).all()
    # applied_gain acts on the next apply_action (robots.py:33) and persists across a Custom env's resets
    cenv = VecEnv("Walker3DCustomEnv-v0", 8, auto_reset=False, seed=1)
    cenv.reset()
    g = np.linspace(0.5, 1.2, 8).astype(np.float32)
    cenv.set_param_v(L.PARAM_APPLIED_GAIN, g)
    np.testing.assert_allclose(task_to_float64(cenv.get_task())[:, 21], g, atol=1e-7)
    cenv.reset()
    np.testing.assert_allclose(task_to_float64(cenv.get_task())[:, 21], g, atol=1e-7)
    cenv.set_param(L.PARAM_APPLIED_GAIN, 1.1)
    np.testing.assert_allclose(task_to_float64(cenv.get_task())[:, 21], 1.1, atol=1e-7)
    # eval mode per env: walk target = (x + 4, 0, 1) for the flagged envs only (env_locomotion.py:115-116)
    cenv.set_param_v(L.PARAM_EVAL_MODE, (np.arange(8) < 4).astype(np.float32))
    cenv.reset()
    tk = task_to_float64(cenv.get_task())
    np.testing.assert_allclose(tk[:4, 14], 4.0); np.testing.assert_allclose(tk[:4, 15], 0.0)
    assert (tk[4:, 14] != 4.0).all()
    env.close(); cenv.close()


@pytest.mark.parametrize("env_id,n,steps", [("Walker3DStepperEnv-v0", 4096, 120), ("CassieEnv-v0", 2048, 12),
                                            ("Walker3DCustomEnv-v0", 8192, 120), ("Walker3DPlannerEnv-v0", 4096, 120)])
def test_properties_at_the_benchmark_sizes(env_id, n, steps):
    """BASELINE.json configs 2, 3 and 4's per-GPU shard at full size: bitwise determinism run to run, finite state, unit
    quaternions, joint limits honoured up to the solver's slack, speed clamp, done envs really restart."""
    import torch
    from mocca_envs_amd.vec_env import VecEnv, compile_model_for
    m = compile_model_for(env_id)
    nj = m.n_joints

    def run():
        env = VecEnv(env_id, n, auto_reset=True, seed=31)
        if "Stepper" in env_id:
            env.set_param(2, 9)
        env.reset()
        g = torch.Generator(device="cuda").manual_seed(17)
        n_done = 0
        for k in range(steps):
            a = torch.rand(n, env.act_dim, device="cuda", generator=g) * 2 - 1
            o, r, d, info = env.step(a)
            n_done += int((d != 0).sum())
        out = (o.clone(), r.clone(), d.clone(), env.get_state().clone(), env.get_task().clone())
        env.close()
        return out, n_done

    (o1, r1, d1, s1, t1), nd1 = run()
    (o2, r2, d2, s2, t2), nd2 = run()
    assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(s1, s2) and torch.equal(t1, t2)
    assert nd1 == nd2 and (nd1 > 0 or "Cassie" in env_id)
    assert torch.isfinite(s1).all() and torch.isfinite(o1).all() and torch.isfinite(r1).all()
    assert torch.allclose(s1[:, 3:7].norm(dim=1), torch.ones(n, device="cuda"), atol=1e-5)
    lo, hi = M.joint_limits(m)
    q = s1[:, 13:13 + nj].cpu().numpy()
    fin = (hi > lo) & (hi - lo < 1e20)          # Cassie's continuous rod joints carry +-1e30
    # limit rows exist only at / past the limit: a joint crosses by at most speed x dt (the 100 rad/s clamp x dt: limbs of a flailing robot do
    # reach it), then it is held and pushed back -- what a limit row does per substep is bounded sample by sample in
    # test_limit_rows_hold_a_joint_at_its_stop; here, at full size, the end state must lie within ONE crossing of the stops
    over = max(float((lo[fin] - q[:, fin]).max()), float((q[:, fin] - hi[fin]).max()))
    print(f"largest limit overshoot {over:.3f} rad")
    assert over < float(m.max_qd) * float(m.dt) + 0.02
    assert (s1[:, 13 + nj:13 + 2 * nj].abs() <= m.max_qd + 1e-3).all()
    ep = t1[:, 9].cpu().numpy()
    assert (ep >= 0).all() and (nd1 == 0 or ep.max() >= 1)          # episode counters advanced where envs finished


@pytest.mark.parametrize("env_id", ["Walker3DCustomEnv-v0", "Walker3DStepperEnv-v0", "LaikagoCustomEnv-v0"])
def test_limit_rows_hold_a_joint_at_its_stop(env_id):
    """The limit law (limit rows only at / past the stop, MoccaModel.limit_at_violation) per SUBSTEP and per joint, on 2048 free-running
    envs with one substep per step: (a) a joint inside its range crosses a stop by at most its own speed x dt -- symplectic Euler, to the
    last bit; (b) a joint that IS past a stop has a row: the solver leaves it no speed that carries it further out -- what it may keep is
    the residual of five Gauss-Seidel sweeps from zero, bounded here in rad/s by what the run shows with a margin of 2 --, and on average
    it is pushed back (the non-contact ERP 0.2 of the gap per substep)."""
    import torch
    from mocca_envs_amd.vec_env import VecEnv, compile_model_for
    m = compile_model_for(env_id)
    m.n_substeps = 1
    nj, dt = m.n_joints, float(m.dt)
    n = 2048
    env = VecEnv(env_id, n, auto_reset=True, seed=12, model_blob=m.to_bytes())
    if "Stepper" in env_id:
        env.set_param(2, 9)
    if "Laikago" in env_id:
        env.set_param(3, 0)
    env.reset()
    lo, hi = (torch.from_numpy(x).cuda() for x in M.joint_limits(m))
    g = torch.Generator(device="cuda").manual_seed(5)
    worst_a, res, back = 0.0, [], []
    for k in range(500):
        s0 = env.get_state().clone()
        _, _, d, _ = env.step(torch.rand(n, env.act_dim, device="cuda", generator=g) * 2 - 1)
        s1 = env.get_state()
        keep = (d == 0)[:, None]
        q0, q1, v1 = s0[:, 13:13 + nj], s1[:, 13:13 + nj], s1[:, 13 + nj:13 + 2 * nj]
        over0, over1 = torch.maximum(lo - q0, q0 - hi), torch.maximum(lo - q1, q1 - hi)
        inside = keep & (over0 <= 0)
        worst_a = max(worst_a, float((over1 - v1.abs() * dt)[inside].max()))
        past = keep & (over0 > 0) & (over1 > 0) & ((q0 > hi) == (q1 > hi))
        if past.any():
            res.append(((over1 - over0) / dt)[past])                         # outward speed the solve left on a joint with an active limit row
            back.append(((over0 - over1) / over0.clamp(min=1e-9))[past & (over0 > 1e-3)])
    res, back = torch.cat(res).cpu().numpy(), torch.cat(back).cpu().numpy()
    print(f"\n{env_id}: inside joints cross by at most speed x dt (+{worst_a:.2e}); {len(res)} (joint, substep) samples past a stop: outward speed left "
          f"median {np.median(res):.3f} p99 {np.percentile(res, 99):.3f} max {res.max():.3f} rad/s; fraction of the gap closed per substep median {np.median(back):.3f}")
    assert worst_a < 1e-5
    assert len(res) > 1000
    # measured (walker, 1.5 M samples): median -0.57 rad/s (pushed back), p99 2.3, max 16 -- five sweeps from zero do not converge a row that
    # competes with contact rows of a flailing robot; a broken limit row would leave the joint's full speed (tens of rad/s at the median)
    assert np.median(res) < 0.0 and np.percentile(res, 99) < 5.0 and res.max() < 35.0
    assert 0.15 < np.median(back) < 0.25        # the non-contact ERP: a fifth of the gap per substep
    env.close()
