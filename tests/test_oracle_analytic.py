"""Closed-form physics cases for the oracle's rigid-body substep (its parity with PyBullet is unpinned, DESIGN.md section 4;
these pin it to mechanics instead).  Tiny purpose-built models compiled with the product's own model compiler:

  * resting contact carries exactly m g; a block on a soft plank sinks m g / (n k)      (bullet_objects.py:64-72 erp / cfm)
  * Coulomb friction: a block on a plank tilted below atan(mu) sticks, above it slides with g (sin - mu cos)
  * a hinge driven into its limit by a constant torque stops at the limit (no overshoot beyond the ERP slack)
  * a compound pendulum swings with the textbook period; a damped spinning link decays with c / I
  * two crossing capsules: contact point, normal, depth of the self-collision narrow phase; the impulse separates them
  * a block on a height field of constant slope: plane normal, m g cos / (n k) sink, sticks below atan(mu), slides with g (sin - mu cos)
CPU only.
"""
import numpy as np
import pytest

import dense_reference as D
from mocca_envs_amd import model as M
from mocca_envs_amd.model import GEOM_CAPSULE, GEOM_SPHERE, Body, Geom, Hinge
from oracle.oracle import Oracle

G, DT = 9.8, 1.0 / 240.0


def _block(fric=0.5):
    """A rigid 'table': four radius-2cm corner spheres 20 cm apart below a heavy core sphere (2.28 kg)."""
    c = [Geom(f"c{i}", GEOM_SPHERE, 0.02, (sx * 0.1, sy * 0.1, -0.1), friction=fric)
         for i, (sx, sy) in enumerate([(1, 1), (1, -1), (-1, 1), (-1, -1)])]
    root = Body("block", (0, 0, 0.5), geoms=c + [Geom("core", GEOM_SPHERE, 0.08, (0, 0, 0), friction=fric)])
    m = M.compile_model(root, [], {}, (0, 0, 0.5), [], [], [], self_collision=False)
    m.lin_damp = m.ang_damp = 0.0
    # the closed forms below are those of the CONVERGED contact solve.  The envs' 5 Gauss-Seidel sweeps from zero (no warm start) leave a
    # resting block a steady creep of 1e-3 rad/s (pyramid) / 2.6e-3 (cone) about the vertical and shift a soft-contact equilibrium by
    # 10 %; with 50 sweeps both vanish -- the tests pin the rows' physics, not the truncation
    m.n_iters = 50
    return m


def _oracle(m, task=0):
    o = Oracle(m.to_bytes(), task, 1, "f64")
    st = np.zeros((1, o.state_dim)); st[0, 6] = 1
    o.set_state(st)
    return o, st


def test_resting_contact_carries_the_weight():
    m = _block()
    o, st = _oracle(m)
    st[0, 2] = 0.13                                    # corners 1 cm above the ground
    o.set_state(st)
    o.physics_substeps(0, np.zeros(0), 480)
    s = o.get_state()[0]
    assert abs(s[2] - 0.12) < 2e-4                     # sits ON the plane (Baumgarte erp 0.9: sub-0.2 mm penetration)
    assert np.abs(s[7:13]).max() < 1e-3
    np.testing.assert_allclose(s[13:].sum(), m.mass[0] * G * DT, rtol=1e-6)      # sum of normal impulses per substep = m g dt
    np.testing.assert_allclose(s[13:17], m.mass[0] * G * DT / 4, rtol=0.02)      # shared evenly by the four corners


def _plank_world(m, tilt):
    """Oracle on the Stepper task with plank 0 rolled by `tilt` about x (slope along y) under the origin."""
    o, st = _oracle(m, task=M.TASK_WALKER3D_STEPPER)
    ter = np.zeros((1, 124))
    ter[0, 0:6] = [0, 0, 0, 0, tilt, 0]                 # x y z phi x_tilt y_tilt (env_locomotion.py:441,461-465)
    ter[0, 6:12] = [50, 0, 0, 0, 0, 0]                  # the other two live planks far away
    ter[0, 12:18] = [60, 0, 0, 0, 0, 0]
    ter[0, 120:123] = [0, 1, 2]
    o.set_terrain(ter)
    mdl = D.Model(m)
    bc, Rb, _ = D.live_planks(mdl, ter[0], 1)[0]
    return o, st, bc, Rb, mdl


def _put_on_plank(st, bc, Rb, mdl, lift):
    top = bc + Rb @ np.array([0, 0, mdl.plank_half[2]])
    st[0, 0:3] = top + Rb @ np.array([0, 0, 0.12 + lift])       # corner spheres' lowest points `lift` above the top face
    st[0, 3:7] = D._mat_quat(Rb)
    return st


def test_block_sinks_mg_over_k_into_a_soft_plank():
    """Plank contacts are springs: stiffness 30000, damping 1000 (bullet_objects.py:70-71) -> erp / cfm of the normal rows.
    At rest each of the n = 4 corner contacts carries m g / 4 and is compressed by m g / (4 k)."""
    m = _block()
    assert m.n_iters == 50   # four rows coupled through one rigid body (A_ij = 1/m = 0.44 against cfm 0.21) take more than the envs' 5 sweeps
    o, st, bc, Rb, mdl = _plank_world(m, 0.0)
    o.set_state(_put_on_plank(st, bc, Rb, mdl, 0.0))
    o.physics_substeps(0, np.zeros(0), 1200)
    s = o.get_state()[0]
    assert np.abs(s[7:13]).max() < 1e-3                # at rest (5 Gauss-Seidel sweeps per substep leave a 1e-4 friction residual)
    top_z = (bc + Rb @ np.array([0, 0, mdl.plank_half[2]]))[2]
    sink = top_z - (s[2] - 0.12)
    np.testing.assert_allclose(sink, m.mass[0] * G / (4 * m.plank_stiffness), rtol=0.02)
    np.testing.assert_allclose(s[13:].sum(), m.mass[0] * G * DT, rtol=1e-4)


@pytest.mark.parametrize("deg,sticks", [(15.0, True), (24.0, True), (30.0, False), (38.0, False)])
def test_coulomb_friction_on_an_incline(deg, sticks):
    """mu = plank friction 1.0 x geom friction 0.5 (bullet_objects.py:68): critical slope atan(0.5) = 26.6 deg.  The slope runs
    along one of btPlaneSpace1's friction directions, so the friction pyramid and the Coulomb cone coincide."""
    mu, th = 0.5, np.deg2rad(deg)
    m = _block(fric=mu)
    o, st, bc, Rb, mdl = _plank_world(m, th)
    o.set_state(_put_on_plank(st, bc, Rb, mdl, 0.0))
    o.physics_substeps(0, np.zeros(0), 120)               # settle into the springs
    p0, v0 = o.get_state()[0, 0:3].copy(), o.get_state()[0, 7:10].copy()
    n_sub = 240
    o.physics_substeps(0, np.zeros(0), n_sub)
    s = o.get_state()[0]
    down = Rb @ np.array([0, -1.0, 0])                     # unit vector down the slope (roll about x lifts +y)
    if down[2] > 0:
        down = -down
    travel, speed = (s[0:3] - p0) @ down, (s[7:10] - v0) @ down
    t = n_sub * DT
    if sticks:
        # holds: what is left is the creep of 5 Gauss-Seidel sweeps per substep (mm/s), two orders below the sliding speeds
        assert abs(travel) < 1e-2 and abs(s[7:10] @ down) < 1e-2, (travel, s[7:10] @ down)
    else:
        a = G * (np.sin(th) - mu * np.cos(th))
        np.testing.assert_allclose(speed / t, a, rtol=0.03)
        assert abs((s[0:3] - p0) @ (Rb @ np.array([1.0, 0, 0]))) < 1e-3                          # and only down the slope


@pytest.mark.parametrize("deg,sticks", [(12.0, True), (22.0, True), (32.0, False), (40.0, False)])
def test_block_on_a_sloping_height_field(deg, sticks):
    """The planner envs' terrain (bullet_objects.py:338-393): a height field whose heights rise linearly with y is a plane of slope
    `deg` made of triangles.  The contact normal is the plane's, the block sinks m g cos(theta) / (n k) into the soft ground
    (stiffness 30000, damping 1000), and Coulomb friction (field 1.0 x geom 0.5) holds it below atan(0.5) = 26.6 deg and lets it
    slide with g (sin - mu cos) above -- whatever triangle of whatever cell each corner sphere happens to be over."""
    mu, th = 0.5, np.deg2rad(deg)
    m = _block(fric=mu)
    o, st = _oracle(m, task=M.TASK_WALKER3D_PLANNER)
    scale, npts = 4, 96
    ys = (np.arange(npts) - (npts - 1) / 2) / scale
    field = np.tile((np.tan(th) * ys)[:, None], (1, npts)).astype(np.float64)        # heights[iy][ix]
    o.set_heightfield(field, scale)
    nrm = np.array([0.0, -np.sin(th), np.cos(th)])
    # a probe 5 cm above the surface (the 2 x 2-cell search is exact up to half a cell = 12.5 cm of reach)
    g_probe, n_probe = o.heightfield_probe(np.array([0.3, 0.2, np.tan(th) * 0.2]) + 0.05 * nrm, 0.02)
    np.testing.assert_allclose(n_probe, nrm, atol=1e-5)                               # (the grid is stored as float32)
    np.testing.assert_allclose(g_probe, 0.05 - 0.02, atol=1e-5)
    # the block, tilted with the slope, its corner spheres touching the surface
    c, sn = np.cos(th / 2), np.sin(th / 2)
    Rq = np.array([sn, 0, 0, c])                                                       # rotation by theta about x lifts +y
    st[0, 0:3] = 0.12 * nrm + np.array([0.07, 0.11, np.tan(th) * 0.11])               # 0.12 = corner offset 0.1 + radius 0.02
    st[0, 3:7] = Rq
    o.set_state(st)
    o.physics_substeps(0, np.zeros(0), 120)
    p0, v0 = o.get_state()[0, 0:3].copy(), o.get_state()[0, 7:10].copy()
    n_sub = 240
    o.physics_substeps(0, np.zeros(0), n_sub)
    s = o.get_state()[0]
    down = np.array([0.0, -np.cos(th), -np.sin(th)])
    travel, speed = (s[0:3] - p0) @ down, (s[7:10] - v0) @ down
    if sticks:
        assert abs(travel) < 1e-2 and abs(s[7:10] @ down) < 1e-2, (travel, s[7:10] @ down)
        sink = 0.12 - (s[0:3] - np.array([s[0], s[1], np.tan(th) * s[1]])) @ nrm       # distance of the block centre to the plane
        # (a 0.18 mm compression read off positions while the four corners still trade load: 8 %; the impulse sum below is exact)
        np.testing.assert_allclose(sink, m.mass[0] * G * np.cos(th) / (4 * m.plank_stiffness), rtol=0.08)
        np.testing.assert_allclose(s[13:].sum(), m.mass[0] * G * np.cos(th) * DT, rtol=1e-3)
    else:
        np.testing.assert_allclose(speed / (n_sub * DT), G * (np.sin(th) - mu * np.cos(th)), rtol=0.03)
        assert abs(s[0] - p0[0]) < 1e-3                                                # and only down the slope


def _base_with_arm(axis, lo, hi, arm_dir, damping=0.0, armature=0.0):
    """A 520 kg base sphere resting on four corner points + one light capsule link on a hinge at the top of the base."""
    c = [Geom(f"c{i}", GEOM_SPHERE, 0.02, (sx * 0.4, sy * 0.4, -0.5), friction=1.0)
         for i, (sx, sy) in enumerate([(1, 1), (1, -1), (-1, 1), (-1, -1)])]
    arm = Body("arm", (0, 0, 0.8), hinges=[Hinge("j", axis, lo, hi, 1.0)],
               geoms=[Geom("arm", GEOM_CAPSULE, 0.03, (0, 0, 0), tuple(0.4 * np.array(arm_dir)), group=0, mask=0)])
    root = Body("base", (0, 0, 0.52), geoms=c + [Geom("core", GEOM_SPHERE, 0.5, (0, 0, 0))], children=[arm])
    m = M.compile_model(root, [], {}, (0, 0, 0.52), [], [], [], self_collision=False, joint_damping=damping, joint_armature=armature)
    m.lin_damp = m.ang_damp = 0.0
    return m


def _arm_inertia_about_hinge(m):
    """I of the link about its hinge axis from the blob (parallel axes)."""
    ax = np.array(list(m.jaxis[1]))
    xx, yy, zz, xy, xz, yz = m.inertia[1]
    Ic = np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])
    c = np.array(list(m.com[1]))
    d2 = c @ c - (c @ ax) ** 2
    return ax @ Ic @ ax + m.mass[1] * d2, m.mass[1], c


@pytest.mark.parametrize("at_violation", [1, 0])
def test_hinge_driven_into_its_limit_stops_there(at_violation):
    """`limit_at_violation = 1` (the blobs' default, Bullet's `if (penetration > 0) continue`): no row before the joint is at the limit, so it
    crosses by speed x dt, is pushed back with the non-contact ERP and settles on the stop.  0: a row from a predicted gap of limit_slack
    on stops it AT the limit."""
    m = _base_with_arm((0, 0, 1), -30, 30, (1, 0, 0))     # vertical axis: gravity does not load the joint
    assert m.limit_at_violation == 1
    m.limit_at_violation = at_violation
    o, st = _oracle(m)
    st[0, 2] = 0.52
    o.set_state(st)
    hi = np.deg2rad(30)
    qs, qds = [], []
    for k in range(480):
        o.physics_substeps(0, np.array([2.0]), 1)        # 2 N m on ~0.06 kg m^2: reaches the stop in ~0.25 s at ~8 rad/s
        s = o.get_state()[0]
        qs.append(s[13]); qds.append(s[14])
    qs, qds = np.array(qs), np.array(qds)
    assert qds.max() > 4.0                                # it did arrive at speed
    if at_violation:
        k = int(np.argmax(qs > hi))                       # the substep that crossed: by at most its speed x dt, and it did cross
        assert 0 < qs.max() - hi <= qds[k] * DT * 1.001 + 1e-9, (qs.max() - hi, qds[k] * DT)
        assert (qs[k + 1:] < qs[k] + 1e-9).all()          # the row that exists from then on lets it go no further
    else:
        assert qs.max() < hi + 1e-3, qs.max() - hi        # no overshoot beyond a milliradian
    assert abs(qs[-1] - hi) < 1e-3 and np.abs(qds[-100:]).max() < (0.2 if at_violation else 0.05)   # and rests against the stop under the torque


def test_compound_pendulum_period_and_joint_damping():
    m = _base_with_arm((0, 1, 0), -170, 170, (0, 0, -1))   # hangs down, swings about y
    I, mass, c = _arm_inertia_about_hinge(m)
    T_exact = 2 * np.pi * np.sqrt(I / (mass * G * abs(c[2])))
    o, st = _oracle(m)
    st[0, 2] = 0.52
    st[0, 13] = 0.05                                        # small amplitude
    o.set_state(st)
    o.physics_substeps(0, np.zeros(1), 1)
    q, ts = [], []
    for k in range(int(3.2 * T_exact / DT)):
        o.physics_substeps(0, np.zeros(1), 1)
        q.append(o.get_state()[0, 13]); ts.append((k + 2) * DT)
    q, ts = np.array(q), np.array(ts)
    up = np.where((q[:-1] < 0) & (q[1:] >= 0))[0]           # upward zero crossings, linearly interpolated
    tc = ts[up] + DT * (-q[up]) / (q[up + 1] - q[up])
    assert len(tc) >= 3
    T_sim = np.diff(tc).mean()
    np.testing.assert_allclose(T_sim, T_exact, rtol=0.01)   # base recoil (mass ratio 1/500) + O(dt) integrator: < 1 %
    assert np.abs(q).max() < 0.05 * 1.05                    # symplectic Euler: amplitude bounded
    # joint damping: a spinning link on a vertical hinge decays as exp(-c t / I)
    md = _base_with_arm((0, 0, 1), -1e5, 1e5, (1, 0, 0), damping=0.1)
    Id, _, _ = _arm_inertia_about_hinge(md)
    o, st = _oracle(md)
    st[0, 2] = 0.52; st[0, 14] = 5.0
    o.set_state(st)
    o.physics_substeps(0, np.zeros(1), 240)
    np.testing.assert_allclose(o.get_state()[0, 14], 5.0 * np.exp(-0.1 / Id * 1.0), rtol=0.02)


def test_two_crossing_capsules_self_contact():
    """Siblings on one base (parents are excluded, robots.py:259-264): capsule A along x, capsule B along y, axes 9 cm apart,
    radii 5 cm -> 1 cm deep contact midway, normal along z from B to A.  Each link is hinged 30 cm away from the crossing, so
    the normal row can act: Baumgarte asks for a separating speed of erp * depth / dt."""
    a = Body("a", (-0.3, 0, 0.09), hinges=[Hinge("ja", (0, 1, 0), -90, 90, 1.0)],
             geoms=[Geom("ga", GEOM_CAPSULE, 0.05, (0, 0, 0), (0.6, 0, 0), friction=0.0)])
    b = Body("b", (0, -0.3, 0.0), hinges=[Hinge("jb", (1, 0, 0), -90, 90, 1.0)],
             geoms=[Geom("gb", GEOM_CAPSULE, 0.05, (0, 0, 0), (0, 0.6, 0), friction=0.0)])
    root = Body("base", (0, 0, 5.0), geoms=[Geom("core", GEOM_SPHERE, 0.1, (0, 0, -1.0), group=0, mask=0)], children=[a, b])
    m = M.compile_model(root, [], {}, (0, 0, 5.0), [], [], [], self_collision=True)
    m.gravity = 0.0; m.lin_damp = m.ang_damp = 0.0
    assert m.n_pairs == 1
    o, st = _oracle(m)
    st[0, 2] = 5.0
    o.set_state(st)
    o.physics_substeps(0, np.zeros(2), 1)
    c = o.last_contacts()
    assert len(c) == 1
    ba, bb = int(c[0][0]), int(c[0][1])
    assert {ba, bb} == {1, 2} and int(c[0][2]) == -1
    sign = 1.0 if ba == 1 else -1.0                      # normal points from the second body to the first
    np.testing.assert_allclose(c[0][6:9], [0, 0, sign], atol=1e-7)           # blob constants are fp32
    np.testing.assert_allclose(c[0][3:6], [0, 0, 0.045], atol=1e-7)          # midway between the two surface points, rel. base origin
    assert abs(c[0][9] - 0.01) < 1e-7 and c[0][10] == 0.0                  # frictionless skins: mu_a * mu_b = 0
    lam, kind = o.last_lambda()
    assert [int(k) for k in kind] == [1, 2, 2] and lam[0] > 0 and np.abs(lam[1:]).max() == 0.0
    s = o.get_state()[0]
    qd_a, qd_b = s[15], s[16]
    # A's contact point is 0.3 m along +x of its y-hinge (z = -0.3 sin q), B's 0.3 m along +y of its x-hinge (z = +0.3 sin q);
    # whatever the base does moves both coincident points alike
    v_sep = -0.3 * qd_a - 0.3 * qd_b
    np.testing.assert_allclose(v_sep, 0.9 * 0.01 / DT, rtol=1e-6)


def test_contact_manifold_keeps_the_corners_of_a_plate():
    """MoccaModel.manifold_max (Cassie's toes): of nine support points in a 3 x 3 grid under one link, all within the
    margin, the four-point manifold keeps the corners -- the deepest point, the one farthest from it and the farthest to either side of
    the line through those two -- whichever corner is deepest; the plate then rests on them carrying m g, a quarter each when level."""
    pts = [(sx * 0.1, sy * 0.1, -0.1) for sx in (-1, 0, 1) for sy in (-1, 0, 1)]
    geoms = [Geom(f"p{i}", GEOM_SPHERE, 0.02, p, friction=0.5) for i, p in enumerate(pts)]
    root = Body("plate", (0, 0, 0.5), geoms=geoms + [Geom("core", GEOM_SPHERE, 0.08, (0, 0, 0), group=0, mask=0)])
    m = M.compile_model(root, [], {}, (0, 0, 0.5), [], [], [], self_collision=False)
    m.lin_damp = m.ang_damp = 0.0
    m.manifold_max = 4
    m.n_iters = 50                                                  # the resting state below is the converged solve's (see _block)
    corners = {0, 2, 6, 8}
    rng = np.random.default_rng(0)
    for trial in range(6):
        o, st = _oracle(m)
        mg = float(m.slot_margin[0])                                # the plate's relative contact breaking threshold (a few mm)
        assert 0.002 < mg < 0.01
        tilt = rng.normal(0, 0.15 * mg / 0.14, 2)                   # a random corner is the deepest (corners 14 cm from the centre)
        q = np.array([tilt[0] / 2, tilt[1] / 2, 0.0, 1.0]); q /= np.linalg.norm(q)
        st[0, 2], st[0, 3:7] = 0.12 + 0.4 * mg, q                   # all nine within the margin, none penetrating much
        o.set_state(st)
        o.physics_substeps(0, np.zeros(0), 1)
        kept = {int(c[2]) for c in o.last_contacts()}
        assert kept == corners, (trial, kept)
        dbg = o.get_debug()[0]
        assert dbg[2] == 4 and (int(dbg[3]) & 0x1FF) == sum(1 << k for k in corners)      # the slot mask is the manifold's, not all nine
    o, st = _oracle(m)
    st[0, 2] = 0.12 + 0.4 * mg
    o.set_state(st)
    o.physics_substeps(0, np.zeros(0), 480)
    s = o.get_state()[0]
    assert abs(s[2] - 0.12) < 2e-4 and np.abs(s[7:13]).max() < 1e-3
    np.testing.assert_allclose(s[13:13 + 9].sum(), m.mass[0] * G * DT, rtol=1e-6)
    np.testing.assert_allclose(s[13:13 + 9][sorted(corners)], m.mass[0] * G * DT / 4, rtol=0.03)
    assert np.abs(s[13:13 + 9][[1, 3, 4, 5, 7]]).max() == 0.0       # the five inner / edge points carry nothing: they are not contacts
    m.manifold_max = 0                                              # switched off: all nine are contacts again
    o, st = _oracle(m)
    st[0, 2] = 0.12 + 0.4 * mg
    o.set_state(st)
    o.physics_substeps(0, np.zeros(0), 1)
    assert len(o.last_contacts()) == 9


def test_link_damping_is_btmultibodys_law_on_every_link():
    """btMultiBody drags the base AND every link: force m v (k + k |v|) through the COM, torque Ic w (k + k |w|), k = 0.04 (its
    computeAccelerationsArticulatedBodyAlgorithmMultiDof, "adding damping terms (only)"; restated from the published source as recalled,
    [UNVERIFIED-BULLET]).  Closed forms: a mechanism translating as a whole decelerates by k (1 + |v|) v whatever its mass distribution
    (every link at the same velocity), and leaves its joints alone; a body spinning about a principal axis by k (1 + |w|) w."""
    m = M.compile_walker3d()
    assert abs(m.lin_damp - 0.04) < 1e-7 and abs(m.ang_damp - 0.04) < 1e-7
    m.gravity = 0.0
    o = Oracle(m.to_bytes(), M.TASK_WALKER3D_CUSTOM, 1, "f64")
    o.reset(seed=0)
    st = o.get_state()
    nj = m.n_joints
    st[0, 2] = 5.0                                   # far from the ground
    st[0, 7:13] = 0; st[0, 13 + nj:13 + 2 * nj] = 0
    v0 = np.array([1.5, -2.0, 0.5])
    st[0, 7:10] = v0
    o.set_state(st)
    o.physics_substeps(0, np.zeros(nj), 1)
    s1 = o.get_state()[0]
    k = float(m.lin_damp) * (1 + np.linalg.norm(v0))         # the blob holds float32 numbers
    np.testing.assert_allclose(s1[7:10], v0 * (1 - float(m.dt) * k), rtol=0, atol=1e-12)
    assert np.abs(s1[10:13]).max() < 1e-12 and np.abs(s1[13 + nj:13 + 2 * nj]).max() < 1e-10   # a uniform field: no relative motion
    # one rigid body spinning about a principal axis, at rest otherwise (the block's COM is not its frame origin: the force term vanishes
    # only because the COM itself is at rest, so spin it about the axis through the COM -- z, on which the COM lies)
    b = _block()
    b.lin_damp = b.ang_damp = 0.04
    b.gravity = 0.0
    o, st = _oracle(b)
    st[0, 2] = 5.0
    st[0, 12] = 3.0                                   # w_z
    o.set_state(st)
    o.physics_substeps(0, np.zeros(0), 1)
    s1 = o.get_state()[0]
    np.testing.assert_allclose(s1[12], 3.0 * (1 - float(b.dt) * float(b.ang_damp) * (1 + 3.0)), rtol=0, atol=1e-12)
    assert np.abs(s1[7:12]).max() < 1e-12


@pytest.mark.parametrize("cone", [1, 0])
def test_sliding_friction_is_a_cone_or_a_pyramid(cone):
    """A block sliding along the diagonal of the two friction directions (btPlaneSpace1 of a vertical normal: -y and x).  With Bullet's
    implicit cone friction (the blobs' default, `friction_cone`) the pair of friction impulses is clipped to the circle: the block
    decelerates by mu g along its velocity.  With the pyramid (`enableConeFriction = 0`) each direction is clipped on its own: sqrt(2) mu g."""
    m = _block()
    assert m.friction_cone == 1
    m.friction_cone = cone
    mu = float(m.ground_friction) * 0.5
    o, st = _oracle(m)
    st[0, 2] = 0.12
    o.set_state(st)
    o.physics_substeps(0, np.zeros(0), 60)                  # settle
    st = o.get_state()
    v0 = np.array([1.5, 1.5, 0.0])
    st[0, 7:10] = v0
    o.set_state(st)
    n = 48
    o.physics_substeps(0, np.zeros(0), n)
    v1 = o.get_state()[0, 7:10]
    decel = (np.linalg.norm(v0) - np.linalg.norm(v1[:2])) / (n * DT)
    want = mu * G * (1.0 if cone else np.sqrt(2.0))
    np.testing.assert_allclose(decel, want, rtol=0.02)
    assert abs(v1[0] - v1[1]) < 1e-3 * np.linalg.norm(v1)   # still along the diagonal
