"""An INDEPENDENT f64 restatement of one physics substep, in dense maximal-textbook form, for pinning the oracle.

oracle/mocca_oracle.c (and the HIP kernel after it) computes a substep with O(n) recursions: Featherstone's articulated
body algorithm about the moving base origin, unit-impulse responses by tree sweeps, a Delassus matrix assembled from
them.  Kernel and oracle share one author's reading of those recursions, so agreement between them proves nothing about
the recursions themselves.  This module recomputes the same substep with none of that machinery:

  * forward kinematics in absolute world coordinates, written for complex numbers;
  * every Jacobian by COMPLEX-STEP differentiation of that FK (machine-precision derivatives, no analytic Jacobian
    formulas, no spatial vectors): J[:, k] = Im x(config + i h e_k) / h;
  * the mass matrix from kinetic energy  M = sum_b m Jc^T Jc + Jw^T (R I R^T) Jw  (+ joint armature);
  * velocity-product terms from the acceleration of points along the constant-velocity path (central differences of
    the complex-step Jacobians), Newton-Euler per body, projected with the Jacobians (Kane / d'Alembert);
  * unconstrained velocity  nu* = nu + dt M^-1 (Q - h)  by a dense solve;
  * contact detection with its own closest-point routines (candidate enumeration instead of sequential clamping);
  * constraint rows as dense Jacobian rows, Delassus matrix  A = J M^-1 J^T  by a dense solve, the same projected
    Gauss-Seidel ORDER and iteration count (that order is part of the algorithm being checked);
  * symplectic Euler + quaternion exponential map.

What is shared with the oracle is only the MEANING of the blob's fields (include/mocca_model.h) and the published
constants of the algorithm (ERP, CFM, margin, warm-start factor, btPlaneSpace1 friction directions, row order).
tests/test_oracle_dense.py holds the oracle to this reference at ~1e-9.

TEST INFRASTRUCTURE ONLY (never imported by mocca_envs_amd/).
"""
from __future__ import annotations

import numpy as np

H = 1e-30  # complex step


def _rot_axis(axis, th):
    """Rodrigues; works for complex th."""
    ax = np.asarray(axis, dtype=complex if np.iscomplexobj(th) else float)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def _quat_mat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def _expm_so3(w):
    th = np.linalg.norm(w)
    if th < 1e-14:
        return np.eye(3) + _skew(w)
    K = _skew(w / th)
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


class Model:
    """Plain-numpy view of a MoccaModel blob."""

    def __init__(self, m):
        nb = self.nb = m.n_bodies
        self.nj = m.n_joints
        self.nd = 6 + self.nj
        self.parent = [m.parent[b] for b in range(nb)]
        self.jpos = np.array([list(m.jpos[b]) for b in range(nb)], float)
        self.jrot = np.array([list(m.jrot[b]) for b in range(nb)], float).reshape(nb, 3, 3)
        self.jaxis = np.array([list(m.jaxis[b]) for b in range(nb)], float)
        self.jlo = np.array([m.jlo[b] for b in range(nb)], float)
        self.jhi = np.array([m.jhi[b] for b in range(nb)], float)
        self.jdamp = np.array([m.jdamp[b] for b in range(nb)], float)
        self.jarm = np.array([m.jarm[b] for b in range(nb)], float)
        self.mass = np.array([m.mass[b] for b in range(nb)], float)
        self.com = np.array([list(m.com[b]) for b in range(nb)], float)
        self.Il = np.zeros((nb, 3, 3))
        for b in range(nb):
            xx, yy, zz, xy, xz, yz = m.inertia[b]
            self.Il[b] = [[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]]
        self.anc = [[] for _ in range(nb)]
        for b in range(1, nb):
            self.anc[b] = self.anc[self.parent[b]] + [b]
        self.geoms = [dict(body=m.g_body[g], capsule=m.g_type[g] == 1, slot=m.g_slot[g], terrain=bool(m.g_terrain[g]),
                           radius=float(m.g_radius[g]), p=[np.array(list(m.g_p1[g]), float), np.array(list(m.g_p2[g]), float)],
                           friction=float(m.g_friction[g]),
                           margin=float(m.slot_margin[m.g_slot[g]])) for g in range(m.n_geoms)]   # the link's relative threshold, as quantised in the blob
        self.pairs = [(m.pair_a[k], m.pair_b[k]) for k in range(m.n_pairs)]
        self.closures = [dict(a=m.cl_body_a[c], b=m.cl_body_b[c], pa=np.array(list(m.cl_point_a[c]), float),
                              pb=np.array(list(m.cl_point_b[c]), float)) for c in range(m.n_closures)]
        for k in ("gravity", "dt", "n_iters", "erp", "erp_noncontact", "friction_cone", "limit_at_violation", "contact_margin", "lin_damp", "ang_damp", "max_qd", "warmstart", "ground_friction",
                  "plank_friction", "plank_stiffness", "plank_damping", "limit_slack", "plank_com_z", "max_contacts", "max_rows", "n_slots", "manifold_max", "linear_slop", "sweep_alternate"):
            setattr(self, k, getattr(m, k))
        self.gravity, self.dt = float(np.float32(self.gravity)), float(np.float32(self.dt))
        self.plank_half = np.array(list(m.plank_half), float)


class State:
    def __init__(self, pos, quat, vel, omg, q, qd, warm):
        self.pos, self.quat, self.vel, self.omg = (np.array(x, float) for x in (pos, quat, vel, omg))
        self.q, self.qd, self.warm = np.array(q, float), np.array(qd, float), np.array(warm, float)  # q, qd indexed by body (entry 0 unused)

    @classmethod
    def from_row(cls, mdl: Model, row):
        nj = mdl.nj
        q = np.concatenate([[0.0], row[13:13 + nj]])
        qd = np.concatenate([[0.0], row[13 + nj:13 + 2 * nj]])
        return cls(row[0:3], row[3:7], row[7:10], row[10:13], q, qd, row[13 + 2 * nj:13 + 2 * nj + mdl.n_slots])

    def to_row(self, mdl: Model):
        return np.concatenate([self.pos, self.quat, self.vel, self.omg, self.q[1:], self.qd[1:], self.warm])

    def nu(self):
        return np.concatenate([self.omg, self.vel, self.qd[1:]])


def fk(mdl: Model, pos, R0, q):
    """Absolute world frames of every body: rotation R[b], origin o[b].  Complex-safe."""
    dt = complex if (np.iscomplexobj(pos) or np.iscomplexobj(R0) or np.iscomplexobj(q)) else float
    R = np.zeros((mdl.nb, 3, 3), dt)
    o = np.zeros((mdl.nb, 3), dt)
    R[0], o[0] = R0, pos
    for b in range(1, mdl.nb):
        p = mdl.parent[b]
        R[b] = R[p] @ mdl.jrot[b] @ _rot_axis(mdl.jaxis[b], q[b])
        o[b] = o[p] + R[p] @ mdl.jpos[b]
    return R, o


def _perturbed(mdl, st, k):
    """Configuration displaced by i*H along generalised direction k of nu = [omega_world, v_base_origin, qd]."""
    pos, R0, q = st.pos.astype(complex), _quat_mat(st.quat).astype(complex), st.q.astype(complex)
    if k < 3:
        e = np.zeros(3); e[k] = 1
        R0 = (np.eye(3) + 1j * H * _skew(e)) @ R0      # rotation about the base origin, world axis k (first order is all a complex step sees)
    elif k < 6:
        pos[k - 3] += 1j * H
    else:
        q[k - 5] += 1j * H
    return fk(mdl, pos, R0, q)


def jacobians(mdl: Model, st: State, points):
    """points: list of (body, local point).  Returns world positions X[n][3], point Jacobians Jp[n][3][nd], and per-body
    angular Jacobians Jw[nb][3][nd], rotations R[nb]."""
    R, o = fk(mdl, st.pos, _quat_mat(st.quat), st.q)
    X = np.array([o[b] + R[b] @ pl for b, pl in points]) if points else np.zeros((0, 3))
    Jp = np.zeros((len(points), 3, mdl.nd))
    Jw = np.zeros((mdl.nb, 3, mdl.nd))
    for k in range(mdl.nd):
        Rk, ok = _perturbed(mdl, st, k)
        for n, (b, pl) in enumerate(points):
            Jp[n, :, k] = (ok[b] + Rk[b] @ pl).imag / H
        for b in range(mdl.nb):
            W = (Rk[b].imag / H) @ R[b].T         # dR R^T = [omega]x
            Jw[b, :, k] = [W[2, 1], W[0, 2], W[1, 0]]
    return X, Jp, Jw, R


def mass_matrix(mdl, st):
    X, Jc, Jw, R = jacobians(mdl, st, [(b, mdl.com[b]) for b in range(mdl.nb)])
    M = np.zeros((mdl.nd, mdl.nd))
    for b in range(mdl.nb):
        Iw = R[b] @ mdl.Il[b] @ R[b].T
        M += mdl.mass[b] * Jc[b].T @ Jc[b] + Jw[b].T @ Iw @ Jw[b]
    for b in range(1, mdl.nb):
        M[5 + b, 5 + b] += mdl.jarm[b]
    return M, X, Jc, Jw, R


def _along_path(mdl, st, t):
    """The state reached after time t at CONSTANT generalised velocity (omega, v_origin, qd fixed)."""
    R0 = _expm_so3(st.omg * t) @ _quat_mat(st.quat)
    s = State(st.pos + st.vel * t, st.quat, st.vel, st.omg, st.q + st.qd * t, st.qd, st.warm)
    s._R0 = R0
    return s


def unconstrained_velocity(mdl: Model, st: State, tau):
    """nu* = nu + dt M^-1 (Q - h): gravity, joint torques, joint damping, base damping, velocity-product terms."""
    M, Xc, Jc, Jw, R = mass_matrix(mdl, st)
    nu = st.nu()
    # acceleration of each COM / angular acceleration of each body at zero generalised acceleration: d/dt [J(q(t)) nu] along the
    # constant-velocity path, by central differences of the complex-step Jacobians
    eps = 1e-6

    def jn(t):
        s = _along_path(mdl, st, t)
        R0 = s._R0
        Rr, orr = fk(mdl, s.pos, R0, s.q)
        vc = np.zeros((mdl.nb, 3)); wb = np.zeros((mdl.nb, 3))
        for k in range(mdl.nd):
            pos, R0c, q = s.pos.astype(complex), R0.astype(complex), s.q.astype(complex)
            if k < 3:
                e = np.zeros(3); e[k] = 1
                R0c = (np.eye(3) + 1j * H * _skew(e)) @ R0c
            elif k < 6:
                pos[k - 3] += 1j * H
            else:
                q[k - 5] += 1j * H
            Rk, ok = fk(mdl, pos, R0c, q)
            for b in range(mdl.nb):
                vc[b] += (ok[b] + Rk[b] @ mdl.com[b]).imag / H * nu[k]
                W = (Rk[b].imag / H) @ Rr[b].T
                wb[b] += np.array([W[2, 1], W[0, 2], W[1, 0]]) * nu[k]
        return vc, wb

    vcp, wbp = jn(eps)
    vcm, wbm = jn(-eps)
    a_bias, al_bias = (vcp - vcm) / (2 * eps), (wbp - wbm) / (2 * eps)
    h = np.zeros(mdl.nd)
    Q = np.zeros(mdl.nd)
    for b in range(mdl.nb):
        Iw = R[b] @ mdl.Il[b] @ R[b].T
        wb = Jw[b] @ nu
        h += Jc[b].T @ (mdl.mass[b] * a_bias[b]) + Jw[b].T @ (Iw @ al_bias[b] + np.cross(wb, Iw @ wb))
        Q += Jc[b].T @ np.array([0, 0, -mdl.gravity * mdl.mass[b]])
    for b in range(1, mdl.nb):
        Q[5 + b] += tau[b] - mdl.jdamp[b] * st.qd[b]
    # link damping (btMultiBody: base and every link): force m vc (k + k |vc|) through the COM, torque Ic w (k + k |w|)
    for b in range(mdl.nb):
        Iw = R[b] @ mdl.Il[b] @ R[b].T
        vc, wb = Jc[b] @ nu, Jw[b] @ nu
        Q -= Jc[b].T @ (mdl.lin_damp * (1 + np.linalg.norm(vc)) * mdl.mass[b] * vc)
        Q -= Jw[b].T @ (mdl.ang_damp * (1 + np.linalg.norm(wb)) * (Iw @ wb))
    return nu + mdl.dt * np.linalg.solve(M, Q - h), M


def _closest_seg_seg(p1, q1, p2, q2):
    """Closest points of two segments by candidate enumeration: the interior stationary point if it lies in the unit square,
    else the best of the four edges (each a point-to-segment projection)."""
    d1, d2, r = q1 - p1, q2 - p2, p1 - p2
    a, e, b, c, f = d1 @ d1, d2 @ d2, d1 @ d2, d1 @ r, d2 @ r
    cand = []
    den = a * e - b * b
    if den > 1e-12 * max(a * e, 1e-300) and a > 0 and e > 0:
        s, t = (b * f - c * e) / den, (a * f - b * c) / den
        if 0 <= s <= 1 and 0 <= t <= 1:
            cand.append((s, t))
    clamp = lambda x: min(1.0, max(0.0, x))
    for s in (0.0, 1.0):
        t = clamp((f + b * s) / e) if e > 0 else 0.0
        cand.append((s, t))
    for t in (0.0, 1.0):
        s = clamp((b * t - c) / a) if a > 0 else 0.0
        cand.append((s, t))
    s, t = min(cand, key=lambda st_: np.linalg.norm((p1 + d1 * st_[0]) - (p2 + d2 * st_[1])))
    return p1 + d1 * s, p2 + d2 * t


def _plane_space(n):  # btPlaneSpace1 (Bullet's published friction-direction convention)
    if abs(n[2]) > 0.7071067811865475244:
        a = n[1] * n[1] + n[2] * n[2]; k = 1 / np.sqrt(a)
        t1 = np.array([0, -n[2] * k, n[1] * k])
        t2 = np.array([a * k, -n[0] * t1[2], n[0] * t1[1]])
    else:
        a = n[0] * n[0] + n[1] * n[1]; k = 1 / np.sqrt(a)
        t1 = np.array([-n[1] * k, n[0] * k, 0])
        t2 = np.array([-n[2] * t1[1], n[2] * t1[0], a * k])
    return t1, t2


def _euler_mat(roll, pitch, yaw):
    return _rot_axis([0, 0, 1], yaw) @ _rot_axis([0, 1, 0], pitch) @ _rot_axis([1, 0, 0], roll)


def _closest_on_triangle(p, a, b, c):
    """Closest point of triangle abc to p, NOT by Voronoi regions (the oracle's way): the unconstrained minimiser of
    |a + s (b - a) + t (c - a) - p|^2 if it lies inside, else the best of the three edges by clamped projection."""
    e1, e2, d = b - a, c - a, p - a
    G = np.array([[e1 @ e1, e1 @ e2], [e1 @ e2, e2 @ e2]])
    s, t = np.linalg.solve(G, [e1 @ d, e2 @ d])
    if s >= 0 and t >= 0 and s + t <= 1:
        return a + s * e1 + t * e2
    best, bd = None, 1e300
    for u, v in ((a, b), (b, c), (c, a)):
        w = v - u
        k = min(1.0, max(0.0, (p - u) @ w / (w @ w)))
        q = u + k * w
        if (p - q) @ (p - q) < bd:
            best, bd = q, (p - q) @ (p - q)
    return best


def heightfield_surface(data, scale, x, y):
    """(height of the piecewise-linear surface under (x, y), unit normal of the triangle there) or None outside the grid."""
    rows, cols = data.shape
    X = (np.arange(cols) - (cols - 1) / 2) / scale
    Y = (np.arange(rows) - (rows - 1) / 2) / scale
    i, j = int(np.searchsorted(X, x, side="right")) - 1, int(np.searchsorted(Y, y, side="right")) - 1
    if i < 0 or j < 0 or i > cols - 2 or j > rows - 2:
        return None
    u, v = (x - X[i]) * scale, (y - Y[j]) * scale
    P = lambda ii, jj: np.array([X[ii], Y[jj], data[jj, ii]])
    tri = (P(i, j), P(i + 1, j), P(i, j + 1)) if u + v <= 1 else (P(i + 1, j), P(i + 1, j + 1), P(i, j + 1))
    tn = np.cross(tri[1] - tri[0], tri[2] - tri[0]); tn /= np.linalg.norm(tn)
    h = tri[0][2] - (tn[0] * (x - tri[0][0]) + tn[1] * (y - tri[0][1])) / tn[2]
    return h, tn


def heightfield_window(scale, reach):
    """Cells each way of the nearest grid point that a sphere of reach = radius + margin can touch (its centre is at most half a cell from
    that point): ceil(reach x scale + 1/2); 1e-6: a reach of exactly half a cell stays at 1."""
    return max(1, int(np.ceil(reach * scale + 0.5 - 1e-6)))


def heightfield_gap(data, scale, C, radius, window=None, margin=0.0):
    """Signed gap and normal of a sphere against a height field (tests only; conventions in include/mocca.h mocca_set_heightfield):
    explicit vertex coordinate arrays; centre below the surface: distance to the plane of the triangle above it; otherwise every triangle
    of the cells within `window` cells of the nearest grid point (default: heightfield_window(scale, radius + margin))."""
    if window is None:
        window = heightfield_window(scale, radius + margin)
    rows, cols = data.shape
    X = (np.arange(cols) - (cols - 1) / 2) / scale
    Y = (np.arange(rows) - (rows - 1) / 2) / scale
    if C[0] < X[0] - 1 / scale or C[0] > X[-1] + 1 / scale or C[1] < Y[0] - 1 / scale or C[1] > Y[-1] + 1 / scale:
        return 1e30, np.array([0.0, 0.0, 1.0])
    under = heightfield_surface(data, scale, C[0], C[1])
    if under is not None and C[2] < under[0]:
        return (C[2] - under[0]) * under[1][2] - radius, under[1]       # distance to the plane = vertical distance x n_z
    iv, jv = int(np.floor((C[0] - X[0]) * scale + 0.5)), int(np.floor((C[1] - Y[0]) * scale + 0.5))
    gap, n = 1e30, np.array([0.0, 0.0, 1.0])
    for j in range(jv - window, jv + window):
        for i in range(iv - window, iv + window):
            if i < 0 or j < 0 or i > cols - 2 or j > rows - 2:
                continue
            V = lambda ii, jj: np.array([X[ii], Y[jj], data[jj, ii]])
            for tri in ((V(i, j), V(i + 1, j), V(i, j + 1)), (V(i + 1, j), V(i + 1, j + 1), V(i, j + 1))):
                q = _closest_on_triangle(C, *tri)
                d = C - q
                dist = np.linalg.norm(d)
                if dist - radius < gap:
                    tn = np.cross(tri[1] - tri[0], tri[2] - tri[0])
                    gap, n = dist - radius, (d / dist if dist > 1e-9 else tn / np.linalg.norm(tn))
    return gap, n


def detect_contacts(mdl: Model, st: State, planks=None, heightfield=None):
    """planks: None (flat ground z = 0) or list of (box centre, rotation, is_target) for the live planks; heightfield: (data, scale) of
    the planner envs.  Returns contacts in the oracle's priority order: terrain slots, then self pairs; capped at max_contacts."""
    R, o = fk(mdl, st.pos, _quat_mat(st.quat), st.q)
    out, slot_mask, n_self = [], 0, 0
    for g in mdl.geoms:
        if not g["terrain"]:
            continue
        for e in range(2 if g["capsule"] else 1):
            C = o[g["body"]] + R[g["body"]] @ g["p"][e]
            if heightfield is not None:
                gap, n = heightfield_gap(heightfield[0], heightfield[1], C, g["radius"], margin=g["margin"])
                kk, cc, dt = mdl.plank_stiffness, mdl.plank_damping, mdl.dt
                mu, erp, cfm = mdl.plank_friction * g["friction"], dt * kk / (dt * kk + cc), 1 / (dt * kk + cc) / dt
            elif planks is None:
                gap, n = C[2] - g["radius"], np.array([0.0, 0.0, 1.0])
                mu, erp, cfm = mdl.ground_friction * g["friction"], mdl.erp, 0.0
            else:
                gap = 1e30
                for bc, Rb, _ in planks:
                    l = Rb.T @ (C - bc)
                    qc = np.clip(l, -mdl.plank_half, mdl.plank_half)
                    if np.all(np.abs(l) <= mdl.plank_half):          # centre inside: leave through the nearest face
                        dpt = mdl.plank_half - np.abs(l)
                        ax = int(np.argmin(dpt))
                        nl = np.zeros(3); nl[ax] = 1.0 if l[ax] >= 0 else -1.0
                        dist = -dpt[ax]
                    else:
                        dist = np.linalg.norm(l - qc); nl = (l - qc) / dist
                    if dist - g["radius"] < gap:
                        gap, n = dist - g["radius"], Rb @ nl
                kk, cc, dt = mdl.plank_stiffness, mdl.plank_damping, mdl.dt
                mu, erp, cfm = mdl.plank_friction * g["friction"], dt * kk / (dt * kk + cc), 1 / (dt * kk + cc) / dt
            if gap < g["margin"]:              # the link's relative contact breaking threshold
                slot_mask |= 1 << (g["slot"] + e)
                out.append(dict(a=g["body"], b=-1, slot=g["slot"] + e, P=C - g["radius"] * n, n=n, depth=-gap, mu=mu, erp=erp, cfm=cfm))
    if getattr(mdl, "manifold_max", 0) > 0:   # 4-point manifold per link: deepest, farthest from it, farthest to either side of that line
        keep = np.ones(len(out), bool)
        for body in sorted({c["a"] for c in out}):
            idx = [i for i, c in enumerate(out) if c["a"] == body]
            if len(idx) <= 4:
                continue
            Pm = np.array([out[i]["P"] for i in idx]); dep = np.array([out[i]["depth"] for i in idx])
            k1 = int(np.argmax(dep))                                    # (argmax returns the first of equals: the lower slot)
            dist = np.linalg.norm(Pm - Pm[k1], axis=1); dist[k1] = -1
            k2 = int(np.argmax(dist))
            area = np.cross(Pm - Pm[k1], Pm[k2] - Pm[k1]) @ out[idx[k1]]["n"]
            area[[k1, k2]] = 0
            chosen = {k1, k2}
            if area.max() > 0:
                chosen.add(int(np.argmax(area)))
            if area.min() < 0:
                chosen.add(int(np.argmin(area)))
            for k, i in enumerate(idx):
                keep[i] = k in chosen
        for i, c in enumerate(out):
            if not keep[i]:
                slot_mask &= ~(1 << c["slot"])
        out = [c for i, c in enumerate(out) if keep[i]]
    if len(out) > mdl.max_contacts:      # more terrain contacts than the solver holds: the deepest stay (ties: lower slot), in slot order
        order = sorted(range(len(out)), key=lambda i: (-out[i]["depth"], i))[:mdl.max_contacts]
        out = [out[i] for i in sorted(order)]
    for ga, gb in mdl.pairs:
        A, B = mdl.geoms[ga], mdl.geoms[gb]
        a1, a2 = (o[A["body"]] + R[A["body"]] @ A["p"][k] for k in (0, 1))
        b1, b2 = (o[B["body"]] + R[B["body"]] @ B["p"][k] for k in (0, 1))
        ca, cb = _closest_seg_seg(a1, a2, b1, b2)
        d = ca - cb
        dist = np.linalg.norm(d)
        gap = dist - A["radius"] - B["radius"]
        if gap < min(A["margin"], B["margin"]) and dist > 1e-9:
            n_self += 1
            if len(out) < mdl.max_contacts:
                n = d / dist
                out.append(dict(a=A["body"], b=B["body"], slot=-1, P=0.5 * ((ca - A["radius"] * n) + (cb + B["radius"] * n)), n=n,
                                depth=-gap, mu=A["friction"] * B["friction"], erp=mdl.erp, cfm=0.0))
    return out, R, o, slot_mask, n_self


def substep(mdl: Model, st: State, tau, planks=None, heightfield=None):
    """One physics substep.  Returns (new State, info dict with rows / impulses / contacts)."""
    dt = mdl.dt
    contacts, R, o, slot_mask, n_self = detect_contacts(mdl, st, planks, heightfield)
    nus, M = unconstrained_velocity(mdl, st, tau)
    # material points the rows act on
    pts = []
    for c in contacts:
        pts.append((c["a"], R[c["a"]].T @ (c["P"] - o[c["a"]])))
        if c["b"] >= 0:
            pts.append((c["b"], R[c["b"]].T @ (c["P"] - o[c["b"]])))
    for cl in mdl.closures:
        pts += [(cl["a"], cl["pa"]), (cl["b"], cl["pb"])]
    X, Jp, _, _ = jacobians(mdl, st, pts)
    rows = []   # dict(J, bias, cfm, lo, hi, lam, kind, normal, mu, slot)
    for b in range(1, mdl.nb):
        for side, sgn in ((0, 1.0), (1, -1.0)):
            gap = st.q[b] - mdl.jlo[b] if side == 0 else mdl.jhi[b] - st.q[b]
            if (gap > 0 if mdl.limit_at_violation else gap + dt * sgn * nus[5 + b] >= mdl.limit_slack) or len(rows) >= mdl.max_rows:
                continue
            J = np.zeros(mdl.nd); J[5 + b] = sgn
            rows.append(dict(J=J, bias=mdl.erp_noncontact * (-gap) / dt if gap < 0 else -gap / dt, cfm=0.0, lo=0.0, hi=1e30, lam=0.0, kind=0, slot=-1))
    n_limit = len(rows)
    ip = len(pts) - 2 * len(mdl.closures)
    for ci, cl in enumerate(mdl.closures):
        Pa, Pb = X[ip + 2 * ci], X[ip + 2 * ci + 1]
        for ax in range(3):
            if len(rows) >= mdl.max_rows:
                break
            rows.append(dict(J=Jp[ip + 2 * ci][ax] - Jp[ip + 2 * ci + 1][ax], bias=mdl.erp_noncontact * (Pb[ax] - Pa[ax]) / dt, cfm=0.0, lo=-1e30, hi=1e30,
                             lam=0.0, kind=3, slot=-1))
    nc = min(len(contacts), (mdl.max_rows - len(rows)) // 3)
    first_normal = len(rows)
    cj = []
    ip = 0
    for c in contacts:
        Ja = Jp[ip]; ip += 1
        if c["b"] >= 0:
            Ja = Ja - Jp[ip]; ip += 1
        cj.append(Ja)
    for i in range(nc):
        c = contacts[i]
        lam0 = mdl.warmstart * st.warm[c["slot"]] if c["slot"] >= 0 else 0.0
        dep = c["depth"] - float(np.float32(mdl.linear_slop))      # penetration = distance + slop
        rows.append(dict(J=c["n"] @ cj[i], bias=c["erp"] * dep / dt if dep > 0 else dep / dt, cfm=c["cfm"], lo=0.0, hi=1e30,
                         lam=lam0, kind=1, slot=c["slot"]))
    for i in range(nc):
        c = contacts[i]
        for tdir in _plane_space(c["n"]):
            rows.append(dict(J=tdir @ cj[i], bias=0.0, cfm=0.0, lam=0.0, kind=2, normal=first_normal + i, mu=c["mu"], slot=-1))
    nr = len(rows)
    lam = np.array([r["lam"] for r in rows])
    if nr:
        J = np.array([r["J"] for r in rows])
        Mi = np.linalg.solve(M, J.T)              # nd x nr
        A = J @ Mi
        w = J @ nus + A @ lam                      # warm-start impulses act before the first iteration
        for it in range(mdl.n_iters):
            skip = False
            rev = bool(mdl.sweep_alternate) and not (it & 1)     # Bullet: the non-contact rows last-to-first in the even iterations
            for ro in range(nr):
                if skip:
                    skip = False
                    continue
                r = first_normal - 1 - ro if (rev and ro < first_normal) else ro
                row = rows[r]
                if mdl.friction_cone and row["kind"] == 2:      # implicit cone friction: the pair from one velocity state, clipped to the circle
                    lim = row["mu"] * lam[row["normal"]]
                    skip = True
                    if not lam[row["normal"]] > 0:               # Bullet: `if (totalImpulse > 0)`, else the pair is left as it is
                        continue
                    cand = []
                    for q in (r, r + 1):
                        den = A[q, q] + rows[q]["cfm"]
                        cand.append(lam[q] + ((rows[q]["bias"] - w[q] - rows[q]["cfm"] * lam[q]) / den if den > 1e-12 else 0.0))
                    rad = np.hypot(*cand)
                    sc = lim / rad if rad > lim else 1.0
                    for q, sq in zip((r, r + 1), cand):
                        new = sq * sc
                        w += A[:, q] * (new - lam[q])
                        lam[q] = new
                    skip = True
                    continue
                lo, hi = (row["lo"], row["hi"]) if row["kind"] != 2 else (-row["mu"] * lam[row["normal"]], row["mu"] * lam[row["normal"]])
                if row["kind"] == 2 and not lam[row["normal"]] > 0:   # pyramid: the same test
                    continue
                den = A[r, r] + row["cfm"]
                dl = (row["bias"] - w[r] - row["cfm"] * lam[r]) / den if den > 1e-12 else 0.0
                new = min(hi, max(lo, lam[r] + dl))
                w += A[:, r] * (new - lam[r])
                lam[r] = new
        nu = nus + Mi @ lam
    else:
        nu = nus.copy()
    warm = np.zeros(mdl.n_slots)
    for r, row in enumerate(rows):
        if row["slot"] >= 0:
            warm[row["slot"]] = lam[r]
    qd = np.clip(nu[5:], -mdl.max_qd, mdl.max_qd); qd[0] = 0.0   # entry 0 = placeholder for the base
    omg, vel = nu[0:3], nu[3:6]
    q = st.q + dt * qd
    pos = st.pos + dt * vel
    Rn = _expm_so3(omg * dt) @ _quat_mat(st.quat)
    new = State(pos, _mat_quat(Rn), vel, omg, q, qd, warm)
    return new, dict(rows=nr, n_limit=n_limit, nc=nc, lam=lam, kinds=[r["kind"] for r in rows], contacts=contacts, nu_star=nus,
                     slot_mask=slot_mask, n_self=n_self, Rn=Rn)


def _mat_quat(R):
    """(x, y, z, w) of a rotation matrix (w >= 0 branch is enough for comparisons through the matrix)."""
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    if w > 1e-6:
        return np.array([(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w])
    x = np.sqrt(max(0.0, 1 + R[0, 0] - R[1, 1] - R[2, 2])) / 2
    return np.array([x, (R[0, 1] + R[1, 0]) / (4 * x), (R[0, 2] + R[2, 0]) / (4 * x), (R[2, 1] - R[1, 2]) / (4 * x)])


def live_planks(mdl: Model, terrain_row, next_step_index):
    """The three live planks of the Stepper from the oracle's terrain record (20 x 6 table + 3 plank rows)."""
    table, info = np.asarray(terrain_row[:120]).reshape(20, 6), np.asarray(terrain_row[120:123]).astype(int)
    out = []
    for k in range(3):
        x, y, z, phi, xt, yt = table[info[k]]
        Rb = _euler_mat(xt, yt, phi)
        cz = mdl.plank_com_z
        bc = np.array([x, y, z]) + Rb @ np.array([0, 0, -mdl.plank_half[2] - cz]) + np.array([0, 0, cz])
        out.append((bc, Rb, k == next_step_index % 3))
    return out
