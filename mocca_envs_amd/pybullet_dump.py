"""Model blob from a REAL PyBullet session: `tools/dump_pybullet_trace.py`, run on any machine
that has pybullet, records the multibody Bullet actually built from the robot file -- per link `getJointInfo`
(names, types, axes, limits, damping, parent frames, parent indices) and `getDynamicsInfo` (mass, local inertia
diagonal, inertial frame).  `from_pybullet_dump` turns that record into a `MoccaModel`: every number this project's
model compiler had to *assume* about Bullet's importer (link masses, inertial frames and principal inertias, the
joint frames, damping; DESIGN.md "Model assumptions") is replaced by what Bullet reported, link by link.

What stays from the compiled template (`model.compile_*`): the tree (checked against the dump), the collision geoms
(re-expressed in the dump's link frames: the geometry is the robot file's, the frames are Bullet's), the
self-collision pair list, feet, physics and task constants.

PyBullet conventions used (pybullet quick-start guide, getJointInfo / getDynamicsInfo):
  * a link's frame in the API is its centre-of-mass (inertial) frame C; the URDF/MJCF link frame L satisfies
    C = L o (localInertialPos, localInertialOrn);
  * parentFramePos / parentFrameOrn: the joint (= child link) frame expressed in the PARENT's inertial frame, so at
    q = 0   L_child = C_parent o (parentFramePos, parentFrameOrn);
  * jointAxis is given in the child link frame L; the base pose/velocity the API reports is that of the base's C frame;
  * localInertiaDiagonal is expressed in C's axes.
Quaternions are (x, y, z, w).
"""
from __future__ import annotations

from typing import Dict, Mapping, Optional, Sequence

import numpy as np

from . import model as M

JOINT_REVOLUTE, JOINT_FIXED, JOINT_POINT2POINT = 0, 4, 5
# solver parameters from_pybullet_dump takes from the record (as "engine_<name>"); tools/dump_pybullet_trace.py writes the ones pybullet reports and,
# for the ones the session itself set, the values it set
ENGINE_KEYS = ("erp", "contactERP", "numSolverIterations", "contactBreakingThreshold", "enableConeFriction")


def _qmat(q) -> np.ndarray:
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


class _T:
    """Rigid transform x -> R x + t."""

    def __init__(self, R=None, t=None):
        self.R = np.eye(3) if R is None else np.asarray(R, float)
        self.t = np.zeros(3) if t is None else np.asarray(t, float)

    def __matmul__(self, o: "_T") -> "_T":
        return _T(self.R @ o.R, self.R @ o.t + self.t)

    def inv(self) -> "_T":
        return _T(self.R.T, -self.R.T @ self.t)

    def apply(self, p) -> np.ndarray:
        return self.R @ np.asarray(p, float) + self.t


def _template_frames(m: M.MoccaModel):
    """Body frames of the template at q = 0, relative to its base frame."""
    fr = [_T()]
    for b in range(1, m.n_bodies):
        p = m.parent[b]
        fr.append(fr[p] @ _T(np.array(list(m.jrot[b])).reshape(3, 3), list(m.jpos[b])))
    return fr


def from_pybullet_dump(dump: Mapping[str, np.ndarray], template: M.MoccaModel, joint_names: Sequence[str],
                       armature: Optional[float] = None) -> M.MoccaModel:
    """`dump`: the arrays written by tools/dump_pybullet_trace.py (an open .npz works).  `template`: the compiled blob of
    the same robot.  `joint_names[b - 1]`: Bullet joint name of template body b (e.g. model.WALKER3D_JOINT_NAMES).
    `armature`: None keeps the template's joint armature (Bullet's API does not expose what its importer did with it)."""
    names = [str(n) for n in dump["joint_names"]]
    jtype = np.asarray(dump["joint_type"]).astype(int)
    parent = np.asarray(dump["parent_index"]).astype(int)          # -1 = base
    n_links = len(names)
    mass = np.asarray(dump["mass"], float)                         # index 0 = base, 1 + j = link j
    inertia_diag = np.asarray(dump["local_inertia_diag"], float)
    ipos, iorn = np.asarray(dump["inertial_pos"], float), np.asarray(dump["inertial_orn"], float)
    pf_pos, pf_orn = np.asarray(dump["parent_frame_pos"], float), np.asarray(dump["parent_frame_orn"], float)
    axis = np.asarray(dump["joint_axis"], float)
    if not (len(mass) == n_links + 1 and len(parent) == n_links):
        raise ValueError("dump arrays are inconsistent")

    # link frames L_j and inertial frames C_j at q = 0, relative to the base's inertial frame (= our base frame)
    I_of = lambda k: _T(_qmat(iorn[k]), ipos[k])                   # noqa: E731  (k = 0 base, 1 + j link j)
    C = {-1: _T()}
    L = {}
    for j in range(n_links):                                       # Bullet lists links parents-first
        if parent[j] >= j:
            raise ValueError("links are not in parents-first order")
        L[j] = C[parent[j]] @ _T(_qmat(pf_orn[j]), pf_pos[j])
        C[j] = L[j] @ I_of(1 + j)

    # template body b <-> Bullet link carrying the joint of that name; every other link must be rigidly attached (fixed)
    link_of = {}
    for b in range(1, template.n_bodies):
        nm = joint_names[b - 1]
        if nm not in names:
            raise ValueError(f"joint {nm!r} is not in the dump")
        j = names.index(nm)
        if jtype[j] != JOINT_REVOLUTE:
            raise ValueError(f"joint {nm!r} is not a hinge in the dump")
        link_of[b] = j
    body_of_link = {j: b for b, j in link_of.items()}

    def owner(j: int) -> int:
        """Template body a link moves with: itself if it carries an actuated hinge, else its nearest such ancestor (0 = base)."""
        while j >= 0 and j not in body_of_link:
            if jtype[j] != JOINT_FIXED:
                raise ValueError(f"link {names[j]!r} moves on a joint the template does not have")
            j = parent[j]
        return body_of_link[j] if j >= 0 else 0

    for b, j in link_of.items():
        if owner(parent[j]) != template.parent[b]:
            raise ValueError(f"tree mismatch at {names[j]!r}: dump parent body {owner(parent[j])}, template {template.parent[b]}")

    out = M.MoccaModel.from_bytes(template.to_bytes())
    # solver parameters of the recorded session (format >= 2 files written after round 3 carry getPhysicsEngineParameters())
    if "engine_erp" in dump:
        out.erp_noncontact = float(dump["engine_erp"])              # infoGlobal.m_erp: joint limits, point-to-point closures
    if "engine_contactERP" in dump:
        out.erp = float(dump["engine_contactERP"])                  # infoGlobal.m_erp2: contact rows (setDefaultContactERP)
    if "engine_numSolverIterations" in dump:
        out.n_iters = int(dump["engine_numSolverIterations"])
    if "engine_contactBreakingThreshold" in dump:                   # gContactBreakingThreshold: the factor of the links' relative margins
        f = float(dump["engine_contactBreakingThreshold"]) / float(template.contact_margin)
        out.contact_margin = float(dump["engine_contactBreakingThreshold"])
        for g in range(out.n_geoms):
            out.g_margin[g] = template.g_margin[g] * f
    if "engine_enableConeFriction" in dump:
        out.friction_cone = int(dump["engine_enableConeFriction"])   # 0: pyramid (SOLVER_DISABLE_IMPLICIT_CONE_FRICTION)
    if "engine_contactSlop" in dump:                                # infoGlobal.m_linearSlop (setPhysicsEngineParameter(contactSlop=...)); absent: the blob's own
        out.linear_slop = float(dump["engine_contactSlop"])
    missing = [k for k in ENGINE_KEYS if "engine_" + k not in dump]
    if missing:
        # pybullet.getPhysicsEngineParameters() is not guaranteed to report every solver parameter (older builds: fixedTimeStep, numSubSteps,
        # numSolverIterations, useRealTimeSimulation, gravity only): what is absent keeps this project's [UNVERIFIED-BULLET] default, and the
        # trace test then pins the default, not Bullet's value -- say so instead of implying the opposite
        import warnings
        warnings.warn("PyBullet dump: engine parameter(s) %s not recorded; the blob keeps the compiled defaults (erp_noncontact %.2f, contact erp %.2f, "
                      "friction_cone %d, contact margins x %.3f)" % (", ".join(missing), out.erp_noncontact, out.erp, out.friction_cone, out.contact_margin))
    if any(out.margin_code(g) >= 255 for g in range(out.n_geoms)):
        import warnings
        warnings.warn("PyBullet dump: a contact margin saturates the 8-bit code of the slot records (31 mm); the kernel uses the capped value")
    for key in ("rolling_friction", "spinning_friction", "restitution"):
        if key in dump and np.abs(np.asarray(dump[key], float)).max() > 0:
            raise ValueError(f"the dump reports non-zero {key} on a robot link: not modelled by this stepper (DESIGN.md section 9)")
    frame = {0: _T()}
    frame.update({b: L[j] for b, j in link_of.items()})
    # the template's base frame is the robot file's root BODY frame (DESIGN.md "Model assumptions"); Bullet's base frame is the
    # root link's inertial frame C_base = L_base o I_base, so in the dump's coordinates the template's frames sit at inv(I_base) o .
    base_link = I_of(0).inv()
    tf = [base_link @ t for t in _template_frames(template)]
    for b in range(1, out.n_bodies):
        j = link_of[b]
        rel = frame[out.parent[b]].inv() @ frame[b]
        for k in range(3):
            out.jpos[b][k] = rel.t[k]
            out.jaxis[b][k] = axis[j][k]
        for k in range(9):
            out.jrot[b][k] = rel.R.reshape(-1)[k]
        out.jlo[b], out.jhi[b] = float(dump["joint_limits"][j][0]), float(dump["joint_limits"][j][1])
        out.jdamp[b] = float(dump["joint_damping"][j])
        if armature is not None:
            out.jarm[b] = armature
        # the hinge axis must be the template's, seen from the new frame (same robot file): a sign flip would silently
        # reverse the joint coordinate
        ax_new = frame[b].R @ axis[j]
        ax_old = tf[b].R @ np.array(list(template.jaxis[b]))
        if ax_new @ ax_old < 0.99:
            raise ValueError(f"hinge axis of {names[j]!r} differs from the template's")

    # inertial parameters: every Bullet link contributes to the template body it moves with
    acc = {b: [0.0, np.zeros(3), []] for b in range(out.n_bodies)}
    for k in range(n_links + 1):
        j = k - 1
        b = 0 if j < 0 else owner(j)
        Cw = C[j]
        rel = frame[b].inv() @ Cw                                   # inertial frame of the link in the body's frame
        Ic = rel.R @ np.diag(inertia_diag[k]) @ rel.R.T
        acc[b][0] += mass[k]
        acc[b][1] += mass[k] * rel.t
        acc[b][2].append((mass[k], rel.t, Ic))
    for b in range(out.n_bodies):
        mb = acc[b][0]
        com = acc[b][1] / mb if mb > 0 else np.zeros(3)
        I = np.zeros((3, 3))
        for mk, ck, Ik in acc[b][2]:
            d = ck - com
            I += Ik + mk * ((d @ d) * np.eye(3) - np.outer(d, d))
        out.mass[b] = mb
        for k in range(3):
            out.com[b][k] = com[k]
        for k, (r, c) in enumerate([(0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2)]):
            out.inertia[b][k] = I[r, c]

    # geometry: the template's geoms (the robot file's shapes) re-expressed in the dump's link frames
    for g in range(out.n_geoms):
        b = out.g_body[g]
        re = frame[b].inv() @ tf[b]
        for src, dst in ((template.g_p1[g], out.g_p1[g]), (template.g_p2[g], out.g_p2[g])):
            p = re.apply(list(src))
            for k in range(3):
                dst[k] = p[k]
    for f in range(out.n_feet):
        fb = out.foot_body[f]
        if all(abs(template.foot_point[f][k] - template.com[fb][k]) < 1e-9 for k in range(3)):
            for k in range(3):
                out.foot_point[f][k] = out.com[fb][k]            # getLinkState()[0] = the foot link's centre of mass
        else:
            p = (frame[fb].inv() @ tf[fb]).apply(list(template.foot_point[f]))
            for k in range(3):
                out.foot_point[f][k] = p[k]
    # loop closures (Cassie: createConstraint(JOINT_POINT2POINT) tarsus <-> achilles rod, env_cassie.py:114-137): the record's
    # `constraints` rows [parent link, child link, joint type, parent pivot xyz, child pivot xyz] -- pivots in each link's INERTIAL frame,
    # as createConstraint takes them -- become the blob's pivots in its body frames
    if template.n_closures > 0:
        if "constraints" not in dump:
            raise ValueError("the template has loop closures but the dump records no constraints")
        rows = np.asarray(dump["constraints"], float).reshape(-1, 9)
        used = set()
        for k in range(template.n_closures):
            hit = None
            for r, row in enumerate(rows):
                la, lb = int(row[0]), int(row[1])
                if r in used or int(row[2]) != JOINT_POINT2POINT:
                    continue
                if (owner(la), owner(lb)) == (template.cl_body_a[k], template.cl_body_b[k]):
                    hit = (r, la, lb, row[3:6], row[6:9], False)
                elif (owner(lb), owner(la)) == (template.cl_body_a[k], template.cl_body_b[k]):
                    hit = (r, lb, la, row[6:9], row[3:6], True)
                if hit:
                    break
            if hit is None:
                raise ValueError(f"no point-to-point constraint between the links of closure {k} in the dump")
            r, la, lb, pa, pb, _ = hit
            used.add(r)
            pa_b = (frame[template.cl_body_a[k]].inv() @ C[la]).apply(pa)
            pb_b = (frame[template.cl_body_b[k]].inv() @ C[lb]).apply(pb)
            for i in range(3):
                out.cl_point_a[k][i], out.cl_point_b[k][i] = pa_b[i], pb_b[i]
    # init_pos / init_quat are kept: the reference resets with resetBasePositionAndOrientation (bullet_utils.py:97-102), which
    # places Bullet's base frame -- now this blob's base frame -- at those values
    return out.finalize_tables()


def link_bodies(dump: Mapping[str, np.ndarray], template: M.MoccaModel, joint_names: Sequence[str]) -> np.ndarray:
    """[n_links + 1] blob body each Bullet link moves with (index 0 = Bullet's base, 1 + j = link j): itself if the link carries one of
    the template's hinges, else its nearest such ancestor."""
    names = [str(n) for n in dump["joint_names"]]
    parent = np.asarray(dump["parent_index"]).astype(int)
    body_of_link = {names.index(joint_names[b - 1]): b for b in range(1, template.n_bodies)}
    out = np.zeros(len(names) + 1, int)
    for j in range(len(names)):
        k = j
        while k >= 0 and k not in body_of_link:
            k = parent[k]
        out[1 + j] = body_of_link[k] if k >= 0 else 0
    return out


def body_frames(m: M.MoccaModel, state_row) -> list:
    """World frames of the blob's bodies at a state row (base pos 3, quat 4, ..., q at 13..): forward kinematics in numpy."""
    st = np.asarray(state_row, float)
    fr = [_T(_qmat(st[3:7]), st[0:3])]
    for b in range(1, m.n_bodies):
        ax, q = np.array(list(m.jaxis[b])), st[13 + b - 1]
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        Rq = np.eye(3) + np.sin(q) * K + (1 - np.cos(q)) * (K @ K)
        fr.append(fr[m.parent[b]] @ _T(np.array(list(m.jrot[b])).reshape(3, 3) @ Rq, list(m.jpos[b])))
    return fr


def warm_start_from_contacts(m: M.MoccaModel, bodies_of_links: np.ndarray, state_row, contact_rows, dt: Optional[float] = None) -> np.ndarray:
    """[n_slots] warm-start normal impulses for the step that FOLLOWS `state_row`, from the contact points Bullet reported after the step
    that produced it (tools/dump_pybullet_trace.py `contact_points`: link, other link, position xyz, normal xyz, normal force).  Bullet
    warm-starts its solver with the impulses of the frame before; a teacher-forced state alone does not carry them.  A ground contact
    (or plank: other link < 0) contact on link l is credited to the terrain slot of l's body whose geom end point, moved one radius
    against the reported normal, lies closest to the contact position; the impulse of the last substep is force x dt.  Self contacts (other link >= 0) have no slot in this solver (they start from zero here too)."""
    dt = float(m.dt) if dt is None else dt
    warm = np.zeros(m.n_slots)
    fr = None
    for row in np.asarray(contact_rows, float):
        link, other, force = int(row[0]), int(row[1]), row[8]
        if link < -1 or other >= 0 or force <= 0:
            continue
        body = int(bodies_of_links[link + 1])
        fr = fr or body_frames(m, state_row)
        best, best_d = -1, 1e30
        for g in range(m.n_geoms):
            if m.g_body[g] != body or not m.g_terrain[g]:
                continue
            ends = [list(m.g_p1[g])] + ([list(m.g_p2[g])] if m.g_type[g] == M.GEOM_CAPSULE else [])
            for e, pl in enumerate(ends):
                c = fr[body].apply(pl)
                d = np.linalg.norm(c - m.g_radius[g] * row[5:8] - row[2:5])   # contact position on the robot = sphere centre - r n
                if d < best_d:
                    best, best_d = m.g_slot[g] + e, d
        if best >= 0:
            warm[best] += force * dt
    return warm


def synthetic_dump(m: M.MoccaModel, joint_names: Sequence[str], fixed_children: Dict[int, float] = None,
                   base_axes_aligned: bool = False, link_names: Optional[Sequence[str]] = None, all_axes_aligned: bool = False,
                   fixed_prefix: str = "jointfix_", fixed_link_names: Optional[Dict[int, str]] = None) -> Dict[str, np.ndarray]:
    """The record tools/dump_pybullet_trace.py WOULD write for a Bullet multibody equal to blob `m` (tests: loader round trip).
    Inertial frames are the principal-axes frames at the COM, as Bullet reports them.  `fixed_children`: {body: fraction}
    splits that fraction of the body's mass off into an extra FIXED link (exercises the merge of fixed links), named
    `fixed_link_names[body]` if given (Laikago's toe links are such fixed children of the lower legs).
    `base_axes_aligned`: the base link's inertial frame keeps the link's axes (its inertia's off-diagonal terms are dropped), so that
    resetting Bullet's base to the identity orientation stands the robot up -- what an MJCF import whose root body carries no inertial
    rotation gives."""
    nb = m.n_bodies
    fr = _template_frames(m)

    def principal(b, mass_scale=1.0):
        xx, yy, zz, xy, xz, yz = m.inertia[b]
        if (b == 0 and base_axes_aligned) or all_axes_aligned:   # (all_axes_aligned: every inertial frame keeps its link's axes -- a createConstraint
            return np.array([xx, yy, zz]) * mass_scale, np.eye(3)   # pivot given in "the COM frame" then means the same point as in the URDF's link axes)
        Im = np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]]) * mass_scale
        w, V = np.linalg.eigh(Im)
        if np.linalg.det(V) < 0:
            V[:, 0] = -V[:, 0]
        return w, V

    def mat_quat(R):
        w = np.sqrt(max(0.0, 1 + np.trace(R))) / 2
        if w > 1e-8:
            return np.array([(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w])
        x = np.sqrt(max(0.0, 1 + R[0, 0] - R[1, 1] - R[2, 2])) / 2
        if x > 1e-8:
            return np.array([x, (R[0, 1] + R[1, 0]) / (4 * x), (R[0, 2] + R[2, 0]) / (4 * x), (R[2, 1] - R[1, 2]) / (4 * x)])
        y = np.sqrt(max(0.0, 1 - R[0, 0] + R[1, 1] - R[2, 2])) / 2
        if y > 1e-8:
            return np.array([(R[0, 1] + R[1, 0]) / (4 * y), y, (R[1, 2] + R[2, 1]) / (4 * y), (R[0, 2] - R[2, 0]) / (4 * y)])
        return np.array([0.0, 0.0, 1.0, 0.0])

    fixed_children = fixed_children or {}
    # base: Bullet's base frame is its inertial frame; our blob's base frame may have com != 0 -- shift everything
    w0, V0 = principal(0)
    C0 = _T(V0, list(m.com[0]))                                    # base inertial frame in OUR base frame
    rows = []                                                      # (name, type, parent link, L frame (our base coords), body, mass frac)
    link_index = {0: -1}
    for b in range(1, nb):
        rows.append(dict(name=joint_names[b - 1], type=JOINT_REVOLUTE, parent=link_index[m.parent[b]], L=fr[b], body=b, frac=1.0 - fixed_children.get(b, 0.0),
                         link=link_names[b - 1] if link_names is not None else joint_names[b - 1] + "_link"))
        link_index[b] = len(rows) - 1
        if b in fixed_children:
            rows.append(dict(name=f"{fixed_prefix}{b}", type=JOINT_FIXED, parent=link_index[b], L=fr[b], body=b, frac=fixed_children[b],
                             link=(fixed_link_names or {}).get(b, f"fixed_part_{b}")))
    n = len(rows)
    out = dict(joint_names=np.array([r["name"] for r in rows]), link_names=np.array([r["link"] for r in rows]),
               joint_type=np.array([r["type"] for r in rows]), parent_index=np.array([r["parent"] for r in rows]),
               joint_damping=np.zeros(n), joint_limits=np.zeros((n, 2)), joint_axis=np.zeros((n, 3)),
               parent_frame_pos=np.zeros((n, 3)), parent_frame_orn=np.zeros((n, 4)),
               mass=np.zeros(n + 1), local_inertia_diag=np.zeros((n + 1, 3)), inertial_pos=np.zeros((n + 1, 3)), inertial_orn=np.zeros((n + 1, 4)),
               n_links=np.array(n))
    out["mass"][0], out["local_inertia_diag"][0] = m.mass[0], w0
    out["inertial_pos"][0], out["inertial_orn"][0] = C0.t, mat_quat(C0.R)        # base inertial frame in the root body (link) frame
    Cw = {-1: C0}
    for j, r in enumerate(rows):
        b = r["body"]
        w, V = principal(b, r["frac"])
        Ij = _T(V, list(m.com[b]))                                  # inertial frame in the link frame (a split-off fixed part shares COM and axes)
        Cw[j] = r["L"] @ Ij
        F = Cw[r["parent"]].inv() @ r["L"]                          # (all frames in the template's base coordinates; only relatives are stored)
        out["parent_frame_pos"][j], out["parent_frame_orn"][j] = F.t, mat_quat(F.R)
        out["mass"][1 + j], out["local_inertia_diag"][1 + j] = m.mass[b] * r["frac"], w
        out["inertial_pos"][1 + j], out["inertial_orn"][1 + j] = Ij.t, mat_quat(Ij.R)
        if r["type"] == JOINT_REVOLUTE:
            out["joint_axis"][j] = list(m.jaxis[b])
            out["joint_limits"][j] = [m.jlo[b], m.jhi[b]]
            out["joint_damping"][j] = m.jdamp[b]
    out["_base_inertial_in_template_base"] = np.concatenate([C0.t, C0.R.reshape(-1)])
    # the session's solver parameters, as the tool writes them (engine_*): here the blob's own
    out.update(engine_erp=np.array(float(m.erp_noncontact)), engine_contactERP=np.array(float(m.erp)), engine_numSolverIterations=np.array(float(m.n_iters)),
               engine_contactBreakingThreshold=np.array(float(m.contact_margin)), engine_enableConeFriction=np.array(float(m.friction_cone)),
               engine_contactSlop=np.array(float(m.linear_slop)))
    if m.n_closures > 0:   # createConstraint rows: parent link, child link, type, pivots in the links' inertial frames
        cons = []
        for k in range(m.n_closures):
            ja, jb = link_index[m.cl_body_a[k]], link_index[m.cl_body_b[k]]
            Ia = _T(principal(m.cl_body_a[k])[1], list(m.com[m.cl_body_a[k]])) if ja >= 0 else C0
            Ib = _T(principal(m.cl_body_b[k])[1], list(m.com[m.cl_body_b[k]])) if jb >= 0 else C0
            cons.append([ja, jb, JOINT_POINT2POINT, *Ia.inv().apply(list(m.cl_point_a[k])), *Ib.inv().apply(list(m.cl_point_b[k]))])
        out["constraints"] = np.array(cons, float)
    return out
