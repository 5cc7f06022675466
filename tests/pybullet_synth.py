"""Test infrastructure: the record tools/dump_pybullet_trace.py WOULD write (format_version 2), synthesised without PyBullet.

The multibody part comes from pybullet_dump.synthetic_dump (a compiled blob re-expressed in PyBullet's frame conventions, optionally
with fixed links split off and with mass on the intermediate links of the multi-hinge joints); the traces come from the f64 oracle
running the blob LOADED BACK from that record -- so the oracle reproduces them exactly and the HIP path to fp32 accuracy.  That says
nothing about Bullet; it proves that loader, harness and both steppers are wired for the day a real file arrives."""
import numpy as np

from mocca_envs_amd import model as M
from mocca_envs_amd import pybullet_dump as PD
from oracle.oracle import Oracle

NJ = 21
MAX_CP = 24


def massive_template(mass: float = 0.08, inertia: float = 2e-4) -> M.MoccaModel:
    """Walker3D with mass and inertia on the nine intermediate links of its multi-hinge joints (what Bullet's MJCF importer may report:
    SURVEY Appendix D [UNVERIFIED-BULLET]); the compiled model keeps them massless."""
    m = M.compile_walker3d()
    n = 0
    for b in range(1, m.n_bodies):
        if m.mass[b] == 0.0:
            m.mass[b] = mass
            for k in range(3):
                m.inertia[b][k] = inertia * (1 + 0.3 * k)
                m.com[b][k] = 0.01 * (k - 1)
            n += 1
    assert n == 9
    return m.finalize_tables()


def upright_reset(m: M.MoccaModel, rec, template: M.MoccaModel) -> M.MoccaModel:
    """The synthetic record puts Bullet's base frame at the root link's COM with its PRINCIPAL axes (the most general thing
    getDynamicsInfo can report); the loader keeps init_pos / init_quat, because the reference resets Bullet's base frame to those
    values (bullet_utils.py:97-102) whatever that frame is.  For the synthetic robot to start in the template's physical pose -- upright,
    feet on the ground -- its reset pose must be the template's, expressed for the moved frame."""
    import dense_reference as D
    base = np.asarray(rec["_base_inertial_in_template_base"])
    C0t, C0R = base[:3], base[3:].reshape(3, 3)
    R0 = D._quat_mat(np.array(list(template.init_quat)))
    p, q = np.array(list(template.init_pos)) + R0 @ C0t, D._mat_quat(R0 @ C0R)
    for k in range(3):
        m.init_pos[k] = p[k]
    for k in range(4):
        m.init_quat[k] = q[k]
    return m


def _contact_rows(o: Oracle, m: M.MoccaModel, link_of_body, base_pos):
    """Bullet-style contact rows of the oracle's last substep: link, other, position (world), normal, normal force = impulse / dt."""
    rows = np.zeros((MAX_CP, 9))
    rows[:, 0] = -2
    lam, kind = o.last_lambda()
    normals = lam[kind == 1]
    for k, c in enumerate(o.last_contacts()[:MAX_CP]):
        a, b = int(c[0]), int(c[1])
        rows[k] = [link_of_body[a], -1 if b < 0 else link_of_body[b], *(c[3:6] + base_pos), *c[6:9], normals[k] / m.dt if k < len(normals) else 0.0]
    return rows


def synthetic_record(template: M.MoccaModel = None, fixed_children=None, n_trace: int = 40, n_free: int = 120, seed: int = 0):
    """(record dict in the dump tool's format, the blob loaded from it)."""
    tm = template or M.compile_walker3d()
    g = PD.synthetic_dump(tm, M.WALKER3D_JOINT_NAMES, fixed_children=fixed_children or {2: 0.25})
    m = upright_reset(PD.from_pybullet_dump(g, tm, M.WALKER3D_JOINT_NAMES), g, tm)
    bodies = PD.link_bodies(g, tm, M.WALKER3D_JOINT_NAMES)
    link_of_body = {0: -1}
    for j in range(len(bodies) - 1):
        link_of_body.setdefault(int(bodies[1 + j]), j)   # the first link of a body is the one that carries its hinge
    gains = np.array([m.gain[b] for b in range(1, NJ + 1)])
    o = Oracle(m.to_bytes(), 0, 1, "f64")
    o.reset(seed=seed)
    rng = np.random.default_rng(seed)
    before, after, torques, cps = [], [], [], []
    for t in range(n_trace):
        a = rng.uniform(-1, 1, NJ).astype(np.float32)
        before.append(o.get_state()[0, :13 + 2 * NJ].copy())
        o.step(a[None])
        st = o.get_state()[0]
        after.append(st[:13 + 2 * NJ].copy()); torques.append(gains * a)
        cps.append(_contact_rows(o, m, link_of_body, st[0:3]))
    rec = dict(g, format_version=np.array(2), before=np.array(before), after=np.array(after), torques=np.array(torques),
               contact_points=np.array(cps))
    for tag, scale in (("free", 1.0), ("free03", 0.3)):
        o = Oracle(m.to_bytes(), 0, 1, "f64")
        o.reset(seed=seed)
        st0 = np.zeros((1, o.state_dim))
        st0[0, :3] = list(m.init_pos); st0[0, 3:7] = list(m.init_quat)
        st0[0, 13:13 + NJ] = [m.init_q[b] for b in range(1, NJ + 1)]
        o.set_state(st0)
        rng = np.random.default_rng(0)
        states, actions = [o.get_state()[0, :13 + 2 * NJ].copy()], []
        for t in range(n_free):
            a = (scale * rng.uniform(-1, 1, NJ)).astype(np.float32)
            o.step(a[None])
            states.append(o.get_state()[0, :13 + 2 * NJ].copy()); actions.append(a.astype(np.float64))
        rec[f"{tag}_states"], rec[f"{tag}_actions"] = np.array(states), np.array(actions)
    return rec, m


# ---------------------------------------------------------------------------------------------------------------- Cassie (BASELINE config 4)
def cassie_rows(g, m, key):
    """State rows of a Cassie record (columns 13..: q then qd of `state_joint_names`) in the BLOB's body order, padded to the blob's state width."""
    jn, _ = M.cassie_joint_names()
    cols = [list(map(str, g["state_joint_names"])).index(n) for n in jn]
    src = np.asarray(g[key], float)
    ns = len(g["state_joint_names"])
    out = np.zeros((len(src), M.MoccaModel.state_dim(m) if False else 13 + 2 * m.n_joints + m.n_slots))
    out[:, :13] = src[:, :13]
    out[:, 13:13 + m.n_joints] = src[:, 13:13 + ns][:, cols]
    out[:, 13 + m.n_joints:13 + 2 * m.n_joints] = src[:, 13 + ns:13 + 2 * ns][:, cols]
    return out


def cassie_jvel(g, key):
    """[N][14] filtered / finite-difference joint speeds in the blob's ordered-joint order (env_cassie.py ordered_joints)."""
    cols = [list(map(str, g["ordered_joint_names"])).index(n) for n in M.CASSIE_ORDERED_JOINTS]
    return np.asarray(g[key], float)[:, cols]


def synthetic_record_cassie(n_trace: int = 12, n_free: int = 10, seed: int = 0, action_scale: float = 0.1):
    """(record in the format of tools/dump_pybullet_trace.py's Cassie section, the blob loaded from it): multibody from the compiled Cassie
    blob in PyBullet's conventions (one fixed link split off, the two createConstraint rows), traces = CassieEnv.step of the f64 oracle ON
    THE LOADED BLOB -- teacher-forced env steps (state, jvel, action) -> (state, jvel) and one free-running rollout."""
    tm = M.compile_cassie()
    jn, ln = M.cassie_joint_names()
    g = PD.synthetic_dump(tm, jn, fixed_children={3: 0.2}, base_axes_aligned=True, link_names=ln, fixed_prefix="fixed_extra_")
    m = upright_reset(PD.from_pybullet_dump(g, tm, jn), g, tm)
    g = dict(g, format_version=np.array(3), robot=np.array("cassie"),
             state_joint_names=np.array([n for n, t in zip(g["joint_names"], g["joint_type"]) if int(t) != PD.JOINT_FIXED]),
             ordered_joint_names=np.array(M.CASSIE_ORDERED_JOINTS), cas_action_scale=np.array(action_scale))
    nj = m.n_joints

    def run(n, restart):
        o = Oracle(m.to_bytes(), M.TASK_CASSIE, 1, "f64")
        o.reset(seed=seed)
        rng = np.random.default_rng(seed + (0 if restart else 1))
        rows = {k: [] for k in ("before", "jvel_before", "action", "after", "jvel_after")}
        for t in range(n):
            a = (action_scale * rng.uniform(-1, 1, 10)).astype(np.float32)
            rows["before"].append(o.get_state()[0, :13 + 2 * nj].copy()); rows["jvel_before"].append(o.get_task()[0, 24:38].copy()); rows["action"].append(a.astype(np.float64))
            _, _, d, _ = o.step(a[None])
            rows["after"].append(o.get_state()[0, :13 + 2 * nj].copy()); rows["jvel_after"].append(o.get_task()[0, 24:38].copy())
            if restart and d[0]:
                o.reset(seed=seed)
        return {k: np.array(v) for k, v in rows.items()}

    tr = run(n_trace, True)
    for k, v in tr.items():
        g["cas_" + k] = v
    fr = run(n_free, False)
    g["casfree_states"] = np.concatenate([fr["before"][:1], fr["after"]])
    g["casfree_jvel"] = np.concatenate([fr["jvel_before"][:1], fr["jvel_after"]])
    g["casfree_actions"] = fr["action"]
    return g, m
