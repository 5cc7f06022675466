"""Model compiler: robot description -> flat ``MoccaModel`` blob (include/mocca_model.h).

This is the host-side replacement for Bullet's MJCF/SDF/URDF importers that the
reference reaches through ``loadMJCF`` (/root/reference/mocca_envs/robots.py:102),
``loadSDF`` (bullet_utils.py:365-368) and ``getJointInfo`` (bullet_utils.py:197-199).

The Walker3D description below is this project's own table of the numbers in
``data/robots/walker3d.xml`` (line numbers cited per entry); it is data, not a
copy of the XML.  ``tests/test_model.py`` re-parses the reference XML when it is
present and checks every number against this table.

Assumptions about how Bullet turns the MJCF into a multibody are listed in
DESIGN.md ("Model assumptions"); all of them only affect numbers in the blob,
never the kernels, so a blob dumped from a real PyBullet session
(``tools/dump_pybullet_trace.py``, loaded by ``pybullet_dump.from_pybullet_dump``) can be loaded instead of this compiler's output.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

MAX_BODIES = 24
MAX_GEOMS = 32
MAX_PAIRS = 192
MAX_FEET = 4
MAX_SLOTS = 40
MAGIC = 0x41434F4D
VERSION = 13

GEOM_SPHERE, GEOM_CAPSULE = 0, 1
TASK_WALKER3D_CUSTOM, TASK_WALKER3D_STEPPER, TASK_CASSIE, TASK_WALKER3D_PLANNER = 0, 1, 2, 3
TASK_WORDS = 40
TASKF_NEVER_DONE, TASKF_RESET_TAIL_ZERO, TASKF_BODY_CONTACT, TASKF_QUADRUPED_STEPPER = 1, 2, 4, 8  # MoccaModel.task_flags (include/mocca_model.h)
TASKF_STALE_RESET_CONTACTS = 16   # Stepper reset() reads the contact manifolds of the episode before (env_locomotion.py:484-499): the reference's behaviour
PLANK_BOX, PLANK_CYLINDER = 0, 1
MAX_PLANKS = 4
MAX_CLOSURES = 2
MAX_CTRL = 16
STATE_BASE = 13
TERRAIN_STEPS = 20


class MoccaModel(C.Structure):
    """ctypes mirror of ``struct MoccaModel`` (include/mocca_model.h)."""

    _fields_ = [
        ("magic", C.c_uint32),
        ("version", C.c_uint32),
        ("n_bodies", C.c_int32),
        ("n_joints", C.c_int32),
        ("n_geoms", C.c_int32),
        ("n_pairs", C.c_int32),
        ("n_feet", C.c_int32),
        ("n_slots", C.c_int32),
        ("parent", C.c_int32 * MAX_BODIES),
        ("anc_mask", C.c_uint32 * MAX_BODIES),
        ("depth", C.c_int32 * MAX_BODIES),
        ("jpos", (C.c_float * 3) * MAX_BODIES),
        ("jrot", (C.c_float * 9) * MAX_BODIES),
        ("jaxis", (C.c_float * 3) * MAX_BODIES),
        ("jlo", C.c_float * MAX_BODIES),
        ("jhi", C.c_float * MAX_BODIES),
        ("jdamp", C.c_float * MAX_BODIES),
        ("jarm", C.c_float * MAX_BODIES),
        ("gain", C.c_float * MAX_BODIES),
        ("mass", C.c_float * MAX_BODIES),
        ("com", (C.c_float * 3) * MAX_BODIES),
        ("inertia", (C.c_float * 6) * MAX_BODIES),
        ("g_body", C.c_int32 * MAX_GEOMS),
        ("g_type", C.c_int32 * MAX_GEOMS),
        ("g_slot", C.c_int32 * MAX_GEOMS),
        ("g_terrain", C.c_int32 * MAX_GEOMS),
        ("g_foot", C.c_int32 * MAX_GEOMS),
        ("g_radius", C.c_float * MAX_GEOMS),
        ("g_p1", (C.c_float * 3) * MAX_GEOMS),
        ("g_p2", (C.c_float * 3) * MAX_GEOMS),
        ("g_friction", C.c_float * MAX_GEOMS),
        ("pair_a", C.c_int32 * MAX_PAIRS),
        ("pair_b", C.c_int32 * MAX_PAIRS),
        ("foot_body", C.c_int32 * MAX_FEET),
        ("foot_point", (C.c_float * 3) * MAX_FEET),
        ("gravity", C.c_float),
        ("dt", C.c_float),
        ("n_substeps", C.c_int32),
        ("n_iters", C.c_int32),
        ("erp", C.c_float),
        ("contact_margin", C.c_float),
        ("lin_damp", C.c_float),
        ("ang_damp", C.c_float),
        ("max_qd", C.c_float),
        ("warmstart", C.c_float),
        ("ground_friction", C.c_float),
        ("plank_friction", C.c_float),
        ("plank_stiffness", C.c_float),
        ("plank_damping", C.c_float),
        ("plank_half", C.c_float * 3),
        ("limit_slack", C.c_float),
        ("plank_com_z", C.c_float),
        ("init_q", C.c_float * MAX_BODIES),
        ("init_pos", C.c_float * 3),
        ("control_dt", C.c_float),
        ("termination_height", C.c_float),
        ("electricity_cost", C.c_float),
        ("stall_torque_cost", C.c_float),
        ("joints_at_limit_cost", C.c_float),
        ("max_episode_steps", C.c_int32),
        ("mirror_right", C.c_int32 * 9),
        ("mirror_left", C.c_int32 * 9),
        ("mirror_neg", C.c_int32 * 2),
        ("n_mirror_side", C.c_int32),
        ("n_mirror_neg", C.c_int32),
        ("max_contacts", C.c_int32),
        ("max_rows", C.c_int32),
        ("n_closures", C.c_int32),
        ("cl_body_a", C.c_int32 * MAX_CLOSURES),
        ("cl_body_b", C.c_int32 * MAX_CLOSURES),
        ("cl_point_a", (C.c_float * 3) * MAX_CLOSURES),
        ("cl_point_b", (C.c_float * 3) * MAX_CLOSURES),
        ("n_ctrl", C.c_int32),
        ("n_llc", C.c_int32),
        ("ctrl_body", C.c_int32 * MAX_CTRL),
        ("ctrl_kp", C.c_float * MAX_CTRL),
        ("ctrl_kd", C.c_float * MAX_CTRL),
        ("ctrl_base", C.c_float * MAX_CTRL),
        ("torque_limit", C.c_float * MAX_BODIES),
        ("n_ordered", C.c_int32),
        ("ordered_body", C.c_int32 * MAX_CTRL),
        ("ctrl_oidx", C.c_int32 * MAX_CTRL),
        ("jvel_alpha", C.c_float),
        ("alive_height", C.c_float),
        ("cassie_target", C.c_float * 3),
        ("init_quat", C.c_float * 4),
        ("task_flags", C.c_int32),
        ("n_planks", C.c_int32),
        ("lookbehind", C.c_int32),
        ("plank_shape", C.c_int32),
        ("step_radius", C.c_float),
        ("init_step_separation", C.c_float),
        ("dist_range", C.c_float * 2),
        ("pitch_range_deg", C.c_float),
        ("yaw_range_deg", C.c_float),
        ("tilt_range_deg", C.c_float),
        ("step_bonus_smoothness", C.c_float),
        ("term_height_cur", C.c_float * 2),
        ("gain_cur", C.c_float * 2),
        ("init_vel", C.c_float * 3),
        ("planar", C.c_int32),
        ("cassie_mode", C.c_int32),
        ("cassie_rsi", C.c_int32),
        ("residual_control", C.c_int32),
        ("rod_body", C.c_int32 * 4),
        ("mocap_w", C.c_float * 6),
        ("mocap_speed", C.c_float),
        ("g_torso", C.c_int32 * MAX_GEOMS),
        ("target_range", C.c_float),
        ("fall_z", C.c_float),
        ("manifold_max", C.c_int32),
        ("erp_noncontact", C.c_float),
        ("friction_cone", C.c_int32),
        ("limit_at_violation", C.c_int32),
        ("linear_slop", C.c_float),
        ("sweep_alternate", C.c_int32),
        ("reserved_", C.c_int32 * 2),
        ("slot_tab", (C.c_float * 4) * MAX_SLOTS),
        ("gp_tab", (C.c_float * 4) * (2 * MAX_GEOMS)),
        ("pair_tab", (C.c_float * 4) * MAX_PAIRS),
        ("slot_margin", C.c_float * MAX_SLOTS),
        ("pair_margin", C.c_float * MAX_PAIRS),
        ("g_margin", C.c_float * MAX_GEOMS),
    ]

    def to_bytes(self) -> bytes:
        return bytes(memoryview(self))

    @classmethod
    def from_bytes(cls, b: bytes) -> "MoccaModel":
        if len(b) != C.sizeof(cls):
            raise ValueError(f"model blob is {len(b)} bytes, expected {C.sizeof(cls)}")
        m = cls.from_buffer_copy(b)
        if m.magic != MAGIC or m.version != VERSION:
            raise ValueError("bad model blob magic/version")
        return m

    def margin_code(self, g: int) -> int:
        """The geom's contact margin in units of 2^-13 m (0.122 mm), 8 bits: what the kernels decode from the slot record (the tables
        slot_margin / pair_margin hold the same, decoded -- every implementation compares against the quantised value)."""
        gm = self.g_margin[g] if self.g_margin[g] > 0 else self.contact_margin
        return int(min(255, max(1, round(gm / MARGIN_UNIT))))

    def finalize_tables(self) -> "MoccaModel":
        """Fill the derived lookup tables from the primary fields (call after any edit of geoms / pairs)."""
        def bits(i: int) -> float:
            return float(np.array([i & 0xFFFFFFFF], dtype=np.uint32).view(np.float32)[0])
        # joint record 0 is the identity: the kinematics walk composes it for the path positions past a body's depth
        for k in range(9):
            self.jrot[0][k] = 1.0 if k in (0, 4, 8) else 0.0
        for k in range(3):
            self.jpos[0][k] = 0.0
            self.jaxis[0][k] = 1.0 if k == 2 else 0.0
        for g in range(self.n_geoms):
            ne = 2 if self.g_type[g] == GEOM_CAPSULE else 1
            b = self.g_body[g]
            for e in range(2):
                p = self.g_p2[g] if e else self.g_p1[g]
                for k in range(3):
                    self.gp_tab[2 * g + e][k] = p[k]
                self.gp_tab[2 * g + e][3] = bits(b)
            mq = self.margin_code(g)
            for e in range(ne):
                sl = self.g_slot[g] + e
                self.slot_margin[sl] = mq * MARGIN_UNIT
                self.slot_tab[sl][0] = self.g_radius[g]
                self.slot_tab[sl][1] = self.g_friction[g]
                self.slot_tab[sl][2] = bits(b | (g << 8) | (e << 16) | (mq << 17) | ((1 if self.g_terrain[g] else 0) << 25) |
                                            ((self.g_foot[g] + 1) << 26) | ((1 if self.g_torso[g] else 0) << 29))
                self.slot_tab[sl][3] = bits(self.anc_mask[b])
        for k in range(self.n_pairs):
            ga, gb = self.pair_a[k], self.pair_b[k]
            assert MAX_GEOMS <= 32 and MAX_BODIES <= 32
            pm = min(self.margin_code(ga), self.margin_code(gb))
            self.pair_tab[k][0] = bits(ga | (gb << 5) | (self.g_body[ga] << 10) | (self.g_body[gb] << 15) | (pm << 20))
            self.pair_tab[k][1] = self.g_radius[ga]
            self.pair_tab[k][2] = self.g_radius[gb]
            # broad-phase reach of the pair: half lengths + radii (constants of the two geoms; the contact margin is added at run
            # time), padded by 1e-6 relative + 1e-6 m so that fp32 rounding can only widen the conservative test
            half = lambda g: 0.5 * math.sqrt(sum((float(self.g_p2[g][i]) - float(self.g_p1[g][i])) ** 2 for i in range(3)))
            self.pair_tab[k][3] = (half(ga) + half(gb) + self.g_radius[ga] + self.g_radius[gb]) * (1.0 + 1e-6) + 1e-6
            self.pair_margin[k] = min(self.margin_code(ga), self.margin_code(gb)) * MARGIN_UNIT
        return self

    @property
    def state_dim(self) -> int:
        return STATE_BASE + 2 * self.n_joints + self.n_slots


def bullet_fidelity(m: "MoccaModel") -> "MoccaModel":
    """The blob with the three solver details that only the 64-row ACCURACY instance of the step kernel carries, switched to Bullet's side as
    recalled [UNVERIFIED-BULLET]: no contact / row cap within the wave's reach (64 rows, 20 contacts), the non-contact rows swept in alternating
    direction (`sweep_alternate`), pybullet's default contact slop (1e-5 m).  Slower by design (17 KB of LDS per env, two waves per SIMD); what a
    real PyBullet trace should be compared with first."""
    m.max_rows, m.max_contacts = 64, 20
    m.sweep_alternate = 1
    m.linear_slop = 1e-5
    return m


def relative_margins(factor: float, groups) -> dict:
    """Bullet's relative contact breaking threshold per collision object: `groups` maps a link id to a list of
    (kind, radius, p1, p2, mass, com) geoms given in ONE frame whose axes are the link's; returns {link id: factor x getAngularMotionDisc()},
    the disc = half diagonal of the AABB of the link's shapes + distance of the AABB's centre from the link's COM (the inertial frame
    Bullet hangs the shapes on)."""
    out = {}
    for lid, gs in groups.items():
        lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
        mass, mom = 0.0, np.zeros(3)
        for kind, r, p1, p2, gm, gc in gs:
            for p in (np.asarray(p1, float), np.asarray(p2, float)):
                lo, hi = np.minimum(lo, p - r), np.maximum(hi, p + r)
            mass += gm
            mom += gm * np.asarray(gc, float)
        com = mom / mass if mass > 0 else 0.5 * (lo + hi)
        out[lid] = factor * (0.5 * float(np.linalg.norm(hi - lo)) + float(np.linalg.norm(0.5 * (lo + hi) - com)))
    return out


# --------------------------------------------------------------------------
# Robot description format (this project's own)
# --------------------------------------------------------------------------
DEG = math.pi / 180.0
DENSITY = 1000.0  # MJCF default geom density; walker3d.xml sets none


@dataclass
class Hinge:
    name: str
    axis: Tuple[float, float, float]
    lo_deg: float
    hi_deg: float
    gain: float  # robots.py:234-256 power_coef (base_power = 1.0)


@dataclass
class Geom:
    name: str
    kind: int  # GEOM_SPHERE / GEOM_CAPSULE
    radius: float
    p1: Tuple[float, float, float]
    p2: Optional[Tuple[float, float, float]] = None
    group: int = 3  # MJCF contype   (walker3d.xml:5 default 3)
    mask: int = 3   # MJCF conaffinity
    friction: float = 1.2  # walker3d.xml:5 friction="1.2 0.1 0.1" (lateral)


@dataclass
class Body:
    name: str
    pos: Tuple[float, float, float]  # in parent body frame
    anchor: Tuple[float, float, float] = (0.0, 0.0, 0.0)  # hinge anchor(s) in this body frame
    quat_wxyz: Tuple[float, float, float, float] = (1.0, 0.0, 0.0, 0.0)
    hinges: List[Hinge] = field(default_factory=list)
    geoms: List[Geom] = field(default_factory=list)
    children: List["Body"] = field(default_factory=list)


def _leg(side: str, y: float, sx: float) -> Body:
    """walker3d.xml:32-46 (right) / :48-62 (left).  `sx` flips hip_x / hip_z axes (:49-50)."""
    foot = Body(
        f"{side}_foot", (0, 0, -0.49), anchor=(0, 0, 0.07),
        hinges=[Hinge(f"{side}_ankle", (0, 1, 0), -20, 40, 60)],
        geoms=[
            Geom(f"{side}_foot_1", GEOM_CAPSULE, 0.045, (-0.04, 0.02, 0.07), (0.18, 0.03, 0.07)),
            Geom(f"{side}_foot_2", GEOM_CAPSULE, 0.045, (-0.04, -0.02, 0.07), (0.18, -0.03, 0.07)),
        ],
    )
    shin = Body(
        f"{side}_shin", (0, 0, -0.363), anchor=(0, 0, 0.02),
        hinges=[Hinge(f"{side}_knee", (0, -1, 0), -150, 0, 90)],
        geoms=[Geom(f"{side}_shin1", GEOM_CAPSULE, 0.055, (0, 0, 0), (0, 0, -0.34))],
        children=[foot],
    )
    return Body(
        f"{side}_thigh", (0, y, -0.04), anchor=(0, 0, 0.06),
        hinges=[
            Hinge(f"{side}_hip_x", (sx, 0, 0), -25, 5, 80),
            Hinge(f"{side}_hip_z", (0, 0, sx), -40, 35, 60),
            Hinge(f"{side}_hip_y", (0, 1, 0), -100, 20, 100),
        ],
        geoms=[
            Geom(f"{side}_hip", GEOM_SPHERE, 0.08, (0, 0, 0.06)),
            Geom(f"{side}_thigh1", GEOM_CAPSULE, 0.065, (0, 0, 0), (0, 0, -0.30)),
        ],
        children=[shin],
    )


def _arm(side: str, sy: float, sx: float) -> Body:
    """walker3d.xml:66-78 (right) / :79-91 (left); `sx` flips shoulder_x/z and elbow axes."""
    hand = Body(f"{side}_hand", (0, sy * 0.30, 0),
                geoms=[Geom(f"{side}_hand", GEOM_SPHERE, 0.04, (0, 0, 0))])
    lower = Body(
        f"{side}_lower_arm", (0, sy * 0.28, 0),
        hinges=[Hinge(f"{side}_elbow", (0, 0, sx), 0, 120, 60)],
        geoms=[Geom(f"{side}_larm", GEOM_CAPSULE, 0.035, (0, 0, 0), (0, sy * 0.25, 0))],
        children=[hand],
    )
    return Body(
        f"{side}_upper_arm", (0, sy * 0.23, 0.08),
        hinges=[
            Hinge(f"{side}_shoulder_x", (sx, 0, 0), -60, 100, 60),
            Hinge(f"{side}_shoulder_z", (0, 0, sx), -35, 120, 60),
            Hinge(f"{side}_shoulder_y", (0, 1, 0), -60, 60, 50),
        ],
        geoms=[Geom(f"{side}_uarm1", GEOM_CAPSULE, 0.035, (0, 0, 0), (0, sy * 0.25, 0))],
        children=[lower],
    )


def walker3d_description() -> Body:
    """The Walker3D tree (walker3d.xml:16-92), joint order == robots.py:282-288 indices."""
    pelvis = Body(
        "pelvis", (0, 0, -0.16), anchor=(0, 0, 0.1), quat_wxyz=(1.0, 0.0, -0.002, 0.0),  # :29
        hinges=[Hinge("abdomen_x", (1, 0, 0), -25, 25, 60)],  # :30
        geoms=[Geom("butt", GEOM_SPHERE, 0.11, (0, 0, 0.1), group=1, mask=1)],  # :31
        children=[_leg("right", -0.11, 1.0), _leg("left", 0.11, -1.0)],
    )
    waist = Body(
        "waist", (0, 0, -0.240), anchor=(0, 0, 0.065),  # :25-27
        hinges=[Hinge("abdomen_z", (0, 0, 1), -35, 35, 60), Hinge("abdomen_y", (0, 1, 0), -80, 15, 80)],
        geoms=[Geom("waist", GEOM_SPHERE, 0.09, (0, 0, 0.07), group=2, mask=2)],  # :28
        children=[pelvis],
    )
    head = Body("head", (0, 0, 0.25), geoms=[Geom("head", GEOM_SPHERE, 0.1, (0, 0, 0))])  # :17-19
    torso = Body("torso", (0, 0, 0), geoms=[Geom("torso1", GEOM_SPHERE, 0.14, (0, 0, 0), group=1, mask=1)])  # :20-22
    root = Body(
        "walker3d", (0, 0, 1.32),  # :16
        geoms=[
            Geom("right_shoulder", GEOM_SPHERE, 0.05, (0, -0.15, 0.08)),  # :23
            Geom("left_shoulder", GEOM_SPHERE, 0.05, (0, 0.15, 0.08)),  # :24
        ],
        children=[head, torso, waist, _arm("right", -1.0, 1.0), _arm("left", 1.0, -1.0)],
    )
    return root


# --------------------------------------------------------------------------
# geometry helpers
# --------------------------------------------------------------------------
def quat_wxyz_to_mat(q: Sequence[float]) -> np.ndarray:
    w, x, y, z = np.asarray(q, dtype=np.float64) / np.linalg.norm(q)
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])


def _geom_inertial(g: Geom) -> Tuple[float, np.ndarray, np.ndarray]:
    """mass, com, inertia tensor about com (frame of p1/p2) for a solid sphere / capsule."""
    r = g.radius
    if g.kind == GEOM_SPHERE:
        m = DENSITY * 4.0 / 3.0 * math.pi * r ** 3
        return m, np.asarray(g.p1, float), np.eye(3) * (0.4 * m * r * r)
    p1, p2 = np.asarray(g.p1, float), np.asarray(g.p2, float)
    L = float(np.linalg.norm(p2 - p1))
    u = (p2 - p1) / L
    m_cyl = DENSITY * math.pi * r * r * L
    m_hs = DENSITY * 2.0 / 3.0 * math.pi * r ** 3
    m = m_cyl + 2 * m_hs
    i_ax = 0.5 * m_cyl * r * r + 2 * (0.4 * m_hs * r * r)
    i_tr = m_cyl * (L * L / 12.0 + r * r / 4.0) + 2 * (
        (83.0 / 320.0) * m_hs * r * r + m_hs * (L / 2.0 + 3.0 * r / 8.0) ** 2)
    uu = np.outer(u, u)
    return m, 0.5 * (p1 + p2), i_ax * uu + i_tr * (np.eye(3) - uu)


def _compose_inertial(parts: List[Tuple[float, np.ndarray, np.ndarray]]):
    m = sum(p[0] for p in parts)
    if m == 0.0:
        return 0.0, np.zeros(3), np.zeros((3, 3))
    c = sum(p[0] * p[1] for p in parts) / m
    I = np.zeros((3, 3))
    for mi, ci, Ii in parts:
        d = ci - c
        I += Ii + mi * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
    return m, c, I


@dataclass
class _FlatBody:
    name: str
    parent: int
    jpos: np.ndarray
    jrot: np.ndarray
    hinge: Optional[Hinge]
    geoms: List[Tuple[Geom, np.ndarray, np.ndarray, int]]  # geom, p1, p2 (flat body frame), bullet link id
    bullet_link: int


def _flatten(root: Body, base_ref: str):
    """Expand multi-hinge bodies into chains, merge hinge-less children into their parent.

    Returns flat bodies plus, for the self-collision filter, the Bullet-style
    link tree in which hinge-less children stay separate links attached by fixed
    joints ("jointfix", robots.py:167).
    """
    flat: List[_FlatBody] = []
    bl_parent: List[int] = []  # Bullet-style link parents (index = bullet link id)

    # base reference point: origin of the base frame inside the root body frame
    own = [_geom_inertial(g) for g in root.geoms]
    if base_ref == "own_com":
        base_origin = _compose_inertial(own)[1]
    elif base_ref == "body_frame":
        base_origin = np.zeros(3)
    else:
        raise ValueError(base_ref)

    def add_geoms(fb: _FlatBody, body: Body, R: np.ndarray, t: np.ndarray, bl: int):
        for g in body.geoms:
            p1 = R @ np.asarray(g.p1, float) + t
            p2 = R @ np.asarray(g.p2 if g.p2 is not None else g.p1, float) + t
            fb.geoms.append((g, p1, p2, bl))

    def merge_fixed(fi: int, body: Body, R: np.ndarray, t: np.ndarray, bl_par: int):
        """hinge-less child `body` rigidly attached to flat body `fi`: x_fi = R x + t."""
        bl_parent.append(bl_par)
        bl = len(bl_parent) - 1
        add_geoms(flat[fi], body, R, t, bl)
        for ch in body.children:
            Rc = R @ quat_wxyz_to_mat(ch.quat_wxyz)
            tc = R @ np.asarray(ch.pos, float) + t
            if ch.hinges:
                add_hinged(ch, fi, Rc, tc, bl)
            else:
                merge_fixed(fi, ch, Rc, tc, bl)

    def add_hinged(body: Body, parent_flat: int, R: np.ndarray, t: np.ndarray, bl_par: int):
        """`body` frame expressed in the parent flat-body frame: x_par = R x + t (at q = 0)."""
        anchor = np.asarray(body.anchor, float)
        last = parent_flat
        bl = bl_par
        for k, h in enumerate(body.hinges):
            if k == 0:
                jpos, jrot = R @ anchor + t, R
            else:
                jpos, jrot = np.zeros(3), np.eye(3)
            bl_parent.append(bl)
            bl = len(bl_parent) - 1
            fb = _FlatBody(h.name, last, jpos, jrot, h, [], bl)
            flat.append(fb)
            last = len(flat) - 1
        fb = flat[last]
        # the last link of the chain carries the MJCF body: body coords -> link coords = x - anchor
        add_geoms(fb, body, np.eye(3), -anchor, bl)
        for ch in body.children:
            Rc = quat_wxyz_to_mat(ch.quat_wxyz)
            tc = np.asarray(ch.pos, float) - anchor
            if ch.hinges:
                add_hinged(ch, last, Rc, tc, bl)
            else:
                merge_fixed(last, ch, Rc, tc, bl)

    bl_parent.append(-1)
    base = _FlatBody(root.name, -1, np.zeros(3), np.eye(3), None, [], 0)
    flat.append(base)
    add_geoms(base, root, np.eye(3), -base_origin, 0)
    for ch in root.children:
        Rc = quat_wxyz_to_mat(ch.quat_wxyz)
        tc = np.asarray(ch.pos, float) - base_origin
        if ch.hinges:
            add_hinged(ch, 0, Rc, tc, 0)
        else:
            merge_fixed(0, ch, Rc, tc, 0)
    return flat, bl_parent, base_origin


def _bullet_ancestors(bl_parent: List[int], a: int) -> set:
    out = set()
    while a >= 0:
        out.add(a)
        a = bl_parent[a]
    return out


# MJCF contype / conaffinity become Bullet's collision filter group / mask.  MuJoCo lets two geoms collide when
# (contype_a & conaffinity_b) OR (contype_b & conaffinity_a); with Bullet's default AND rule walker2d.xml
# (contype 1, conaffinity 0 -- "no self collision, but stand on the floor") could not touch the ground plane at all, yet
# PyBullet's 2-D walkers do.  The OR rule is therefore what is modelled, for terrain and for self pairs.  [UNVERIFIED-BULLET]
TERRAIN_GROUP = 2           # btBroadphaseProxy::StaticFilter
TERRAIN_MASK = ~2 & 0xFFFF  # AllFilter ^ StaticFilter


def filters_collide(group_a: int, mask_a: int, group_b: int, mask_b: int) -> bool:
    return bool(group_a & mask_b) or bool(group_b & mask_a)


# step objects of the Stepper envs (bullet_objects.py:86-103, data/objects/steps/*.urdf): (shape, unscaled half extents of base + cover
# -- box: x y z; cylinder: radius radius half-height --, unscaled z of the base link's inertial frame, scale per unit step_radius)
PLANK_CLASSES = {
    "LargePlank": (PLANK_BOX, (0.5, 10.0, 0.25), -0.275, 2.0),      # plank_large.urdf: boxes 1 x 20 x (0.45 + 0.05), scale 2 * width
    "Plank": (PLANK_BOX, (0.5, 0.75, 0.25), -0.275, 2.0),           # plank.urdf: boxes 1 x 1.5 x (0.45 + 0.05), scale 2 * width
    "Pillar": (PLANK_CYLINDER, (1.0, 1.0, 0.5), -0.55, 1.0),        # pillar.urdf: cylinders r 1, length 0.9 + 0.1, scale = radius
}


def set_stepper_params(m: "MoccaModel", *, quadruped: bool = False, plank_class: str = "LargePlank") -> "MoccaModel":
    """Class attributes of Walker3DStepperEnv (env_locomotion.py:338-351,367-385) or LaikagoStepperEnv (:894-926) into the blob."""
    if plank_class not in PLANK_CLASSES:
        raise ValueError(f"unknown plank_class {plank_class!r}; the reference has {sorted(PLANK_CLASSES)}")
    m.n_planks, m.lookbehind = (4, 2) if quadruped else (3, 1)
    m.step_radius = 0.16 if quadruped else 0.25
    m.init_step_separation = 0.45 if quadruped else 0.75
    m.dist_range[0], m.dist_range[1] = (0.45, 0.75) if quadruped else (0.65, 1.25)
    m.pitch_range_deg, m.yaw_range_deg, m.tilt_range_deg = (20.0, 20.0, 10.0) if quadruped else (30.0, 20.0, 15.0)
    m.step_bonus_smoothness = 6.0 if quadruped else 1.0
    m.term_height_cur[0], m.term_height_cur[1] = (0.20, 0.0) if quadruped else (0.75, 0.45)
    m.gain_cur[0], m.gain_cur[1] = (1.0, 1.0) if quadruped else (1.0, 1.2)
    shape, half, com_z, per_radius = PLANK_CLASSES[plank_class]
    scale = per_radius * m.step_radius
    m.plank_shape = shape
    for k in range(3):
        m.plank_half[k] = half[k] * scale
    m.plank_com_z = com_z * scale                  # BaseStep._pos_offset (bullet_objects.py:62)
    m.plank_friction, m.plank_stiffness, m.plank_damping = 1.0, 30000.0, 1000.0      # bullet_objects.py:64-72
    # reset() -> calc_feet_state() on the manifolds of the episode before (env_locomotion.py:484-499); only the Stepper task reads the flag
    m.task_flags |= TASKF_STALE_RESET_CONTACTS
    return m


def compile_model(
    root: Body,
    foot_names: Sequence[str],
    init_q_by_name: Dict[str, float],
    init_pos: Sequence[float],
    mirror_right: Sequence[int],
    mirror_left: Sequence[int],
    mirror_neg: Sequence[int],
    *,
    base_ref: str = "body_frame",
    joint_damping: float = 0.0,
    joint_armature: float = 0.0,
    self_collision: bool = True,
    init_quat_xyzw: Sequence[float] = (0.0, 0.0, 0.0, 1.0),
    link_mass: Optional[Dict[str, float]] = None,
    plank_class: str = "LargePlank",
    torso_name: Optional[str] = None,
) -> MoccaModel:
    flat, bl_parent, _ = _flatten(root, base_ref)
    nb = len(flat)
    assert nb <= MAX_BODIES
    m = MoccaModel()
    m.magic, m.version = MAGIC, VERSION
    m.n_bodies, m.n_joints = nb, nb - 1

    geoms = []
    for b, fb in enumerate(flat):
        m.parent[b] = fb.parent
        if b == 0:
            m.anc_mask[0], m.depth[0] = 0, 0
        else:
            m.anc_mask[b] = m.anc_mask[fb.parent] | (1 << b)
            m.depth[b] = m.depth[fb.parent] + 1
            for k in range(3):
                m.jpos[b][k] = fb.jpos[k]
                m.jaxis[b][k] = fb.hinge.axis[k]
            for k in range(9):
                m.jrot[b][k] = fb.jrot.reshape(-1)[k]
            m.jlo[b] = fb.hinge.lo_deg * DEG
            m.jhi[b] = fb.hinge.hi_deg * DEG
            m.jdamp[b] = joint_damping
            m.jarm[b] = joint_armature
            m.gain[b] = fb.hinge.gain
            m.init_q[b] = init_q_by_name.get(fb.hinge.name, 0.0)
        parts = []
        for g, p1, p2, bl in fb.geoms:
            gg = Geom(g.name, g.kind, g.radius, tuple(p1), tuple(p2), g.group, g.mask, g.friction)
            parts.append(_geom_inertial(gg))
            geoms.append((b, gg, bl))
        mass, com, I = _compose_inertial(parts)
        if link_mass and fb.name in link_mass:
            # changeDynamics(mass=...) on a link (robots.py:507-510): Bullet re-derives the link's local inertia from its
            # collision shape at the new mass, i.e. scales it with the mass.                    [UNVERIFIED-BULLET]
            I, mass = I * (link_mass[fb.name] / mass), link_mass[fb.name]
        m.mass[b] = mass
        for k in range(3):
            m.com[b][k] = com[k]
        for k, (i, j) in enumerate([(0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2)]):
            m.inertia[b][k] = I[i, j]

    assert len(geoms) <= MAX_GEOMS
    m.n_geoms = len(geoms)
    slot = 0
    for gi, (b, g, bl) in enumerate(geoms):
        m.g_body[gi], m.g_type[gi], m.g_radius[gi] = b, g.kind, g.radius
        m.g_friction[gi] = g.friction
        for k in range(3):
            m.g_p1[gi][k] = g.p1[k]
            m.g_p2[gi][k] = g.p2[k]
        m.g_slot[gi] = slot
        slot += 1 if g.kind == GEOM_SPHERE else 2
        m.g_terrain[gi] = int(filters_collide(g.group, g.mask, TERRAIN_GROUP, TERRAIN_MASK))
    assert slot <= MAX_SLOTS
    m.n_slots = slot
    # Bullet's relative contact breaking threshold, per Bullet link (a hinge-less child body is a link of its own there)
    groups = {}
    for gi, (b, g, bl) in enumerate(geoms):
        gm, gc, _ = _geom_inertial(g)
        groups.setdefault((b, bl), []).append((g.kind, g.radius, g.p1, g.p2, gm, gc))
    rel = relative_margins(CONTACT_BREAKING_THRESHOLD, groups)
    for gi, (b, g, bl) in enumerate(geoms):
        m.g_margin[gi] = rel[(b, bl)]

    # self-collision candidate pairs: URDF_USE_SELF_COLLISION |
    # URDF_USE_SELF_COLLISION_EXCLUDE_ALL_PARENTS (robots.py:259-264) + group/mask filter
    pairs = []
    if self_collision:
        for i in range(len(geoms)):
            for j in range(i + 1, len(geoms)):
                bi, gi_, li = geoms[i]
                bj, gj_, lj = geoms[j]
                if li == lj:
                    continue
                if li in _bullet_ancestors(bl_parent, lj) or lj in _bullet_ancestors(bl_parent, li):
                    continue
                if not filters_collide(gi_.group, gi_.mask, gj_.group, gj_.mask):
                    continue
                if bi == bj:
                    continue  # rigidly attached in our tree (e.g. head vs torso): cannot move relative
                pairs.append((i, j))
    assert len(pairs) <= MAX_PAIRS, len(pairs)
    m.n_pairs = len(pairs)
    for k, (i, j) in enumerate(pairs):
        m.pair_a[k], m.pair_b[k] = i, j

    names = [fb.name for fb in flat]
    last_link_of_body = {}
    # foot_names are MJCF body names; the link carrying the body is its last hinge
    def find_body(b: Body, nm: str) -> Optional[Body]:
        if b.name == nm:
            return b
        for ch in b.children:
            r = find_body(ch, nm)
            if r is not None:
                return r
        return None

    m.n_feet = len(foot_names)
    for k, fn in enumerate(foot_names):
        fbdy = find_body(root, fn)
        m.foot_body[k] = names.index(fbdy.hinges[-1].name)
    for gi in range(m.n_geoms):   # MJCF feet: every geom of the foot body belongs to the foot link
        m.g_foot[gi] = next((k for k in range(m.n_feet) if m.foot_body[k] == m.g_body[gi]), -1)
    if torso_name is not None:    # robot_torso_name (env_locomotion.py:992): the link that carries the MJCF body of that name
        tb = names.index(find_body(root, torso_name).hinges[-1].name)
        for gi, (b, g, bl) in enumerate(geoms):
            m.g_torso[gi] = int(b == tb and bl == flat[tb].bullet_link)
    m.target_range, m.fall_z = 16.0, -5.0     # env_locomotion.py:1060, :1108
    for k in range(m.n_feet):     # ... and the foot link is the body itself: its COM is what getLinkState reports
        for i in range(3):
            m.foot_point[k][i] = m.com[m.foot_body[k]][i]

    # physics parameters
    m.gravity = 9.8                 # env_base.py:80
    m.dt = 1.0 / 240.0              # env_base.py:81 with env_locomotion.py:39-41
    m.n_substeps = 4                # env_locomotion.py:41
    m.n_iters = 5                   # bullet_utils.py:340
    m.erp = 0.9                     # bullet_utils.py:345: setDefaultContactERP = infoGlobal.m_erp2, the contact rows' ERP
    m.erp_noncontact = ERP_NONCONTACT
    m.friction_cone = FRICTION_CONE
    m.limit_at_violation = LIMIT_AT_VIOLATION
    m.contact_margin = CONTACT_BREAKING_THRESHOLD   # gContactBreakingThreshold: the FACTOR of the relative thresholds in g_margin (above)
    m.lin_damp = 0.04               # [UNVERIFIED-BULLET] btMultiBody default; applies to the base and to every link, with the quadratic term
    m.ang_damp = 0.04
    m.max_qd = 100.0                # [UNVERIFIED-BULLET] maxCoordinateVelocity
    m.warmstart = WARMSTART
    m.ground_friction = 0.8         # bullet_utils.py:371
    set_stepper_params(m, plank_class=plank_class)   # plank geometry, terrain ranges, curricula (LargePlank: 0.5 x 10 x 0.25 m slab)
    m.limit_slack = 0.05
    m.max_contacts = 12
    m.max_rows = 48

    for k in range(3):
        m.init_pos[k] = init_pos[k]
    for k in range(4):
        m.init_quat[k] = init_quat_xyzw[k]
    m.control_dt = 1.0 / 60.0       # env_locomotion.py:39
    m.termination_height = 0.7      # env_locomotion.py:44
    m.electricity_cost = 4.5        # env_locomotion.py:54
    m.stall_torque_cost = 0.225     # env_locomotion.py:55
    m.joints_at_limit_cost = 0.1    # env_locomotion.py:56
    m.max_episode_steps = 1000      # __init__.py:55
    m.n_mirror_side, m.n_mirror_neg = len(mirror_right), len(mirror_neg)
    for k, v in enumerate(mirror_right):
        m.mirror_right[k] = v
    for k, v in enumerate(mirror_left):
        m.mirror_left[k] = v
    for k, v in enumerate(mirror_neg):
        m.mirror_neg[k] = v
    return m.finalize_tables()


WALKER3D_JOINT_NAMES = [
    "abdomen_z", "abdomen_y", "abdomen_x",
    "right_hip_x", "right_hip_z", "right_hip_y", "right_knee", "right_ankle",
    "left_hip_x", "left_hip_z", "left_hip_y", "left_knee", "left_ankle",
    "right_shoulder_x", "right_shoulder_z", "right_shoulder_y", "right_elbow",
    "left_shoulder_x", "left_shoulder_z", "left_shoulder_y", "left_elbow",
]


# Bullet link that each of those joints carries: the last hinge of an MJCF body carries the body (and bears its name -- the reference
# finds its parts by it: `parts["right_foot"]`, `parts["waist"]`, robots.py:232, env_locomotion.py:992); the hinges before it carry the
# massless intermediate links of the multi-hinge joints  [UNVERIFIED-BULLET naming of the dummy links]
WALKER3D_LINK_NAMES = [
    "link_dummy_abdomen_z", "waist", "pelvis",
    "link_dummy_right_hip_x", "link_dummy_right_hip_z", "right_thigh", "right_shin", "right_foot",
    "link_dummy_left_hip_x", "link_dummy_left_hip_z", "left_thigh", "left_shin", "left_foot",
    "link_dummy_right_shoulder_x", "link_dummy_right_shoulder_z", "right_upper_arm", "right_lower_arm",
    "link_dummy_left_shoulder_x", "link_dummy_left_shoulder_z", "left_upper_arm", "left_lower_arm",
]


def walker3d_running_start() -> Dict[str, float]:
    """robots.py:296-302."""
    q = np.zeros(21)
    q[[5, 6]] = -np.pi / 8
    q[10] = np.pi / 10
    q[[13, 17]] = np.pi / 3
    q[14] = -np.pi / 6
    q[18] = np.pi / 6
    q[[16, 20]] = np.pi / 3
    return {n: float(v) for n, v in zip(WALKER3D_JOINT_NAMES, q)}


def _planner_params(m: "MoccaModel") -> "MoccaModel":
    """Walker3DPlannerEnv's class attributes and its terrain's contact parameters (env_locomotion.py:982-996, bullet_objects.py:386-393)."""
    m.termination_height = 0.5                                                       # :995
    m.plank_friction, m.plank_stiffness, m.plank_damping = 1.0, 30000.0, 1000.0      # HeightField.reload changeDynamics
    return m.finalize_tables()


def compile_walker3d(task: int = TASK_WALKER3D_CUSTOM, **kw) -> MoccaModel:
    """Walker3D blob.  `task` changes the initial base position (env_locomotion.py:339, :989) and, for the planner env, the
    termination height and the torso link."""
    init_pos = {TASK_WALKER3D_CUSTOM: (0.0, 0.0, 1.32), TASK_WALKER3D_STEPPER: (0.3, 0.0, 1.32), TASK_WALKER3D_PLANNER: (-15.5, -15.5, 1.32)}[task]
    if task == TASK_WALKER3D_PLANNER:
        kw.setdefault("torso_name", "waist")                # robot_torso_name, :992
    # walker3d.xml:4 <joint armature="0.01" damping=".1">.  The armature is what keeps the
    # three-hinge shoulder (massless intermediate links, z range up to 120 deg, walker3d.xml:67-69)
    # away from the gimbal singularity where ABA's D_i -> 0; see DESIGN.md "Model assumptions".
    kw.setdefault("joint_damping", 0.1)
    kw.setdefault("joint_armature", 0.01)
    m = compile_model(
        walker3d_description(),
        foot_names=["right_foot", "left_foot"],             # robots.py:232
        init_q_by_name=walker3d_running_start(),
        init_pos=init_pos,                                  # robots.py:276
        mirror_right=[3, 4, 5, 6, 7, 13, 14, 15, 16],       # robots.py:282-284
        mirror_left=[8, 9, 10, 11, 12, 17, 18, 19, 20],     # robots.py:285-287
        mirror_neg=[0, 2],                                  # robots.py:288
        **kw,
    )
    return _planner_params(m) if task == TASK_WALKER3D_PLANNER else m


def walker3d_crawl() -> Dict[str, float]:
    """robots.py:309-315 ("crawl")."""
    q = np.zeros(21)
    q[[13, 17]] = np.pi / 2
    q[[14, 18]] = np.pi / 2
    q[[16, 20]] = np.pi / 3
    q[[5, 10]] = -np.pi / 2
    q[[6, 11]] = -120 * DEG
    q[[7, 12]] = -20 * DEG
    return {n: float(v) for n, v in zip(WALKER3D_JOINT_NAMES, q)}


_WALKER3D_MIRROR = dict(mirror_right=[3, 4, 5, 6, 7, 13, 14, 15, 16], mirror_left=[8, 9, 10, 11, 12, 17, 18, 19, 20],
                        mirror_neg=[0, 2])  # robots.py:282-288 (inherited by Child3D and Mike)


def compile_child3d(**kw) -> MoccaModel:
    """Child3D for Child3DCustomEnv (robots.py:326-335, env_locomotion.py:317-324): the Walker3D tree at child scale,
    gains x0.4, starting on all fours 0.38 m up with the base pitched 90 degrees, fallen below 0.1 m."""
    from .mjcf_tables import child3d_description
    kw.setdefault("joint_damping", 0.1)     # child3d.xml:4
    kw.setdefault("joint_armature", 0.01)
    h = math.sqrt(0.5)
    m = compile_model(child3d_description(), foot_names=["right_foot", "left_foot"], init_q_by_name=walker3d_crawl(),
                      init_pos=(0.0, 0.0, 0.38),                 # robots.py:335
                      init_quat_xyzw=(0.0, h, 0.0, h),           # getQuaternionFromEuler([0, 90 deg, 0]), robots.py:316-318
                      **_WALKER3D_MIRROR, **kw)
    m.termination_height = 0.1                                   # env_locomotion.py:320
    return m.finalize_tables()


def compile_mike(**kw) -> MoccaModel:
    """Mike for MikeStepperEnv (robots.py:474-510, env_locomotion.py:843-845): Walker3D tree, passive abdomen, halved
    arm gains, waist link forced to 8 kg, starting at (0.3, 0, 1.0)."""
    from .mjcf_tables import mike_description
    kw.setdefault("joint_damping", 0.1)     # mike.xml:4
    kw.setdefault("joint_armature", 0.01)
    planner = kw.pop("planner", False)      # MikePlannerEnv (env_locomotion.py:1131-1133): starts at (-15.5, -15.5, 1.05)
    if planner:
        kw.setdefault("torso_name", "waist")
    m = compile_model(mike_description(), foot_names=["right_foot", "left_foot"],
                      init_q_by_name=walker3d_running_start(),  # env_locomotion.py:360
                      init_pos=(-15.5, -15.5, 1.05) if planner else (0.3, 0.0, 1.0),   # :1133 / :845
                      link_mass={"abdomen_y": 8.0},           # the link carrying MJCF body "waist", robots.py:507-510
                      **_WALKER3D_MIRROR, **kw)
    return _planner_params(m) if planner else m


def _compile_planar(description, damping: float, armature: float, self_collision: bool, mirror_right, mirror_left, **kw) -> MoccaModel:
    """Walker2D / Crab2D for the planar Custom envs (env_locomotion.py:285-314).

    The MJCF root carries slide-x / slide-z / hinge-y "ignore*" joints (robots.py:163-165): in Bullet the pelvis is a
    link behind them on a world-fixed base.  Here the pelvis is the floating base: every hinge axis is +-y and every
    geom lies in the y = 0 plane, so the motion stays planar by symmetry; what differs from the 3-D walkers is that no
    base damping acts (Bullet damps the multibody BASE, which is the fixed anchor there).  The reported base point is
    the pelvis link's COM (robot_body = parts["pelvis"], robots.py:105-106; getLinkState()[0]).
    Start height: the rest pose of the file (feet just above the ground).  What robot_init_position = [0, 0, 1.35]
    (env_locomotion.py:287) does to a fixed-base multibody whose MJCF body sits at z = -1.35 cannot be observed without
    PyBullet.                                                                                     [UNVERIFIED-BULLET]
    """
    root = description()
    kw.setdefault("joint_damping", damping)
    kw.setdefault("joint_armature", armature)
    m = compile_model(root, foot_names=["foot", "foot_left"], init_q_by_name={}, init_pos=root.pos,
                      mirror_right=mirror_right, mirror_left=mirror_left, mirror_neg=[], self_collision=self_collision, **kw)
    m.lin_damp = m.ang_damp = 0.0
    m.task_flags |= TASKF_NEVER_DONE | TASKF_RESET_TAIL_ZERO
    return m.finalize_tables()


def compile_walker2d(**kw) -> MoccaModel:
    """robots.py:338-369; loaded without the self-collision flags (:358)."""
    from .mjcf_tables import walker2d_description
    return _compile_planar(walker2d_description, 0.1, 0.01, False, [1, 2, 3], [4, 5, 6], **kw)


def compile_crab2d(**kw) -> MoccaModel:
    """robots.py:372-404; crab2d.xml:4 armature 1, damping 1; self-collision flags on (:390-394)."""
    from .mjcf_tables import crab2d_description
    return _compile_planar(crab2d_description, 1.0, 1.0, True, [0, 1, 2], [3, 4, 5], **kw)


# --------------------------------------------------------------------------
# Cassie (env_cassie.py:13-282) from this project's table of the URDF numbers
# --------------------------------------------------------------------------
def _rpy_mat(rpy) -> np.ndarray:
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


CASSIE_ORDERED_JOINTS = [s % side for side in ("left", "right") for s in (
    "hip_abduction_%s", "hip_rotation_%s", "hip_flexion_%s", "knee_joint_%s", "knee_to_shin_%s", "ankle_joint_%s",
    "toe_joint_%s")]
# env_cassie.py:20-39
CASSIE_BASE_ANGLES = [0.035615837, -0.01348790, 0.391940848, -0.95086160, -0.08376049, 1.305643634, -1.61174064] * 2
CASSIE_ROD_ANGLES = {"fixed_left_achilles_rod_joint_z": -0.8967891835, "fixed_left_achilles_rod_joint_y": 0.063947468,
                     "fixed_right_achilles_rod_joint_z": -0.8967891835, "fixed_right_achilles_rod_joint_y": -0.063947468}
CASSIE_POWER = {"hip_abduction": 112.5, "hip_rotation": 112.5, "hip_flexion": 195.2, "knee_joint": 195.2,
                "knee_to_shin": 200.0, "ankle_joint": 200.0, "toe_joint": 45.0}  # env_cassie.py:41-56
CASSIE_DAMPING = [1, 1, 1, 1, 0.1, 0, 1] * 2                                    # env_cassie.py:57
CASSIE_POWERED = [0, 1, 2, 3, 6, 7, 8, 9, 10, 13]                               # env_cassie.py:59
CASSIE_SPRINGS = [4, 11]                                                        # env_cassie.py:60
CASSIE_KP = np.array([100, 100, 88, 96, 50, 100, 100, 88, 96, 50, 400, 400]) / 1.9  # env_cassie.py:292-317


# Bullet defaults the reference never touches (bullet_utils.py:338-350 sets the time step, 5 iterations, 4 substeps and the CONTACT erp only),
# restated from the published source as recalled  [UNVERIFIED-BULLET]:
#  * rows that are not contacts (joint limits, point-to-point closures) take infoGlobal.m_erp = 0.2, not the contact ERP (m_erp2) that
#    setDefaultContactERP(0.9) sets: btMultiBodyConstraint::fillMultiBodyConstraint ("split impulse is not implemented yet for
#    btMultiBody*": erp = infoGlobal.m_erp), btMultiBodyJointLimitConstraint::createConstraintRows;
#  * multibody contact rows do NOT warm start: btMultiBodyConstraintSolver::setupMultiBodyContactConstraint, "disable warmstarting for
#    btMultiBody, it has issues gaining energy (==explosion)", `if (0)`; later versions gate it behind SOLVER_USE_ARTICULATED_WARMSTARTING,
#    which pybullet's default solver mode does not contain.  (Rounds 1-3 early used 0.85, btContactSolverInfo's rigid-body factor.)
#  * the two friction rows of a contact are coupled ("implicit cone friction", Bullet >= 2.87): both candidate impulses from the same
#    velocity state, the pair clipped to the circle of radius mu * lambda_n (btMultiBodyConstraintSolver::resolveConeFrictionConstraintRows);
#    pybullet: setPhysicsEngineParameter(enableConeFriction=0) selects the pyramid, "cone is default".  (Pyramid until round 3 late.)
#  * a joint-limit row exists only while the joint is at or past its limit: btMultiBodyJointLimitConstraint::createConstraintRows,
#    "//todo: consider adding some safety threshold here / if (penetration > 0) continue;" -- the joint crosses the limit by up to
#    speed x dt and is pushed back with the non-contact ERP.  (Until round 3 late: a row from a predicted gap of 0.05 rad on, which stops the
#    joint AT the limit -- the older form of that file, whose positive-gap branch `velocityError = -penetration / dt` is still in the source.)
#  * contacts open within the RELATIVE breaking threshold 0.02 x getAngularMotionDisc() of the smaller of the two collision objects
#    (btCollisionDispatcher::getNewManifold, CD_USE_RELATIVE_CONTACT_BREAKING_THRESHOLD on by default; MoccaModel.g_margin): 3 - 6 mm for a
#    walker's links.  (Until round 3 late: 20 mm for every pair.)
CONTACT_BREAKING_THRESHOLD = 0.02
MARGIN_UNIT = 2.0 ** -13      # slot_tab carries a geom's margin as an 8-bit multiple of this (0.122 mm; up to 31 mm)
ERP_NONCONTACT = 0.2
WARMSTART = 0.0
FRICTION_CONE = 1
LIMIT_AT_VIOLATION = 1

CASSIE_PLAIN, CASSIE_PHASE_MOCCA, CASSIE_PHASE_MIRROR = 0, 1, 2   # MoccaModel.cassie_mode (include/mocca_model.h)
# Cassie2D (env_cassie.py:279-282) loads cassie_collide_2d.urdf.  The class's path (data/cassie/urdf/) does not exist in the reference's
# tree; the file it means lies beside the 3-D one, data/robots/cassie/urdf/cassie_collide_2d.urdf.  It differs from cassie_collide.urdf in
# two ways only: (1) a massless chain  world-fixed root -prismatic z-> -prismatic x-> -continuous y-> pelvis  (:449-468) in place of the
# floating base -- the pelvis keeps x, z and pitch, which is what the three planar rows of this blob enforce (MoccaModel.planar); (2) the
# limits of both hips' abduction and rotation joints shrink to +-0.01 rad (:479,486,535,542), which holds the legs in the sagittal plane.
CASSIE_2D_LIMITS = {"hip_abduction_left": (-0.01, 0.01), "hip_rotation_left": (-0.01, 0.01),
                    "hip_abduction_right": (-0.01, 0.01), "hip_rotation_right": (-0.01, 0.01)}


def cassie_joint_names():
    """(joint names, link names) of the Cassie blob's bodies 1 .. 18, in blob order: the URDF's moving joints, depth-first as compile_cassie
    walks them (a joint named "fixed_*_achilles_rod_joint_*" is a CONTINUOUS joint: the reference only keeps it out of its ordered joints,
    env_cassie.py:189).  What pybullet_dump.from_pybullet_dump matches a PyBullet record against."""
    from . import cassie_table as CT
    kids: Dict[str, list] = {}
    for j in CT.JOINTS:
        kids.setdefault(j["parent"], []).append(j)
    jn, ln = [], []

    def walk(link):
        for j in kids.get(link, []):
            if j["type"] != "fixed":
                jn.append(j["name"]); ln.append(j["child"])
            walk(j["child"])

    walk("pelvis")
    return jn, ln


def compile_cassie(planar: bool = False, power_coef: float = 1.0, residual_control: bool = True, mode: int = CASSIE_PLAIN,
                   rsi: bool = True) -> MoccaModel:
    """Cassie blob: URDF tree with inertia from file (env_cassie.py:81-99), two point-to-point loop closures
    (:114-137), per-joint damping (:57,197-201), torque limits (:41-56), the PD gains of CassieEnv (:292-319).
    Ground contact: 12 support points of each toe's convex hull (radius-0 spheres); other meshes and mesh-mesh
    self collision are not modelled (the episode ends when the pelvis is 0.6 m above the lower toe, :406-412)."""
    from . import cassie_table as CT
    kids: Dict[str, list] = {}
    for j in CT.JOINTS:
        kids.setdefault(j["parent"], []).append(j)
    bodies = []  # dict(name, parent, jpos, jrot, axis, lo, hi, joint, parts[], points[])

    def inertial(link, R, t):
        e = CT.LINKS[link]
        if e["mass"] == 0.0:
            return None
        Ri = R @ _rpy_mat(e["rpy"])
        xx, yy, zz, xy, xz, yz = e["inertia"]
        I = np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])
        return e["mass"], R @ np.asarray(e["com"], float) + t, Ri @ I @ Ri.T

    def visit(link, body, R, t):
        part = inertial(link, R, t)
        if part is not None:
            bodies[body]["parts"].append(part)
        if link.endswith("_toe"):
            side = link.split("_")[0]
            for p in CT.TOE_POINTS[side]:
                bodies[body]["points"].append((R @ np.asarray(p, float) + t, CT.LINKS[link]["friction"] or 1.0))
        bodies[body]["links"][link] = (R.copy(), t.copy())
        for j in kids.get(link, []):
            Rj, tj = R @ _rpy_mat(j["rpy"]), R @ np.asarray(j["xyz"], float) + t
            if j["type"] == "fixed":
                visit(j["child"], body, Rj, tj)
            else:
                lo = j["lower"] if j["lower"] is not None else -1e30
                hi = j["upper"] if j["upper"] is not None else 1e30
                if planar and j["name"] in CASSIE_2D_LIMITS:
                    lo, hi = CASSIE_2D_LIMITS[j["name"]]
                bodies.append(dict(name=j["name"], parent=body, jpos=tj, jrot=Rj, axis=np.asarray(j["axis"], float),
                                   lo=lo, hi=hi, parts=[], points=[], links={}))
                visit(j["child"], len(bodies) - 1, np.eye(3), np.zeros(3))

    bodies.append(dict(name="pelvis", parent=-1, jpos=np.zeros(3), jrot=np.eye(3), axis=None, lo=0, hi=0, parts=[],
                       points=[], links={}))
    visit("pelvis", 0, np.eye(3), np.zeros(3))
    nb = len(bodies)
    m = MoccaModel()
    m.magic, m.version = MAGIC, VERSION
    m.n_bodies, m.n_joints = nb, nb - 1
    names = [b["name"] for b in bodies]
    g = 0
    for b, bd in enumerate(bodies):
        m.parent[b] = bd["parent"]
        if b:
            m.anc_mask[b] = m.anc_mask[bd["parent"]] | (1 << b)
            m.depth[b] = m.depth[bd["parent"]] + 1
            for k in range(3):
                m.jpos[b][k] = bd["jpos"][k]
                m.jaxis[b][k] = bd["axis"][k] / np.linalg.norm(bd["axis"])
            for k in range(9):
                m.jrot[b][k] = bd["jrot"].reshape(-1)[k]
            m.jlo[b], m.jhi[b] = bd["lo"], bd["hi"]
            stem = bd["name"].rsplit("_", 1)[0]
            m.torque_limit[b] = power_coef * CASSIE_POWER.get(stem, 0.0)      # base_power * power_coef[name], env_cassie.py:192-195
            m.init_q[b] = CASSIE_ROD_ANGLES.get(bd["name"], 0.0)
        mass, com, I = _compose_inertial(bd["parts"])
        m.mass[b] = mass
        for k in range(3):
            m.com[b][k] = com[k]
        for k, (i, j) in enumerate([(0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2)]):
            m.inertia[b][k] = I[i, j]
        for p, fr in bd["points"]:
            m.g_body[g], m.g_type[g], m.g_radius[g], m.g_slot[g], m.g_terrain[g] = b, GEOM_SPHERE, 0.0, g, 1
            m.g_friction[g] = fr
            for k in range(3):
                m.g_p1[g][k] = m.g_p2[g][k] = p[k]
            g += 1
    m.n_geoms = m.n_slots = g
    m.n_pairs = 0
    m.n_feet = 2
    # relative contact breaking threshold of each toe (one convex mesh = one collision object): from the AABB of its hull points
    groups = {}
    for gi in range(m.n_geoms):
        bb = m.g_body[gi]
        groups.setdefault(bb, []).append((GEOM_SPHERE, 0.0, list(m.g_p1[gi]), list(m.g_p1[gi]), 1.0, list(m.com[bb])))
    rel = relative_margins(CONTACT_BREAKING_THRESHOLD, groups)
    for gi in range(m.n_geoms):
        m.g_margin[gi] = rel[m.g_body[gi]]
    m.foot_body[0], m.foot_body[1] = names.index("toe_joint_right"), names.index("toe_joint_left")  # env_cassie.py:72
    for gi in range(m.n_geoms):
        m.g_foot[gi] = next((k for k in range(2) if m.foot_body[k] == m.g_body[gi]), -1)
    for k in range(2):
        for i in range(3):
            m.foot_point[k][i] = m.com[m.foot_body[k]][i]
    # ordered joints, controller
    m.n_ordered = len(CASSIE_ORDERED_JOINTS)
    for k, n in enumerate(CASSIE_ORDERED_JOINTS):
        b = names.index(n)
        m.ordered_body[k] = b
        m.init_q[b] = CASSIE_BASE_ANGLES[k]
        m.jdamp[b] = CASSIE_DAMPING[k]
    ctrl = CASSIE_POWERED + CASSIE_SPRINGS
    m.n_ctrl = len(ctrl)
    for k, oi in enumerate(ctrl):
        m.ctrl_body[k] = m.ordered_body[oi]
        m.ctrl_oidx[k] = oi
        m.ctrl_kp[k] = CASSIE_KP[k]
        m.ctrl_kd[k] = CASSIE_KP[k] / 10.0
        # residual_control (:434-443): the action is added to the nominal angles of the powered joints, or to zero
        m.ctrl_base[k] = CASSIE_BASE_ANGLES[oi] if (k < len(CASSIE_POWERED) and residual_control) else 0.0
    # loop closures tarsus <-> achilles rod (env_cassie.py:114-137)
    m.n_closures = 2
    for k, (side, z) in enumerate((("left", 0.00711836), ("right", -0.00711836))):
        m.cl_body_a[k] = names.index("ankle_joint_%s" % side)               # link *_tarsus
        m.cl_body_b[k] = names.index("fixed_%s_achilles_rod_joint_y" % side)  # link *_achilles_rod
        # createConstraint frames are relative to each link's CENTRE-OF-MASS frame: with the COM offsets the two
        # pivots coincide to 2.9 mm in the nominal pose (and the rod pivot lands at 0.5012 m, the rod's length)
        ca, cb = CT.LINKS["%s_tarsus" % side]["com"], CT.LINKS["%s_achilles_rod" % side]["com"]
        for i, v in enumerate((-0.22735404, 0.05761813, z)):
            m.cl_point_a[k][i] = ca[i] + v
        for i, v in enumerate((0.254001, 0.0, 0.0)):
            m.cl_point_b[k][i] = cb[i] + v
    # physics: env_cassie.py:287-289 control_step 0.03 / llc 50 / sim_frame_skip 1
    m.gravity, m.dt, m.n_substeps, m.n_iters, m.erp = 9.8, 0.03 / 50, 1, 5, 0.9
    m.erp_noncontact = ERP_NONCONTACT   # the two point-to-point closures (btMultiBodyPoint2Point -> fillMultiBodyConstraint) and the limits
    m.friction_cone = FRICTION_CONE
    m.limit_at_violation = LIMIT_AT_VIOLATION
    m.n_llc = 50
    m.contact_margin, m.lin_damp, m.ang_damp, m.max_qd, m.warmstart = CONTACT_BREAKING_THRESHOLD, 0.04, 0.04, 100.0, WARMSTART
    m.ground_friction = 0.8
    m.limit_slack, m.max_contacts, m.max_rows = 0.05, 12, 48
    # each toe is ONE convex mesh in cassie_collide.urdf: Bullet keeps at most 4 contact points per pair of collision objects
    # (btPersistentManifold)  [UNVERIFIED-BULLET] -- of the twelve hull points of a toe on the ground four make contacts, not twelve
    # (which also overran the solver's 12-contact cap in 84 % of the substeps of a standing robot, profiles/archive/r03_cap_pressure.jsonl)
    m.manifold_max = 4
    m.init_pos[0], m.init_pos[1], m.init_pos[2] = 0.0, 0.0, 1.085            # env_cassie.py:17
    m.init_quat[3] = 1.0
    m.control_dt = 0.03
    m.max_episode_steps = 1000                                               # __init__.py:18-22
    m.jvel_alpha = min(10 / 50, 1)                                           # env_cassie.py:319
    m.alive_height = 0.6                                                     # env_cassie.py:406-412
    m.cassie_target[0], m.cassie_target[1], m.cassie_target[2] = 1000.0, 0.0, 0.0  # env_cassie.py:366
    m.planar = int(planar)                                                   # env_cassie.py:326-341 (Cassie2DEnv-v0, __init__.py:24-29)
    # mocap / phase variants (env_cassie.py:481-660)
    m.cassie_mode, m.cassie_rsi, m.residual_control = int(mode), int(bool(rsi)), int(bool(residual_control))
    for k, n in enumerate(("fixed_right_achilles_rod_joint_z", "fixed_right_achilles_rod_joint_y",
                           "fixed_left_achilles_rod_joint_z", "fixed_left_achilles_rod_joint_y")):   # resetJoints, :591-596
        m.rod_body[k] = names.index(n)
    if mode != CASSIE_PLAIN:
        # CassieMocapRewEnv.__init__ (:483-493): the joint terms share what the four fixed weights leave
        w = {"SpeedRew": 0.1, "CoMRew": 0.02 if planar else 0.05, "OrientationRew": 0.0 if planar else 0.05, "AngularSpeedRew": 0.1}
        wleft = 1 - sum(w.values())
        w["JPosRew"], w["JVelRew"] = wleft / 5 * 4, wleft / 5
        for k, n in enumerate(("SpeedRew", "JPosRew", "JVelRew", "OrientationRew", "AngularSpeedRew", "CoMRew")):
            m.mocap_w[k] = w[n]
        m.mocap_speed = 0.8                                                  # :498
        m.init_vel[0], m.init_vel[1], m.init_vel[2] = 0.8, 0.0, 0.0          # CassieMoccaEnv.initial_velocity, :552
    return m.finalize_tables()


LAIKAGO_JOINTS = ["%s_%s" % (leg, j) for leg in ("FR", "FL", "RR", "RL")
                  for j in ("hip_motor_2_chassis_joint", "upper_leg_2_hip_motor_joint", "lower_leg_2_upper_leg_joint")]  # robots.py:561-574
LAIKAGO_FEET = ["toeFR", "toeFL", "toeRR", "toeRL"]                                                       # robots.py:559
LAIKAGO_SHAPE_MARGIN = 0.001  # collision margin PyBullet gives URDF mesh shapes; enters the box inertia  [UNVERIFIED-BULLET]


def compile_laikago(stepper: bool = False, plank_class: str = "LargePlank") -> MoccaModel:
    """Laikago for LaikagoCustomEnv (robots.py:554-656, env_locomotion.py:854-890) from mocca_envs_amd/laikago_table.py.

    * The URDF is y-up; its chassis INERTIAL frame (rpy -1.57 -1.57 0) is what stands the robot up: PyBullet's base pose is
      the pose of that frame, and the env resets it to the identity.  The model's base frame therefore is the chassis
      inertial frame (origin at its COM), everything else is expressed in it.
    * The file's inertia tensors are zero and the robot is loaded without URDF_USE_INERTIA_FROM_FILE (robots.py:595-600):
      every link gets the box inertia of its collision shape's bounding box in its inertial frame (what Bullet's
      calculateLocalInertia does for convex hulls / compounds), spheres 0.4 m r^2.                [UNVERIFIED-BULLET]
    * Ground contact: the four toe spheres (the feet) and 28 support points of the link meshes' convex hulls (radius-0
      spheres, no foot index): touching the ground with any of those ends the episode.  Mesh-mesh self collision
      (URDF_USE_SELF_COLLISION) is not modelled.
    """
    from . import laikago_table as LT
    kids: Dict[str, list] = {}
    for j in LT.JOINTS:
        kids.setdefault(j["parent"], []).append(j)
    bodies = []  # dict(name, parent, jpos, jrot, axis, lo, hi, parts[], geoms[])

    def visit(link, body, R, t):
        e = LT.LINKS[link]
        Ri = R @ _rpy_mat(e["rpy"])
        if e["mass"] > 0.0:
            if e["sphere"] is not None:
                I = np.eye(3) * 0.4 * e["mass"] * e["sphere"]["radius"] ** 2
            else:
                hx, hy, hz = (h + LAIKAGO_SHAPE_MARGIN for h in e["box_half"])
                I = Ri @ np.diag([hy * hy + hz * hz, hx * hx + hz * hz, hx * hx + hy * hy]) @ Ri.T * (e["mass"] / 3.0)
            bodies[body]["parts"].append((e["mass"], R @ np.asarray(e["com"], float) + t, I))
        fr = e["friction"] if e["friction"] is not None else 1.0
        if e["sphere"] is not None:
            bodies[body]["geoms"].append((R @ np.asarray(e["sphere"]["center"], float) + t, e["sphere"]["radius"], fr, link))
        for p in e["points"]:
            bodies[body]["geoms"].append((R @ np.asarray(p, float) + t, 0.0, fr, None))
        for j in kids.get(link, []):
            Rj, tj = R @ _rpy_mat(j["rpy"]), R @ np.asarray(j["xyz"], float) + t
            if j["type"] == "fixed":
                visit(j["child"], body, Rj, tj)
            else:
                bodies.append(dict(name=j["name"], parent=body, jpos=tj, jrot=Rj, axis=np.asarray(j["axis"], float),
                                   lo=j["lower"], hi=j["upper"], parts=[], geoms=[]))
                visit(j["child"], len(bodies) - 1, np.eye(3), np.zeros(3))

    ch = LT.LINKS["chassis"]
    R0 = _rpy_mat(ch["rpy"]).T
    bodies.append(dict(name="chassis", parent=-1, jpos=np.zeros(3), jrot=np.eye(3), axis=None, lo=0, hi=0, parts=[], geoms=[]))
    visit("chassis", 0, R0, -R0 @ np.asarray(ch["com"], float))
    names = [b["name"] for b in bodies]
    assert names[1:] == LAIKAGO_JOINTS, names      # URDF order == ordered_joints order (robots.py:609-626)
    nb = len(bodies)
    m = MoccaModel()
    m.magic, m.version = MAGIC, VERSION
    m.n_bodies, m.n_joints = nb, nb - 1
    g = 0
    m.n_feet = 4
    for b, bd in enumerate(bodies):
        m.parent[b] = bd["parent"]
        if b:
            m.anc_mask[b] = m.anc_mask[bd["parent"]] | (1 << b)
            m.depth[b] = m.depth[bd["parent"]] + 1
            for k in range(3):
                m.jpos[b][k] = bd["jpos"][k]
                m.jaxis[b][k] = bd["axis"][k] / np.linalg.norm(bd["axis"])
            for k in range(9):
                m.jrot[b][k] = bd["jrot"].reshape(-1)[k]
            m.jlo[b], m.jhi[b] = bd["lo"], bd["hi"]
            m.gain[b] = 40.0                                                          # robots.py:561-574, base_power 1
            m.init_q[b] = -math.pi / 6 if bd["name"].endswith("lower_leg_2_upper_leg_joint") else 0.0   # "running_start", :654-655
        mass, com, I = _compose_inertial(bd["parts"])
        m.mass[b] = mass
        for k in range(3):
            m.com[b][k] = com[k]
        for k, (i, j) in enumerate([(0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2)]):
            m.inertia[b][k] = I[i, j]
        for p, rad, fr, toe in bd["geoms"]:
            m.g_body[g], m.g_type[g], m.g_radius[g], m.g_slot[g], m.g_terrain[g] = b, GEOM_SPHERE, rad, g, 1
            m.g_friction[g] = fr
            m.g_foot[g] = LAIKAGO_FEET.index(toe) if toe in LAIKAGO_FEET else -1
            if toe in LAIKAGO_FEET:      # the foot link is the toe, a fixed child: its COM is the sphere centre
                m.foot_body[LAIKAGO_FEET.index(toe)] = b
                for k in range(3):
                    m.foot_point[LAIKAGO_FEET.index(toe)][k] = p[k]
            for k in range(3):
                m.g_p1[g][k] = m.g_p2[g][k] = p[k]
            g += 1
    assert g <= MAX_GEOMS
    m.n_geoms = m.n_slots = g
    m.n_pairs = 0
    # relative contact breaking thresholds: a toe sphere is a collision object of its own (fixed child link); the hull points of the
    # mesh links are grouped by the body they move with
    groups = {}
    for gi in range(m.n_geoms):
        bb, rr, pp = m.g_body[gi], m.g_radius[gi], list(m.g_p1[gi])
        key = ("toe", gi) if m.g_foot[gi] >= 0 else ("body", bb)
        groups.setdefault(key, []).append((GEOM_SPHERE, rr, pp, pp, 1.0, pp if m.g_foot[gi] >= 0 else list(m.com[bb])))
    rel = relative_margins(CONTACT_BREAKING_THRESHOLD, groups)
    for gi in range(m.n_geoms):
        m.g_margin[gi] = rel[("toe", gi) if m.g_foot[gi] >= 0 else ("body", m.g_body[gi])]
    # physics: control_step 1/60, sim_frame_skip 8 (env_locomotion.py:856-858) -> 8 substeps of 1/480 s
    m.gravity, m.dt, m.n_substeps, m.n_iters, m.erp = 9.8, 1.0 / 480.0, 8, 5, 0.9
    m.erp_noncontact = ERP_NONCONTACT
    m.friction_cone = FRICTION_CONE
    m.limit_at_violation = LIMIT_AT_VIOLATION
    m.contact_margin, m.lin_damp, m.ang_damp, m.max_qd, m.warmstart = CONTACT_BREAKING_THRESHOLD, 0.04, 0.04, 100.0, WARMSTART
    m.ground_friction = 0.8
    m.limit_slack, m.max_contacts, m.max_rows = 0.05, 12, 48
    m.init_pos[0], m.init_pos[1], m.init_pos[2] = 0.0, 0.0, 0.56                      # env_locomotion.py:864
    m.init_quat[3] = 1.0
    m.control_dt = 1.0 / 60.0
    m.termination_height = 0.0                                                        # :862
    m.electricity_cost, m.stall_torque_cost, m.joints_at_limit_cost = 4.5, 0.225, 0.1
    m.max_episode_steps = 1000
    right, left = [0, 1, 2, 6, 7, 8], [3, 4, 5, 9, 10, 11]                            # robots.py:578-580
    m.n_mirror_side, m.n_mirror_neg = len(right), 0
    for k, v in enumerate(right):
        m.mirror_right[k] = v
    for k, v in enumerate(left):
        m.mirror_left[k] = v
    m.task_flags |= TASKF_BODY_CONTACT
    if stepper:
        # LaikagoStepperEnv (env_locomotion.py:893-979): sim_frame_skip 4 -> 4 substeps of 1/240 s, start at (0.25, 0, 0.53) moving
        # at (0.5, 0, 0.25), four live planks of step_radius 0.16, its own reward / termination (MOCCA_TASKF_QUADRUPED_STEPPER)
        m.dt, m.n_substeps = 1.0 / 240.0, 4
        m.init_pos[0], m.init_pos[1], m.init_pos[2] = 0.25, 0.0, 0.53
        m.init_vel[0], m.init_vel[1], m.init_vel[2] = 0.5, 0.0, 0.25
        set_stepper_params(m, quadruped=True, plank_class=plank_class)
        m.task_flags = (m.task_flags & ~TASKF_BODY_CONTACT) | TASKF_QUADRUPED_STEPPER
    return m.finalize_tables()


def joint_limits(m: MoccaModel) -> Tuple[np.ndarray, np.ndarray]:
    nj = m.n_joints
    lo = np.array([m.jlo[b] for b in range(1, nj + 1)], dtype=np.float32)
    hi = np.array([m.jhi[b] for b in range(1, nj + 1)], dtype=np.float32)
    return lo, hi


# --------------------------------------------------------------------------
# compile-time topology for the HIP kernels
# --------------------------------------------------------------------------
def topology_header(m: MoccaModel, name: str = "Walker3D") -> str:
    """C++ header with the tree as constexpr tables (mocca_envs_amd/csrc/topo_*.h).

    The step kernel unrolls its sweeps over bodies, so parent/ancestor indices must be
    compile-time constants (runtime-indexed register arrays would go to scratch).  Every
    numeric model parameter stays in the runtime blob.
    """
    nb = m.n_bodies
    parent = [m.parent[b] for b in range(nb)]
    depth = [m.depth[b] for b in range(nb)]
    maxd = max(depth)
    children = [[c for c in range(nb) if parent[c] == b] for b in range(nb)]
    maxc = max(len(c) for c in children)
    # Level tables for the ABA inward pass (8 lanes per body, one slot of 8 lanes per body of a level).  A body keeps
    # the slot of one of its children ("carried" child) so that a serial chain -- a leg, an arm -- stays on the same
    # lanes from leaf to root and its articulated inertia travels in registers; other children go through LDS.
    by_depth = [[b for b in range(nb) if depth[b] == d] for d in range(maxd + 1)]
    maxw = max(4, max(len(l) for l in by_depth[1:]))  # the level pass is written for four 8-lane slots
    slot_of, carry = {}, [-1] * nb
    levels = [[-1] * maxw for _ in range(maxd + 1)]
    for d in range(maxd, 0, -1):
        taken = set()
        for b in by_depth[d]:                       # continuations first: inherit the slot of the lowest-index child
            for c in sorted(children[b]):
                if slot_of[c] not in taken:
                    slot_of[b], carry[b] = slot_of[c], c
                    taken.add(slot_of[c])
                    break
        for b in by_depth[d]:                       # leaves and bodies whose children's slots were all taken
            if b not in slot_of:
                slot_of[b] = min(set(range(maxw)) - taken)
                taken.add(slot_of[b])
        for b in by_depth[d]:
            levels[d][slot_of[b]] = b
    path = []
    for b in range(nb):
        p, cur = [], b
        while cur > 0:
            p.append(cur)
            cur = parent[cur]
        p = p[::-1]
        path.append(p + [-1] * (maxd - len(p)))

    def arr(vals):
        return "{" + ", ".join(str(v) for v in vals) + "}"

    lines = [
        "// GENERATED by mocca_envs_amd.model.topology_header() -- do not edit.",
        "// Kinematic tree of %s as compile-time tables (numbers live in the MoccaModel blob)." % name,
        "#pragma once",
        "// runtime-indexed copies (per-lane lookups)",
        "__device__ static const signed char kPath%s[%d][%d] = {" % (name, nb, maxd),
    ]
    lines += ["  %s," % arr(p) for p in path]
    lines += ["};", "__device__ static const signed char kChild%s[%d][%d] = {" % (name, nb, maxc)]
    # children in DESCENDING order: the oracle accumulates b = nb-1 .. 1 into parent[b]
    lines += ["  %s," % arr(sorted(c, reverse=True) + [-1] * (maxc - len(c))) for c in children]
    lines += ["};", "__device__ static const signed char kLevel%s[%d][%d] = {" % (name, maxd + 1, maxw)]
    lines += ["  %s," % arr(l) for l in levels]
    lines += [
        "};",
        "__device__ static const unsigned long long kPathPk%s[%d] = %s;" % (
            name, nb, arr(["0x%xull" % sum(((path[b][k] if path[b][k] >= 0 else 0) << (5 * k)) for k in range(maxd)) for b in range(nb)])),
        "struct Topo%s {" % name,
        "  static constexpr int NB = %d;        // bodies incl. floating base" % nb,
        "  static constexpr int NJ = %d;        // hinges" % (nb - 1),
        "  static constexpr int ND = %d;        // generalised velocities" % (nb + 5),
        "  static constexpr int MAXD = %d;      // tree depth" % maxd,
        "  static constexpr int MAXCH = %d;     // children per body" % maxc,
        "  static constexpr int MAXW = %d;      // bodies per level" % maxw,
        "  static constexpr int NG = %d;        // geoms" % m.n_geoms,
        "  static constexpr int NSLOT = %d;     // terrain contact slots" % m.n_slots,
        "  static constexpr int NPAIR = %d;     // self-collision candidate pairs" % m.n_pairs,
        "  static constexpr int NCLOS = %d;     // point-to-point loop closures" % m.n_closures,
        "  static constexpr int NFEET = %d;     // feet (entries of the observation, robots.py:74-86)" % m.n_feet,
        "  // constexpr functions (implicitly __host__ __device__ under hipcc) fold after unrolling",
        "  static constexpr int parent(int b) { constexpr int t[%d] = %s; return t[b]; }" % (nb, arr(parent)),
        "  static constexpr int depth(int b) { constexpr int t[%d] = %s; return t[b]; }" % (nb, arr(depth)),
        "  static constexpr unsigned anc_mask(int b) { constexpr unsigned t[%d] = %s; return t[b]; }"
        % (nb, arr(["0x%xu" % m.anc_mask[b] for b in range(nb)])),
        "  static __device__ __forceinline__ int path(int b, int k) { return kPath%s[b][k]; }" % name,
        "  static __device__ __forceinline__ int child(int b, int k) { return kChild%s[b][k]; }" % name,
        "  static __device__ __forceinline__ int level(int d, int s) { return kLevel%s[d][s]; }" % name,
        "  // compile-time versions: with the loops over k / d unrolled these cost a few VALU ops and NO memory access",
        "  // (a dependent global table load per tree level / path step is what dominated the latency-bound phases)",
        "  static constexpr int clevel(int d, int s) { constexpr int t[%d][%d] = {%s}; return t[d][s]; }"
        % (maxd + 1, maxw, ", ".join(arr(l) for l in levels)),
        "  // links without mass or inertia (the intermediate links of multi-hinge joints): a tree level that holds only such links skips the",
        "  // reads of its (all-zero) link inertia and bias force in the ABA inward pass; mocca_create() checks the blob against this table",
        "  static constexpr bool massless(int b) { constexpr bool t[%d] = %s; return b >= 0 && t[b]; }"
        % (nb, arr(["true" if (m.mass[b] == 0.0 and all(m.inertia[b][i] == 0.0 for i in range(6)) and b > 0) else "false" for b in range(nb)])),
        "  // the child whose articulated inertia stays in registers (same slot, next level), -1 if none",
        "  static constexpr int ccarry(int b) { constexpr int t[%d] = %s; return b < 0 ? -1 : t[b]; }" % (nb, arr(carry)),
        "  static constexpr int cchild(int b, int k) { constexpr int t[%d][%d] = {%s}; return b < 0 ? -1 : t[b][k]; }"
        % (nb, maxc, ", ".join(arr(sorted(c, reverse=True) + [-1] * (maxc - len(c))) for c in children)),
        "  // the lane's own root->body path, 5 bits per step (0 = past the end: joints are numbered from 1): loaded once per kernel, two VGPRs, then",
        "  // every path step is a v_bfe -- no table access inside the walks",
        "  static __device__ __forceinline__ unsigned long long path_packed(int b) { return kPathPk%s[b]; }" % name,
        "};",
        "",
    ]
    return "\n".join(lines)


def write_topology_headers(outdir: Optional[str] = None) -> None:
    import os
    outdir = outdir or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    with open(os.path.join(outdir, "topo_walker3d.h"), "w") as f:
        f.write(topology_header(compile_walker3d(), "Walker3D"))
    with open(os.path.join(outdir, "topo_cassie.h"), "w") as f:
        f.write(topology_header(compile_cassie(), "Cassie"))
    with open(os.path.join(outdir, "topo_walker2d.h"), "w") as f:
        f.write(topology_header(compile_walker2d(), "Walker2D"))
    with open(os.path.join(outdir, "topo_crab2d.h"), "w") as f:
        f.write(topology_header(compile_crab2d(), "Crab2D"))
    with open(os.path.join(outdir, "topo_laikago.h"), "w") as f:
        f.write(topology_header(compile_laikago(), "Laikago"))


if __name__ == "__main__":
    write_topology_headers()
