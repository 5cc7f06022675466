"""The Cassie mocap / phase envs (env_cassie.py:481-660) on the CPU oracle, against vectors the reference's own classes produced.

tests/golden/make_golden_cassie_mocap.py ran `CassiePhaseMoccaEnv` / `CassiePhaseMirrorEnv` (their real reset / step / pd_control /
compute_rewards / get_obs) over this project's f64 physics, with the re-created CassieTrajectory standing in for the module the
reference tree lacks.  The oracle's own task layer must reproduce every observation, reward and done flag."""
import os

import numpy as np
import pytest

from mocca_envs_amd import model as M
from mocca_envs_amd.trajectory import CassieTrajectory
from oracle.oracle import Oracle

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "cassie_mocap_reference.npz"))
MODES = {"mocca": M.CASSIE_PHASE_MOCCA, "mirror": M.CASSIE_PHASE_MIRROR}


def _oracle(tag, precision="f64", n=1):
    m = M.compile_cassie(planar=True, mode=MODES[tag])
    o = Oracle(m.to_bytes(), M.TASK_CASSIE, n, precision)
    tr = CassieTrajectory()
    o.set_trajectory(tr.table(), tr.max_time(), 0.03)
    return m, o


def test_trajectory_facts():
    tr = CassieTrajectory()
    assert len(tr) == 1682 and abs(tr.max_time() - 0.8405) < 1e-6
    assert tr.table().shape == (1682, 32)
    # the cycle is periodic and left / right are half a period apart (to the accuracy of a recorded gait)
    assert np.abs(tr.joint_angles(0.0) - tr.joint_angles(tr.max_time() - 1e-9)).max() < 0.02
    assert np.abs(tr.joint_angles(0.0)[:7] - tr.joint_angles(tr.max_time() / 2)[7:]).max() < 0.06
    # speeds are the time derivative of the angles
    fd = np.gradient(tr.angles, tr.time, axis=0)
    assert (np.sqrt(((fd - tr.speeds) ** 2).mean(0)) / np.sqrt((tr.speeds ** 2).mean(0))).max() < 0.1
    # lookup: floor of the fractional frame, periodic
    assert tr.index(0.0) == 0 and tr.index(tr.max_time() + 0.00051) == 1 and tr.index(0.00049) == 0


def test_rod_angles_close_the_loops():
    """rod_joint_angles(t) come from a least-squares loop closure on this project's model: resetting to any frame of the motion
    leaves the two four-bar loops closed to about a centimetre (the nominal pose: 3 mm)."""
    m, o = _oracle("mocca")
    worst = 0.0
    for istep in (0, 350, 700, 1050, 3971, 9999):
        o.set_tape(np.array([(istep + 0.5) / 10000.0]))
        o.reset(seed=0)
        fr = o.link_frames(0, m.n_bodies)
        for c in range(m.n_closures):
            a, b = m.cl_body_a[c], m.cl_body_b[c]
            pa = fr[a, 9:12] + fr[a, :9].reshape(3, 3) @ np.array(list(m.cl_point_a[c]))
            pb = fr[b, 9:12] + fr[b, :9].reshape(3, 3) @ np.array(list(m.cl_point_b[c]))
            worst = max(worst, float(np.linalg.norm(pa - pb)))
        assert o.get_task()[0, 39] == istep
    assert worst < 0.012


def _mocap_fk(m, tr, f, base_R=None, base_p=None):
    """Body frames (R, p) of frame f of the reference's walking cycle on the compiled Cassie blob: the 14 recorded joint angles, the 4
    fitted rod angles, the pelvis at (base_R, base_p) (identity / origin by default)."""
    def rot(axis, q):
        a = axis / np.linalg.norm(axis)
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        return np.eye(3) + np.sin(q) * K + (1 - np.cos(q)) * (K @ K)
    nb = m.n_bodies
    q = np.zeros(nb)
    for k in range(14):
        q[m.ordered_body[k]] = tr.angles[f, k]
    for k in range(4):
        q[tr.rod_bodies[k]] = tr.rods[f, k]
    R, p = [None] * nb, [None] * nb
    R[0], p[0] = (np.eye(3) if base_R is None else base_R), (np.zeros(3) if base_p is None else base_p)
    for b in range(1, nb):
        pa = m.parent[b]
        R[b] = R[pa] @ np.array(m.jrot[b][:]).reshape(3, 3) @ rot(np.array(m.jaxis[b][:]), q[b])
        p[b] = p[pa] + R[pa] @ np.array(m.jpos[b][:])
    return R, p


def test_every_mocap_frame_closes_the_four_bars_to_a_centimetre():
    """What the reference's Cassie data says about THIS project's compiled geometry, part 1 (env_cassie.py:114-137: the two point-to-point
    closures tarsus <-> achilles rod).  For each of the 1 682 frames of data/robots/cassie/mocap (joint angles fitted to the URDF robot),
    with the rod's two free angles at their least-squares optimum, the distance between the closure's two pivots on this blob is the
    part of the four-bar the URDF tree cannot represent: the heel spring's deflection (a fixed joint in the URDF).  Bounded here over the
    whole cycle -- tools/gen_cassie_mocap.py only printed it: worst 10.9 mm, mean 4.7 mm; a wrong link offset or joint axis in the tarsus /
    rod chain (lengths of 0.12 .. 0.50 m) would show as centimetres."""
    m = M.compile_cassie()
    tr = CassieTrajectory()
    res = np.zeros((len(tr), 2))
    for f in range(len(tr)):
        R, p = _mocap_fk(m, tr, f)
        for c in range(2):
            a, b = m.cl_body_a[c], m.cl_body_b[c]
            res[f, c] = np.linalg.norm(p[a] + R[a] @ np.array(m.cl_point_a[c][:]) - p[b] - R[b] @ np.array(m.cl_point_b[c][:]))
    print(f"\nclosure residual over {len(tr)} frames [mm]: worst {1e3 * res.max(0)}, mean {1e3 * res.mean(0)}")
    assert res.max() < 0.012 and res.mean() < 0.006
    assert abs(res[:, 0].mean() - res[:, 1].mean()) < 0.001          # left and right loops are mirror images walking the same gait


def test_stance_foot_of_every_mocap_frame_stands_on_the_ground():
    """Part 2: the recorded pelvis pose (stepdata.bin qpos[0:7], fixture tests/golden/cassie_mocap_base.npz made by
    make_golden_cassie_base.py) + the recorded joint angles through THIS blob's kinematics (tree, joint frames, axes) put the lowest point
    of the stance foot's toe hull (cassie_table.TOE_POINTS, the blob's contact geoms) on the ground plane z = 0 in every frame of the
    walking cycle: within [-1.5 mm, +3 mm] (measured: -0.7 .. +2.1 mm, mean 0.9 mm) over a 1.0 m leg chain of seven joints.  The swing
    foot clears 10 cm, each foot stands for about half the cycle.  These are the only reference-held numbers that reach Cassie's compiled
    geometry; an error in any link offset, joint axis sign or toe point would break the bound by centimetres."""
    m = M.compile_cassie()
    tr = CassieTrajectory()
    B = np.load(os.path.join(os.path.dirname(__file__), "golden", "cassie_mocap_base.npz"))
    assert len(B["time"]) == len(tr) and np.abs(B["time"] - tr.time).max() == 0.0
    low = np.full((len(tr), 2), 1e9)
    for f in range(len(tr)):
        w, x, y, z = B["base_quat_wxyz"][f]
        Rb = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                       [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                       [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        R, p = _mocap_fk(m, tr, f, Rb, B["base_pos"][f])
        for g in range(m.n_geoms):
            b, k = m.g_body[g], m.g_foot[g]
            assert k in (0, 1) and m.g_radius[g] == 0.0
            low[f, k] = min(low[f, k], (p[b] + R[b] @ np.array(m.g_p1[g][:]))[2])
    stance = low.min(1)
    print(f"\nstance foot's lowest toe point over {len(tr)} frames [mm]: {1e3 * stance.min():.2f} .. {1e3 * stance.max():.2f}, mean {1e3 * stance.mean():.2f}; "
          f"swing apex [mm]: {1e3 * low.max(0)}")
    assert -0.0015 < stance.min() and stance.max() < 0.003
    assert low.max(0).min() > 0.10                                  # both feet swing: 10.6 / 10.7 cm at the apex
    on = low < 0.003
    assert 0.4 < on[:, 0].mean() < 0.7 and 0.4 < on[:, 1].mean() < 0.7 and on.any(1).all()


@pytest.mark.parametrize("tag", ["mocca", "mirror"])
def test_blob_constants_match_the_reference_classes(tag):
    m, _ = _oracle(tag)
    np.testing.assert_allclose(list(m.mocap_w), G[f"{tag}_weights"], rtol=1e-6)
    np.testing.assert_allclose(list(m.init_vel), G[f"{tag}_initial_velocity"], rtol=1e-7)
    assert m.cassie_rsi == 1 and m.residual_control == 1 and m.planar == 1


@pytest.mark.parametrize("tag", ["mocca", "mirror"])
@pytest.mark.parametrize("ep", [0, 1, 2])
def test_episode_matches_the_reference_code(tag, ep):
    m, o = _oracle(tag)
    istep0 = int(G[f"{tag}_ep{ep}_istep0"])
    o.set_tape(np.array([(istep0 + 0.5) / 10000.0]))      # np_random.randint(0, 10000) of the recorded reset
    obs = o.reset(seed=0)
    o.set_tape(None)
    gobs, grew, gdone = G[f"{tag}_ep{ep}_obs"], G[f"{tag}_ep{ep}_rew"], G[f"{tag}_ep{ep}_done"]
    np.testing.assert_allclose(obs[0], gobs[0], rtol=0, atol=2e-6)
    nd = 13 + 2 * m.n_joints
    np.testing.assert_allclose(o.get_state()[0][:nd], G[f"{tag}_ep{ep}_state"][0][:nd], atol=1e-7)
    for t, a in enumerate(G[f"{tag}_ep{ep}_actions"]):
        obs, rew, done, _ = o.step(a.astype(np.float32)[None])
        # (entries 26..39 are finite-difference joint speeds, (q' - q) / 0.03 of float32 angles: one ulp of an angle is 2e-6 of speed)
        # ... and the episode runs free: 50 solves per step whose `if (normal impulse > 0)` friction switch turns last-bit differences of the
        # two drivers (the reference's Python PD loop in float64 here, the oracle's C loop there) into 1e-5 now and then
        np.testing.assert_allclose(obs[0], gobs[t + 1], rtol=0, atol=3e-5, err_msg=f"step {t}")
        assert abs(float(rew[0]) - grew[t]) < 5e-6, (t, rew, grew[t])
        assert (int(done[0]) & 1) == gdone[t]
        assert o.get_task()[0, 39] == G[f"{tag}_ep{ep}_istep"][t + 1]


def test_recorded_episodes_reach_the_mirrored_branch():
    n = sum(int((G[f"mirror_ep{ep}_obs"][:, 40:42].max(1) > 0.5).sum()) for ep in range(3))
    flipped = sum(int((G[f"mocca_ep{ep}_obs"][:, 40] > 0.5).sum()) for ep in range(3))
    assert n > 0 and flipped > 5


def test_mirror_swaps_and_negates_like_the_reference():
    """Same state, same phase: the mirrored class's observation is the plain one with left <-> right swapped and the lateral
    entries negated (index lists of env_cassie.py:554-571, read back from the reference objects)."""
    left, right = list(G["mirror_mi_left_obs_inds"]), list(G["mirror_mi_right_obs_inds"])
    neg = list(G["mirror_mi_neg_obs_inds"]) + list(G["mirror_mi_sideneg_obs_inds"])
    _, oa = _oracle("mocca")
    _, ob = _oracle("mirror")
    seen = set()
    for istep in (800, 100):
        for o in (oa, ob):
            o.set_tape(np.array([(istep + 0.5) / 10000.0]))
        a, b = oa.reset(seed=0)[0], ob.reset(seed=0)[0]
        if a[40] > 0.5:
            ref = a.copy()
            ref[left + right] = a[right + left]
            ref[neg] *= -1
            np.testing.assert_array_equal(b, ref)
        else:
            np.testing.assert_array_equal(b, a)
        seen.add(bool(a[40] > 0.5))
    assert seen == {True, False}


def test_f32_oracle_follows_the_f64_one():
    m, o64 = _oracle("mirror", "f64")
    _, o32 = _oracle("mirror", "f32")
    for o in (o64, o32):
        o.set_tape(np.array([(3971 + 0.5) / 10000.0]))
        o.reset(seed=0)
        o.set_tape(None)
    rng = np.random.default_rng(0)
    for t in range(3):
        a = (0.1 * rng.uniform(-1, 1, (1, 10))).astype(np.float32)
        o32.set_state(o64.get_state()); o32.set_task(o64.get_task())
        x, r, d, _ = o64.step(a)
        y, s, e, _ = o32.step(a)
        assert np.abs(x - y).max() < 5e-3 and abs(float(r[0] - s[0])) < 5e-3 and d[0] == e[0]


def test_phase_ids_are_registered_like_the_reference():
    import mocca_envs_amd as pkg
    assert pkg.REGISTERED["CassiePhaseMocca2DEnv-v0"] == ("mocca_envs_amd.envs:CassiePhaseMoccaEnv", {"planar": True})    # __init__.py:31-36
    assert pkg.REGISTERED["CassiePhaseMirror2DEnv-v0"] == ("mocca_envs_amd.envs:CassiePhaseMirrorEnv", {"planar": True})  # :38-43
