"""Host-side mirror of the reference's reset / terrain / mirror logic vs golden vectors captured from it."""
import numpy as np

from mocca_envs_amd import host_logic as H
from mocca_envs_amd import model as M


class TapeRNG:
    """Same stand-in RandomState tests/golden/make_golden.py gave the reference: draws pop a recorded tape."""

    def __init__(self, tape):
        self.tape, self.pos = np.asarray(tape), 0

    def _pop(self, n=None):
        if n is None:
            self.pos += 1
            return float(self.tape[self.pos - 1])
        self.pos += n
        return self.tape[self.pos - n:self.pos].copy()

    def rand(self):
        return self._pop()

    def uniform(self, low=0.0, high=1.0, size=None):
        return low + (high - low) * (self._pop() if size is None else self._pop(int(size)))

    def choice(self, seq):
        return seq[0] if self._pop() < 0.5 else seq[1]


def test_custom_reset_draws_match_reference(golden):
    m = M.compile_walker3d()
    for ep in range(int(golden["custom_n_episodes"])):
        g = lambda k: golden[f"custom_ep{ep}_{k}"]
        rng = TapeRNG(g("tape"))
        dist, angle, stop = H.randomize_target(rng, bool(g("eval_mode")))
        q, mirrored = H.reset_pose(rng, m, True)
        assert abs(dist - float(g("reset_dist"))) < 1e-12 and abs(angle - float(g("reset_angle"))) < 1e-12
        assert stop == float(g("reset_stop_frames")) and int(mirrored) == int(g("reset_mirrored"))
        np.testing.assert_allclose(q, g("reset_q"), atol=1e-7)  # model limits are fp32-rounded


def test_stepper_reset_draws_match_reference(golden):
    m = M.compile_walker3d(M.TASK_WALKER3D_STEPPER)
    for ep in range(int(golden["stepper_n_episodes"])):
        g = lambda k: golden[f"stepper_ep{ep}_{k}"]
        rng = TapeRNG(g("tape"))
        q, mirrored = H.reset_pose(rng, m, True)
        table = H.generate_step_placements(rng, int(g("curriculum")))
        np.testing.assert_allclose(q, g("reset_q"), atol=1e-7)
        np.testing.assert_allclose(table, g("terrain"), atol=1e-12)
        assert abs(H.applied_gain(int(g("curriculum"))) - float(g("applied_gain"))) < 1e-12


def test_terrain_generator(golden):
    for cur in (0, 5, 9):
        table = H.generate_step_placements(TapeRNG(golden[f"terrain_c{cur}_tape"]), cur)
        np.testing.assert_allclose(table, golden[f"terrain_c{cur}_table"], atol=1e-12)


def test_curricula(golden):
    np.testing.assert_allclose([H.terminal_height(c) for c in range(10)], golden["terminal_height_curriculum"])
    np.testing.assert_allclose([H.applied_gain(c) for c in range(10)], golden["applied_gain_curriculum"])


def test_mirror_indices(golden):
    for name, stepper, task in (("custom", False, M.TASK_WALKER3D_CUSTOM), ("stepper", True, M.TASK_WALKER3D_STEPPER)):
        got = H.mirror_indices(M.compile_walker3d(task), stepper)
        for k, v in zip(["neg_obs", "right_obs", "left_obs", "neg_act", "right_act", "left_act"], got):
            np.testing.assert_array_equal(v, golden[f"mirror_{name}_{k}"])


def test_registration_surface(golden):
    """ids, entry points and episode caps of the envs this package serves == the reference's."""
    import mocca_envs_amd
    from mocca_envs_amd import gym_shim
    ref = dict(zip(golden["registered_ids"], golden["registered_max_steps"]))
    for env_id in mocca_envs_amd.REGISTERED:
        assert env_id in ref and int(ref[env_id]) == 1000
    try:
        import gym  # noqa: F401
    except ImportError:
        for env_id in mocca_envs_amd.REGISTERED:
            assert gym_shim.registry.env_specs[env_id].max_episode_steps == 1000
    assert int(golden["obs_dim_custom"]) == 52 and int(golden["obs_dim_stepper"]) == 65 and int(golden["act_dim"]) == 21


def test_time_limit_shim():
    from mocca_envs_amd import gym_shim

    class Dummy(gym_shim.Env):
        observation_space = action_space = gym_shim.Box(-np.ones(1), np.ones(1))

        def reset(self):
            return np.zeros(1)

        def step(self, a):
            return np.zeros(1), 0.0, False, {}
    env = gym_shim.TimeLimit(Dummy(), 3)
    env.reset()
    assert [env.step(0)[2] for _ in range(3)] == [False, False, True]


def test_shim_seeding_hashes_the_seed_like_classic_gym():
    """gym.utils.seeding.np_random (gym <= 0.21) seeds numpy with the 32-bit words of SHA-512(str(seed))[:8], not with the seed:
    the shim must do the same or `env.seed(s)` starts another stream than under the reference's gym."""
    import hashlib
    import struct
    from mocca_envs_amd import gym_shim as g
    rng, seed = g._np_random(5)
    assert seed == 5
    lo, hi = struct.unpack("<II", hashlib.sha512(b"5").digest()[:8])
    want = np.random.RandomState(); want.seed([lo, hi])
    assert rng.rand() == want.rand()
    assert g._np_random(5)[0].rand() != np.random.RandomState(5).rand()
    assert g._np_random(2 ** 64 + 5)[0].rand() == g._np_random(5)[0].rand()       # create_seed folds ints modulo 2^64
    r0, s0 = g._np_random(None)
    assert isinstance(s0, int) and 0 <= s0 < 2 ** 64
    import pytest
    with pytest.raises(ValueError):
        g._np_random(-1)
