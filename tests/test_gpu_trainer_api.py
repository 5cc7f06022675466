"""`mocca_envs_amd.trainer_api`: the VecPyTorch-shaped surface SymmetricRL / ALLSTEPS drive (/root/reference/README.md:33-39), over one VecEnv.
Checked against a plain VecEnv stepped beside it with the same seed and actions: observations and rewards bit for bit; Monitor's episode
return / length, TimeLimitMask's `bad_transition` and the Stepper's `steps_reached` (env_locomotion.py:562-566) against a hand-kept ledger.
Needs a real MI355X: -m gpu."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env_id,sub_batches", [("Walker3DCustomEnv-v0", 1), ("Walker3DStepperEnv-v0", 2), ("Walker2DCustomEnv-v0", 1)])
def test_trainer_surface_matches_a_hand_kept_ledger(env_id, sub_batches):
    import torch
    from mocca_envs_amd.trainer_api import make_vec_envs
    from mocca_envs_amd.vec_env import VecEnv
    n = 64
    steps = 1100 if "2D" in env_id else 260      # Walker2DCustomEnv never terminates (env_locomotion.py:302-309): only the TimeLimit ends its episodes
    envs = make_vec_envs(env_id, seed=4, num_processes=n, log_dir=None, sub_batches=sub_batches, terminal_observation=True)
    ref = VecEnv(env_id, n, auto_reset=True, seed=4, terminal_obs=True)
    assert envs.num_envs == n and envs.observation_space.shape == (ref.obs_dim,) and envs.action_space.shape == (ref.act_dim,)
    assert float(envs.action_space.high.max()) == 1.0 and float(envs.action_space.low.min()) == -1.0
    obs = envs.reset()
    assert torch.equal(obs, ref.reset()) and obs.dtype == torch.float32 and obs.is_cuda
    g = torch.Generator(device="cuda").manual_seed(9)
    ret, length = np.zeros(n), np.zeros(n, int)
    n_eps = n_bad = 0
    for t in range(steps):
        a = torch.rand(n, ref.act_dim, device="cuda", generator=g) * 2 - 1
        obs, rew, done, infos = envs.step(a)
        o2, r2, d2, i2 = ref.step(a)
        assert torch.equal(obs, o2) and torch.equal(rew, r2.unsqueeze(1)) and rew.shape == (n, 1)
        assert isinstance(done, np.ndarray) and done.dtype == bool and (done == (d2.cpu().numpy() != 0)).all()
        assert len(infos) == n
        ret += r2.cpu().numpy(); length += 1
        d2h, i2h = d2.cpu().numpy(), i2.cpu().numpy()
        masks, bad = envs.masks.cpu().numpy()[:, 0], envs.bad_masks.cpu().numpy()[:, 0]
        for i, info in enumerate(infos):            # the trainers' own loop over the N dicts
            if done[i]:
                assert abs(info["episode"]["r"] - ret[i]) < 1e-3 * (1 + abs(ret[i])) and info["episode"]["l"] == length[i]
                assert ("bad_transition" in info) == (d2h[i] == 2) and masks[i] == 0.0 and bad[i] == (0.0 if d2h[i] == 2 else 1.0)
                assert torch.equal(info["terminal_observation"], ref.terminal_obs[i])
                if "Stepper" in env_id:
                    assert info["steps_reached"] == i2h[i]
                n_eps += 1; n_bad += d2h[i] == 2
                ret[i], length[i] = 0.0, 0
            else:
                assert info == {} and masks[i] == 1.0 and bad[i] == 1.0
        assert sorted(k for k, _ in infos.finished()) == list(np.nonzero(done)[0])
    assert n_eps > n // 2
    if "2D" in env_id:
        assert n_bad == n_eps > 0                   # every Walker2D episode ends by the TimeLimit
    # the curriculum and mirror calls the trainers make
    if "Stepper" in env_id:
        envs.set_env_params({"curriculum": 5})
        assert envs.env_method("set_env_params", {"curriculum": 7}) == [None] * n
    if "2D" not in env_id:
        assert len(envs.get_mirror_indices()) == 6
    envs.close(); ref.close()
