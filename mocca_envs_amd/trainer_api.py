"""The vectorised env surface the reference's trainers drive, over the GPU stepper.

SymmetricRL and ALLSTEPS / SteppingStone (/root/reference/README.md:33-39) are PPO trainers in the pytorch-a2c-ppo-acktr mould: they build
`envs = make_vec_envs(env_name, seed, num_processes, log_dir)` -- N single-env processes (`gym.make(env_name)` + `Monitor` + a TimeLimit mask)
behind baselines' `ShmemVecEnv`, wrapped in `VecPyTorch` -- and then only ever touch this surface:

    obs = envs.reset()                                   # float tensor [N, obs_dim] on the trainer's device
    obs, reward, done, infos = envs.step(action)         # reward [N, 1]; done: numpy bool [N]; infos: N dicts
    for info in infos: info["episode"]["r"]              # Monitor's episode return / length, in the step that ends the episode
    "bad_transition" in info                             # TimeLimitMask: the episode was cut by max_episode_steps, not terminated
    envs.observation_space / action_space / num_envs, envs.close()
    env.unwrapped.get_mirror_indices()                   # on a dummy env (SymmetricRL); set_env_params({"curriculum": k}) (ALLSTEPS)

`TorchVecEnv` is that surface with the N processes replaced by ONE `VecEnv` (or `SubBatchedVecEnv`): finished envs are reset inside the
launch and `obs` already holds the next episode's first observation (baselines' VecEnv contract); episode return and length are accumulated
on the device; only the done flags (N bytes) and the finished envs' statistics cross to the host per step.  A trainer that wants no host
traffic at all reads `masks` / `bad_masks` (float tensors [N, 1] on the device, what the PPO loop builds from `done` / `infos`) and skips `infos`
(the list-like `infos` only builds dicts for the envs that finished; `infos.finished()` iterates just those).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import gym_shim
from .multi import make_vec_env


class _Infos:
    """List-of-dicts view of one step's infos: {} for envs that go on, Monitor / TimeLimitMask keys for envs that finished."""

    def __init__(self, n: int, finished: dict):
        self._n, self._fin = n, finished

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        return self._fin.get(i, {})

    def __iter__(self):
        fin, empty = self._fin, {}
        return (fin.get(i, empty) for i in range(self._n))

    def finished(self):
        """(env index, info dict) of the envs whose episode ended in this step -- what a trainer's logging loop is after."""
        return self._fin.items()


class TorchVecEnv:
    """`VecPyTorch`-shaped env batch on one MI355X (module docstring).  kwargs go to the env class (`plank_class=...`) / `VecEnv`
    (`max_rows=...`); `sub_batches=k` steps the batch as k sub-batches on their own HIP streams (`multi.SubBatchedVecEnv`)."""

    def __init__(self, env_id: str, num_envs: int, seed: int = 0, device: Optional[int] = None, sub_batches: int = 1,
                 terminal_observation: bool = False, **kw):
        self.venv = make_vec_env(env_id, num_envs, sub_batches=sub_batches, seed=seed, auto_reset=True, terminal_obs=terminal_observation,
                                 **({"device": device} if device is not None else {}), **kw)
        self.env_id, self.num_envs = env_id, int(num_envs)
        self.device = self.venv.device
        high = np.inf * np.ones(self.venv.obs_dim, dtype=np.float32)
        self.observation_space = gym_shim.Box(-high, high, dtype=np.float32)                      # robots.py:18-29, env_locomotion.py:58-60
        self.action_space = gym_shim.Box(-np.ones(self.venv.act_dim, np.float32), np.ones(self.venv.act_dim, np.float32), dtype=np.float32)
        self._ret = torch.zeros(num_envs, device=self.device)
        self._len = torch.zeros(num_envs, dtype=torch.int32, device=self.device)
        self.masks = torch.ones(num_envs, 1, device=self.device)        # 0 where the episode ended in the last step
        self.bad_masks = torch.ones(num_envs, 1, device=self.device)    # 0 where it ended by the TimeLimit only ("bad_transition")
        self._want_terminal = bool(terminal_observation)
        self._pack = torch.zeros(num_envs, 5, device=self.device)
        self._pack_host = torch.zeros(num_envs, 5).pin_memory()
        from . import model as M
        self._stepper = self.venv.task_id == M.TASK_WALKER3D_STEPPER

    # ---- the VecPyTorch surface ----
    def reset(self) -> torch.Tensor:
        self._ret.zero_(); self._len.zero_()
        return self.venv.reset()

    def step(self, actions: torch.Tensor):
        actions = actions.to(device=self.device, dtype=torch.float32)
        obs, rew, done, kinfo = self.venv.step(actions.contiguous())
        self._ret += rew
        self._len += 1
        ended = done != 0
        truncated_only = done == 2                                       # bit1 = TimeLimit, bit0 = terminated (include/mocca.h)
        self.masks = (~ended).float().unsqueeze(1)
        self.bad_masks = (~truncated_only).float().unsqueeze(1)
        # ONE download per step: done flags + the statistics of the envs that finished, in one pinned image (N x 5 floats)
        torch.stack([ended.float(), self._ret, self._len.float(), truncated_only.float(), kinfo.float()], dim=1, out=self._pack)
        self._pack_host.copy_(self._pack, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        h = self._pack_host.numpy()
        done_host = h[:, 0] != 0.0
        finished = {}
        if done_host.any():
            self._ret *= self.masks[:, 0]                                # (after the download was queued: same stream)
            self._len *= (~ended).to(self._len.dtype)
            for i in np.nonzero(done_host)[0].tolist():
                info = {"episode": {"r": float(h[i, 1]), "l": int(h[i, 2])}}
                if h[i, 3] != 0.0:
                    info["bad_transition"] = True
                    info["TimeLimit.truncated"] = True
                if self._stepper:
                    info["steps_reached"] = int(h[i, 4])          # Walker3DStepperEnv.step's info at done (env_locomotion.py:562-566)
                if self._want_terminal:
                    info["terminal_observation"] = self.venv.terminal_obs[i]
                finished[i] = info
        return obs, rew.unsqueeze(1), done_host, _Infos(self.num_envs, finished)

    def step_async(self, actions):      # baselines' two-phase form
        self._pending = self.step(actions)

    def step_wait(self):
        out, self._pending = self._pending, None
        return out

    def close(self):
        self.venv.close()

    def seed(self, seed: int):
        return self.venv.seed(seed)

    # ---- what the trainers reach through `envs.venv` / a dummy env ----
    def get_mirror_indices(self):
        return self.venv.get_mirror_indices()

    def set_env_params(self, params_dict):
        self.venv.set_env_params(params_dict)

    def set_robot_params(self, params_dict):
        self.venv.set_robot_params(params_dict)

    def env_method(self, name: str, *args, **kwargs):
        """baselines' `venv.env_method(name, ...)`: one result per env for the methods the reference's envs expose batch-wide."""
        if name in ("set_env_params", "set_robot_params", "evaluation_mode", "get_mirror_indices", "seed"):
            return [getattr(self, name)(*args, **kwargs)] * self.num_envs
        raise AttributeError(f"env_method({name!r}) has no batched counterpart")

    def evaluation_mode(self, on=True):
        self.venv.evaluation_mode(on)


def make_vec_envs(env_name: str, seed: int, num_processes: int, log_dir=None, device=None, **kw) -> TorchVecEnv:
    """Same call as the trainers' `common.envs_utils.make_vec_envs(env_name, seed, num_processes, log_dir)`; `log_dir` (Monitor's csv) is
    accepted and ignored -- episode statistics arrive through `infos`."""
    dev = None
    if device is not None:
        dev = torch.device(device).index if not isinstance(device, int) else device
    return TorchVecEnv(env_name, num_processes, seed=seed, device=dev, **kw)
