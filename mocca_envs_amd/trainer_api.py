"""The vectorised env surface the reference's trainers drive, over the GPU stepper.

SymmetricRL and ALLSTEPS / SteppingStone (/root/reference/README.md:33-39) are PPO trainers in the pytorch-a2c-ppo-acktr mould: they build
`envs = make_vec_envs(env_name, seed, num_processes, log_dir)` -- N single-env processes (`gym.make(env_name)` + `Monitor` + a TimeLimit mask)
behind baselines' `ShmemVecEnv`, wrapped in `VecPyTorch` -- and then only ever touch this surface:

    obs = envs.reset()                                   # float tensor [N, obs_dim] on the trainer's device
    obs, reward, done, infos = envs.step(action)         # reward [N, 1]; done: bool array [N]; infos: N dicts
    for info in infos: info["episode"]["r"]              # Monitor's episode return / length, in the step that ends the episode
    "bad_transition" in info                             # TimeLimitMask: the episode was cut by max_episode_steps
    envs.observation_space / action_space / num_envs, envs.close()
    env.unwrapped.get_mirror_indices()                   # on a dummy env (SymmetricRL); set_env_params({"curriculum": k}) (ALLSTEPS)

`TorchVecEnv` is that surface with the N processes replaced by ONE `VecEnv` (or `SubBatchedVecEnv`), and with Monitor and the TimeLimit mask
INSIDE the step kernel (`VecEnv.episode_stats`, include/mocca.h mocca_set_episode_stats): the launch accumulates every env's episode return,
writes the PPO loop's `masks` / `bad_masks` columns on the device and, for the few envs that finished, a 16-byte record straight into pinned
host memory.  `step()` therefore is one kernel launch and one event record: no torch arithmetic, no copy, NO SYNCHRONISE.  `done` and `infos`
are lazy views of that step's records -- they wait for THAT step's launch, and only when the trainer looks at them (iterate / index /
np.asarray), and then behave like the bool array and the list of dicts of the contract above.  A loop that logs `infos` one step late
(after it has issued the next step) never lets the GPU run dry.  A trainer that wants no host traffic at all reads `envs.masks` / `envs.bad_masks`
([N, 1] float tensors on the device, what the PPO loop builds from `done` / `infos`), `envs.done` (uint8 [N], bit0 terminated, bit1
TimeLimit) and `envs.episode_totals` ([4] on the device: sums of return, length, episodes, truncated episodes since it last zeroed them),
and never touches `done` / `infos`; that loop can be captured in a `torch.cuda.CUDAGraph` (tests/test_gpu_trainer_api.py).
"""
from __future__ import annotations

import inspect
import weakref
from typing import Optional

import numpy as np
import torch

from . import gym_shim
from .multi import make_vec_env


class _StepRecords:
    """The episode records of ONE step (serial k), read from the pinned ring on first use."""

    __slots__ = ("_env", "serial", "_done", "_fin", "_rows", "_idx", "_term", "_stepper", "__weakref__")

    def __init__(self, env: "TorchVecEnv", serial: int):
        self._env, self.serial, self._done, self._fin, self._rows, self._idx, self._term = env, serial, None, None, None, None, None
        self._stepper = env._stepper

    def materialise(self):
        """Wait for THIS step's launch and copy its records out of the ring slot (numpy only: no per-env Python work yet)."""
        if self._done is not None:
            return
        env = self._env
        if env._events is not None:
            env._events[self.serial % env._slots].synchronize()  # the launch of THIS step has completed (later ones may still be queued or running)
        else:
            env.synchronize()                                 # (record_events=False: wait for everything issued so far)
        rec = env._ring[self.serial % env._slots]            # [N][4] int32 view of the pinned slot
        self._idx = np.nonzero(rec[:, 0] == np.array(self.serial, np.uint32).view(np.int32))[0]
        self._rows = rec[self._idx]                          # (a copy: the slot is rewritten `slots` steps from now)
        done = np.zeros(env.num_envs, dtype=bool)
        done[self._idx] = True
        if env._want_terminal and self._idx.size:            # owned copies of the finished envs' terminal observations, gathered once
            self._term = env.venv.terminal_obs.index_select(0, torch.from_numpy(self._idx).to(env.device))
        self._done = done
        self._env = None

    def episodes(self) -> dict:
        """The finished envs of the step as arrays: env index, Monitor's r and l, the TimeLimit flag, the step's info word."""
        self.materialise()
        r = self._rows
        flags = r[:, 3].view(np.uint32) if r.size else np.zeros(0, np.uint32)
        return {"env": self._idx, "r": np.ascontiguousarray(r[:, 1]).view(np.float32), "l": r[:, 2], "truncated": (flags & 2) != 0, "info": flags >> 8}

    def fin(self) -> dict:
        """{env index: info dict} of the finished envs, built on first use."""
        if self._fin is None:
            e = self.episodes()
            self._fin = {}
            for j, i in enumerate(e["env"].tolist()):
                info = {"episode": {"r": float(e["r"][j]), "l": int(e["l"][j])}}
                if e["truncated"][j]:                        # TimeLimitMask: done at max_episode_steps (whether or not it also terminated)
                    info["bad_transition"] = True
                    info["TimeLimit.truncated"] = True
                if self._stepper:
                    info["steps_reached"] = int(e["info"][j])     # Walker3DStepperEnv.step's info at done (env_locomotion.py:562-566)
                if self._term is not None:
                    info["terminal_observation"] = self._term[j]
                self._fin[i] = info
        return self._fin


class _LazyDone:
    """`done` of one step: a bool array [N] that is fetched when first looked at (np.asarray(done), iteration, indexing, any attribute of
    numpy.ndarray)."""

    __slots__ = ("_rec",)

    def __init__(self, rec: _StepRecords):
        self._rec = rec

    def numpy(self) -> np.ndarray:
        self._rec.materialise()
        return self._rec._done

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def __len__(self):
        return len(self._rec._done) if self._rec._done is not None else self._rec._env.num_envs

    def __iter__(self):
        return iter(self.numpy().tolist())      # Python bools: a trainer's `for d in done` comprehension runs twice as fast over them

    def __getitem__(self, i):
        return self.numpy()[i]

    def __getattr__(self, name):        # .any(), .sum(), .dtype, .shape, .nonzero(), .astype(...), ...
        return getattr(self.numpy(), name)

    def __invert__(self):
        return ~self.numpy()

    def __and__(self, o):
        return self.numpy() & o

    def __or__(self, o):
        return self.numpy() | o

    def __eq__(self, o):
        return self.numpy() == o

    def __ne__(self, o):
        return self.numpy() != o

    __hash__ = None

    def __repr__(self):
        return f"LazyDone({self.numpy()!r})"


class _Infos:
    """List-of-dicts view of one step's infos: {} for envs that go on, Monitor / TimeLimitMask keys for envs that finished."""

    __slots__ = ("_n", "_rec")

    def __init__(self, n: int, rec: _StepRecords):
        self._n, self._rec = n, rec

    def _finished(self) -> dict:
        return self._rec.fin()

    def episodes(self) -> dict:
        """The same information without a dict per env: {"env": indices of the envs that finished in this step, "r": their episode returns,
        "l": lengths, "truncated": ended at the TimeLimit ("bad_transition"), "info": the step's info word (Stepper: steps_reached)} as
        numpy arrays -- what a logging loop over thousands of envs wants."""
        return self._rec.episodes()

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        return self._finished().get(i, {})

    def __iter__(self):
        out = [{}] * self._n                    # ONE shared empty dict for the envs that go on (the verbatim loops only read it)
        for i, info in self._finished().items():
            out[i] = info
        return iter(out)

    def finished(self):
        """(env index, info dict) of the envs whose episode ended in this step -- what a trainer's logging loop is after."""
        return self._finished().items()


class TorchVecEnv:
    """`VecPyTorch`-shaped env batch on one MI355X (module docstring).  kwargs go to the env class (`plank_class=...`) / `VecEnv`
    (`max_rows=...`); `sub_batches=k` steps the batch as k sub-batches on their own HIP streams (`multi.SubBatchedVecEnv`).
    `record_slots`: how many steps' episode records the pinned ring holds; a step's `done` / `infos` that are still referenced when
    their slot comes up for rewriting are fetched then, so they stay correct however late they are read.  `eager_done=True` returns
    `done` as a real numpy array (one synchronise per step -- for code that insists on `isinstance(done, np.ndarray)`); `record_events=False`
    drops the per-step event (for loops that read `masks` / `episode_totals` only)."""

    def __init__(self, env_id: str, num_envs: int, seed: int = 0, device: Optional[int] = None, sub_batches: int = 1,
                 terminal_observation: bool = False, record_slots: int = 8, eager_done: bool = False, record_events: bool = True, **kw):
        self.venv = make_vec_env(env_id, num_envs, sub_batches=sub_batches, seed=seed, auto_reset=True, terminal_obs=terminal_observation,
                                 **({"device": device} if device is not None else {}), **kw)
        self.env_id, self.num_envs = env_id, int(num_envs)
        self.device = self.venv.device
        high = np.inf * np.ones(self.venv.obs_dim, dtype=np.float32)
        self.observation_space = gym_shim.Box(-high, high, dtype=np.float32)                      # robots.py:18-29, env_locomotion.py:58-60
        self.action_space = gym_shim.Box(-np.ones(self.venv.act_dim, np.float32), np.ones(self.venv.act_dim, np.float32), dtype=np.float32)
        ep = self.venv.episode_stats(True, slots=int(record_slots))
        self.masks = ep["masks"].unsqueeze(1)              # [N, 1], 0 where the episode ended in the last step; REWRITTEN IN PLACE by every step
        self.bad_masks = ep["bad_masks"].unsqueeze(1)      # [N, 1], 0 where it ended at the TimeLimit ("bad_transition")
        self.episode_totals = ep["totals"]                 # [4] sums of return / length / episodes / truncated episodes; zero it when you like
        self.done = self.venv.done                         # uint8 [N] on the device: bit0 terminated, bit1 TimeLimit
        self._ring = ep["records"].numpy()                 # [slots][N][4] int32 over the pinned host ring
        self._slots, self._k = ep["slots"], ep["first_serial"]
        self._live = [None] * self._slots                  # weak references to the lazy records of the last `slots` steps
        # an event behind each step's launch: what that step's lazy records wait for.  record_events=False saves the record (a trainer that
        # never looks at `done` / `infos`): records then wait for everything issued so far when they are looked at
        self._events = [torch.cuda.Event() for _ in range(self._slots)] if record_events else None
        self._rew2 = self.venv.rew.unsqueeze(1)
        self._want_terminal = bool(terminal_observation)
        self._eager = bool(eager_done)
        from . import model as M
        self._stepper = self.venv.task_id == M.TASK_WALKER3D_STEPPER

    def synchronize(self):
        """Wait for every launch issued so far (the streams the env steps on)."""
        if hasattr(self.venv, "synchronize"):
            self.venv.synchronize()
        torch.cuda.current_stream(self.device).synchronize()

    # ---- the VecPyTorch surface ----
    def reset(self) -> torch.Tensor:
        return self.venv.reset()       # (the reset kernel zeroes the envs' running returns: Monitor.reset)

    def step(self, actions: torch.Tensor, into: Optional[dict] = None):
        """`into` (one handle only): {"obs": [N, obs_dim], "reward": [N, 1], "masks": [N, 1], "bad_masks": [N, 1]} -- any subset -- tensors of the
        TRAINER's storage that this step's launch writes directly (row t + 1 of the rollout buffers: PPO reads its next policy input from
        there anyway), instead of this object's own buffers: no copy kernels between the env and the storage.  The returned `obs` / `reward`
        are those tensors; `envs.masks` / `envs.bad_masks` keep pointing at the last buffers handed in."""
        k = self._k
        slot = k % self._slots
        old = self._live[slot]
        if old is not None:
            old = old()
            if old is not None:
                old.materialise()      # its slot is about to be rewritten: fetch it now (its launch finished long ago)
        if actions.device != self.device or actions.dtype != torch.float32 or not actions.is_contiguous():
            actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
        rew = self._rew2
        if into:
            if not hasattr(self.venv, "lib"):
                raise NotImplementedError("step(into=...) needs one handle (sub_batches=1)")
            if "masks" in into or "bad_masks" in into:
                self.masks, self.bad_masks = into.get("masks", self.masks), into.get("bad_masks", self.bad_masks)
                self.venv.episode_masks_into(self.masks, self.bad_masks)
            rew = into.get("reward", rew)
            obs = self.venv.step(actions, obs_out=into.get("obs"), rew_out=into.get("reward"))[0]
        else:
            obs = self.venv.step(actions)[0]
        if self._events is not None:
            self._events[slot].record(torch.cuda.current_stream(self.device))
        self._k = k + 1 if k < 0xFFFFFFFF else 1
        rec = _StepRecords(self, k)
        self._live[slot] = weakref.ref(rec)
        return obs, rew, (_LazyDone(rec).numpy() if self._eager else _LazyDone(rec)), _Infos(self.num_envs, rec)

    def capture_rollout(self, policy, num_steps: int, sink=None, warmup: int = 2, into=None):
        """The collection phase as ONE CUDA graph: `num_steps` x { action = policy(obs); env.step(action); sink(t, obs, reward, masks, bad_masks,
        action) } captured once, replayed with `.replay()` (returns the `torch.cuda.CUDAGraph`).  `policy` maps the observation tensor [N, obs_dim] to
        actions [N, act_dim] with torch ops only (no host reads; `policy(obs, t)` is called with the step index if it takes two arguments);
        `sink` copies what the trainer keeps into ITS pre-allocated rollout storage (`rollouts.obs[t + 1].copy_(obs)` ...: `obs`, `reward` [N, 1], `masks` / `bad_masks` [N, 1] are this env's persistent buffers, rewritten
        by every step).  `mocca_step` keeps no host state per launch (ABI 7), so a replay advances the envs exactly as `num_steps` calls of
        `step()` would -- bit for bit (tests/test_gpu_trainer_api.py).  Episode statistics of a replayed rollout: `episode_totals` (the lazy
        `done` / `infos` of `step()` do not exist inside a graph).  `warmup` eager iterations run first on a side stream, as torch requires before a
        capture: they advance the envs too.  `into`: a callable t -> the `into` dict of `step()`, each with an "obs" entry -- launch t of the
        graph writes its observations / rewards / masks straight into those tensors (row t + 1 of the trainer's storage) and `policy` reads
        `into(t - 1)["obs"]`; `into(-1)["obs"]` is the storage's row 0, which must hold the current observation before every replay (PPO's
        `rollouts.after_update()` copies the last row there; this call leaves it filled): the rollout needs no copy kernels at all."""
        if not hasattr(self.venv, "lib"):
            raise NotImplementedError("capture_rollout needs one handle (sub_batches=1): sub-batches step on streams of their own")
        venv, dev = self.venv, self.device
        obs, rew = venv.obs, self._rew2
        if len(inspect.signature(policy).parameters) >= 2:        # policy(obs, t): e.g. to write its action into the storage's row t
            act_of = policy
        else:
            act_of = lambda o, t: policy(o)

        def body(t):
            if into is None:
                action = act_of(obs, t)
                venv.step(action)
                o, r = obs, rew
            else:
                action = act_of(into(t - 1)["obs"], t)
                d = into(t)
                if "masks" in d or "bad_masks" in d:
                    self.masks, self.bad_masks = d.get("masks", self.masks), d.get("bad_masks", self.bad_masks)
                    venv.episode_masks_into(self.masks, self.bad_masks)
                o, r = venv.step(action, obs_out=d.get("obs"), rew_out=d.get("reward"))[:2]
            if sink is not None:
                sink(t, o, r, self.masks, self.bad_masks, action)

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            if into is not None:
                into(-1)["obs"].copy_(obs)
            for t in range(warmup):
                body(t)
            if into is not None and warmup > 0:
                into(-1)["obs"].copy_(into(warmup - 1)["obs"])      # row 0 <- the observation the warm-up ended on
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph), torch.no_grad():
            for t in range(num_steps):
                body(t)
        return graph

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
    accepted and ignored -- episode statistics arrive through `infos` / `envs.episode_totals`."""
    dev = None
    if device is not None:
        dev = torch.device(device).index if not isinstance(device, int) else device
    return TorchVecEnv(env_name, num_processes, seed=seed, device=dev, **kw)
