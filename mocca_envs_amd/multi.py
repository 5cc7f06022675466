"""One env batch behind several handles: sub-batches of one GPU that step independently, and shards on several GPUs.

Both keep the reference's structure -- every env is its own world (one Bullet client per env, /root/reference/mocca_envs/env_base.py:55),
no env reads another's state -- and `VecEnv`'s results: the parts carry the GLOBAL env ids of the batch (`env_offset`), random draws are
keyed by them, so the batch does not depend on how it is cut (tests/test_gpu_multi.py: bit for bit against the single handle).

`SubBatchedVecEnv` -- the GPU's batch as k sub-batches, each with a handle and a HIP stream of its own.  A synchronous launch of N envs
lasts as long as its slowest wave while most of the chip idles behind it; with sub-batches the tail of one launch is
filled by the next launch of ANOTHER sub-batch.  The trainer loop that gets it is the double-buffered one:

    env = SubBatchedVecEnv("Walker3DCustomEnv-v0", 8192, sub_batches=2, max_rows=32)
    env.reset()
    while training:
        for i in range(env.n_parts):
            obs, rew, done, info = env.wait(i)        # torch's current stream now sees sub-batch i's last step
            act = policy(obs)                         # ... while the other sub-batches are stepping on their streams
            env.step_async(i, act)                    # launch on sub-batch i's stream, ordered after `act`

`ShardedVecEnv` -- one process, one shard per device (what the reference's single-process trainers, README.md:33-39, can call): one
handle + stream per GPU, no host synchronisation between devices inside `step`, no collective; `gather=True` copies observations,
rewards and done flags to one device (peer-to-peer copies, 1.7 MB per GPU and step at 8192 envs).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch

from . import lib as _lib
from .vec_env import VecEnv


class _Parts:
    """N envs as contiguous ranges [lo_k, hi_k) behind one VecEnv each; setters fan out, getters concatenate."""

    parts: List[VecEnv]
    streams: List[torch.cuda.Stream]
    slices: List[slice]

    @property
    def n_parts(self) -> int:
        return len(self.parts)

    def _build(self, env_id, counts, devices, seed, env_offset, auto_reset, terminal_obs, kw):
        self.env_id, self.n_envs = env_id, int(sum(counts))
        self.parts, self.streams, self.slices = [], [], []
        lo = 0
        for cnt, dev in zip(counts, devices):
            st = torch.cuda.Stream(device=dev)
            e = VecEnv(env_id, cnt, device=dev, auto_reset=auto_reset, seed=seed, env_offset=env_offset + lo, **kw)
            e.stream = st                   # every call of this handle goes to its own stream from here on
            self.parts.append(e); self.streams.append(st); self.slices.append(slice(lo, lo + cnt))
            lo += cnt
        p0 = self.parts[0]
        self.model, self.task_id = p0.model, p0.task_id
        self.obs_dim, self.act_dim, self.state_dim = p0.obs_dim, p0.act_dim, p0.state_dim
        self.seed_value = int(seed)

    # ---- ordering between a part's stream and torch's current stream on the part's device (stream-level, never the host) ----
    def _before(self, i: int):
        """work queued on sub-batch i's stream from now on runs after what the caller has queued on the current stream"""
        self.streams[i].wait_stream(torch.cuda.current_stream(self.parts[i].device))

    def _after(self, i: int):
        """what the caller queues on the current stream from now on runs after sub-batch i's stream"""
        torch.cuda.current_stream(self.parts[i].device).wait_stream(self.streams[i])

    # ---- the reference's env-level setters (env_base.py:103-118, env_locomotion.py:76-77), for all parts ----
    def set_param(self, pid: int, value: float):
        for e in self.parts:
            e.set_param(pid, value)

    def set_param_v(self, pid: int, values, broadcast: bool = False):
        v = torch.as_tensor(values, dtype=torch.float32).reshape(-1)
        if v.numel() != (1 if broadcast else self.n_envs):
            raise ValueError("values must hold one float per env (or one float with broadcast=True)")
        for e, sl in zip(self.parts, self.slices):
            e.set_param_v(pid, v if broadcast else v[sl], broadcast)

    def set_env_params(self, params_dict):
        for k, v in params_dict.items():
            if k == "curriculum":
                self.set_param(_lib.PARAM_CURRICULUM, float(v)) if np.ndim(v) == 0 else self.set_param_v(_lib.PARAM_CURRICULUM, v)

    def set_robot_params(self, params_dict):
        if "applied_gain" in params_dict:
            g = params_dict["applied_gain"]
            self.set_param(_lib.PARAM_APPLIED_GAIN, float(g)) if np.ndim(g) == 0 else self.set_param_v(_lib.PARAM_APPLIED_GAIN, g)

    def evaluation_mode(self, on=True):
        self.set_param(_lib.PARAM_EVAL_MODE, 1.0 if on else 0.0) if np.ndim(on) == 0 else self.set_param_v(_lib.PARAM_EVAL_MODE, on)

    def get_mirror_indices(self):
        return self.parts[0].get_mirror_indices()

    def seed(self, seed: int, rewind: bool = True):
        for e in self.parts:
            e.seed(seed, rewind)
        self.seed_value = int(seed)
        return [seed]

    def kernel_info(self) -> dict:
        return self.parts[0].kernel_info()

    def synchronize(self):
        for st in self.streams:
            st.synchronize()

    def close(self):
        for e in getattr(self, "parts", []):
            e.close()
        self.parts = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- snapshots: concatenated on the first part's device ----
    def _cat(self, getter: str) -> torch.Tensor:
        outs = [getattr(e, getter)() for e in self.parts]       # VecEnv's getters order their stream against the current one themselves
        dev = self.parts[0].device
        return torch.cat([o.to(dev) for o in outs], dim=0)

    def get_state(self) -> torch.Tensor:
        return self._cat("get_state")

    def get_task(self) -> torch.Tensor:
        return self._cat("get_task")

    def set_state(self, st) -> None:
        st = torch.as_tensor(st, dtype=torch.float32).reshape(self.n_envs, self.state_dim)
        for e, sl in zip(self.parts, self.slices):
            e.set_state(st[sl])

    def set_task(self, t: torch.Tensor) -> None:
        for e, sl in zip(self.parts, self.slices):
            e.set_task(t[sl])


class SubBatchedVecEnv(_Parts):
    """`n_envs` envs of one GPU as `sub_batches` independently stepping handles (module docstring).  `obs`, `rew`, `done`, `info` (and
    `terminal_obs`) are whole-batch tensors; each sub-batch writes its rows.  step() is the synchronous convenience (all sub-batches
    launched together, waited for together); the throughput comes from step_async() / wait()."""

    def __init__(self, env_id: str = "Walker3DCustomEnv-v0", n_envs: int = 8192, sub_batches: int = 2, device: Optional[int] = None,
                 auto_reset: bool = True, seed: int = 0, env_offset: int = 0, terminal_obs: bool = False, **kw):
        if sub_batches < 1 or n_envs % sub_batches:
            raise ValueError("sub_batches must divide n_envs")
        dev = torch.cuda.current_device() if device is None else int(device)
        per = n_envs // sub_batches
        self._build(env_id, [per] * sub_batches, [dev] * sub_batches, seed, env_offset, auto_reset, terminal_obs, kw)
        self.device_index, self.device = dev, torch.device("cuda", dev)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.obs = torch.zeros(self.n_envs, self.obs_dim, **f32)
        self.rew = torch.zeros(self.n_envs, **f32)
        self.done = torch.zeros(self.n_envs, dtype=torch.uint8, device=self.device)
        self.info = torch.zeros(self.n_envs, dtype=torch.int32, device=self.device)
        for e, sl in zip(self.parts, self.slices):      # the handles write straight into their rows of the whole-batch tensors
            e.obs, e.rew, e.done, e.info = self.obs[sl], self.rew[sl], self.done[sl], self.info[sl]
        self.terminal_obs = None
        if terminal_obs:
            self.terminal_obs = torch.zeros(self.n_envs, self.obs_dim, **f32)
            for e, sl in zip(self.parts, self.slices):
                e.keep_terminal_obs(True, buffer=self.terminal_obs[sl])
        torch.cuda.synchronize(self.device)

    def episode_stats(self, on: bool = True, slots: int = 4) -> Optional[dict]:
        """`VecEnv.episode_stats` for the whole batch: one set of buffers, every sub-batch writes its rows (the sub-batches' serials advance
        together as long as every step() / round of step_async() steps each of them once)."""
        if not on:
            for e in self.parts:
                e.episode_stats(False)
            self.ep = None
            return None
        f32 = dict(dtype=torch.float32, device=self.device)
        masks, bad = torch.ones(self.n_envs, **f32), torch.ones(self.n_envs, **f32)
        totals = torch.zeros(4, **f32)
        records = torch.zeros(int(slots), self.n_envs, 4, dtype=torch.int32).pin_memory()
        torch.cuda.synchronize(self.device)     # the buffers are filled on torch's current stream, the parts read them on their own
        firsts = {e.episode_stats(True, masks=masks[sl], bad_masks=bad[sl], totals=totals, records=records, row0=sl.start)["first_serial"]
                  for e, sl in zip(self.parts, self.slices)}
        if len(firsts) != 1:
            raise _lib.MoccaError("the sub-batches' episode serials differ: attach episode_stats before stepping them separately")
        self.ep = dict(masks=masks, bad_masks=bad, totals=totals, records=records, row0=0, slots=int(slots), first_serial=firsts.pop())
        return self.ep

    def reset(self, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        for e, sl in zip(self.parts, self.slices):
            e.reset(None if mask is None else mask[sl])
        return self.obs

    def step_async(self, i: int, actions: torch.Tensor, ordered: bool = True) -> None:
        """Launch one env.step of sub-batch i on ITS stream and return at once.  actions: [n_envs / k, act_dim] float32, contiguous, on
        the device.  ordered=True makes the launch wait for what torch's current stream has queued so far (the policy that produced
        `actions`, and whatever still reads sub-batch i's rows of obs / rew / done -- the launch overwrites them); ordered=False skips that: the
        caller vouches that the actions are ready (e.g. a pre-computed tape) AND that nothing queued on the current stream still reads those rows."""
        e = self.parts[i]
        if actions.device != e.device or actions.dtype != torch.float32 or not actions.is_contiguous():
            actions = actions.to(device=e.device, dtype=torch.float32).contiguous()     # on the current stream, ordered below
            ordered = True
        if ordered:
            self._before(i)
        actions.record_stream(self.streams[i])   # torch's allocator must not hand this memory out again before sub-batch i's launch has read it
        e.step(actions)

    def wait(self, i: int):
        """Order torch's current stream after sub-batch i's last launch (no host synchronisation) and return its rows
        (obs, rew, done, info) -- views into the whole-batch tensors, valid until sub-batch i's next step_async."""
        self._after(i)
        sl = self.slices[i]
        return self.obs[sl], self.rew[sl], self.done[sl], self.info[sl]

    def step(self, actions: torch.Tensor):
        if actions.shape != (self.n_envs, self.act_dim):
            raise ValueError(f"actions must be [{self.n_envs}, {self.act_dim}]")
        actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
        for i, sl in enumerate(self.slices):
            self.step_async(i, actions[sl])
        for i in range(self.n_parts):
            self._after(i)
        return self.obs, self.rew, self.done, self.info


class ShardedVecEnv(_Parts):
    """`n_envs` envs as one shard per entry of `devices` (a device may be listed twice: two shards on one GPU), one process.
    step(actions): `actions` is a list of per-shard tensors (each on its shard's device) or one [n_envs, act_dim] tensor on any device
    (its rows are copied to the shards).  Returns per-shard lists, or -- gather=True -- whole-batch tensors on `gather_device`."""

    def __init__(self, env_id: str = "Walker3DCustomEnv-v0", n_envs: int = 8 * 8192, devices: Optional[Sequence[int]] = None,
                 auto_reset: bool = True, seed: int = 0, env_offset: int = 0, gather: bool = False, gather_device: Optional[int] = None,
                 terminal_obs: bool = False, **kw):
        devices = list(range(torch.cuda.device_count())) if devices is None else [int(d) for d in devices]
        if not devices or n_envs % len(devices):
            raise ValueError("the number of devices must divide n_envs")
        per = n_envs // len(devices)
        self.devices = devices
        self._build(env_id, [per] * len(devices), devices, seed, env_offset, auto_reset, terminal_obs, kw)
        if terminal_obs:
            for e in self.parts:
                e.keep_terminal_obs(True)
        self.gather = bool(gather)
        self.gather_device = torch.device("cuda", devices[0] if gather_device is None else int(gather_device))
        if self.gather:
            gd = self.gather_device
            self.obs = torch.zeros(self.n_envs, self.obs_dim, dtype=torch.float32, device=gd)
            self.rew = torch.zeros(self.n_envs, dtype=torch.float32, device=gd)
            self.done = torch.zeros(self.n_envs, dtype=torch.uint8, device=gd)
            self.info = torch.zeros(self.n_envs, dtype=torch.int32, device=gd)
        for d in set(devices):
            torch.cuda.synchronize(d)

    def _collect(self):
        if not self.gather:
            return ([e.obs for e in self.parts], [e.rew for e in self.parts], [e.done for e in self.parts], [e.info for e in self.parts])
        for e, sl in zip(self.parts, self.slices):      # peer-to-peer copies, ordered by torch between the two devices' current streams
            self.obs[sl].copy_(e.obs, non_blocking=True); self.rew[sl].copy_(e.rew, non_blocking=True)
            self.done[sl].copy_(e.done, non_blocking=True); self.info[sl].copy_(e.info, non_blocking=True)
        return self.obs, self.rew, self.done, self.info

    def reset(self):
        for e in self.parts:
            e.reset()
        return self._collect()[0]

    def step_async(self, i: int, actions: torch.Tensor) -> None:
        e = self.parts[i]
        if actions.device != e.device or actions.dtype != torch.float32 or not actions.is_contiguous():
            with torch.cuda.device(e.device):
                actions = actions.to(device=e.device, dtype=torch.float32, non_blocking=True).contiguous()
        self._before(i)
        actions.record_stream(self.streams[i])   # (see SubBatchedVecEnv.step_async)
        e.step(actions)

    def wait(self, i: int):
        self._after(i)
        e = self.parts[i]
        return e.obs, e.rew, e.done, e.info

    def step(self, actions):
        if isinstance(actions, torch.Tensor):
            if actions.shape != (self.n_envs, self.act_dim):
                raise ValueError(f"actions must be [{self.n_envs}, {self.act_dim}]")
            actions = [actions[sl] for sl in self.slices]
        if len(actions) != self.n_parts:
            raise ValueError("one action tensor per shard")
        for i, a in enumerate(actions):
            self.step_async(i, a)
        for i in range(self.n_parts):
            self._after(i)
        return self._collect()


def make_vec_env(env_id: str, n_envs: int, sub_batches: int = 1, devices: Optional[Sequence[int]] = None, **kw):
    """`VecEnv` / `SubBatchedVecEnv` / `ShardedVecEnv` by what is asked for."""
    if devices is not None and len(devices) > 1:
        if sub_batches != 1:
            raise ValueError("sub_batches and several devices do not combine here: shard first, sub-batch each shard's VecEnv")
        return ShardedVecEnv(env_id, n_envs, devices=devices, **kw)
    if devices is not None and len(devices) == 1:
        kw.setdefault("device", devices[0])
    if sub_batches > 1:
        return SubBatchedVecEnv(env_id, n_envs, sub_batches=sub_batches, **kw)
    return VecEnv(env_id, n_envs, **kw)
