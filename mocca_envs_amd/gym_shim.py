"""Minimal stand-in for the parts of classic `gym` the reference uses, for machines without gym.

Only used when `import gym` fails (this image has neither gym nor gymnasium).  Surface:
gym.Env, gym.spaces.Box, gym.utils.seeding.np_random, gym.envs.registration.{register, registry}, gym.make
with a TimeLimit wrapper honouring max_episode_steps -- what /root/reference/mocca_envs/__init__.py:5-116
and env_base.py:164-166 rely on.
"""
from __future__ import annotations

import hashlib
import importlib
import os
import struct
import types

import numpy as np


class Box:
    def __init__(self, low, high, dtype=np.float32, shape=None):
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        self.dtype = np.dtype(dtype)
        self.shape = self.low.shape if shape is None else tuple(shape)
        self._rng = np.random.RandomState()

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return self._rng.uniform(lo, hi).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def seed(self, seed=None):
        self._rng = np.random.RandomState(seed)
        return [seed]


class Env:
    metadata = {}
    reward_range = (-float("inf"), float("inf"))
    spec = None

    @property
    def unwrapped(self):
        return self

    def close(self):
        pass


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.observation_space = env.observation_space
        self.action_space = env.action_space
        self.metadata = env.metadata

    def __getattr__(self, name):
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def reset(self, **kw):
        return self.env.reset(**kw)

    def step(self, a):
        return self.env.step(a)

    def seed(self, s=None):
        return self.env.seed(s)

    def close(self):
        return self.env.close()


class TimeLimit(Wrapper):
    """done=True once max_episode_steps steps have elapsed (old-gym 4-tuple API)."""

    def __init__(self, env, max_episode_steps):
        super().__init__(env)
        self._max_episode_steps = max_episode_steps
        self._elapsed_steps = 0

    def reset(self, **kw):
        self._elapsed_steps = 0
        return self.env.reset(**kw)

    def step(self, a):
        obs, rew, done, info = self.env.step(a)
        self._elapsed_steps += 1
        if self._elapsed_steps >= self._max_episode_steps:
            info = dict(info)
            info["TimeLimit.truncated"] = not done
            done = True
        return obs, rew, done, info


class EnvSpec:
    def __init__(self, id, entry_point, max_episode_steps=None, kwargs=None):
        self.id, self.entry_point, self.max_episode_steps, self.kwargs = id, entry_point, max_episode_steps, kwargs or {}


class _Registry:
    def __init__(self):
        self.env_specs = {}


registry = _Registry()


def register(id, entry_point, max_episode_steps=None, kwargs=None, **_):
    registry.env_specs[id] = EnvSpec(id, entry_point, max_episode_steps, kwargs)


def make(id, **kwargs):
    if ":" in id:  # "package:EnvId" form (reference test_env.py:13-14)
        mod, id = id.split(":")
        importlib.import_module(mod)
    spec = registry.env_specs[id]
    ep = spec.entry_point
    if isinstance(ep, str):
        mod, cls = ep.split(":")
        ep = getattr(importlib.import_module(mod), cls)
    env = ep(**{**spec.kwargs, **kwargs})
    env.spec = spec
    if spec.max_episode_steps:
        env = TimeLimit(env, spec.max_episode_steps)
    return env


def _bigint_from_bytes(b: bytes) -> int:
    pad = 4 - len(b) % 4           # gym pads a whole extra word when the length already is a multiple of 4: harmless zeros
    b += b"\0" * pad
    words = struct.unpack("{}I".format(len(b) // 4), b)
    return sum(v << (32 * i) for i, v in enumerate(words))


def hash_seed(seed=None, max_bytes=8) -> int:
    """gym.utils.seeding.hash_seed (gym <= 0.21, the API generation the reference targets: 4-tuple step, `env.seed()`;
    the dependency is un-pinned in /root/reference/setup.py:11): SHA-512 of the decimal string, first 8 bytes, little endian."""
    if seed is None:
        seed = create_seed(max_bytes=max_bytes)
    return _bigint_from_bytes(hashlib.sha512(str(seed).encode("utf8")).digest()[:max_bytes])


def create_seed(a=None, max_bytes=8) -> int:
    if a is None:
        return _bigint_from_bytes(os.urandom(max_bytes))
    if isinstance(a, str):
        a = a.encode("utf8") + hashlib.sha512(a.encode("utf8")).digest()
        return _bigint_from_bytes(a[:max_bytes])
    if isinstance(a, (int, np.integer)):
        return int(a) % 2 ** (8 * max_bytes)
    raise TypeError("Invalid type for seed: {} ({})".format(type(a), a))


def _np_random(seed=None):
    """gym.utils.seeding.np_random of gym <= 0.21: the seed is HASHED before it reaches numpy's Mersenne Twister
    (RandomState.seed of the 32-bit words of hash_seed(seed)), so `env.seed(5)` here starts the stream the reference's
    `env.seed(5)` starts under that gym.  Returns (RandomState, seed) like gym."""
    if seed is not None and not (isinstance(seed, (int, np.integer)) and 0 <= seed):
        raise ValueError("Seed must be a non-negative integer or omitted, not {}".format(seed))
    seed = create_seed(seed)
    big = hash_seed(seed)
    words = []
    while big > 0:
        big, mod = divmod(big, 2 ** 32)
        words.append(mod)
    rng = np.random.RandomState()
    rng.seed(words or [0])
    return rng, seed


def as_module() -> types.ModuleType:
    """Assemble a module object shaped like `gym`."""
    gym = types.ModuleType("gym")
    spaces = types.ModuleType("gym.spaces")
    spaces.Box = Box
    utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")
    seeding.np_random = _np_random
    utils.seeding = seeding
    envs = types.ModuleType("gym.envs")
    registration = types.ModuleType("gym.envs.registration")
    registration.registry, registration.register, registration.EnvSpec = registry, register, EnvSpec
    envs.registration = registration
    wrappers = types.ModuleType("gym.wrappers")
    wrappers.TimeLimit = TimeLimit
    gym.Env, gym.Wrapper, gym.spaces, gym.utils, gym.envs, gym.wrappers = Env, Wrapper, spaces, utils, envs, wrappers
    gym.make, gym.register = make, register
    gym.__shim__ = True
    return gym
