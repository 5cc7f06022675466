"""mocca_envs_amd: MI355X-native vectorised locomotion stepper behind the mocca_envs gym surface.

Importing the package registers the reference's env ids (mocca_envs/__init__.py:18-116) that have a GPU
stepper.  The physics runs only in hand-written HIP kernels (libmocca_hip.so); there is no CPU fallback.
"""
from __future__ import annotations

import os

current_dir = os.path.dirname(os.path.realpath(__file__))

REGISTERED = {
    # id -> (entry point, kwargs); max_episode_steps = 1000 for all (reference __init__.py:55,61)
    "Walker3DCustomEnv-v0": ("mocca_envs_amd.envs:Walker3DCustomEnv", {}),
    "Walker3DStepperEnv-v0": ("mocca_envs_amd.envs:Walker3DStepperEnv", {}),
    "CassieEnv-v0": ("mocca_envs_amd.envs:CassieEnv", {}),
    "Child3DCustomEnv-v0": ("mocca_envs_amd.envs:Child3DCustomEnv", {}),
    "MikeStepperEnv-v0": ("mocca_envs_amd.envs:MikeStepperEnv", {}),
    "Walker2DCustomEnv-v0": ("mocca_envs_amd.envs:Walker2DCustomEnv", {}),
    "Crab2DCustomEnv-v0": ("mocca_envs_amd.envs:Crab2DCustomEnv", {}),
    "LaikagoCustomEnv-v0": ("mocca_envs_amd.envs:LaikagoCustomEnv", {}),
    "LaikagoStepperEnv-v0": ("mocca_envs_amd.envs:LaikagoStepperEnv", {}),
    "Cassie2DEnv-v0": ("mocca_envs_amd.envs:CassieEnv", {"planar": True}),   # reference __init__.py:24-29
    "CassiePhaseMocca2DEnv-v0": ("mocca_envs_amd.envs:CassiePhaseMoccaEnv", {"planar": True}),     # :31-36
    "CassiePhaseMirror2DEnv-v0": ("mocca_envs_amd.envs:CassiePhaseMirrorEnv", {"planar": True}),   # :38-43
    "Walker3DPlannerEnv-v0": ("mocca_envs_amd.envs:Walker3DPlannerEnv", {}),                       # :82-86
    "MikePlannerEnv-v0": ("mocca_envs_amd.envs:MikePlannerEnv", {}),                               # :88-92
}


def register(id, **kvargs):
    """Idempotent registration, same contract as the reference's (mocca_envs/__init__.py:5-9)."""
    try:
        import gym  # type: ignore
    except Exception:
        from . import gym_shim as gym  # noqa: N813
        if id in gym.registry.env_specs:
            return None
        return gym.register(id, **kvargs)
    reg = gym.envs.registration.registry
    specs = getattr(reg, "env_specs", reg)
    if id in specs:
        return None
    return gym.envs.registration.register(id, **kvargs)


for _id, (_ep, _kw) in REGISTERED.items():
    register(id=_id, entry_point=_ep, max_episode_steps=1000, kwargs=_kw)


def make(id, **kwargs):
    """gym.make for machines without gym (uses the shim's registry)."""
    try:
        import gym  # type: ignore
        return gym.make(id, **kwargs)
    except ImportError:
        from . import gym_shim
        return gym_shim.make(id, **kwargs)
