"""ctypes binding of libmocca_hip.so (include/mocca.h).

There is no CPU fallback: if the HIP library is missing or does not load, every
entry point raises -- the product path is the hand-written gfx950 kernels only.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# MOCCA_LIB_PATH selects another build of the same HIP library (A/B kernel experiments); never a CPU fallback
LIB_PATH = os.environ.get("MOCCA_LIB_PATH") or os.path.join(HERE, "libmocca_hip.so")

ABI_VERSION = 7
PARAM_AUTO_RESET, PARAM_EVAL_MODE, PARAM_CURRICULUM, PARAM_RANDOM_POSE, PARAM_HOST_RETARGET, PARAM_SEED, PARAM_ENV_OFFSET, PARAM_APPLIED_GAIN, PARAM_RANDOM_REWARD = 0, 1, 2, 3, 4, 5, 6, 7, 8
PARAM_ISSUE_PRIORITY = 9   # timing only: row-count thresholds of the step kernel's issue priorities, t1 + 64 t2 + 4096 t3
PARAM_PERSIST_IMPULSES = 10  # keep the last substep's normal impulses in the state record although the blob does not warm-start (diagnostic)
PARAM_KERNEL_VARIANT = 11    # timing only: 1 forces the 48-row step-kernel instance for a blob that would run the compact one, 2 the 64-row one for any blob
PARAM_ORDER_EVERY = 12       # timing only: every K-th step re-sorts the launch order, heaviest envs (most constraint rows) first
PARAM_PACE_TICKS = 13        # timing only: pace priorities (target ticks per env.step of a wave) instead of the row-count priorities
DEBUG_WORDS = 20

# every symbol include/mocca.h declares: (name, restype, argtypes)
_vp, _i, _u64, _sz, _d = C.c_void_p, C.c_int, C.c_uint64, C.c_size_t, C.c_double
SYMBOLS = {
    "mocca_abi_version": (_i, []),
    "mocca_model_sizeof": (_sz, []),
    "mocca_create": (_i, [_vp, _sz, _i, _i, _i, C.POINTER(_vp)]),
    "mocca_destroy": (_i, [_vp]),
    "mocca_n_envs": (_i, [_vp]),
    "mocca_obs_dim": (_i, [_vp]),
    "mocca_act_dim": (_i, [_vp]),
    "mocca_state_dim": (_i, [_vp]),
    "mocca_reset": (_i, [_vp, _vp, _u64, _vp, _vp]),
    "mocca_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mocca_task_step": (_i, [_vp] * 10),
    "mocca_set_draw_tape": (_i, [_vp, _vp, _i]),
    "mocca_observe": (_i, [_vp, _vp, _vp]),
    "mocca_get_state": (_i, [_vp, _vp, _vp]),
    "mocca_set_state": (_i, [_vp, _vp, _vp]),
    "mocca_get_task": (_i, [_vp, _vp, _vp]),
    "mocca_set_task": (_i, [_vp, _vp, _vp]),
    "mocca_get_terrain": (_i, [_vp, _vp, _vp]),
    "mocca_set_terrain": (_i, [_vp, _vp, _vp]),
    "mocca_set_param": (_i, [_vp, _i, _d]),
    "mocca_set_param_v": (_i, [_vp, _i, _vp, _i, _vp]),
    "mocca_set_seed": (_i, [_vp, _u64]),
    "mocca_set_debug_buffer": (_i, [_vp, _vp]),
    "mocca_set_terminal_obs_buffer": (_i, [_vp, _vp]),
    "mocca_set_episode_stats": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _sz]),
    "mocca_episode_serial": (C.c_uint32, [_vp]),
    "mocca_set_trajectory": (_i, [_vp, _vp, _i, _d, _d]),
    "mocca_set_heightfield": (_i, [_vp, _vp, _i, _i, _d]),
    "mocca_is_diagnostic_build": (_i, []),
    "mocca_kernel_info": (_i, [_vp] + [C.POINTER(_i)] * 5),
    "mocca_last_error": (C.c_char_p, [_vp]),
}

_lib = None


class MoccaError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the HIP library; raises (never falls back) when it is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MoccaError(
            f"{LIB_PATH} not found: build it with `python -m mocca_envs_amd.build` "
            "(or __graft_entry__.build()); there is no CPU fallback")
    # PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64; libmocca_hip.so names the same sonames.  Whichever is loaded
    # first serves both, so torch goes first: one HIP runtime per process, and torch's device pointers / streams are valid in it.
    # (Loading this library first and torch afterwards left mocca_create without a visible device.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
        fn.restype, fn.argtypes = res, args
    if lib.mocca_abi_version() != ABI_VERSION:
        raise MoccaError("libmocca_hip.so ABI version mismatch; rebuild")
    if lib.mocca_is_diagnostic_build() and not os.environ.get("MOCCA_ALLOW_DIAGNOSTIC_BUILD"):
        # -DMOCCA_SKIP_* / MOCCA_DUMMY_VALU / MOCCA_STAMPS builds (tools/ablate*.sh, tools/stamps.py) skip phases of the
        # physics or add timing stores: never the product
        raise MoccaError(f"{LIB_PATH} is a diagnostic build (results wrong or slow by construction); "
                         "set MOCCA_ALLOW_DIAGNOSTIC_BUILD=1 to load it for profiling")
    _lib = lib
    return lib


def check(rc: int, handle=None) -> None:
    if rc != 0:
        msg = load().mocca_last_error(handle)
        raise MoccaError(f"libmocca_hip error {rc}: {msg.decode() if msg else ''}")
