"""ctypes binding of the CPU oracle (oracle/mocca_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke(); never by the product package ``mocca_envs_amd``.
Physics parity with PyBullet is *unpinned* (see the header of mocca_oracle.c).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
TASK_WORDS = 40
TERRAIN_WORDS = 20 * 6 + 4   # 20 x 6 step table + MOCCA_MAX_PLANKS live-plank rows


def build(force: bool = False) -> None:
    """Compile liboracle_f32.so / liboracle_f64.so with gcc (oracle/Makefile)."""
    libs = [os.path.join(_HERE, f"liboracle_{p}.so") for p in ("f32", "f64")]
    src = os.path.join(_HERE, "mocca_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "mocca_model.h")
    stale = force or any(
        (not os.path.exists(l)) or os.path.getmtime(l) < max(os.path.getmtime(src), os.path.getmtime(hdr))
        for l in libs
    )
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def _load(precision: str) -> C.CDLL:
    path = os.path.join(_HERE, f"liboracle_{precision}.so")
    if os.environ.get("MOCCA_ORACLE_SANITIZED"):     # tools/sanitize_oracle.sh: the ASan / UBSan build of the same source
        path = os.path.join(_HERE, "_san", f"liboracle_{precision}.so")
    elif not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    lib.orc_create.restype = C.c_void_p
    lib.orc_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.orc_destroy.argtypes = [C.c_void_p]
    for name in ("orc_obs_dim", "orc_state_dim", "orc_last_rows", "orc_act_dim"):
        getattr(lib, name).argtypes = [C.c_void_p]
        getattr(lib, name).restype = C.c_int
    lib.orc_set_param.argtypes = [C.c_void_p, C.c_int, C.c_double]
    lib.orc_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.orc_step.argtypes = [C.c_void_p] * 6
    for name in ("orc_get_state", "orc_set_state", "orc_get_task", "orc_set_task",
                 "orc_get_terrain", "orc_set_terrain"):
        getattr(lib, name).argtypes = [C.c_void_p, C.c_void_p]
    lib.orc_rollout.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.orc_task_step.argtypes = [C.c_void_p] * 8
    lib.orc_task_step_feet.argtypes = [C.c_void_p] * 9
    lib.orc_set_tape.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.orc_physics_substeps.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    lib.orc_forward_dynamics.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    lib.orc_minv_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_link_frames.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.orc_link_velocities.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.orc_last_contacts.argtypes = [C.c_void_p, C.c_void_p]
    lib.orc_last_contacts.restype = C.c_int
    lib.orc_get_debug.argtypes = [C.c_void_p, C.c_void_p]
    lib.orc_clear_debug.argtypes = [C.c_void_p]
    lib.orc_last_lambda.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_last_lambda.restype = C.c_int
    lib.orc_set_trajectory.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double]
    lib.orc_set_trajectory.restype = C.c_int
    lib.orc_set_heightfield.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double]
    lib.orc_set_heightfield.restype = C.c_int
    lib.orc_heightfield_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p]
    lib.orc_heightfield_probe.restype = C.c_double
    lib.orc_height_at.argtypes = [C.c_void_p, C.c_double, C.c_double]
    lib.orc_height_at.restype = C.c_double
    return lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


PARAM_AUTO_RESET, PARAM_EVAL_MODE, PARAM_CURRICULUM, PARAM_RANDOM_POSE, PARAM_RANDOM_REWARD = 0, 1, 2, 3, 8


class Oracle:
    """N independent environments stepped serially on the host."""

    def __init__(self, model_blob: bytes, task_id: int, n_envs: int, precision: str = "f32"):
        self.lib = _load(precision)
        self.precision = precision
        self._blob = C.create_string_buffer(model_blob, len(model_blob))
        self.h = self.lib.orc_create(self._blob, len(model_blob), task_id, n_envs)
        if not self.h:
            raise RuntimeError("orc_create rejected the model blob")
        self.n_envs = n_envs
        self.obs_dim = self.lib.orc_obs_dim(self.h)
        self.state_dim = self.lib.orc_state_dim(self.h)
        self.act_dim = self.lib.orc_act_dim(self.h)
        from mocca_envs_amd.model import MoccaModel   # layout only (the oracle may import the product, never the reverse)
        self.n_feet = MoccaModel.from_bytes(model_blob).n_feet

    def close(self):
        if self.h:
            self.lib.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_param(self, pid: int, v: float):
        self.lib.orc_set_param(self.h, pid, float(v))

    def set_trajectory(self, table: np.ndarray, max_time: float, control_step: float):
        """The reference motion of the Cassie mocap / phase envs: [n_frames][32] float32 (copied)."""
        t = np.ascontiguousarray(table, np.float32)
        assert t.ndim == 2 and t.shape[1] == 32
        if self.lib.orc_set_trajectory(self.h, _p(t), t.shape[0], float(max_time), float(control_step)) != 0:
            raise ValueError("orc_set_trajectory rejected the table")

    def set_heightfield(self, data: np.ndarray, scale: float):
        """The planner envs' terrain: data[rows][cols] heights (x along the columns), `scale` grid points per metre (copied)."""
        d = np.ascontiguousarray(data, np.float32)
        assert d.ndim == 2
        if self.lib.orc_set_heightfield(self.h, _p(d), d.shape[0], d.shape[1], float(scale)) != 0:
            raise ValueError("orc_set_heightfield rejected the grid")

    def heightfield_probe(self, centre, radius: float, margin: float = 0.0):
        """(gap, normal) of a sphere against the attached height field; the search window follows radius + margin."""
        c, n = np.ascontiguousarray(centre, np.float64), np.zeros(3)
        return self.lib.orc_heightfield_probe(self.h, _p(c), float(radius), float(margin), _p(n)), n

    def height_at(self, x: float, y: float) -> float:
        return self.lib.orc_height_at(self.h, float(x), float(y))

    def reset(self, seed: int = 0, mask: Optional[np.ndarray] = None) -> np.ndarray:
        obs = np.zeros((self.n_envs, self.obs_dim), np.float32)
        if mask is not None:
            mask = np.ascontiguousarray(mask, np.uint8)
        self.lib.orc_reset(self.h, _p(mask) if mask is not None else None, seed, _p(obs))
        return obs

    def step(self, act: np.ndarray):
        act = np.ascontiguousarray(act, np.float32)
        obs = np.zeros((self.n_envs, self.obs_dim), np.float32)
        rew = np.zeros(self.n_envs, np.float32)
        done = np.zeros(self.n_envs, np.uint8)
        info = np.zeros(self.n_envs, np.int32)
        self.lib.orc_step(self.h, _p(act), _p(obs), _p(rew), _p(done), _p(info))
        return obs, rew, done, info

    def rollout(self, tape: np.ndarray, steps: int) -> None:
        """`steps` env.steps inside one C call (the interpreter lock is released for all of it), actions from a looped tape
        [tape_len][N][act_dim] float32; observations / rewards are discarded: bench.py's all-cores CPU baseline."""
        tape = np.ascontiguousarray(tape, np.float32).reshape(-1, self.n_envs, self.act_dim)
        self.lib.orc_rollout(self.h, _p(tape), tape.shape[0], int(steps))

    def task_step(self, act: np.ndarray, touch: np.ndarray, target: Optional[np.ndarray] = None,
                  body_touch: Optional[np.ndarray] = None):
        """Task logic only on the injected post-physics state (golden tests).  touch / target: [N][n_feet] foot contact
        query results; body_touch [N]: a non-foot link touches the ground (LaikagoCustomEnv)."""
        act = np.ascontiguousarray(act, np.float32)
        nf = self.n_feet
        touch = np.ascontiguousarray(touch, np.int32).reshape(self.n_envs, nf)
        target = np.zeros_like(touch) if target is None else np.ascontiguousarray(target, np.int32).reshape(self.n_envs, nf)
        body = None if body_touch is None else np.ascontiguousarray(body_touch, np.int32).reshape(self.n_envs)
        obs = np.zeros((self.n_envs, self.obs_dim), np.float32)
        rew = np.zeros(self.n_envs, np.float32)
        done = np.zeros(self.n_envs, np.uint8)
        info = np.zeros(self.n_envs, np.int32)
        self.lib.orc_task_step_feet(self.h, _p(act), _p(touch), _p(target), _p(body) if body is not None else None,
                                    _p(obs), _p(rew), _p(done), _p(info))
        return obs, rew, done, info

    def set_tape(self, tape: Optional[np.ndarray]):
        """Feed the next random draws from `tape` (uniforms in [0,1)) instead of Philox."""
        if tape is None:
            self._tape = None
            self.lib.orc_set_tape(self.h, None, 0)
        else:
            self._tape = np.ascontiguousarray(tape, np.float64)
            self.lib.orc_set_tape(self.h, _p(self._tape), len(self._tape))

    def get_state(self) -> np.ndarray:
        st = np.zeros((self.n_envs, self.state_dim), np.float64)
        self.lib.orc_get_state(self.h, _p(st))
        return st

    def set_state(self, st: np.ndarray):
        st = np.ascontiguousarray(st, np.float64).reshape(self.n_envs, self.state_dim)
        self.lib.orc_set_state(self.h, _p(st))

    def get_task(self) -> np.ndarray:
        t = np.zeros((self.n_envs, TASK_WORDS), np.float64)
        self.lib.orc_get_task(self.h, _p(t))
        return t

    def set_task(self, t: np.ndarray):
        t = np.ascontiguousarray(t, np.float64).reshape(self.n_envs, TASK_WORDS)
        self.lib.orc_set_task(self.h, _p(t))

    def get_terrain(self) -> np.ndarray:
        t = np.zeros((self.n_envs, TERRAIN_WORDS), np.float64)
        self.lib.orc_get_terrain(self.h, _p(t))
        return t

    def set_terrain(self, t: np.ndarray):
        t = np.ascontiguousarray(t, np.float64).reshape(self.n_envs, TERRAIN_WORDS)
        self.lib.orc_set_terrain(self.h, _p(t))

    # ---- physics probes -------------------------------------------------
    def physics_substeps(self, env: int, tau: np.ndarray, n: int):
        tau = np.ascontiguousarray(tau, np.float64)
        self.lib.orc_physics_substeps(self.h, env, _p(tau), n)

    def forward_dynamics(self, env: int, tau: np.ndarray, with_bias: bool = True) -> np.ndarray:
        tau = np.ascontiguousarray(tau, np.float64)
        out = np.zeros(6 + len(tau), np.float64)
        self.lib.orc_forward_dynamics(self.h, env, _p(tau), int(with_bias), _p(out))
        return out

    def minv_apply(self, f: np.ndarray) -> np.ndarray:
        f = np.ascontiguousarray(f, np.float64)
        out = np.zeros_like(f)
        self.lib.orc_minv_apply(self.h, _p(f), _p(out))
        return out

    def link_frames(self, env: int, n_bodies: int) -> np.ndarray:
        out = np.zeros((n_bodies, 15), np.float64)
        self.lib.orc_link_frames(self.h, env, _p(out))
        return out

    def link_velocities(self, env: int, n_bodies: int) -> np.ndarray:
        out = np.zeros((n_bodies, 6), np.float64)
        self.lib.orc_link_velocities(self.h, env, _p(out))
        return out

    def last_contacts(self) -> np.ndarray:
        out = np.zeros((24, 11), np.float64)
        n = self.lib.orc_last_contacts(self.h, _p(out))
        return out[:n]

    def last_rows(self) -> int:
        return self.lib.orc_last_rows(self.h)

    def last_lambda(self):
        """(impulses, kinds) of the rows of the last substep solved (kind: 0 limit, 1 normal, 2 friction, 3 closure)."""
        lam, kind = np.zeros(64, np.float64), np.zeros(64, np.int32)
        n = self.lib.orc_last_lambda(self.h, _p(lam), _p(kind))
        return lam[:n], kind[:n]

    def get_debug(self) -> np.ndarray:
        """Debug record of every env, [N][20] int32 (include/mocca.h MOCCA_DBG_*): words 0..11 the active set of the last substep
        (rows, contacts, slot / limit masks, PGS clamp mask and signature), words 12..15 cumulative cap pressure."""
        out = np.zeros((self.n_envs, 20), np.int32)
        self.lib.orc_get_debug(self.h, _p(out))
        return out

    def clear_debug(self):
        self.lib.orc_clear_debug(self.h)
