"""Batched environments on one MI355X: the trainer-facing surface over libmocca_hip.so.

`VecEnv` keeps observations / rewards / done flags in PyTorch-ROCm tensors (device memory
and the current HIP stream are the only things torch is used for) and advances all N
environments with one kernel launch per `step()`.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np
import torch

from . import lib as _lib
from . import model as M

TASKS = {
    "Walker3DCustomEnv-v0": M.TASK_WALKER3D_CUSTOM,
    "Walker3DStepperEnv-v0": M.TASK_WALKER3D_STEPPER,
    "CassieEnv-v0": M.TASK_CASSIE,
    # same tree as Walker3D (child3d.xml / mike.xml): new model blobs on the Walker3D kernels
    "Child3DCustomEnv-v0": M.TASK_WALKER3D_CUSTOM,
    "MikeStepperEnv-v0": M.TASK_WALKER3D_STEPPER,
    # planar robots (walker2d.xml / crab2d.xml): own topologies, Custom task with the quirks of env_locomotion.py:285-314
    "Walker2DCustomEnv-v0": M.TASK_WALKER3D_CUSTOM,
    "Crab2DCustomEnv-v0": M.TASK_WALKER3D_CUSTOM,
    # quadruped (laikago_toes_limits.urdf): own topology, four feet, Custom task ending on body contact
    "LaikagoCustomEnv-v0": M.TASK_WALKER3D_CUSTOM,
    # the quadruped on four live planks (env_locomotion.py:893-979): Laikago topology x Stepper task
    "LaikagoStepperEnv-v0": M.TASK_WALKER3D_STEPPER,
    # CassieEnv(planar=True), __init__.py:24-29: the base held in the x-z plane by three bilateral rows
    "Cassie2DEnv-v0": M.TASK_CASSIE,
    # the planar mocap / phase envs (__init__.py:31-43, env_cassie.py:481-660): Cassie topology, mocap targets and reward
    "CassiePhaseMocca2DEnv-v0": M.TASK_CASSIE,
    "CassiePhaseMirror2DEnv-v0": M.TASK_CASSIE,
    # the walkers on the height field (env_locomotion.py:982-1133): Walker3D / Mike tree x Planner task; the kernel takes the 21 joint
    # actions of the (external) base controller, the planner's 15 numbers are the caller's business
    "Walker3DPlannerEnv-v0": M.TASK_WALKER3D_PLANNER,
    "MikePlannerEnv-v0": M.TASK_WALKER3D_PLANNER,
}
# class attributes of the reference envs that are device parameters here
_DEFAULT_PARAMS = {"LaikagoCustomEnv-v0": {_lib.PARAM_RANDOM_POSE: 0},    # robot_random_start = False, env_locomotion.py:863
                   "LaikagoStepperEnv-v0": {_lib.PARAM_RANDOM_POSE: 0}}   # :899

# Since round 4 the step kernel sets its issue priorities from each wave's PACE (PARAM_PACE_TICKS, default: self-calibrating -- 2.6 % to 8.7 %
# faster than the tables below on every env id, profiles/archive/r04_pace_envs.jsonl); the row-count thresholds only serve a handle's first launch
# (no pace sample yet) and handles that switch the pace off.
# Issue-priority thresholds of the step kernel (PARAM_ISSUE_PRIORITY; timing only, results do not depend on them): constraint-row counts
# above which a wave runs at priority 1 / 2 / 3.  The best set follows the batch's row distribution -- measured per env id with
# a threshold sweep on one MI355X (profiles/archive/r03_prio_sweep_v13.txt: the blob v13 physics hold 5.7 rows per substep on the flat-ground walker
# instead of 12.7, and the thresholds of round 2 had stopped selecting anything: -5.5 % on the launch for re-reading them off the new
# distribution); ids not listed keep the library's default (4, 7, 12: the flat-ground walker).  The stepping-stone walkers carry more rows as
# the curriculum rises: their thresholds grow with it (x 1.7 at curriculum 9).
_ISSUE_PRIORITY = {"LaikagoCustomEnv-v0": (2, 4, 7), "LaikagoStepperEnv-v0": (2, 4, 7), "Child3DCustomEnv-v0": (6, 11, 18),
                   "CassieEnv-v0": (18, 23, 27), "Cassie2DEnv-v0": (24, 29, 33), "CassiePhaseMocca2DEnv-v0": (24, 29, 33),
                   "CassiePhaseMirror2DEnv-v0": (24, 29, 33), "Walker3DPlannerEnv-v0": (10, 16, 24), "MikePlannerEnv-v0": (10, 16, 24),
                   "Walker2DCustomEnv-v0": (6, 9, 14), "Crab2DCustomEnv-v0": (8, 12, 18)}
_ISSUE_PRIORITY_CURRICULUM = {"Walker3DStepperEnv-v0": (7, 11, 17), "MikeStepperEnv-v0": (7, 11, 17)}
_ISSUE_PRIORITY_CURRICULUM_GAIN = 0.7   # thresholds x (1 + gain * curriculum / 9)


def _pack_prio(t):
    return int(t[0]) + 64 * int(t[1]) + 4096 * int(t[2])


_MODELS = {
    "Walker3DCustomEnv-v0": lambda **kw: M.compile_walker3d(M.TASK_WALKER3D_CUSTOM, **kw),
    "Walker3DStepperEnv-v0": lambda **kw: M.compile_walker3d(M.TASK_WALKER3D_STEPPER, **kw),   # kw: plank_class = LargePlank | Plank | Pillar
    "CassieEnv-v0": lambda **kw: M.compile_cassie(**kw),
    "Cassie2DEnv-v0": lambda **kw: M.compile_cassie(planar=True, **kw),
    "CassiePhaseMocca2DEnv-v0": lambda **kw: M.compile_cassie(planar=kw.pop("planar", True), mode=M.CASSIE_PHASE_MOCCA, **kw),
    "CassiePhaseMirror2DEnv-v0": lambda **kw: M.compile_cassie(planar=kw.pop("planar", True), mode=M.CASSIE_PHASE_MIRROR, **kw),
    "LaikagoStepperEnv-v0": lambda **kw: M.compile_laikago(stepper=True, **kw),
    "Child3DCustomEnv-v0": M.compile_child3d,
    "MikeStepperEnv-v0": M.compile_mike,
    "Walker2DCustomEnv-v0": M.compile_walker2d,
    "Crab2DCustomEnv-v0": M.compile_crab2d,
    "LaikagoCustomEnv-v0": M.compile_laikago,
    "Walker3DPlannerEnv-v0": lambda **kw: M.compile_walker3d(M.TASK_WALKER3D_PLANNER, **kw),
    "MikePlannerEnv-v0": lambda **kw: M.compile_mike(planner=True, **kw),
}
_DEFAULT_ENV_OF_TASK = {M.TASK_WALKER3D_CUSTOM: "Walker3DCustomEnv-v0", M.TASK_WALKER3D_STEPPER: "Walker3DStepperEnv-v0",
                        M.TASK_CASSIE: "CassieEnv-v0"}


def compile_model_for(env_or_task, **kw) -> M.MoccaModel:
    """Model blob of a registered env id (or, for the three task ids, of the task's original robot)."""
    env_id = env_or_task if isinstance(env_or_task, str) else _DEFAULT_ENV_OF_TASK[int(env_or_task)]
    return _MODELS[env_id](**kw)


class VecEnv:
    """N independent copies of a registered env id, stepped together.

    step(actions[N, 21]) -> (obs[N, obs_dim], reward[N], done[N] uint8, info[N] int32), all on
    `device`.  done bit0 = terminated (reference `self.done`), bit1 = TimeLimit (1000 steps,
    /root/reference/mocca_envs/__init__.py:55).  With auto_reset=True a finished env is reset
    inside the same launch and `obs` is the first observation of its next episode; with
    terminal_obs=True `self.terminal_obs[i]` then holds the observation of env i's FINAL state
    (what the reference's step() returns together with done, env_locomotion.py:128-141 -- the
    value a PPO-style trainer bootstraps from on a TimeLimit truncation); rows of envs that did
    not finish in this step keep their old content.
    """

    def __init__(self, env_id: str = "Walker3DCustomEnv-v0", n_envs: int = 1, device: Optional[int] = None,
                 auto_reset: bool = True, seed: int = 0, model_blob: Optional[bytes] = None, env_offset: int = 0,
                 terminal_obs: bool = False, max_rows: Optional[int] = None, max_contacts: Optional[int] = None, **model_kw):
        if env_id not in TASKS:
            raise KeyError(f"{env_id!r} has no GPU stepper yet; available: {sorted(TASKS)}")
        if not torch.cuda.is_available():
            raise _lib.MoccaError("no HIP device visible: the stepper only runs on the GPU (no CPU fallback)")
        self.lib = _lib.load()
        self.stream = None    # None: every call goes to torch's CURRENT stream; set to a torch.cuda.Stream to pin this handle's work to it
        self.env_id, self.task_id, self.n_envs = env_id, TASKS[env_id], int(n_envs)
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        if model_blob is None:
            self.model = compile_model_for(env_id, **model_kw)
            model_blob = self.model.to_bytes()
        else:
            self.model = M.MoccaModel.from_bytes(model_blob)
        if max_rows is not None or max_contacts is not None:
            # Solver caps of this batch (MoccaModel.max_rows / max_contacts; Bullet has neither).  A tree without loop closures whose caps are
            # <= 32 rows / <= 10 contacts runs the COMPACT instance of the step kernel (include/mocca.h MOCCA_PARAM_KERNEL_VARIANT: less LDS
            # per env, more resident waves -- what batches beyond one residency round, > 4096 envs per GPU, want); an env that asks for
            # more in a substep keeps its deepest contacts, exactly as under the default 48 / 12 caps (how often: tools/cap_pressure.py).
            # Caps beyond 48 rows / 12 contacts (up to 64 / 20: every lane of the wave a row) run the ACCURACY instance (mocca_r64.hip: 17 KB of LDS
            # per env, two waves per SIMD) -- Bullet has no cap; `max_rows=64` alone asks for its 20 contacts too.
            if max_rows is not None:
                self.model.max_rows = int(max_rows)
            if max_contacts is not None:
                self.model.max_contacts = int(max_contacts)
            elif int(self.model.max_rows) > 48:
                self.model.max_contacts = min(20, int(self.model.max_rows) // 3)
            else:
                self.model.max_contacts = min(int(self.model.max_contacts), int(self.model.max_rows) // 3)
            model_blob = self.model.to_bytes()
        if self.lib.mocca_model_sizeof() != len(model_blob):
            raise _lib.MoccaError("MoccaModel layout mismatch between model.py and libmocca_hip.so")
        self._blob = C.create_string_buffer(model_blob, len(model_blob))
        h = C.c_void_p()
        _lib.check(self.lib.mocca_create(self._blob, len(model_blob), self.task_id, self.n_envs, self.device_index, C.byref(h)))
        self.h = h
        self.obs_dim = self.lib.mocca_obs_dim(h)
        self.act_dim = self.lib.mocca_act_dim(h)
        self.state_dim = self.lib.mocca_state_dim(h)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.obs = torch.zeros(self.n_envs, self.obs_dim, **f32)
        self.rew = torch.zeros(self.n_envs, **f32)
        self.done = torch.zeros(self.n_envs, dtype=torch.uint8, device=self.device)
        self.info = torch.zeros(self.n_envs, dtype=torch.int32, device=self.device)
        self.seed_value = int(seed)
        self.set_param(_lib.PARAM_AUTO_RESET, 1 if auto_reset else 0)
        self.env_offset = int(env_offset)
        self.set_param(_lib.PARAM_ENV_OFFSET, self.env_offset)
        for pid, val in _DEFAULT_PARAMS.get(env_id, {}).items():
            self.set_param(pid, val)
        if env_id in _ISSUE_PRIORITY:
            self.set_param(_lib.PARAM_ISSUE_PRIORITY, _pack_prio(_ISSUE_PRIORITY[env_id]))
        self.terminal_obs = None
        if terminal_obs:
            self.keep_terminal_obs(True)
        self.ep = None        # episode_stats(): Monitor / TimeLimitMask inside the launch
        self.height_field = None
        if self.task_id == M.TASK_WALKER3D_PLANNER:
            from .terrain import load_height_field   # self.terrain.reload(data="height_field_map_0.npy"), env_locomotion.py:1015-1021
            self.set_heightfield(*load_height_field())
        self.trajectory = None
        if self.task_id == M.TASK_CASSIE and self.model.cassie_mode != M.CASSIE_PLAIN:
            from .trajectory import CassieTrajectory   # self.traj = CassieTrajectory(), env_cassie.py:576
            self.set_trajectory(CassieTrajectory())

    # ------------------------------------------------------------------
    def _stream(self) -> C.c_void_p:
        if self.stream is not None:      # a handle bound to a stream of its own (sub-batches that step independently: multi.SubBatchedVecEnv)
            return C.c_void_p(self.stream.cuda_stream)
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # A handle pinned to a stream of its own (`self.stream`) runs its C calls there, while the torch operations around them (uploads,
    # .contiguous(), the caller's reads of what a getter returns) run on torch's CURRENT stream.  Every method except step() orders the
    # two, stream against stream, never through the host: _in() before a call that consumes caller tensors, _out() after a call whose
    # result the caller will read.  step() is the hot path and stays unordered on a pinned stream: the caller orders it (SubBatchedVecEnv's
    # step_async / wait do), or passes ready, contiguous float32 device tensors and reads the outputs after a synchronize.
    def _in(self):
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream(self.device))

    def _out(self):
        if self.stream is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def _sync(self):
        (self.stream if self.stream is not None else torch.cuda.current_stream(self.device)).synchronize()

    def close(self):
        if getattr(self, "h", None):
            torch.cuda.synchronize(self.device)
            self.lib.mocca_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_param(self, pid: int, value: float):
        _lib.check(self.lib.mocca_set_param(self.h, pid, float(value)), self.h)
        if pid == _lib.PARAM_CURRICULUM and self.env_id in _ISSUE_PRIORITY_CURRICULUM:   # timing only, see _ISSUE_PRIORITY
            k = 1.0 + _ISSUE_PRIORITY_CURRICULUM_GAIN * min(9.0, max(0.0, float(value))) / 9.0
            base = _ISSUE_PRIORITY_CURRICULUM[self.env_id]
            _lib.check(self.lib.mocca_set_param(self.h, _lib.PARAM_ISSUE_PRIORITY, float(_pack_prio([min(63, round(k * t)) for t in base]))), self.h)

    def set_trajectory(self, traj, control_step: float = 0.03):
        """Attach the reference motion of the Cassie mocap / phase envs (include/mocca.h mocca_set_trajectory); the table is
        copied into the handle.  control_step = CassieEnv.control_step (env_cassie.py:287)."""
        tab = np.ascontiguousarray(traj.table(), np.float32)
        _lib.check(self.lib.mocca_set_trajectory(self.h, tab.ctypes.data_as(C.c_void_p), tab.shape[0], float(traj.max_time()),
                                                 float(control_step)), self.h)
        self.trajectory = traj

    def set_heightfield(self, heights, scale: float):
        """Attach the terrain of the planner envs (include/mocca.h mocca_set_heightfield): heights[rows][cols], x along the columns,
        `scale` grid points per metre; copied into the handle, one grid for all envs."""
        hf = np.ascontiguousarray(heights, np.float32)
        if hf.ndim != 2:
            raise ValueError("heights must be a [rows][cols] grid")
        _lib.check(self.lib.mocca_set_heightfield(self.h, hf.ctypes.data_as(C.c_void_p), hf.shape[0], hf.shape[1], float(scale)), self.h)
        self.height_field = (hf, float(scale))

    # ---- the reference's env-level setters, batched (env_base.py:103-118, env_locomotion.py:76-77,224-282) ----
    def set_env_params(self, params_dict):
        """`set_env_params({"curriculum": k})`: one value for all envs or one per env (takes effect at each env's next reset; the
        terminal height follows it at once, env_locomotion.py:628).  Unknown keys are ignored, like the reference's hasattr test."""
        for k, v in params_dict.items():
            if k == "curriculum":
                if np.ndim(v) == 0:
                    self.set_param(_lib.PARAM_CURRICULUM, float(v))
                else:
                    self.set_param_v(_lib.PARAM_CURRICULUM, v)

    def set_robot_params(self, params_dict):
        """`set_robot_params({"applied_gain": g})` (env_base.py:108-115): scalar or one value per env; acts on the next apply_action."""
        if "applied_gain" in params_dict:
            g = params_dict["applied_gain"]
            if np.ndim(g) == 0:
                self.set_param(_lib.PARAM_APPLIED_GAIN, float(g))
            else:
                self.set_param_v(_lib.PARAM_APPLIED_GAIN, g)

    def evaluation_mode(self, on=True):
        """Walker3DCustomEnv.evaluation_mode() (env_locomotion.py:76-77): fixed target 4 m ahead; scalar or one flag per env."""
        if np.ndim(on) == 0:
            self.set_param(_lib.PARAM_EVAL_MODE, 1.0 if on else 0.0)
        else:
            self.set_param_v(_lib.PARAM_EVAL_MODE, on)

    def get_mirror_indices(self):
        """The six index lists SymmetricRL consumes (env_locomotion.py:224-282 / :761-840); see symmetry.MirrorTransform."""
        from . import host_logic as H
        if self.task_id == M.TASK_CASSIE:
            # the Cassie mocap / phase envs publish a DICT of index lists as a class attribute (env_cassie.py:536-571, :627-629), not the walkers'
            # six-tuple; CassieEnv itself publishes nothing
            if self.model.cassie_mode == M.CASSIE_PLAIN:
                raise NotImplementedError("CassieEnv has no mirror indices in the reference (env_cassie.py:284-479)")
            import copy
            from .envs import CassieMoccaEnv
            mi = copy.deepcopy(CassieMoccaEnv.mirror_indices)
            mi["left_obs_inds"] += [40]
            mi["right_obs_inds"] += [41]
            return mi
        return H.mirror_indices(self.model, stepper=self.task_id == M.TASK_WALKER3D_STEPPER)

    def set_param_v(self, pid: int, values, broadcast: bool = False):
        """Per-env curriculum / eval_mode / applied_gain (include/mocca.h mocca_set_param_v); values: [N] (or [1] with broadcast)."""
        v = torch.as_tensor(values, dtype=torch.float32).to(self.device).contiguous().reshape(-1)
        if v.numel() != (1 if broadcast else self.n_envs):
            raise ValueError("values must hold one float per env (or one float with broadcast=True)")
        self._in()
        _lib.check(self.lib.mocca_set_param_v(self.h, pid, C.c_void_p(v.data_ptr()), int(broadcast), self._stream()), self.h)
        self._out()       # `v` may be freed (and its memory reused on the current stream) as soon as this returns

    def seed(self, seed: int, rewind: bool = True):
        """Philox key of the in-kernel draws (gym's env.seed(s), env_base.py:164-166).  With rewind (default) the per-env episode
        counters go back to "before the first episode", so that seed(s) followed by reset() replays exactly what a fresh VecEnv created
        with seed=s produces -- the gym contract.  rewind=False only re-keys the stream: the envs continue with new random numbers."""
        self.seed_value = int(seed) & 0xFFFFFFFFFFFFFFFF
        _lib.check(self.lib.mocca_set_seed(self.h, self.seed_value), self.h)
        if rewind:
            tk = self.get_task()
            tk[:, 9] = -1      # episode: the next reset is episode 0 (draws are keyed by (seed, global env id, episode, draw))
            tk[:, 10] = 0      # draw counter
            self.set_task(tk)
        return [seed]

    def set_draw_tape(self, tape) -> None:
        """Uniforms that replace the Philox draws of reset() / task_step(), [N][n] (None detaches); golden replays only."""
        if tape is None:
            self._tape = None
            _lib.check(self.lib.mocca_set_draw_tape(self.h, None, 0), self.h)
            return
        self._tape = torch.as_tensor(tape, dtype=torch.float32).to(self.device).contiguous().reshape(self.n_envs, -1)
        _lib.check(self.lib.mocca_set_draw_tape(self.h, C.c_void_p(self._tape.data_ptr()), self._tape.shape[1]), self.h)

    def keep_terminal_obs(self, on: bool = True, buffer: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """Attach (or detach) the terminal-observation buffer [N][obs_dim] (include/mocca.h mocca_set_terminal_obs_buffer); `buffer`:
        a caller-owned contiguous float32 [N][obs_dim] tensor on the device (e.g. this handle's rows of a larger batch's buffer)."""
        if on and buffer is not None:
            if buffer.shape != (self.n_envs, self.obs_dim) or buffer.dtype != torch.float32 or not buffer.is_contiguous() or buffer.device != self.device:
                raise ValueError("terminal-observation buffer must be a contiguous float32 [n_envs, obs_dim] tensor on the env's device")
            self.terminal_obs = buffer
        else:
            self.terminal_obs = torch.zeros(self.n_envs, self.obs_dim, dtype=torch.float32, device=self.device) if on else None
        _lib.check(self.lib.mocca_set_terminal_obs_buffer(self.h, C.c_void_p(self.terminal_obs.data_ptr()) if on else None), self.h)
        return self.terminal_obs

    def episode_stats(self, on: bool = True, slots: int = 4, masks=None, bad_masks=None, totals=None, records=None, row0: int = 0) -> Optional[dict]:
        """Monitor + TimeLimitMask inside the launch (include/mocca.h mocca_set_episode_stats): per step and env the PPO loop's `masks` /
        `bad_masks` columns (float32 [N] on the device, 0.0 where the episode ended / ended with the TimeLimit bit), device-side `totals`
        [4] (sums of return, length, episodes, truncated episodes) and, for the envs that finished, a 16-byte record {serial, return,
        length, done bits | info << 8} written by the kernel straight into PINNED HOST memory: `records` int32 [slots][N][4], the k-th
        step() after this call (k = 1, 2, ...) writes slot k % slots with serial k.  Nothing is copied and nothing synchronises; read a
        slot after the stream work of its step has completed.  The four buffers are allocated here unless passed in (a sub-batch gets
        its rows of a whole batch's buffers: `records` is then the whole ring and `row0` this handle's first row).  on=False detaches."""
        if not on:
            _lib.check(self.lib.mocca_set_episode_stats(self.h, None, None, None, None, 0, 0), self.h)
            self.ep = None
            return None
        n = self.n_envs
        f32 = dict(dtype=torch.float32, device=self.device)
        masks = torch.ones(n, **f32) if masks is None else masks
        bad_masks = torch.ones(n, **f32) if bad_masks is None else bad_masks
        totals = torch.zeros(4, **f32) if totals is None else totals
        if records is None:
            records = torch.zeros(int(slots), n, 4, dtype=torch.int32).pin_memory()
        for t, shape in ((masks, (n,)), (bad_masks, (n,)), (totals, (4,))):
            if tuple(t.shape) != shape or t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
                raise ValueError("episode_stats: masks / bad_masks must be contiguous float32 [n_envs] and totals float32 [4] on the env's device")
        if records.dim() != 3 or records.shape[2] != 4 or records.dtype != torch.int32 or not records.is_contiguous() or \
                row0 < 0 or row0 + n > records.shape[1] or not (records.is_cuda or records.is_pinned()):
            raise ValueError("episode_stats: records must be a contiguous int32 [slots][rows >= row0 + n_envs][4] tensor in pinned host (or device) memory")
        self._sync()      # no launch of this handle is in flight while its buffers change
        _lib.check(self.lib.mocca_set_episode_stats(self.h, C.c_void_p(masks.data_ptr()), C.c_void_p(bad_masks.data_ptr()), C.c_void_p(totals.data_ptr()),
                                                    C.c_void_p(records.data_ptr() + 16 * row0), records.shape[0], 16 * records.shape[1]), self.h)
        self.ep = dict(masks=masks, bad_masks=bad_masks, totals=totals, records=records, row0=row0, slots=records.shape[0],
                       first_serial=int(self.lib.mocca_episode_serial(self.h)))
        return self.ep

    def set_debug(self, on: bool = True) -> Optional[torch.Tensor]:
        """Attach (or detach) the per-env debug record: [N][16] int32, words MOCCA_DBG_* (0..11 the active set of the last substep
        incl. the solver's clamp mask / signature, 12..15 cumulative cap pressure: zero the tensor to restart the count)."""
        self.debug = torch.zeros(self.n_envs, _lib.DEBUG_WORDS, dtype=torch.int32, device=self.device) if on else None
        _lib.check(self.lib.mocca_set_debug_buffer(self.h, C.c_void_p(self.debug.data_ptr()) if on else None), self.h)
        return self.debug

    def task_step(self, actions: torch.Tensor, touch, target=None, body=None):
        """env.step()'s task layer on the stored (post-physics) state with caller-supplied contact flags
        (include/mocca.h mocca_task_step): the golden replays of the reference's scripted episodes."""
        actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
        i32 = lambda x: None if x is None else torch.as_tensor(x, dtype=torch.int32).to(self.device).contiguous()
        touch, target, body = i32(touch), i32(target), i32(body)
        ptr = lambda x: None if x is None else C.c_void_p(x.data_ptr())
        self._in()
        _lib.check(self.lib.mocca_task_step(self.h, C.c_void_p(actions.data_ptr()), ptr(touch), ptr(target), ptr(body),
                                            C.c_void_p(self.obs.data_ptr()), C.c_void_p(self.rew.data_ptr()),
                                            C.c_void_p(self.done.data_ptr()), C.c_void_p(self.info.data_ptr()), self._stream()), self.h)
        self._sync()   # the int32 temporaries above must outlive the launch
        return self.obs, self.rew, self.done, self.info

    def reset(self, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        mp = None
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
            mp = C.c_void_p(mask.data_ptr())
        self._in()
        _lib.check(self.lib.mocca_reset(self.h, mp, self.seed_value, C.c_void_p(self.obs.data_ptr()), self._stream()), self.h)
        self._out()
        return self.obs

    def step(self, actions: torch.Tensor, obs_out: Optional[torch.Tensor] = None,
             rew_out: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """One env.step of all envs: one kernel launch on torch's current stream (or on `self.stream`, unordered against the current
        stream: see _in / _out above).  `obs_out` / `rew_out`: where THIS launch writes its observations [N, obs_dim] / rewards [N] (or
        [N, 1]) instead of `self.obs` / `self.rew` -- e.g. row t + 1 of a trainer's rollout storage, which PPO reads the next policy input
        from anyway: the kernel's outputs need no copy (mocca_step takes the pointers per call).  Contiguous float32 on the env's device."""
        if actions.device != self.device or actions.dtype != torch.float32 or not actions.is_contiguous():
            actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
        if actions.shape != (self.n_envs, self.act_dim):
            raise ValueError(f"actions must be [{self.n_envs}, {self.act_dim}]")
        obs, rew = self.obs if obs_out is None else obs_out, self.rew if rew_out is None else rew_out
        if obs_out is not None or rew_out is not None:
            if obs.shape != (self.n_envs, self.obs_dim) or rew.numel() != self.n_envs or obs.dtype != torch.float32 or rew.dtype != torch.float32 \
                    or not obs.is_contiguous() or not rew.is_contiguous() or obs.device != self.device or rew.device != self.device:
                raise ValueError("obs_out / rew_out must be contiguous float32 [n_envs, obs_dim] / [n_envs] tensors on the env's device")
        _lib.check(self.lib.mocca_step(self.h, C.c_void_p(actions.data_ptr()), C.c_void_p(obs.data_ptr()),
                                       C.c_void_p(rew.data_ptr()), C.c_void_p(self.done.data_ptr()),
                                       C.c_void_p(self.info.data_ptr()), self._stream()), self.h)
        return obs, rew, self.done, self.info

    def episode_masks_into(self, masks: torch.Tensor, bad_masks: torch.Tensor) -> None:
        """Point the in-kernel `masks` / `bad_masks` columns of episode_stats() at other buffers FROM THE NEXT LAUNCH ON (e.g. row t + 1 of a
        trainer's rollout storage); totals and records stay where they are.  No synchronisation: launches in flight keep the pointers they
        were issued with.  Contiguous float32, n_envs elements each, on the env's device."""
        if self.ep is None:
            raise _lib.MoccaError("episode_masks_into needs episode_stats(True) first")
        for t in (masks, bad_masks):
            if t.numel() != self.n_envs or t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
                raise ValueError("masks / bad_masks must be contiguous float32 tensors of n_envs elements on the env's device")
        r = self.ep["records"]
        _lib.check(self.lib.mocca_set_episode_stats(self.h, C.c_void_p(masks.data_ptr()), C.c_void_p(bad_masks.data_ptr()),
                                                    C.c_void_p(self.ep["totals"].data_ptr()), C.c_void_p(r.data_ptr() + 16 * self.ep["row0"]),
                                                    r.shape[0], 16 * r.shape[1]), self.h)

    # ---- host-side callers (the single-env gym classes; a trainer that lives on the host) ----
    def host_mirror(self) -> dict:
        """Lay obs / rew / info / state / task / done out in ONE device buffer with ONE pinned host image, so that a caller on the host
        pays one upload, one download and one synchronize per step (`step_host`) instead of one blocking copy per quantity.  The
        tensors `obs`, `rew`, `done`, `info` become views into that buffer.  Returns the numpy views of the host image."""
        if getattr(self, "_host_np", None) is not None:
            return self._host_np
        n = self.n_envs
        parts = [("obs", n * self.obs_dim, torch.float32, np.float32, (n, self.obs_dim)), ("rew", n, torch.float32, np.float32, (n,)),
                 ("info", n, torch.int32, np.int32, (n,)), ("state", n * self.state_dim, torch.float32, np.float32, (n, self.state_dim)),
                 ("task", n * M.TASK_WORDS, torch.int32, np.int32, (n, M.TASK_WORDS))]
        words = sum(p[1] for p in parts)
        total = 4 * words + ((n + 3) // 4) * 4
        self._pack = torch.zeros(total, dtype=torch.uint8, device=self.device)
        self._pack_host = torch.zeros(total, dtype=torch.uint8).pin_memory()
        host = self._pack_host.numpy()
        dev, hnp, off = {}, {}, 0
        for name, cnt, tdt, ndt, shape in parts:
            dev[name] = self._pack[off:off + 4 * cnt].view(tdt).view(*shape)
            hnp[name] = host[off:off + 4 * cnt].view(ndt).reshape(shape)
            off += 4 * cnt
        dev["done"], hnp["done"] = self._pack[off:off + n], host[off:off + n]
        dev["obs"].copy_(self.obs); dev["rew"].copy_(self.rew); dev["info"].copy_(self.info); dev["done"].copy_(self.done)
        self.obs, self.rew, self.info, self.done = dev["obs"], dev["rew"], dev["info"], dev["done"]
        self._dev_views = dev
        self._act_host = torch.zeros(n, self.act_dim, dtype=torch.float32).pin_memory()
        self._act_dev = torch.zeros(n, self.act_dim, dtype=torch.float32, device=self.device)
        self._host_np = hnp
        return hnp

    def _download(self) -> dict:
        d = self._dev_views
        _lib.check(self.lib.mocca_get_state(self.h, C.c_void_p(d["state"].data_ptr()), self._stream()), self.h)
        _lib.check(self.lib.mocca_get_task(self.h, C.c_void_p(d["task"].data_ptr()), self._stream()), self.h)
        self._out()
        self._pack_host.copy_(self._pack, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return self._host_np

    def step_host(self, actions_np) -> dict:
        """step() for a caller on the host: float32 actions [n_envs, act_dim] in, the host image (obs, rew, done, info AND the state /
        task records after the step) out -- valid until the next call.  One synchronize."""
        hnp = self.host_mirror()
        self._act_host.numpy()[...] = actions_np
        self._act_dev.copy_(self._act_host, non_blocking=True)
        self._in()
        self.step(self._act_dev)
        return self._download()

    def reset_host(self) -> dict:
        self.host_mirror()
        self.reset()
        return self._download()

    def observe_host(self) -> dict:
        self.host_mirror()
        self.observe()
        return self._download()

    def observe(self) -> torch.Tensor:
        """calc_state() + observation tail of the current state, no stepping (include/mocca.h mocca_observe)."""
        self._in()
        _lib.check(self.lib.mocca_observe(self.h, C.c_void_p(self.obs.data_ptr()), self._stream()), self.h)
        self._out()
        return self.obs

    # ---- snapshots (saveState/restoreState role; used by the parity tests) ----
    def get_state(self) -> torch.Tensor:
        st = torch.empty(self.n_envs, self.state_dim, dtype=torch.float32, device=self.device)
        self._in()      # (st's memory may have been in use on the current stream a moment ago)
        _lib.check(self.lib.mocca_get_state(self.h, C.c_void_p(st.data_ptr()), self._stream()), self.h)
        self._out()
        return st

    def set_state(self, st) -> None:
        st = torch.as_tensor(st, dtype=torch.float32).to(self.device).contiguous().reshape(self.n_envs, self.state_dim)
        self._in()
        _lib.check(self.lib.mocca_set_state(self.h, C.c_void_p(st.data_ptr()), self._stream()), self.h)
        self._sync()

    def get_task(self) -> torch.Tensor:
        t = torch.empty(self.n_envs, M.TASK_WORDS, dtype=torch.int32, device=self.device)
        self._in()
        _lib.check(self.lib.mocca_get_task(self.h, C.c_void_p(t.data_ptr()), self._stream()), self.h)
        self._out()
        return t

    def set_task(self, t: torch.Tensor) -> None:
        t = t.to(device=self.device, dtype=torch.int32).contiguous().reshape(self.n_envs, M.TASK_WORDS)
        self._in()
        _lib.check(self.lib.mocca_set_task(self.h, C.c_void_p(t.data_ptr()), self._stream()), self.h)
        self._sync()

    def get_terrain(self) -> torch.Tensor:
        t = torch.empty(self.n_envs, 128, dtype=torch.float32, device=self.device)
        self._in()
        _lib.check(self.lib.mocca_get_terrain(self.h, C.c_void_p(t.data_ptr()), self._stream()), self.h)
        self._out()
        return t

    def set_terrain(self, t) -> None:
        t = torch.as_tensor(t, dtype=torch.float32).to(self.device).contiguous().reshape(self.n_envs, 128)
        self._in()
        _lib.check(self.lib.mocca_set_terrain(self.h, C.c_void_p(t.data_ptr()), self._stream()), self.h)
        self._sync()

    def kernel_info(self) -> dict:
        v = [C.c_int() for _ in range(5)]
        _lib.check(self.lib.mocca_kernel_info(self.h, *[C.byref(x) for x in v]), self.h)
        out = dict(vgprs=v[0].value, lds_bytes=v[2].value, scratch_bytes=v[3].value, max_blocks_per_cu=v[4].value)
        if v[1].value >= 0:      # the HIP runtime reports no scalar-register count (-1)
            out["sgprs"] = v[1].value
        return out


# task-record helpers: the device record is 24 x 32-bit words, floats and ints mixed (mocca_model.h)
TASK_FLOAT_WORDS = (0, 1, 2, 3, 4, 6, 12, 13, 14, 15, 21, 22) + tuple(range(24, 39))


def task_to_float64(t: torch.Tensor) -> np.ndarray:
    """int32 view of the device task record -> float64 array in the oracle's get_task() layout."""
    a = (t.cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)).astype(np.int32)
    out = a.astype(np.float64)
    fl = a.view(np.float32)
    for w in TASK_FLOAT_WORDS:
        out[:, w] = fl[:, w]
    return out


def task_from_float64(a: np.ndarray) -> torch.Tensor:
    a = np.asarray(a, np.float64)
    out = a.astype(np.int32)
    fl = out.view(np.float32)
    for w in TASK_FLOAT_WORDS:
        fl[:, w] = a[:, w].astype(np.float32)
    return torch.from_numpy(out)
