"""Single-environment gym classes with the reference's surface, backed by the HIP stepper.

Drop-in for /root/reference/mocca_envs/env_locomotion.py `Walker3DCustomEnv` / `Walker3DStepperEnv`
(and `EnvBase`, env_base.py:11-201): same ids, spaces, `reset() -> obs`, `step(a) -> (obs, reward, done,
info)` (4-tuple, no auto-reset), `seed`, `close`, `get_mirror_indices`, `evaluation_mode`, `set_env_params`,
`get_env_param`, attributes `robot.mirrored`, `robot.applied_gain`, `curriculum`, `max_curriculum`.
What is replaced is `EnvBase._p` (the pybullet client) and everything the reference did through it.

A facade env is a batch of one on the GPU; trainers that want throughput use `VecEnv` directly.
Rendering (`render=True`, egl, ffmpeg) is out of scope and raises.
"""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - depends on the machine
    import gym  # type: ignore
except Exception:  # this image has no gym: use the shim with the same surface
    from . import gym_shim as _shim
    gym = _shim.as_module()

from . import host_logic as H
from . import model as M


class _CassieRobot:
    """The attributes of env_cassie.Cassie that trainers touch (env_cassie.py:13-79)."""

    foot_names = ["right_toe", "left_toe"]
    powered_joint_inds = [0, 1, 2, 3, 6, 7, 8, 9, 10, 13]
    spring_joint_inds = [4, 11]

    def __init__(self, mdl: M.MoccaModel):
        self.base_joint_angles = list(M.CASSIE_BASE_ANGLES)
        self.base_position = tuple(mdl.init_pos)
        high = np.ones(10)
        self.action_space = gym.spaces.Box(-high, high, dtype=np.float32)
        high = np.inf * np.ones((10 + 4) * 2 + 6)
        self.observation_space = gym.spaces.Box(-high, high, dtype=np.float32)
        self.body_xyz = np.zeros(3)


class _Robot:
    """The attributes of robots.Walker3D that trainers touch (robots.py:13-29,230-290)."""

    foot_names = ["right_foot", "left_foot"]

    def __init__(self, mdl: M.MoccaModel):
        self._mdl = mdl
        self.mirrored = False
        self.applied_gain = 1.0
        self.action_dim = mdl.n_joints
        high = np.ones(self.action_dim)
        self.action_space = gym.spaces.Box(-high, high, dtype=np.float32)
        if mdl.n_feet == 4:
            self.foot_names = ["toeFR", "toeFL", "toeRR", "toeRL"]   # robots.py:559
        self.state_dim = 6 + self.action_dim * 2 + len(self.foot_names)
        high = np.inf * np.ones(self.state_dim)
        self.observation_space = gym.spaces.Box(-high, high, dtype=np.float32)
        self._right_joint_indices = np.array(list(mdl.mirror_right)[: mdl.n_mirror_side], dtype=np.int64)
        self._left_joint_indices = np.array(list(mdl.mirror_left)[: mdl.n_mirror_side], dtype=np.int64)
        self._negation_joint_indices = np.array(list(mdl.mirror_neg)[: mdl.n_mirror_neg], dtype=np.int64)
        self.body_xyz = np.zeros(3)
        self.joint_angles = np.zeros(self.action_dim)
        self.joint_speeds = np.zeros(self.action_dim)
        self.feet_contact = np.zeros(len(self.foot_names), dtype=np.float32)


class EnvBase(gym.Env):
    """env_base.py:11-201 without the Bullet client: the GPU batch-of-one plays `_p`."""

    metadata = {"render.modes": ["human", "rgb_array"]}
    control_step = 1 / 60
    llc_frame_skip = 1
    sim_frame_skip = 4
    env_id = None
    task_id = None

    def __init__(self, render=False, remove_ground=False, use_egl=False, use_ffmpeg=False, device=None, model_kw=None, **kwargs):
        if render or use_egl or use_ffmpeg:
            raise NotImplementedError("rendering is outside the GPU stepper's scope (SURVEY.md section 2.1 #6)")
        if kwargs:
            raise TypeError(f"unexpected arguments {sorted(kwargs)}")
        self.is_rendered = False
        self.metadata = dict(self.metadata)
        self.metadata["video.frames_per_second"] = int(1 / self.control_step)
        from .vec_env import VecEnv  # imports torch; needs the HIP library and a GPU (no CPU fallback)
        self._vec = VecEnv(self.env_id, 1, device=device, auto_reset=False, **(model_kw or {}))
        self.model = self._vec.model
        self.robot = _CassieRobot(self.model) if self.task_id == M.TASK_CASSIE else _Robot(self.model)
        self.seed()

    # ---- gym surface --------------------------------------------------------------------------
    def seed(self, seed=None):
        self.np_random, seed = gym.utils.seeding.np_random(seed)  # env_base.py:164-166
        self.robot.np_random = self.np_random
        return [seed]

    def close(self):
        if getattr(self, "_vec", None) is not None:
            self._vec.close()
            self._vec = None

    def render(self, mode="human"):
        raise NotImplementedError("rendering is outside the GPU stepper's scope")

    def set_env_params(self, params_dict):  # env_base.py:103-106
        for k, v in params_dict.items():
            if hasattr(self, k):
                setattr(self, k, v)

    def get_env_param(self, param_name, default):  # env_base.py:117-118
        return getattr(self, param_name, default)

    def set_robot_params(self, params_dict):  # env_base.py:108-115 (its calc_torque_limits does not exist)
        for k, v in params_dict.items():
            if hasattr(self.robot, k):
                setattr(self.robot, k, v)
        if "applied_gain" in params_dict and hasattr(self.robot, "applied_gain"):   # used by the next apply_action, robots.py:33
            from . import lib as _lib
            self._vec.set_param(_lib.PARAM_APPLIED_GAIN, float(self.robot.applied_gain))

    # ---- helpers --------------------------------------------------------------------------------
    def _push(self, state, task, terrain=None):
        import torch
        from .vec_env import task_from_float64
        self._vec.set_state(torch.from_numpy(np.asarray(state, np.float32)[None]))
        self._vec.set_task(task_from_float64(np.asarray(task, np.float64)[None]))
        if terrain is not None:
            self._vec.set_terrain(torch.from_numpy(np.asarray(terrain, np.float32)[None]))

    def _pull_robot(self):
        st = self._img["state"][0]      # the host image of the last step_host / observe_host / reset_host (one download per call)
        nj = self.model.n_joints
        self.robot.body_xyz = st[0:3].astype(np.float64)
        self.robot.joint_angles = st[13:13 + nj].copy()
        self.robot.joint_speeds = 0.1 * st[13 + nj:13 + 2 * nj]

    def _step_device(self, action):
        import torch
        action = np.asarray(action, dtype=np.float64)
        assert np.isfinite(action).all()  # robots.py:32
        img = self._img = self._vec.step_host(action.astype(np.float32)[None])
        return img["obs"][0].astype(np.float64), float(img["rew"][0]), bool(int(img["done"][0]) & 1), int(img["info"][0])

    def _observe(self):
        self._img = self._vec.observe_host()
        return self._img["obs"][0].astype(np.float64)


class Walker3DCustomEnv(EnvBase):
    """env_locomotion.py:37-282.  Every random draw (reset AND mid-episode re-targeting) comes from
    `self.np_random` with the reference's calls in the reference's order."""

    env_id = "Walker3DCustomEnv-v0"
    task_id = M.TASK_WALKER3D_CUSTOM
    termination_height = 0.7
    robot_random_start = True

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        from . import lib as _lib
        self._vec.set_param(_lib.PARAM_HOST_RETARGET, 1)
        self.eval_mode = False
        self.electricity_cost, self.stall_torque_cost, self.joints_at_limit_cost = 4.5, 0.225, 0.1
        high = np.inf * np.ones(self.robot.observation_space.shape[0] + 2)
        self.observation_space = gym.spaces.Box(-high, high, dtype=np.float32)
        self.action_space = self.robot.action_space

    def evaluation_mode(self):  # env_locomotion.py:76-77
        self.eval_mode = True

    def reset(self):
        from . import lib as _lib
        self._vec.set_param(_lib.PARAM_EVAL_MODE, int(self.eval_mode))
        self.done = False
        self.dist, self.angle, self.stop_frames = H.randomize_target(self.np_random, self.eval_mode)
        self.walk_target = np.array([self.dist * np.cos(self.angle), self.dist * np.sin(self.angle), 1.0])
        self.close_count = 0
        q, self.robot.mirrored = H.reset_pose(self.robot.np_random, self.model, self.robot_random_start)
        self._episode = getattr(self, "_episode", -1) + 1
        task = H.task_record(walk_target=self.walk_target, stop_frames=self.stop_frames, dist=self.dist, angle=self.angle,
                             mirrored=int(self.robot.mirrored), episode=self._episode)
        self._push(H.initial_state(self.model, q), task)
        obs = self._observe()
        self._pull_robot()
        return obs

    def step(self, action):
        from .vec_env import task_to_float64, task_from_float64
        obs, rew, done, _ = self._step_device(action)
        self._pull_robot()
        tk = task_to_float64(self._img["task"])[0]
        self.walk_target, self.close_count = tk[0:3].copy(), int(tk[5])
        if self.close_count >= self.stop_frames:  # env_locomotion.py:214-222, host RandomState like the reference
            self.close_count = 0
            self.dist, self.angle, self.stop_frames = H.randomize_target(self.np_random, self.eval_mode)
            self.walk_target = self.walk_target + self.dist * np.array([np.cos(self.angle), np.sin(self.angle), 0.0])
            st = self._img["state"][0].astype(np.float64)
            yaw = H.yaw_from_quat(*st[3:7])
            dx, dy = self.walk_target[0] - st[0], self.walk_target[1] - st[1]
            ang, dist = np.arctan2(dy, dx) - yaw, np.hypot(dx, dy)
            tk[0:3], tk[5], tk[6], tk[14], tk[15] = self.walk_target, 0, self.stop_frames, self.dist, self.angle
            tk[3], tk[4] = -dist / (1 / 60), np.cos(ang)  # calc_potential :143-158 (scene.dt = 1/60)
            self._vec.set_task(task_from_float64(tk[None]))
            s_, c_ = dist * np.sin(ang), dist * np.cos(ang)
            obs[-2], obs[-1] = s_ / (1 + abs(s_)), c_ / (1 + abs(c_))
        self.done = done
        return obs, rew, done, {}

    def get_mirror_indices(self):
        return H.mirror_indices(self.model, stepper=False)

    @classmethod
    def mirror_indices(cls):
        from .vec_env import compile_model_for
        return H.mirror_indices(compile_model_for(cls.env_id), stepper=False)


class Child3DCustomEnv(Walker3DCustomEnv):
    """env_locomotion.py:317-327: Child3D (robots.py:326-335) from the "crawl" pose, fallen below 0.1 m."""

    env_id = "Child3DCustomEnv-v0"
    termination_height = 0.1


class Walker2DCustomEnv(Walker3DCustomEnv):
    """env_locomotion.py:285-309: the planar walker; reset returns [robot_state, 0, 0] and step never reports done."""

    env_id = "Walker2DCustomEnv-v0"
    robot_init_position = [0, 0, 1.35]

    def reset(self):
        obs = super().reset()
        obs[-2:] = 0.0                      # :299-300
        return obs

    def step(self, action):
        obs, rew, _, info = super().step(action)
        self.done = False                   # :303-305 (the device record is cleared the same way: MOCCA_TASKF_NEVER_DONE)
        return obs, rew, False, info


class Crab2DCustomEnv(Walker2DCustomEnv):
    """env_locomotion.py:312-314."""

    env_id = "Crab2DCustomEnv-v0"


class LaikagoCustomEnv(Walker3DCustomEnv):
    """env_locomotion.py:854-890: the quadruped; 8 substeps of 1/480 s, no random start pose, tall_bonus 0, and the
    episode ends as soon as anything but a foot touches the ground."""

    env_id = "LaikagoCustomEnv-v0"
    sim_frame_skip = 8
    termination_height = 0
    robot_random_start = False
    robot_init_position = [0, 0, 0.56]

    def __init__(self, **kwargs):
        kwargs.pop("random_reward", False)
        kwargs.pop("plank_class", None)
        super().__init__(**kwargs)
        self.curriculum, self.max_curriculum = 0, 9


class Walker3DStepperEnv(EnvBase):
    """env_locomotion.py:330-840."""

    env_id = "Walker3DStepperEnv-v0"
    task_id = M.TASK_WALKER3D_STEPPER
    max_timestep = 1000
    robot_random_start = True
    n_steps, step_radius, rendered_step_count = 20, 0.25, 3
    lookahead, lookbehind, step_param_dim = 2, 1, 5
    plank_class = "LargePlank"          # Pillar, Plank, LargePlank (env_locomotion.py:342, bullet_objects.py:86-103)

    def __init__(self, **kwargs):
        self.random_reward = kwargs.pop("random_reward", False)
        plank = kwargs.pop("plank_class", None)
        if plank is not None:
            if plank not in M.PLANK_CLASSES:    # the reference falls back to the default for unknown names (:356-357)
                plank = self.plank_class
            self.plank_class = plank
        kwargs.pop("remove_ground", None)
        super().__init__(model_kw={"plank_class": self.plank_class}, **kwargs)
        from . import lib as _lib
        # random_reward (:533-547): the eight weights come from THIS env's np_random, like every other draw of the facade
        self._vec.set_param(_lib.PARAM_RANDOM_REWARD, 2 if self.random_reward else 0)
        self.curriculum, self.max_curriculum = 0, 9
        self.terminal_height_curriculum = np.linspace(H._dec(self.model.term_height_cur[0]), H._dec(self.model.term_height_cur[1]), 10)
        self.applied_gain_curriculum = np.linspace(H._dec(self.model.gain_cur[0]), H._dec(self.model.gain_cur[1]), 10)
        self.next_step_index = self.lookbehind
        self.terrain_info = np.zeros((self.n_steps, 6))
        self.robot_obs_dim = self.robot.observation_space.shape[0]
        high = np.inf * np.ones(self.robot_obs_dim + (self.lookahead + self.lookbehind) * self.step_param_dim)
        self.observation_space = gym.spaces.Box(-high, high, dtype=np.float32)
        self.action_space = self.robot.action_space

    def reset(self):
        self.timestep, self.done = 0, False
        cur = min(int(self.curriculum), self.max_curriculum)
        self.robot.applied_gain = H.applied_gain(cur, self.model)
        q, self.robot.mirrored = H.reset_pose(self.robot.np_random, self.model, self.robot_random_start)
        self.terrain_info = H.generate_step_placements(self.np_random, cur, self.model)
        self.next_step_index = self.lookbehind
        self._episode = getattr(self, "_episode", -1) + 1
        terrain = np.zeros(128, np.float32)
        terrain[:120] = self.terrain_info.reshape(-1)
        terrain[120:124] = [0, 1, 2, 3]
        task = H.task_record(next_step_index=self.next_step_index, curriculum=cur, applied_gain=self.robot.applied_gain,
                             mirrored=int(self.robot.mirrored), episode=self._episode, draw=122)
        # self.calc_feet_state() between robot.reset() and randomize_terrain() (:484-499) reads Bullet's manifolds of the LAST frame of
        # the episode before (no stepSimulation since): feet_contact and, with a foot on the cover of what was then the target plank,
        # target_reached_count = 1 carry over (MOCCA_TASKF_STALE_RESET_CONTACTS; the device record of the last step holds both)
        nf = int(self.model.n_feet)
        if (self.model.task_flags & M.TASKF_STALE_RESET_CONTACTS) and getattr(self, "_img", None) is not None and self._episode > 0:
            from .vec_env import task_to_float64
            old = task_to_float64(self._img["task"])[0]
            fc = [old[12], old[13], old[24], old[25]][:nf]
            task[12:14] = fc[:2]
            if nf > 2:
                task[24:26] = fc[2:4]
            cover, old_nsi = int(old[26]), int(old[16])
            if any((cover >> (4 * f + old_nsi % int(self.model.n_planks))) & 1 for f in range(nf)):
                task[17] = 1
        self._vec.set_param(2, cur)
        self._push(H.initial_state(self.model, q), task, terrain)
        obs = self._observe()
        obs[6 + 2 * int(self.model.n_joints):6 + 2 * int(self.model.n_joints) + nf] = 0.0   # robot.reset()'s own calc_state: feet_contact.fill(0) (robots.py:197-200)
        self._pull_robot()
        return obs

    def step(self, action):
        self.timestep += 1
        if self.random_reward:   # np_random.uniform(0.8, 1.2, 8), :533-535: drawn here, handed to the kernel in task words 30..37
            from .vec_env import task_to_float64, task_from_float64
            # the LIVE record, not the host image of the last step: set_robot_params / VecEnv.seed / set_task may have changed the device
            # record since (a stale image written back would silently revert them -- ADVICE r3); one 160-byte download, this mode only
            tk = task_to_float64(self._vec.get_task())
            tk[0, 30:38] = self.np_random.uniform(0.8, 1.2, 8)
            self._vec.set_task(task_from_float64(tk))
        cur = min(int(self.curriculum), self.max_curriculum)
        if cur != getattr(self, "_pushed_curriculum", None):   # terminal height follows self.curriculum at once (:628)
            self._vec.set_param(2, cur)
            self._pushed_curriculum = cur
        obs, rew, done, nsi = self._step_device(action)
        self._pull_robot()
        self.done, self.next_step_index = done, nsi
        info = {"steps_reached": nsi} if done or self.timestep == self.max_timestep - 1 else {}  # :562-566
        return obs, rew, done, info

    def get_mirror_indices(self):
        return H.mirror_indices(self.model, stepper=True)

    @classmethod
    def mirror_indices(cls):
        from .vec_env import compile_model_for
        return H.mirror_indices(compile_model_for(cls.env_id), stepper=True)


class MikeStepperEnv(Walker3DStepperEnv):
    """env_locomotion.py:843-851: Mike (robots.py:474-510) starting at (0.3, 0, 1.0)."""

    env_id = "MikeStepperEnv-v0"


class LaikagoStepperEnv(Walker3DStepperEnv):
    """env_locomotion.py:893-979: the quadruped on four live planks of radius 0.16, two planks of look-behind, started at
    (0.25, 0, 0.53) with velocity (0.5, 0, 0.25); its own posture penalty, doubled progress, time-based early termination and
    body-contact termination (calc_base_reward :928-979) are in the kernel (MOCCA_TASKF_QUADRUPED_STEPPER)."""

    env_id = "LaikagoStepperEnv-v0"
    robot_random_start = False
    robot_init_position = [0.25, 0, 0.53]
    robot_init_velocity = [0.5, 0, 0.25]
    step_radius, rendered_step_count, init_step_separation = 0.16, 4, 0.45
    lookahead, lookbehind = 2, 2
    step_bonus_smoothness = 6


class Walker3DPlannerEnv(EnvBase):
    """env_locomotion.py:982-1128: the walker on the height field, steered by a 15-number plan that a low-level "base controller"
    turns into the 21 joint actions.

    The reference unpickles that controller (`MikePlannerBase.pt`, a torch actor-critic whose class is not in its tree,
    :1022-1033); here it is INJECTED: `base_controller(base_obs[65]) -> (value, action[21])`, the very call the reference makes
    (`self.query_base_controller`, :1093-1094).  Without one, `step` raises.  `load_base_controller(path)` does what the reference
    does when torch and the pickled class are importable."""

    env_id = "Walker3DPlannerEnv-v0"
    task_id = M.TASK_WALKER3D_PLANNER
    robot_random_start = True
    robot_init_position = [-15.5, -15.5, 1.32]
    robot_init_velocity = None
    robot_torso_name = "waist"
    termination_height = 0.5
    action_scale = 2
    base_lookahead, base_lookbehind, base_step_param_dim = 2, 1, 5

    def __init__(self, base_controller=None, **kwargs):
        kwargs.pop("remove_ground", None)
        super().__init__(**kwargs)
        from .terrain import HeightField
        self.terrain = HeightField(H.HEIGHT_FIELD_SIZE, H.HEIGHT_FIELD_SCALE)      # create_terrain, :1011-1021
        self.terrain.reload(data=H.HEIGHT_FIELD_FILE, rng=self.np_random)
        self.query_base_controller = base_controller
        self.robot_obs_dim = self.robot.observation_space.shape[0]
        high = np.inf * np.ones(self.robot_obs_dim + 2)
        self.observation_space = gym.spaces.Box(-high, high, dtype=np.float32)
        high = np.inf * np.ones((self.base_lookahead + self.base_lookbehind) * self.base_step_param_dim)
        self.action_space = gym.spaces.Box(-high, high, dtype=np.float32)

    def load_base_controller(self, filename):
        """The reference's loader (:1022-1033): needs torch and the module that defines the pickled policy class on the path."""
        import torch
        actor_critic = torch.load(filename, map_location="cpu")

        def inference(o):
            with torch.no_grad():
                value, action, _ = actor_critic.act(torch.from_numpy(o).unsqueeze(0), deterministic=True)
                return value.squeeze().numpy(), action.squeeze().numpy()

        self.query_base_controller = inference
        return inference

    def reset(self):
        self.timestep, self.done = 0, False
        q, self.robot.mirrored = H.reset_pose(self.robot.np_random, self.model, self.robot_random_start)
        xy = self.np_random.uniform(-16, 16, 2)                                     # :1060-1062
        z = self.terrain.get_height_at(*xy)
        self.walk_target = np.array((*xy, z), dtype=np.float32)
        self._episode = getattr(self, "_episode", -1) + 1
        task = H.task_record(walk_target=self.walk_target.astype(np.float64), mirrored=int(self.robot.mirrored), episode=self._episode)
        self._push(H.initial_state(self.model, q), task)
        obs = self._observe()
        self.robot_state = obs[:self.robot_obs_dim].copy()
        self._pull_robot()
        return obs

    def step(self, action):
        if self.query_base_controller is None:
            raise RuntimeError("Walker3DPlannerEnv needs a base controller: pass base_controller=callable(base_obs) -> (value, action[21]) "
                               "or call load_base_controller(path)")
        self.timestep += 1
        base_obs = np.concatenate((self.robot_state, np.asarray(action, dtype=np.float64) * self.action_scale))   # :1093
        base_value, base_action = self.query_base_controller(base_obs.astype(np.float32))
        obs, progress, done, _ = self._step_device(base_action)
        self._pull_robot()
        self.robot_state = obs[:self.robot_obs_dim].copy()
        self.progress = progress
        reward = progress + np.log(max(1, float(base_value))) / 3                                                  # :1101
        self.done = done
        return obs, reward, self.done, {}


class MikePlannerEnv(Walker3DPlannerEnv):
    """env_locomotion.py:1131-1133."""

    env_id = "MikePlannerEnv-v0"
    robot_init_position = [-15.5, -15.5, 1.05]


class CassieEnv(EnvBase):
    """env_cassie.py:284-479: CassieEnv-v0 and, with planar=True, Cassie2DEnv-v0.  The reference class is not importable in the
    reference snapshot; this follows its text, pinned by tests/golden/make_golden_cassie.py."""

    env_id = "CassieEnv-v0"
    task_id = M.TASK_CASSIE
    control_step = 0.03
    llc_frame_skip = 50
    sim_frame_skip = 1

    _mocap = False   # the mocap / phase subclasses: rsi and planar are blob numbers of theirs (the env id selects cassie_mode)

    def __init__(self, render=False, planar=False, power_coef=1.0, residual_control=True, rsi=True, **kwargs):
        # planar (Cassie2DEnv-v0, reference __init__.py:24-29): "constrains the robot movement to a 2D plane" (env_cassie.py:333).
        # The reference points at a URDF that is not in its tree (:279-282); here the 3-D robot's base is held in the x-z plane
        # by three bilateral solver rows (Cassie).  power_coef scales every torque limit (:192-195),
        # residual_control=False adds the action to zero instead of the nominal angles (:434-443): both are blob numbers.
        self.planar, self.residual_control = bool(planar), bool(residual_control)
        if self.planar and self.env_id == "CassieEnv-v0":
            self.env_id = "Cassie2DEnv-v0"
        kwargs["model_kw"] = dict(power_coef=float(power_coef), residual_control=self.residual_control)
        if self._mocap:
            kwargs["model_kw"].update(rsi=bool(rsi), planar=self.planar)
        super().__init__(render=render, **kwargs)
        self.rsi = rsi
        high = np.inf * np.ones(self.robot.observation_space.shape[0] + 2)
        self.observation_space = gym.spaces.Box(-high, high, dtype=np.float32)
        self.action_space = self.robot.action_space

    def reset(self, istep=0):
        import torch
        self.done = False
        self.walk_target = np.array([1000.0, 0.0, 0.0])
        img = self._img = self._vec.reset_host()                      # deterministic: nominal pose at rest
        self.robot.body_xyz = img["state"][0, 0:3].astype(np.float64)
        return img["obs"][0].astype(np.float64)

    def step(self, a):
        import torch
        a = np.asarray(a, dtype=np.float64)
        assert np.isfinite(a).all()  # env_cassie.py:226
        img = self._img = self._vec.step_host(a.astype(np.float32)[None])
        self.robot.body_xyz = img["state"][0, 0:3].astype(np.float64)
        self.done = bool(int(img["done"][0]) & 1)
        return img["obs"][0].astype(np.float64), float(img["rew"][0]), self.done, {}


class CassieMoccaEnv(CassieEnv):
    """env_cassie.py:481-627 (CassieMocapRewEnv + CassieMoccaEnv): PD targets, reset poses and the reward follow the recorded walking
    cycle.  The reference reads it through `loadstep.CassieTrajectory`, which is not in its tree; `mocca_envs_amd.trajectory` re-creates
    the class from the data files that are.  Observation: 40 floats (y z, quaternion w x y z, 14 angles,
    velocity, angular velocity, 14 joint speeds)."""

    env_id = "CassiePhaseMocca2DEnv-v0"
    initial_velocity = [0.8, 0, 0]
    _mocap = True
    _obs_dim = 40
    mirror_indices = {   # env_cassie.py:554-571
        "neg_obs_inds": [0, 3, 5, 21, 23, 25],
        "sideneg_obs_inds": [6, 7, 26, 27],
        "com_obs_inds": [1, 2, 4, 20, 22, 24],
        "left_obs_inds": list(range(6, 13)) + list(range(26, 33)),
        "right_obs_inds": list(range(13, 20)) + list(range(33, 40)),
        "com_act_inds": [],
        "left_act_inds": list(range(0, 5)),
        "right_act_inds": list(range(5, 10)),
        "neg_act_inds": [],
        "sideneg_act_inds": [0, 1],
    }

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        high = np.inf * np.ones(self._obs_dim)
        self.observation_space = gym.spaces.Box(-high, high, dtype=np.float32)
        self.traj = self._vec.trajectory
        self.weights = {k: float(self.model.mocap_w[i]) for i, k in enumerate(
            ("SpeedRew", "JPosRew", "JVelRew", "OrientationRew", "AngularSpeedRew", "CoMRew"))}   # :484-493

    def mocap_time(self):  # :359-360
        return self.istep * self.control_step / self.llc_frame_skip

    def base_angles(self):  # :601-602
        return self.traj.joint_angles(self.mocap_time())

    def base_velocities(self):  # :604-605
        return self.traj.joint_speeds(self.mocap_time())

    def reset(self, istep=None):
        import torch
        if istep is None:
            istep = self.np_random.randint(0, 10000)   # :586-587 (the host RandomState, like the reference)
        self.done = False
        self.walk_target = np.array([1000.0, 0.0, 0.0])
        # the kernel's reset takes floor(10000 u) of the episode's first uniform as istep: hand it the one that gives `istep`
        tape = torch.tensor([[(int(istep) + 0.5) / 10000.0]], dtype=torch.float32)
        self._vec.set_draw_tape(tape)
        img = self._img = self._vec.reset_host()
        obs = img["obs"][0].astype(np.float64)
        self._vec.set_draw_tape(None)
        self.istep = int(istep) if self.rsi else 0
        self.robot.body_xyz = img["state"][0, 0:3].astype(np.float64)
        return obs[: self._obs_dim]

    def step(self, a):
        obs, rew, done, info = super().step(a)
        self.istep += self.llc_frame_skip   # pd_control, :381
        return obs[: self._obs_dim], rew, done, info


class CassiePhaseMoccaEnv(CassieMoccaEnv):
    """env_cassie.py:630-642: the same with the two gait phases appended (42 floats)."""

    env_id = "CassiePhaseMocca2DEnv-v0"
    _obs_dim = 42

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        import copy
        self.mirror_indices = copy.deepcopy(self.mirror_indices)
        self.mirror_indices["left_obs_inds"] += [40]
        self.mirror_indices["right_obs_inds"] += [41]


class CassiePhaseMirrorEnv(CassiePhaseMoccaEnv):
    """env_cassie.py:645-660: left and right swapped and the lateral entries negated whenever the left phase is past 0.5."""

    env_id = "CassiePhaseMirror2DEnv-v0"

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        mi = self.mirror_indices
        self.neg_inds = mi["neg_obs_inds"] + mi["sideneg_obs_inds"]
        self.lr_inds = mi["left_obs_inds"] + mi["right_obs_inds"]
        self.rl_inds = mi["right_obs_inds"] + mi["left_obs_inds"]
