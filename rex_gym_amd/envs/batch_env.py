"""RexBatchEnv: N reference environments stepped by one HIP launch.

Stands where the reference puts `BatchEnv` over N `ExternalProcess`-wrapped envs
(rex_gym/agents/tools/batch_env.py:18-115, wrappers.py:294-458): same call shapes
(`len(env)`, `env[i]`, `step(actions[N,A]) -> (obs[N,O], reward[N], done[N], infos)`,
`reset(indices=None)`), but the state never leaves HBM and there is no IPC.

PyTorch is used for device memory and streams only; all compute is in librexsim_hip.so.
"""
import ctypes
import math

import numpy as np

from .. import _lib
from .spaces import Box

OBSERVATION_EPS = 0.01  # rex_gym_env.py:20


def _spaces(task, signal, sim_dt, num_motors=12):
    """Box bounds of the reference envs (part of the drop-in contract)."""
    if task == "walk":      # walk_env.py:104-114
        hi = {"ik": 0.4, "ol": 0.01}[signal]
        dim = {"ik": 2, "ol": 8}[signal]
        action = Box(-np.full(dim, hi), np.full(dim, hi))
        ub = np.array([2 * math.pi, 2 * math.pi, 2 * math.pi / sim_dt, 2 * math.pi / sim_dt])   # walk_env.py:364-378
    elif task == "gallop":  # gallop_env.py:119-130 -- low/high are inverted in the reference
        hi = {"ik": 0.4, "ol": 0.3}[signal]
        dim = {"ik": 2, "ol": 4}[signal]
        action = Box(np.full(dim, hi), -np.full(dim, hi))
        ub = np.array([2 * math.pi, 2 * math.pi, 2 * math.pi / sim_dt, 2 * math.pi / sim_dt] + [2 * math.pi] * num_motors)
    elif task == "turn":    # turn_env.py:100-110
        action = Box(-np.full(2, 0.01), np.full(2, 0.01))
        ub = np.array([2 * math.pi, 2 * math.pi, 2 * math.pi / sim_dt, 2 * math.pi / sim_dt])
    elif task == "standup":  # standup_env.py:99-101,200-204 (signal_type is unused by the env)
        action = Box(-np.full(1, 0.1), np.full(1, 0.1))
        ub = np.array([2 * math.pi, 2 * math.pi, 2 * math.pi / sim_dt, 2 * math.pi / sim_dt])
    elif task == "poses":   # poses_env.py:115-117
        action = Box(-np.full(1, 0.1), np.full(1, 0.1))
        ub = np.array([2 * math.pi, 2 * math.pi, 2 * math.pi / sim_dt, 2 * math.pi / sim_dt])
    else:
        raise ValueError(f"unsupported task {task!r}")
    obs = Box(-(ub + OBSERVATION_EPS), ub + OBSERVATION_EPS)  # rex_gym_env.py:277-278
    return action, obs


class _EnvView:
    """`batch_env[i]`: attribute access to one env, as BatchEnv.__getitem__ gives (batch_env.py:44-46)."""

    def __init__(self, batch, index):
        self._batch, self.index = batch, index
        self.action_space, self.observation_space = batch.action_space, batch.observation_space

    def state(self):
        return self._batch.state[:, self.index].clone()


class StepOut:
    """Output tensors of a step, validated once (RexBatchEnv.bind_out): the tensors and their device pointers."""
    __slots__ = ("obs", "reward", "done", "done_bool", "p_obs", "p_reward", "p_done")

    def __init__(self, obs, reward, done, torch):
        self.obs, self.reward, self.done = obs, reward, done
        self.done_bool = done if done.dtype == torch.bool else done.view(torch.bool)
        self.p_obs, self.p_reward, self.p_done = obs.data_ptr(), reward.data_ptr(), done.data_ptr()


class RexBatchEnv:
    metadata = {"render.modes": []}

    def __init__(self, num_envs, task="walk", signal_type="ik", device=0, seed=0, env_index_base=0,
                 auto_reset=False, max_episode_steps=0, backwards=None, target_position=None,
                 target_orient=None, init_orient=None, base_y=None, base_z=None, base_roll=None, base_pitch=None,
                 base_yaw=None,
                 motor_kp=1.0, motor_kd=0.02, control_time_step=None, action_repeat=None, control_latency=0.0,
                 pd_latency=0.0,
                 solver_iterations=None, solver_residual_threshold=None,
                 range_normalize=False, check_actions=True, terrain_type="plane", terrain_pool=64, terrain_seed=10,
                 mark="base", render=False, stream=None, gait_clock_scale=1.0,
                 distance_weight=None, energy_weight=None, drift_weight=None, shake_weight=None,
                 tasks=None, mass_scale_range=None, friction_range=None, observation_noise_stdev=None,
                 heightfield=None, heightfield_cell=None, heightfield_origin=(0.0, 0.0, 0.0), init_height=None,
                 body_contacts=None, on_rack=False, env_randomizer=None, forward_reward_cap=None, use_angle_in_observation=True, **ignored):
        import torch
        # Reference constructor keywords that only touch the GUI, logging or debugging are accepted and ignored; anything
        # else that would change what the env computes is an error here, not a silent no-op.
        harmless = {"debug", "urdf_version", "num_steps_to_log", "log_path", "terrain_id", "urdf_root", "reflection",
                    "draw_foot_path", "hard_reset"}
        unknown = sorted(set(ignored) - harmless)
        if unknown:
            raise TypeError(f"RexBatchEnv: unsupported keyword(s) {unknown}")
        # EnvRandomizer objects (rex_gym_env.py:225,345-346,400-401): host-side hooks, called with this env at every reset()
        # (`randomize_env`) and step() (`randomize_step`, when the object has one); they reach the robot through `self.rex`
        self._env_randomizers = ([] if not env_randomizer else list(env_randomizer) if isinstance(env_randomizer, (list, tuple))
                                 else [env_randomizer])
        if self._env_randomizers and auto_reset:
            raise ValueError("env_randomizer hooks run on the host inside reset(); with auto_reset the resets happen inside the "
                             "launch -- use mass_scale_range / friction_range (per-reset draws in the kernel) instead")
        self._randomize_indices = None
        if terrain_type in ("hills", "mounts", "maze") and heightfield is None:
            raise NotImplementedError(
                f"terrain_type={terrain_type!r}: the reference loads this field from the pip package pybullet_data "
                "(model/terrain.py:55-78), which is not part of rex-gym; with that package at hand: "
                f"RexBatchEnv(..., terrain_type={terrain_type!r}, **rex_gym_amd.terrain.load_reference_terrain({terrain_type!r}, "
                "pybullet_data.getDataPath())) -- or pass the heights yourself as heightfield=<array [ny, nx] in metres>, "
                "heightfield_cell=(cx, cy), heightfield_origin=(x, y, z), init_height=...")
        if terrain_type not in ("plane", "random", "hills", "mounts", "maze", "custom") or mark not in _lib.MARKS or render:
            raise NotImplementedError("terrain_type must be 'plane', 'random' or a heightfield ('hills', 'mounts', 'maze', "
                                      "'custom' with heightfield=...); mark 'base' or 'arm'; render=False")
        if heightfield is not None and terrain_type in ("plane", "random"):
            raise ValueError("heightfield=... goes with terrain_type 'custom' / 'hills' / 'mounts' / 'maze'")
        if task not in _lib.TASKS or signal_type not in _lib.SIGNALS:
            raise ValueError(f"unsupported task/signal {task}/{signal_type}")
        if tasks is not None and task != "mixed":
            raise ValueError("`tasks` goes with task='mixed'")
        self._torch = torch
        self._L = _lib.lib()   # raises if the HIP library is absent: no fallback
        if not torch.cuda.is_available():
            raise _lib.RexSimError("no HIP device visible to PyTorch: RexBatchEnv needs an MI355X (no CPU fallback)")
        self.num_envs = int(num_envs)
        self.task, self.signal_type = task, signal_type
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        cfg = _lib.RexConfig()
        _lib.check(self._L.rex_default_config(_lib.TASKS[task], _lib.SIGNALS[signal_type], self.num_envs,
                                             ctypes.byref(cfg)), "rex_default_config")
        if action_repeat is not None:
            cfg.action_repeat = int(action_repeat)
            cfg.solver_iterations = int(300 / cfg.action_repeat)        # rex_gym_env.py:184
        if control_time_step is not None:
            cfg.sim_time_step = float(control_time_step) / cfg.action_repeat  # rex_gym_env.py:172
        if solver_iterations is not None:
            cfg.solver_iterations = int(solver_iterations)
        if solver_residual_threshold is not None:
            cfg.solver_residual_threshold = float(solver_residual_threshold)
        if task == "mixed":   # BASELINE.json configs[4]: every env runs one task of the mix, drawn per env (REX_TASK_MIXED)
            names = ("walk", "gallop", "turn") if tasks is None else tuple(tasks)
            if not names or any(t not in ("walk", "gallop", "turn") for t in names):
                raise ValueError("tasks must be a non-empty subset of ('walk', 'gallop', 'turn')")
            cfg.task_mix = sum(1 << _lib.TASKS[t] for t in set(names))
            self.tasks = tuple(t for t in ("walk", "gallop", "turn") if t in names)
        for rng, lo_name, hi_name in ((mass_scale_range, "mass_scale_lo", "mass_scale_hi"), (friction_range, "friction_lo", "friction_hi")):
            if rng is not None:   # per-reset draws of the env_randomizer hook (rex_gym_env.py:345-346)
                setattr(cfg, lo_name, float(rng[0])); setattr(cfg, hi_name, float(rng[1]))
        if observation_noise_stdev is not None:   # Rex(observation_noise_stdev=...), rex.py:22: angle, velocity, torque, rpy, rpy rate
            if len(observation_noise_stdev) != 5:
                raise ValueError("observation_noise_stdev has 5 entries (motor angle, velocity, torque, base rpy, base rpy rate)")
            for k, v in enumerate(observation_noise_stdev):
                cfg.noise_stdev[k] = float(v)
        if init_height is None and terrain_type in ("hills", "mounts"):
            init_height = {"hills": 1.98, "mounts": 0.85}[terrain_type]     # ROBOT_INIT_POSITION, terrain.py:14-20
        if init_height is not None:
            cfg.init_height = float(init_height)
        cfg.on_rack = int(bool(on_rack))               # debug rack: fixed base at [0, 0, 1] (rex.py:269-287)
        if body_contacts is not None:   # link collision boxes vs ground and vs the base body, next to the toe rows (include/rexsim.h);
            cfg.body_contacts = int(bool(body_contacts))   # None: the task's default (on for 'poses')
        cfg.motor_kp, cfg.motor_kd = float(motor_kp), float(motor_kd)
        cfg.gait_clock_scale = float(gait_clock_scale)     # wall-clock seconds per simulated second (gait_planner.py:108-110)
        for name, v in (("distance_weight", distance_weight), ("energy_weight", energy_weight),
                        ("drift_weight", drift_weight), ("shake_weight", shake_weight)):   # rex_gym_env.py:56-59
            if v is not None:
                setattr(cfg, name, float(v))
        if not use_angle_in_observation:       # RexReactiveEnv(use_angle_in_observation=False): the four base words alone (gallop_env.py:344-356)
            if task != "gallop":
                raise ValueError("use_angle_in_observation is a keyword of the gallop env (RexReactiveEnv)")
            cfg.gallop_no_angles = 1
        if forward_reward_cap is not None:     # `min(forward_reward, cap)` in the base reward, rex_gym_env.py:81,525
            cfg.forward_reward_cap = float(forward_reward_cap)
        if task == "mixed" and energy_weight is not None:
            raise ValueError("energy_weight is a per-task constant in a mixed batch (gallop 0.005, walk / turn 0.0005)")
        cfg.backwards = -1 if backwards is None else int(bool(backwards))
        cfg.target_position = 0.0 if not target_position else float(target_position)
        if target_orient:                      # `if not self._target_orient` -> drawn (turn_env.py:137)
            cfg.target_orient = float(target_orient); cfg.orient_fixed |= 1
        if init_orient is not None:            # `if self._init_orient is None` -> drawn (turn_env.py:146)
            cfg.init_orient = float(init_orient); cfg.orient_fixed |= 2
        pose_kw = (base_y, base_z, base_roll, base_pitch, base_yaw)
        if task == "poses" and any(v is not None for v in pose_kw):   # fill_next_pose_and_target, poses_env.py:172-187
            vals = [0.0 if v is None else float(v) for v in pose_kw]
            k = next((i for i, v in enumerate(vals[:4]) if v != 0.0), 4)
            cfg.pose_index, cfg.pose_value = k, vals[k]
        cfg.pd_latency, cfg.control_latency = float(pd_latency or 0.0), float(control_latency or 0.0)
        cfg.range_normalize = int(bool(range_normalize))
        cfg.seed = int(seed) & (2 ** 64 - 1)
        cfg.env_index_base = int(env_index_base)
        cfg.auto_reset = int(bool(auto_reset))
        cfg.max_episode_steps = int(max_episode_steps)
        cfg.mark = _lib.MARKS[mark]
        self.mark = mark
        self.config = cfg
        self.num_motors = self._L.rex_num_motors(ctypes.byref(cfg))    # mark_constants.MARK_DETAILS['motors_num']
        self.state_words = self._L.rex_state_words(ctypes.byref(cfg))
        self.action_dim = self._L.rex_action_dim(ctypes.byref(cfg))
        self.obs_dim = self._L.rex_obs_dim(ctypes.byref(cfg))
        if task == "mixed":   # the widest task of the mix; narrower tasks ignore / zero the tail of their rows
            per = {t: _spaces(t, signal_type, cfg.sim_time_step, self.num_motors) for t in self.tasks}
            self.task_action_spaces = {t: sp[0] for t, sp in per.items()}
            wide_a = max(per.values(), key=lambda sp: sp[0].shape[0])[0]
            wide_o = max(per.values(), key=lambda sp: sp[1].shape[0])[1]
            tight = min(float(np.abs(sp[0].high).min()) for sp in per.values())     # inside every task's Box
            self.action_space, self.observation_space = Box(-np.full(wide_a.shape, tight), np.full(wide_a.shape, tight)), wide_o
        else:
            self.action_space, self.observation_space = _spaces(task, signal_type, cfg.sim_time_step, self.num_motors if use_angle_in_observation else 0)
        if range_normalize:   # RangeNormalize / ClipAction expose [-1, 1] boxes (wrappers.py:205-219)
            self.inner_action_space, self.inner_observation_space = self.action_space, self.observation_space
            self.action_space = Box(-np.ones(self.action_space.shape), np.ones(self.action_space.shape))
            self.observation_space = Box(-np.ones(self.observation_space.shape), np.ones(self.observation_space.shape))
        self.control_time_step = cfg.sim_time_step * cfg.action_repeat
        self.check_actions = bool(check_actions)
        self._act_lo = self._act_hi = None
        self._stream = stream
        with torch.cuda.device(self.device):
            self.state = torch.zeros((self.state_words, self.num_envs), dtype=torch.float32, device=self.device)
            self._obs = torch.zeros((self.num_envs, self.obs_dim), dtype=torch.float32, device=self.device)
            self._reward = torch.zeros(self.num_envs, dtype=torch.float32, device=self.device)
            self._done = torch.zeros(self.num_envs, dtype=torch.uint8, device=self.device)
            self._cmd = torch.zeros((self.num_envs, self.num_motors), dtype=torch.float32, device=self.device)
            handle = ctypes.c_void_p()
            _lib.check(self._L.rex_create(ctypes.byref(cfg), self.device.index, self.state.data_ptr(),
                                          self._stream_ptr(), ctypes.byref(handle)), "rex_create")
        self._h = handle
        self._own_out = StepOut(self._obs, self._reward, self._done, torch)
        self._p_cmd, self._info = self._cmd.data_ptr(), {"action": self._cmd}
        self._needs_reset = True
        from .rex_knobs import RexKnobs
        self.rex = RexKnobs(self)
        if cfg.pd_latency > 0 or cfg.control_latency > 0:   # observation-history ring of the latency model (rex.py:122)
            with torch.cuda.device(self.device):
                self.history = torch.zeros((100 * (3 * self.num_motors + 7), self.num_envs), dtype=torch.float32, device=self.device)
            _lib.check(self._L.rex_set_history(self._h, self.history.data_ptr()), "rex_set_history")
        self.terrain_type = terrain_type
        if heightfield is not None:    # model/terrain.py:55-78 style fields, supplied by the caller
            h = np.asarray(heightfield, dtype=np.float32)
            if h.ndim == 2:
                h = h[None]
            if h.ndim != 3 or heightfield_cell is None:
                raise ValueError("heightfield must be [ny, nx] or [k, ny, nx] (metres) and heightfield_cell=(cx, cy) given")
            ox, oy, oz = (float(v) for v in heightfield_origin)
            mids = np.array([0.5 * (float(f.min()) + float(f.max())) - oz for f in h], dtype=np.float32)   # Bullet centres the shape, then places it at z0
            with torch.cuda.device(self.device):
                self.terrain_heights = torch.from_numpy(np.ascontiguousarray(h)).to(self.device)
                self.terrain_mids = torch.from_numpy(mids).to(self.device)
                _lib.check(self._L.rex_set_heightfield(self._h, self.terrain_heights.data_ptr(), self.terrain_mids.data_ptr(),
                                                       int(h.shape[0]), int(h.shape[2]), int(h.shape[1]), float(heightfield_cell[0]),
                                                       float(heightfield_cell[1]), ox, oy, self._stream_ptr()), "rex_set_heightfield")
        if terrain_type == "random":   # model/terrain.py:32-54 -- a pool of fields instead of one per env
            from ..terrain import random_terrain_pool
            h, m = random_terrain_pool(int(terrain_pool), int(terrain_seed))
            with torch.cuda.device(self.device):
                self.terrain_heights = torch.from_numpy(h).to(self.device)
                self.terrain_mids = torch.from_numpy(m).to(self.device)
                _lib.check(self._L.rex_set_terrain(self._h, self.terrain_heights.data_ptr(), self.terrain_mids.data_ptr(),
                                                   int(terrain_pool), self._stream_ptr()), "rex_set_terrain")

    # ---- plumbing ----
    def _stream_ptr(self):
        s = self._stream if self._stream is not None else self._torch.cuda.current_stream(self.device)
        return ctypes.c_void_p(s.cuda_stream)

    def task_ids(self):
        """task='mixed': the task every env runs, as a device int32 tensor of REX_TASK_* ids (drawn per env from its global
        index, fixed for the env's life -- the same draw the kernels make)."""
        torch = self._torch
        if self.task != "mixed":
            return torch.full((self.num_envs,), _lib.TASKS[self.task], dtype=torch.int32, device=self.device)
        from .philox import philox4x32
        g = np.arange(self.num_envs, dtype=np.uint32) + np.uint32(self.config.env_index_base)
        ctr = np.stack([np.full_like(g, 0xFFFFFFFF), g, np.full_like(g, 2), np.zeros_like(g)])
        out = philox4x32(ctr, np.uint32(self.config.seed & 0xFFFFFFFF), np.uint32(self.config.seed >> 32))
        ids = np.array([_lib.TASKS[t] for t in self.tasks], dtype=np.int32)
        return torch.from_numpy(ids[(out[0] % np.uint32(len(ids))).astype(np.int64)]).to(self.device)

    def _on_stream(self):
        """Context in which host-side staging (dtype / device conversion of actions and indices) is issued: the stream the
        kernels are launched on, so that a RexBatchEnv(stream=...) orders its copies with its launches by itself."""
        import contextlib
        return self._torch.cuda.stream(self._stream) if self._stream is not None else contextlib.nullcontext()

    def __len__(self):
        return self.num_envs

    def __getitem__(self, index):
        return _EnvView(self, index)

    def seed(self, seed=None):
        return [seed]

    def close(self):
        if getattr(self, "_h", None):
            self._torch.cuda.synchronize(self.device)
            self._L.rex_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_event_trace(self, enable=True):
        """Debug (rex_set_event_trace): fold the discrete events of every physics substep -- toe points in reach, the heightfield
        facets under them, joint / arm bounds reached -- into a chained hash per env, and the solver sweep counts into a second
        one, and the events without the arm's bounds into a third.  Returns the device tensor uint32-as-int32 [3, num_envs] the launches
        update (zeroed here); False switches it off."""
        torch = self._torch
        if not enable:
            _lib.check(self._L.rex_set_event_trace(self._h, None), "rex_set_event_trace")
            self._trace = None
            return None
        self._trace = torch.zeros((3, self.num_envs), dtype=torch.int32, device=self.device)
        _lib.check(self._L.rex_set_event_trace(self._h, self._trace.data_ptr()), "rex_set_event_trace")
        return self._trace

    def set_timing(self, enable=True):
        """True / 1: time the last launch (last_step_ms); 2: a ring of the last 256 launches, no sync in between (step_times_ms)."""
        _lib.check(self._L.rex_set_timing(self._h, int(enable)), "rex_set_timing")

    def step_times_ms(self, count=256):
        buf = (ctypes.c_float * int(count))()
        n = self._L.rex_step_times_ms(self._h, buf, int(count))
        _lib.check(min(n, 0), "rex_step_times_ms")
        return list(buf[:n])

    def last_step_ms(self):
        ms = ctypes.c_float()
        _lib.check(self._L.rex_last_step_ms(self._h, ctypes.byref(ms)), "rex_last_step_ms")
        return ms.value

    def set_body_params(self, base_mass_scale=None, leg_mass_scale=None, foot_friction=None):
        """Per-env domain randomisation (the knobs of Rex.SetBaseMasses / SetLegMasses, rex.py:659-692, plus the foot
        friction). Each argument: None (keep), a float, or an [N] tensor/array. The values live in `self.body_params`
        ([3, N] device tensor) and may be edited in place between steps."""
        torch = self._torch
        if getattr(self, "body_params", None) is None:
            self.body_params = torch.tensor([[1.0], [1.0], [0.5]], device=self.device).repeat(1, self.num_envs).contiguous()
            _lib.check(self._L.rex_set_body_params(self._h, self.body_params.data_ptr()), "rex_set_body_params")
        for row, v in enumerate((base_mass_scale, leg_mass_scale, foot_friction)):
            if v is not None:
                self.body_params[row] = torch.as_tensor(v, dtype=torch.float32, device=self.device)
        return self.body_params

    # ---- Gym / BatchEnv surface ----
    def reset(self, indices=None):
        """Reset all envs (indices=None) or the given ones; returns their first observations."""
        torch = self._torch
        if indices is None:
            _lib.check(self._L.rex_reset(self._h, None, 0, self._obs.data_ptr(), self._stream_ptr()), "rex_reset")
            self._needs_reset = False
            self._randomize(None)
            return self._obs.clone()
        with self._on_stream():
            idx = torch.as_tensor(indices, dtype=torch.int32, device=self.device).contiguous()
            obs = torch.empty((idx.numel(), self.obs_dim), dtype=torch.float32, device=self.device)
            if idx.numel():
                _lib.check(self._L.rex_reset(self._h, idx.data_ptr(), idx.numel(), obs.data_ptr(), self._stream_ptr()),
                           "rex_reset")
                self._randomize(idx)
        return obs

    def add_env_randomizer(self, env_randomizer):     # rex_gym_env.py:293-294
        self._env_randomizers.append(env_randomizer)

    def _randomize(self, idx):
        # after Rex.Reset, as in the reference (rex_gym_env.py:341-346): the reset motion is the nominal robot's
        self._randomize_indices = idx
        for r in self._env_randomizers:
            r.randomize_env(self)
        self._randomize_indices = None

    def step(self, actions, out=None):
        """actions [N, action_dim] (device float32 tensor, or array-like). Returns device tensors
        (obs [N,O], reward [N], done [N] bool) and info {'action': motor command [N,12]}.

        out=(obs [N,O] float32, reward [N] float32, done [N] uint8 or bool): the launch writes its results straight into
        these contiguous device tensors -- e.g. slice t of a rollout segment [T, N, ...] that is handed to the learner
        (sharding.gather_rollout) -- instead of the env's own buffers; nothing is copied."""
        torch = self._torch
        if self._needs_reset:
            raise RuntimeError("Must reset environment.")   # wrappers.py:286-288 semantics
        for r in self._env_randomizers:                     # rex_gym_env.py:400-401
            if hasattr(r, "randomize_step"):
                r.randomize_step(self)
        with self._on_stream():
            a = torch.as_tensor(actions, dtype=torch.float32, device=self.device)
            if a.shape != (self.num_envs, self.action_dim):
                raise ValueError(f"actions must have shape {(self.num_envs, self.action_dim)}, got {tuple(a.shape)}")
            a = a.contiguous()
            if self.check_actions:
                # BatchEnv.step: `if not env.action_space.contains(action): raise ValueError` for every env (agents/tools/
                # batch_env.py:76-79), in the Box's own dtype (float32).  Behind the folded ClipAction the space is unbounded
                # (wrappers.py:254-259): only non-finite actions are invalid.  The verdict is read back from the device, i.e. one
                # host synchronisation per step, as in the reference (issued on the env's own stream, behind the staging copy
                # above); throughput loops that have validated their actions beforehand pass check_actions=False.
                bad = ~torch.isfinite(a)
                if not self.config.range_normalize:
                    lo, hi = self._action_bounds()
                    bad |= (a < lo) | (a > hi)
                rows = bad.any(dim=1)
                if bool(rows.any()):
                    i = int(torch.nonzero(rows)[0])
                    raise ValueError(f"Invalid action at index {i}: {a[i].tolist()}")
        if out is None:
            out = self._own_out
        elif not isinstance(out, StepOut):
            out = self.bind_out(*out)
        _lib.check(self._L.rex_step(self._h, a.data_ptr(), out.p_obs, out.p_reward, out.p_done, self._p_cmd, self._stream_ptr()), "rex_step")
        return out.obs, out.reward, out.done_bool, self._info

    def _action_bounds(self):
        if self._act_lo is None:
            torch = self._torch
            self._act_lo = torch.as_tensor(np.minimum(self.action_space.low, self.action_space.high).astype(np.float32), device=self.device)
            self._act_hi = torch.as_tensor(np.maximum(self.action_space.low, self.action_space.high).astype(np.float32), device=self.device)
        return self._act_lo, self._act_hi

    def step_segment(self, actions, out=None, motor_cmd=None):
        """A rollout segment in ONE launch (rex_step_segment): actions [T, N, action_dim] that the caller already holds -- an
        open-loop rollout: random-action throughput runs, a replayed action tape -- gives (obs [T, N, O], reward [T, N],
        done [T, N] bool), bit-identical to T calls of step() on the slices.  A wave goes on to its envs' next step as soon as it
        has finished this one, so the launch ends with the largest sum of a wave's step times instead of paying the slowest wave of
        every step.  The reference has no counterpart (its envs take one action at a time); step() is unchanged.

        out=(obs, reward, done): contiguous device tensors of those shapes (done uint8 or bool) the launch writes into.
        motor_cmd: optional [T, N, num_motors] float32 tensor for info['action'] of every step (without it info['action'] is None:
        a segment does not record the motor commands unless asked to).  Per-step randomizers
        (randomize_step) and the per-step Box test do not fit a segment: envs that use them raise."""
        torch = self._torch
        if self._needs_reset:
            raise RuntimeError("Must reset environment.")
        if any(hasattr(r, "randomize_step") for r in self._env_randomizers):
            raise ValueError("step_segment: a randomizer with randomize_step() acts between steps; use step()")
        with self._on_stream():
            a = torch.as_tensor(actions, dtype=torch.float32, device=self.device)
            if a.dim() != 3 or a.shape[1:] != (self.num_envs, self.action_dim) or a.shape[0] < 1:
                raise ValueError(f"actions must have shape (T, {self.num_envs}, {self.action_dim}), got {tuple(a.shape)}")
            a = a.contiguous()
            T = int(a.shape[0])
            if self.check_actions:     # the Box test of BatchEnv.step on the whole segment up front: one host synchronisation
                bad = ~torch.isfinite(a)
                if not self.config.range_normalize:
                    lo, hi = self._action_bounds()
                    bad |= (a < lo) | (a > hi)
                if bool(bad.any()):
                    t, i = (int(v) for v in torch.nonzero(bad.any(dim=2))[0])
                    raise ValueError(f"Invalid action at step {t}, index {i}: {a[t, i].tolist()}")
            if out is None:
                out = (torch.empty((T, self.num_envs, self.obs_dim), dtype=torch.float32, device=self.device),
                       torch.empty((T, self.num_envs), dtype=torch.float32, device=self.device),
                       torch.empty((T, self.num_envs), dtype=torch.uint8, device=self.device))
            obs, reward, done = out
            for t_, shape, dt in ((obs, (T, self.num_envs, self.obs_dim), (torch.float32,)), (reward, (T, self.num_envs), (torch.float32,)),
                                  (done, (T, self.num_envs), (torch.uint8, torch.bool))):
                if tuple(t_.shape) != shape or t_.dtype not in dt or not t_.is_contiguous() or t_.device != self.device:
                    raise ValueError(f"out tensors must be contiguous on {self.device}: obs {(T, self.num_envs, self.obs_dim)} float32, "
                                     f"reward {(T, self.num_envs)} float32, done {(T, self.num_envs)} uint8 / bool")
            p_cmd = None
            if motor_cmd is not None:
                nm = self._info["action"].shape[1]
                if tuple(motor_cmd.shape) != (T, self.num_envs, nm) or motor_cmd.dtype != torch.float32 or not motor_cmd.is_contiguous() \
                        or motor_cmd.device != self.device:
                    raise ValueError(f"motor_cmd must be a contiguous float32 tensor of shape {(T, self.num_envs, nm)} on {self.device}")
                p_cmd = motor_cmd.data_ptr()
        _lib.check(self._L.rex_step_segment(self._h, T, a.data_ptr(), obs.data_ptr(), reward.data_ptr(), done.data_ptr(), p_cmd, self._stream_ptr()),
                   "rex_step_segment")
        return obs, reward, (done if done.dtype == torch.bool else done.view(torch.bool)), {"action": motor_cmd}

    # ---- the actor inside the launch (rex_set_policy / rex_step_policy / rex_step_segment_policy) ----
    def set_policy(self, w1=None, b1=None, w2=None, b2=None, w3=None, b3=None, logstd=None, obs_mean=None, obs_scale=None, obs_clip=5.0, sample=True, seed=0):
        """Install the actor the closed-loop launches evaluate (include/rexsim.h `RexPolicy`): the reference's ForwardGaussianPolicy
        (agents/scripts/networks.py:66-110) behind its observ filter (agents/ppo/normalize.py:47-66).  Contiguous float32 device
        tensors, INPUT-major weights: w1 [obs_dim, h1], w2 [h1, h2], w3 [h2, action_dim] (the transpose of torch.nn.Linear.weight),
        biases and logstd as vectors, obs_mean / obs_scale [obs_dim] or both None.  The library snapshots them (a packing kernel on the
        env's stream): call set_policy again after the weights or the filter statistics have changed; set_policy(None) removes the policy.
        The env must have been created with range_normalize=True (the agents act through RangeNormalize + ClipAction)."""
        torch = self._torch
        if w1 is None:
            _lib.check(self._L.rex_set_policy(self._h, None, None), "rex_set_policy")
            self._policy = None
            return
        ts = dict(w1=w1, b1=b1, w2=w2, b2=b2, w3=w3, b3=b3, logstd=logstd)
        if (obs_mean is None) != (obs_scale is None):
            raise ValueError("obs_mean and obs_scale go together")
        if obs_mean is not None:
            ts.update(obs_mean=obs_mean, obs_scale=obs_scale)
        for name, t in ts.items():
            if not isinstance(t, torch.Tensor) or t.dtype != torch.float32 or t.device != self.device or not t.is_contiguous():
                raise ValueError(f"set_policy: {name} must be a contiguous float32 tensor on {self.device}")
        h1, h2 = int(w1.shape[1]) if w1.dim() == 2 else -1, int(w2.shape[1]) if w2.dim() == 2 else -1
        want = dict(w1=(self.obs_dim, h1), b1=(h1,), w2=(h1, h2), b2=(h2,), w3=(h2, self.action_dim), b3=(self.action_dim,),
                    logstd=(self.action_dim,), obs_mean=(self.obs_dim,), obs_scale=(self.obs_dim,))
        for name, t in ts.items():
            if tuple(t.shape) != want[name]:
                raise ValueError(f"set_policy: {name} has shape {tuple(t.shape)}, expected {want[name]}")
        pol = _lib.RexPolicy()
        pol.obs_dim, pol.action_dim, pol.hidden1, pol.hidden2 = self.obs_dim, self.action_dim, h1, h2
        for name in ("w1", "b1", "w2", "b2", "w3", "b3", "logstd", "obs_mean", "obs_scale"):
            setattr(pol, "d_" + name, ts[name].data_ptr() if name in ts else None)
        pol.obs_clip, pol.sample, pol.seed = float(obs_clip), int(bool(sample)), int(seed) & (2 ** 64 - 1)
        _lib.check(self._L.rex_set_policy(self._h, ctypes.byref(pol), self._stream_ptr()), "rex_set_policy")
        self._policy = True

    def _policy_blocks(self, T, obs_in, out, action, mean, motor_cmd):
        torch = self._torch
        if self._needs_reset:
            raise RuntimeError("Must reset environment.")
        if getattr(self, "_policy", None) is None:
            raise RuntimeError("no policy set: call set_policy(...) first")
        if any(hasattr(r, "randomize_step") for r in self._env_randomizers):
            raise ValueError("a randomizer with randomize_step() acts between steps on the host; use step()")
        n, lead = self.num_envs, ((T,) if T is not None else ())
        def block(t, shape, dts, name):
            if t is None:
                return torch.empty(lead + shape, dtype=dts[0], device=self.device)
            if tuple(t.shape) != lead + shape or t.dtype not in dts or not t.is_contiguous() or t.device != self.device:
                raise ValueError(f"{name} must be a contiguous {dts[0]} tensor of shape {lead + shape} on {self.device}")
            return t
        if not isinstance(obs_in, torch.Tensor) or tuple(obs_in.shape) != (n, self.obs_dim) or obs_in.dtype != torch.float32 \
                or not obs_in.is_contiguous() or obs_in.device != self.device:
            raise ValueError(f"obs_in must be a contiguous float32 tensor of shape {(n, self.obs_dim)} on {self.device}")
        o, r, d = out if out is not None else (None, None, None)
        o = block(o, (n, self.obs_dim), (torch.float32,), "out obs")
        r = block(r, (n,), (torch.float32,), "out reward")
        d = block(d, (n,), (torch.uint8, torch.bool), "out done")
        action = block(action, (n, self.action_dim), (torch.float32,), "action")
        mean = block(mean, (n, self.action_dim), (torch.float32,), "mean")
        if motor_cmd is not None:
            motor_cmd = block(motor_cmd, (n, self.num_motors), (torch.float32,), "motor_cmd")
        return obs_in, o, r, d, action, mean, motor_cmd

    def step_policy(self, obs_in, out=None, action=None, mean=None, motor_cmd=None):
        """One CLOSED-LOOP env.step() of every env in one launch (rex_step_policy): action = policy(obs_in) -- obs_in [N, obs_dim]
        is what the envs returned last (reset() or the previous step) --, then the step.  Returns (obs, reward, done, info) as
        step(); info['policy_action'] [N, A] is the action the policy took as the agent's memory stores it (before ClipAction /
        RangeNormalize), info['policy_mean'] its mean, info['action'] the motor commands (the tensor passed as motor_cmd, else None)."""
        torch = self._torch
        obs_in, o, r, d, action, mean, motor_cmd = self._policy_blocks(None, obs_in, out, action, mean, motor_cmd)
        _lib.check(self._L.rex_step_policy(self._h, obs_in.data_ptr(), action.data_ptr(), mean.data_ptr(), o.data_ptr(), r.data_ptr(), d.data_ptr(),
                                           motor_cmd.data_ptr() if motor_cmd is not None else None, self._stream_ptr()), "rex_step_policy")
        return o, r, (d if d.dtype == torch.bool else d.view(torch.bool)), {"action": motor_cmd, "policy_action": action, "policy_mean": mean}

    def step_segment_policy(self, num_steps, obs_in, out=None, action=None, mean=None, motor_cmd=None):
        """A closed-loop rollout SEGMENT in one launch (rex_step_segment_policy): T = num_steps consecutive step_policy() calls, step t
        acting on the observation of step t - 1 (obs_in for t = 0), bit-identical to them.  Blocks [T, N, ...]: returns (obs, reward,
        done, info) with info['policy_action'] / ['policy_mean'] [T, N, A].  A caller that keeps obs[T + 1, N, O] passes
        obs_in=obs[0], out=(obs[1:], reward, done): the policy's input of step t is obs[t]."""
        torch = self._torch
        T = int(num_steps)
        if T < 1:
            raise ValueError("num_steps must be at least 1")
        obs_in, o, r, d, action, mean, motor_cmd = self._policy_blocks(T, obs_in, out, action, mean, motor_cmd)
        _lib.check(self._L.rex_step_segment_policy(self._h, T, obs_in.data_ptr(), action.data_ptr(), mean.data_ptr(), o.data_ptr(), r.data_ptr(),
                                                   d.data_ptr(), motor_cmd.data_ptr() if motor_cmd is not None else None, self._stream_ptr()),
                   "rex_step_segment_policy")
        return o, r, (d if d.dtype == torch.bool else d.view(torch.bool)), {"action": motor_cmd, "policy_action": action, "policy_mean": mean}

    def bind_out(self, obs, reward, done):
        """Validate a set of output tensors once (`step(..., out=...)`): returns a StepOut that a loop can pass again and again."""
        torch = self._torch
        for t, shape, dt in ((obs, (self.num_envs, self.obs_dim), (torch.float32,)), (reward, (self.num_envs,), (torch.float32,)),
                             (done, (self.num_envs,), (torch.uint8, torch.bool))):
            if tuple(t.shape) != shape or t.dtype not in dt or not t.is_contiguous() or t.device != self.device:
                raise ValueError(f"out tensors must be contiguous on {self.device}: obs {(self.num_envs, self.obs_dim)} float32, "
                                 f"reward {(self.num_envs,)} float32, done {(self.num_envs,)} uint8 / bool")
        return StepOut(obs, reward, done, torch)
