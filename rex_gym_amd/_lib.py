"""ctypes binding of the C ABI in include/rexsim.h (librexsim_hip.so).

There is NO CPU fallback: if the HIP library is missing or a call fails, this raises.
"""
import ctypes
import os

from . import build as _build

TASKS = {"walk": 0, "gallop": 1, "turn": 2, "poses": 3, "standup": 4, "mixed": 5}
SIGNALS = {"ik": 0, "ol": 1}
MARKS = {"base": 0, "arm": 1}
STATE_WORDS = 54      # mark 'base'; rex_state_words(cfg) for the others
NUM_MOTORS = 12       # mark 'base'; rex_num_motors(cfg) for the others
ABI_VERSION = 6


class RexConfig(ctypes.Structure):
    """Mirror of `struct RexConfig` (include/rexsim.h)."""
    _fields_ = [
        ("abi_version", ctypes.c_int32), ("num_envs", ctypes.c_int32), ("env_index_base", ctypes.c_int32),
        ("task", ctypes.c_int32), ("signal", ctypes.c_int32), ("action_repeat", ctypes.c_int32),
        ("solver_iterations", ctypes.c_int32), ("sim_time_step", ctypes.c_float),
        ("motor_kp", ctypes.c_float), ("motor_kd", ctypes.c_float), ("backwards", ctypes.c_int32),
        ("target_position", ctypes.c_float), ("seed", ctypes.c_uint64), ("auto_reset", ctypes.c_int32),
        ("max_episode_steps", ctypes.c_int32), ("distance_weight", ctypes.c_float),
        ("energy_weight", ctypes.c_float), ("drift_weight", ctypes.c_float), ("shake_weight", ctypes.c_float),
        ("solver_residual_threshold", ctypes.c_float), ("target_orient", ctypes.c_float),
        ("init_orient", ctypes.c_float), ("orient_fixed", ctypes.c_int32), ("pose_index", ctypes.c_int32),
        ("pose_value", ctypes.c_float), ("range_normalize", ctypes.c_int32), ("pd_latency", ctypes.c_float), ("control_latency", ctypes.c_float),
        ("mark", ctypes.c_int32),
        ("gait_clock_scale", ctypes.c_float), ("body_contacts", ctypes.c_int32), ("noise_stdev", ctypes.c_float * 5),
        ("task_mix", ctypes.c_int32), ("mass_scale_lo", ctypes.c_float), ("mass_scale_hi", ctypes.c_float),
        ("friction_lo", ctypes.c_float), ("friction_hi", ctypes.c_float), ("init_height", ctypes.c_float), ("on_rack", ctypes.c_int32),
        ("forward_reward_cap", ctypes.c_float), ("gallop_no_angles", ctypes.c_int32),
    ]


class RexPolicy(ctypes.Structure):
    """Mirror of `struct RexPolicy` (include/rexsim.h): the actor evaluated inside the launch."""
    _fields_ = [
        ("obs_dim", ctypes.c_int32), ("action_dim", ctypes.c_int32), ("hidden1", ctypes.c_int32), ("hidden2", ctypes.c_int32),
        ("d_w1", ctypes.c_void_p), ("d_b1", ctypes.c_void_p), ("d_w2", ctypes.c_void_p), ("d_b2", ctypes.c_void_p),
        ("d_w3", ctypes.c_void_p), ("d_b3", ctypes.c_void_p), ("d_logstd", ctypes.c_void_p),
        ("d_obs_mean", ctypes.c_void_p), ("d_obs_scale", ctypes.c_void_p),
        ("obs_clip", ctypes.c_float), ("sample", ctypes.c_int32), ("seed", ctypes.c_uint64),
    ]


class RexSimError(RuntimeError):
    pass


_lib = None

_SIGS = {
    "rex_default_config": ([ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(RexConfig)], ctypes.c_int),
    "rex_action_dim": ([ctypes.POINTER(RexConfig)], ctypes.c_int),
    "rex_obs_dim": ([ctypes.POINTER(RexConfig)], ctypes.c_int),
    "rex_num_motors": ([ctypes.POINTER(RexConfig)], ctypes.c_int),
    "rex_state_words": ([ctypes.POINTER(RexConfig)], ctypes.c_int),
    "rex_create": ([ctypes.POINTER(RexConfig), ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                    ctypes.POINTER(ctypes.c_void_p)], ctypes.c_int),
    "rex_destroy": ([ctypes.c_void_p], ctypes.c_int),
    "rex_set_body_params": ([ctypes.c_void_p, ctypes.c_void_p], ctypes.c_int),
    "rex_set_history": ([ctypes.c_void_p, ctypes.c_void_p], ctypes.c_int),
    "rex_set_terrain": ([ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p], ctypes.c_int),
    "rex_set_heightfield": ([ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                             ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p], ctypes.c_int),
    "rex_reset": ([ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p], ctypes.c_int),
    "rex_step": ([ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                  ctypes.c_void_p, ctypes.c_void_p], ctypes.c_int),
    "rex_step_segment": ([ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                          ctypes.c_void_p, ctypes.c_void_p], ctypes.c_int),
    "rex_set_policy": ([ctypes.c_void_p, ctypes.POINTER(RexPolicy), ctypes.c_void_p], ctypes.c_int),
    "rex_step_policy": ([ctypes.c_void_p] + [ctypes.c_void_p] * 8, ctypes.c_int),
    "rex_step_segment_policy": ([ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 8, ctypes.c_int),
    "rex_set_timing": ([ctypes.c_void_p, ctypes.c_int], ctypes.c_int),
    "rex_last_step_ms": ([ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)], ctypes.c_int),
    "rex_ik_solve": ([ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                      ctypes.c_void_p], ctypes.c_int),
    "rex_motor_torque": ([ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                          ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p],
                         ctypes.c_int),
    "rex_gait_loop": ([ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                       ctypes.c_void_p], ctypes.c_int),
    "rex_set_event_trace": ([ctypes.c_void_p, ctypes.c_void_p], ctypes.c_int),
    "rex_envs_per_wave": ([ctypes.c_void_p], ctypes.c_int),
    "rex_mixed_slot_map": ([ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int], ctypes.c_int),
    "rex_get_sweeps": ([ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p], ctypes.c_int),
    "rex_step_times_ms": ([ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.c_int], ctypes.c_int),
    "rex_last_error": ([], ctypes.c_char_p),
    "rex_abi_version": ([], ctypes.c_int),
}
EXPORTED_SYMBOLS = tuple(_SIGS)


def lib():
    """Load (once) the HIP library. Raises RexSimError if it is not there -- never falls back."""
    global _lib
    if _lib is None:
        path = _build.LIB_PATH
        if not os.path.exists(path):
            raise RexSimError(
                f"{path} is missing: build it with `python -m rex_gym_amd.build` (needs hipcc). "
                "rex_gym_amd has no CPU fallback.")
        # PyTorch first: it carries its own HIP runtime, and a process must not end up with two of them (the library would
        # bind to the system one if it were loaded before torch, and the second runtime to start sees no device)
        import torch  # noqa: F401
        l = ctypes.CDLL(path)
        l.rex_abi_version.restype = ctypes.c_int
        if l.rex_abi_version() != ABI_VERSION:
            raise RexSimError(f"{path}: ABI version {l.rex_abi_version()}, this package binds version {ABI_VERSION}: rebuild it "
                              "(`python -m rex_gym_amd.build`)")
        for name, (argtypes, restype) in _SIGS.items():
            try:
                fn = getattr(l, name)
            except AttributeError:      # the ABI header and the library disagree
                raise RexSimError(f"{path} does not export {name}: a stale build -- rebuild it (`python -m rex_gym_amd.build`)") from None
            fn.argtypes = argtypes
            fn.restype = restype
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().rex_last_error().decode("utf-8", "replace")
        raise RexSimError(f"{what} failed (code {rc}): {msg}")
