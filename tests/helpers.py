"""Shared helpers of the GPU parity tests: state conversion between the product's float32/bit-cast
SoA layout (include/rexsim.h) and the oracle's numeric layout, and lock-step rollouts."""
import numpy as np

import orclib



def product_state_to_numeric(state_tensor):
    """torch [53, N] float32 (ints bit-cast) -> numpy float64 [53, N] with ints as numbers."""
    raw = state_tensor.detach().cpu().numpy()
    out = raw.astype(np.float64)
    ints = raw.view(np.int32)
    for w in orclib.int_words(raw.shape[0]):
        out[w] = ints[w].astype(np.uint32).astype(np.float64)
    return out


def numeric_to_product_state(num, torch, device):
    raw = np.ascontiguousarray(num, dtype=np.float32).copy()
    ints = raw.view(np.uint32)
    for w in orclib.int_words(raw.shape[0]):
        ints[w] = np.asarray(num[w], dtype=np.uint32)
    return torch.from_numpy(raw).to(device)


def make_pair(task, signal, n, dtype=np.float32, **kw):
    """(RexBatchEnv on the GPU, OracleEnv) with identical configs."""
    from rex_gym_amd import RexBatchEnv
    okw = dict(kw)
    pkw = dict(kw)
    cfg_kw = {}
    for k in ("seed", "auto_reset", "max_episode_steps", "env_index_base", "motor_kp", "motor_kd", "range_normalize", "pd_latency",
              "control_latency", "gait_clock_scale", "body_contacts", "on_rack"):
        if k in okw:
            cfg_kw[k] = okw.pop(k)
    if "backwards" in okw:
        b = okw.pop("backwards")
        cfg_kw["backwards"] = -1 if b is None else int(b)
    for k in ("target_orient", "init_orient"):
        if k in okw and okw[k] is not None:
            cfg_kw[k] = float(okw.pop(k))
            cfg_kw["orient_fixed"] = cfg_kw.get("orient_fixed", 0) | (1 if k == "target_orient" else 2)
    if "base_roll" in okw:
        cfg_kw["pose_index"], cfg_kw["pose_value"] = 2, float(okw.pop("base_roll"))
    if "target_position" in okw:
        t = okw.pop("target_position")
        cfg_kw["target_position"] = 0.0 if not t else float(t)
    terrain = okw.pop("terrain_type", "plane")
    pool, tseed = okw.pop("terrain_pool", 64), okw.pop("terrain_seed", 10)
    mark = okw.pop("mark", "base")
    okw.pop("check_actions", None)     # product-side only: BatchEnv's per-step Box test (tests that leave the Box switch it off)
    cfg_kw["mark"] = {"base": 0, "arm": 1}[mark]
    assert not okw, okw
    env = RexBatchEnv(n, task=task, signal_type=signal, **pkw)
    cfg = orclib.default_config(task, signal, n, **cfg_kw)
    orc = orclib.OracleEnv(cfg, dtype, mark)
    if terrain == "random":
        from rex_gym_amd.terrain import random_terrain_pool
        orc.set_terrain(*random_terrain_pool(pool, tseed))
    return env, orc


def joint_rmse(a, b):
    """RMSE over the 12 joint angles, per env: a, b numeric states [53, N]."""
    nm = {54: 12, 69: 18}[a.shape[0]]
    d = a[orclib.S_Q:orclib.S_Q + nm] - b[orclib.S_Q:orclib.S_Q + nm]
    return np.sqrt(np.mean(d * d, axis=0))
