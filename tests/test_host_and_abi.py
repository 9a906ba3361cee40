"""CPU-side checks that need no GPU: the C-ABI library loads and exports every symbol the header
declares, config defaults mirror the reference, the Gym surface (spaces) matches, and the product
fails loudly without a device (no silent CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from rex_gym_amd import _lib, build
    if not os.path.exists(build.LIB_PATH):
        build.build()
    return _lib


def test_library_exports_every_declared_symbol(L):
    hdr = open(os.path.join(ROOT, "include", "rexsim.h")).read()
    declared = set(re.findall(r"REX_API\s+[\w\s\*]+?\b(rex_\w+)\s*\(", hdr))
    assert len(declared) >= 14
    assert declared == set(L.EXPORTED_SYMBOLS), declared ^ set(L.EXPORTED_SYMBOLS)
    lib = L.lib()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.rex_abi_version() == L.ABI_VERSION == 6


def test_default_config_matches_reference_constants(L):
    lib = L.lib()
    c = L.RexConfig()
    assert lib.rex_default_config(L.TASKS["walk"], L.SIGNALS["ik"], 7, ctypes.byref(c)) == 0
    # walk_env.py:34-40, rex_gym_env.py:25,56-59,172,184
    assert (c.action_repeat, c.solver_iterations, c.num_envs) == (5, 60, 7)
    assert abs(c.sim_time_step - 0.001) < 1e-9 and abs(c.motor_kp - 1.0) < 1e-9 and abs(c.motor_kd - 0.02) < 1e-9
    assert c.backwards == -1 and c.target_position == 0.0
    assert [round(x, 6) for x in (c.distance_weight, c.energy_weight, c.drift_weight, c.shake_weight)] == [1.0, 0.0005, 2.0, 0.005]
    assert lib.rex_action_dim(ctypes.byref(c)) == 2 and lib.rex_obs_dim(ctypes.byref(c)) == 4
    assert lib.rex_default_config(L.TASKS["gallop"], L.SIGNALS["ol"], 3, ctypes.byref(c)) == 0
    assert (c.action_repeat, c.solver_iterations) == (6, 50)          # gallop_env.py:47-48
    assert abs(c.energy_weight - 0.005) < 1e-9                         # gallop_env.py:45
    assert lib.rex_action_dim(ctypes.byref(c)) == 4 and lib.rex_obs_dim(ctypes.byref(c)) == 16
    assert lib.rex_num_motors(ctypes.byref(c)) == 12 and lib.rex_state_words(ctypes.byref(c)) == 54
    c.mark = L.MARKS["arm"]                                           # mark_constants.py MARK_DETAILS['motors_num']
    assert lib.rex_num_motors(ctypes.byref(c)) == 18 and lib.rex_state_words(ctypes.byref(c)) == 69
    assert lib.rex_obs_dim(ctypes.byref(c)) == 22
    assert lib.rex_default_config(L.TASKS["walk"], L.SIGNALS["ol"], 3, ctypes.byref(c)) == 0
    assert lib.rex_action_dim(ctypes.byref(c)) == 8


def test_config_struct_layout_matches_oracle_binding(L):
    import orclib
    assert ctypes.sizeof(L.RexConfig) == ctypes.sizeof(orclib.RexConfig)
    assert [f[0] for f in L.RexConfig._fields_] == [f[0] for f in orclib.RexConfig._fields_]
    oc = orclib.default_config("walk", "ik", 5)
    pc = L.RexConfig()
    L.lib().rex_default_config(0, 0, 5, ctypes.byref(pc))
    for name, _ in L.RexConfig._fields_:
        if name == "reserved":
            continue
        a, b = getattr(oc, name), getattr(pc, name)
        if isinstance(a, ctypes.Array):
            a, b = list(a), list(b)
        assert a == b, name


def test_bad_arguments_return_error_codes(L):
    lib = L.lib()
    c = L.RexConfig()
    assert lib.rex_default_config(9, 0, 4, ctypes.byref(c)) < 0
    assert b"unsupported" in lib.rex_last_error()
    assert lib.rex_default_config(0, 0, 0, ctypes.byref(c)) < 0
    assert lib.rex_step(None, None, None, None, None, None, None) < 0
    assert lib.rex_ik_solve(0, None, None, None, None, None) < 0


def test_segment_entry_point_rejects_bad_arguments_without_a_gpu(L):
    """rex_step / rex_step_segment check their pointers before anything touches a device: null arguments come back as REX_EINVAL
    with a message on any host (the segment's other checks -- num_steps >= 1, 32-bit block offsets -- need a sim, i.e. a GPU)."""
    lib = L.lib()
    assert lib.rex_step(None, None, None, None, None, None, None) < 0 and b"rex_step" in lib.rex_last_error()
    assert lib.rex_step_segment(None, 3, None, None, None, None, None, None) < 0 and b"rex_step_segment" in lib.rex_last_error()


def test_policy_entry_points_reject_bad_arguments_without_a_gpu(L):
    """rex_set_policy / rex_step_policy / rex_step_segment_policy (ABI 6: the actor inside the launch) check their handles and pointers
    before anything touches a device; the RexPolicy mirror has the header's layout (4 ints, 9 pointers, float, int, uint64)."""
    lib = L.lib()
    assert lib.rex_set_policy(None, None, None) < 0 and b"rex_set_policy" in lib.rex_last_error()
    assert lib.rex_step_policy(None, None, None, None, None, None, None, None, None) < 0 and b"rex_step_policy" in lib.rex_last_error()
    assert lib.rex_step_segment_policy(None, 5, None, None, None, None, None, None, None, None) < 0 and b"rex_step_segment_policy" in lib.rex_last_error()
    assert ctypes.sizeof(L.RexPolicy) == 4 * 4 + 9 * 8 + 4 + 4 + 8 and L.RexPolicy.seed.offset == 96 and L.RexPolicy.d_w1.offset == 16
    hdr = open(os.path.join(ROOT, "include", "rexsim.h")).read()
    body = hdr[hdr.index("typedef struct RexPolicy {"):hdr.index("} RexPolicy;")]
    names = re.findall(r"\b(obs_dim|action_dim|hidden1|hidden2|d_w1|d_b1|d_w2|d_b2|d_w3|d_b3|d_logstd|d_obs_mean|d_obs_scale|obs_clip|sample|seed)\b\s*[;,]", body)
    assert names == [f[0] for f in L.RexPolicy._fields_], names


def test_spaces_match_reference_bounds():
    from rex_gym_amd.envs.batch_env import _spaces
    _, o = _spaces("gallop", "ol", 0.001, 0)             # RexReactiveEnv(use_angle_in_observation=False): gallop_env.py:374-377
    assert o.shape == (4,)
    a, o = _spaces("walk", "ik", 0.001)
    assert a.shape == (2,) and np.allclose(a.high, 0.4) and np.allclose(a.low, -0.4)       # walk_env.py:104-114
    assert o.shape == (4,)
    assert np.allclose(o.high, [2 * np.pi + 0.01] * 2 + [2 * np.pi / 0.001 + 0.01] * 2)    # walk_env.py:364-378
    a, _ = _spaces("walk", "ol", 0.001)
    assert a.shape == (8,) and np.allclose(a.high, 0.01)
    a, o = _spaces("gallop", "ol", 0.001)
    assert a.shape == (4,) and np.allclose(a.low, 0.3) and np.allclose(a.high, -0.3)        # inverted, gallop_env.py:128-130
    assert o.shape == (16,)
    assert a.sample().shape == (4,) and np.all(np.abs(a.sample()) <= 0.3 + 1e-6)


def test_box_contains_and_sample():
    from rex_gym_amd import Box
    b = Box(-np.ones(3), np.ones(3))
    assert b.contains(np.zeros(3, np.float32)) and not b.contains(np.full(3, 2.0)) and not b.contains(np.zeros(2))
    b.seed(0)
    assert b.contains(b.sample())


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from rex_gym_amd import RexBatchEnv
    from rex_gym_amd._lib import RexSimError
    with pytest.raises(RexSimError):
        RexBatchEnv(4)


def test_product_never_imports_the_oracle():
    """The oracle is the checker only: nothing under rex_gym_amd/ may reference it."""
    pkg = os.path.join(ROOT, "rex_gym_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "orclib" not in src and "rex_oracle" not in src.replace("oracle/rex_oracle.c", ""), f
                assert "librex_oracle" not in src, f


def test_shard_arithmetic():
    from rex_gym_amd.sharding import Shard
    s = Shard(3, 8, 65536)
    assert s.num_envs == 8192 and s.env_index_base == 24576
    assert s.env_kwargs() == {"num_envs": 8192, "env_index_base": 24576}
    with pytest.raises(ValueError):
        Shard(0, 3, 10)


def test_registered_env_ids_mirror_the_reference():
    """playground/__init__.py:19-58: ids, step limits; RexGo-v0 dangles there and fails here too."""
    from rex_gym_amd.envs import registry
    assert registry.ENV_IDS == {"RexGalloping-v0": ("gallop", 1000), "RexWalk-v0": ("walk", 2500), "RexTurn-v0": ("turn", 1000),
                                "RexStandup-v0": ("standup", 400), "RexPoses-v0": ("poses", 400)}
    with pytest.raises(ModuleNotFoundError):
        registry.make("RexGo-v0")
    with pytest.raises(KeyError):
        registry.make("RexFly-v0")


def test_constructor_keywords_are_checked_before_anything_touches_the_gpu():
    """A caller porting reference code must not get silently different behaviour: keywords that would change what the
    env computes and are not supported raise; GUI / logging keywords of the reference are accepted."""
    from rex_gym_amd import RexBatchEnv
    with pytest.raises(TypeError, match="unsupported keyword"):
        RexBatchEnv(4, task="walk", forward_reward=3.0)
    with pytest.raises(ValueError, match="auto_reset"):
        RexBatchEnv(4, task="walk", env_randomizer=object(), auto_reset=True)
    with pytest.raises(NotImplementedError, match="pybullet_data"):
        RexBatchEnv(4, task="walk", terrain_type="mounts")
    with pytest.raises(ValueError, match="tasks"):
        RexBatchEnv(4, task="walk", tasks=("walk", "turn"))


def test_rex_knobs_map_reference_mass_setters_to_group_scales():
    """`env.rex` of the batched env (envs/rex_knobs.py): the getters / setters an EnvRandomizer uses (rex.py:643-692)."""
    import torch
    from rex_gym_amd.envs.rex_knobs import RexKnobs

    class FakeEnv:
        _torch, device, _randomize_indices = torch, torch.device("cpu"), None
        params = torch.tensor([[1.0], [1.0], [0.5]]).repeat(1, 4)

        def set_body_params(self):
            return self.params

    env = FakeEnv()
    rex = RexKnobs(env)
    base, legs = rex.GetBaseMassesFromURDF(), rex.GetLegMassesFromURDF()
    assert base == [1.20, 0.05, 0.05] and len(legs) == 20 and abs(sum(base) + sum(legs) - 4.52) < 1e-12    # REX_TOTAL_MASS
    rex.SetBaseMasses([1.2 * m for m in base])
    rex.SetLegMasses(np.outer([0.8, 0.9, 1.0, 1.1], legs))              # one row per env
    rex.SetFootFriction(0.3)
    np.testing.assert_allclose(env.params.numpy(), [[1.2] * 4, [0.8, 0.9, 1.0, 1.1], [0.3] * 4], rtol=1e-6)
    env._randomize_indices = torch.tensor([2], dtype=torch.int32)       # a partial reset touches its envs only
    rex.SetBaseMasses(base)
    np.testing.assert_allclose(env.params[0].numpy(), [1.2, 1.2, 1.0, 1.2], rtol=1e-6)
    with pytest.raises(ValueError, match="not the same"):
        rex.SetBaseMasses([1.0, 2.0])
    with pytest.raises(ValueError, match="one factor"):
        rex.SetBaseMasses([1.3, 0.05, 0.05])


def test_config_validation_happens_before_any_device_call():
    """rex_create validates the config first, so a bad one is reported (REX_EINVAL + message) on any host: a NaN
    forward_reward_cap, and a caller-set energy_weight in a REX_TASK_MIXED batch (a per-task constant there; silently
    ignoring the caller's value would contradict 'anything that changes the computation is an error')."""
    import ctypes
    from rex_gym_amd import _lib as L
    lib = L.lib()
    cfg = L.RexConfig()
    assert lib.rex_default_config(L.TASKS["walk"], L.SIGNALS["ik"], 8, ctypes.byref(cfg)) == 0
    assert cfg.forward_reward_cap == float("inf")            # rex_gym_env.py:81
    dummy = (ctypes.c_float * 4)()
    out = ctypes.c_void_p()
    cfg.forward_reward_cap = float("nan")
    assert lib.rex_create(ctypes.byref(cfg), 0, ctypes.cast(dummy, ctypes.c_void_p), None, ctypes.byref(out)) == -1
    assert b"forward_reward_cap" in lib.rex_last_error()
    assert lib.rex_default_config(L.TASKS["mixed"], L.SIGNALS["ik"], 8, ctypes.byref(cfg)) == 0
    cfg.energy_weight = 0.01
    assert lib.rex_create(ctypes.byref(cfg), 0, ctypes.cast(dummy, ctypes.c_void_p), None, ctypes.byref(out)) == -1
    assert b"energy_weight" in lib.rex_last_error()


@pytest.mark.parametrize("n,epw,base,seed", [(96, 4, 0, 4), (2048, 4, 0, 0), (2048, 4, 6144, 0), (5000, 8, 0, 4), (16384, 16, 0, 0), (1, 4, 0, 1), (37, 16, 5, 2)])
def test_mixed_batch_slot_map_places_every_env_once_in_waves_of_one_task(L, n, epw, base, seed):
    """BASELINE configs[4] (per-env mixed tasks): the host decides once which envs of a REX_TASK_MIXED batch share a wave
    (rex_mixed_slot_map, host-only).  Every env sits in exactly one slot; a workgroup's envs all run the workgroup's task,
    which is the task the env's own Philox draw gives it (the draw the oracle and RexBatchEnv.task_ids make); the envs of a
    workgroup come out of one chunk of neighbouring indices (their state words share sectors); workgroup b of a chunk dealt
    to XCD x has b % 8 == x; padding stays below one wave per (chunk, task)."""
    from rex_gym_amd.envs.philox import philox4x32
    lib = L.lib()
    cfg = L.RexConfig()
    assert lib.rex_default_config(L.TASKS["mixed"], L.SIGNALS["ik"], n, ctypes.byref(cfg)) == 0
    cfg.seed = seed; cfg.env_index_base = base; cfg.mark = 1
    nb = lib.rex_mixed_slot_map(ctypes.byref(cfg), epw, None, None, 0)
    assert nb > 0
    slots = np.full(nb * epw, -7, np.int32); tasks = np.full(nb, -7, np.int32)
    assert lib.rex_mixed_slot_map(ctypes.byref(cfg), epw, slots.ctypes.data_as(ctypes.c_void_p), tasks.ctypes.data_as(ctypes.c_void_p), nb) == nb
    assert lib.rex_mixed_slot_map(ctypes.byref(cfg), epw, slots.ctypes.data_as(ctypes.c_void_p), tasks.ctypes.data_as(ctypes.c_void_p), nb - 1) < 0
    assert lib.rex_mixed_slot_map(ctypes.byref(cfg), 5, None, None, 0) < 0
    real = slots[slots >= 0]
    assert sorted(real.tolist()) == list(range(n))                     # every env exactly once
    assert set(slots[slots < 0].tolist()) <= {-1}
    g = np.arange(n, dtype=np.uint32) + np.uint32(base)
    out = philox4x32(np.stack([np.full_like(g, 0xFFFFFFFF), g, np.full_like(g, 2), np.zeros_like(g)]), np.uint32(seed), np.uint32(0))
    mix = np.array([L.TASKS["walk"], L.TASKS["gallop"], L.TASKS["turn"]], np.int32)    # task_mix bits, ascending
    want = mix[(out[0] % np.uint32(3)).astype(np.int64)]
    grid = slots.reshape(nb, epw)
    busy = 0
    for b in range(nb):
        envs = grid[b][grid[b] >= 0]
        if envs.size == 0:
            continue
        busy += 1
        assert grid[b][0] >= 0                                           # the slot the padding slots shadow
        assert (want[envs] == tasks[b]).all()
        assert envs.max() - envs.min() < 64 * epw + 16                   # one chunk
    m = -(-n // (8 * 64 * epw))
    chunk = (-(-n // (8 * m)) + 15) // 16 * 16
    nchunks = -(-n // chunk)
    assert busy * epw - n <= nchunks * 3 * (epw - 1)
    for b in range(nb):                                                  # chunk c goes to XCD c % 8
        envs = grid[b][grid[b] >= 0]
        if envs.size:
            assert (envs[0] // chunk) % 8 == b % 8
