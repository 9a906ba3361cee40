"""Multi-GPU layout of the batched simulator.

Environments are independent units (each reference env is its own Bullet world,
rex_gym_env.py:228-231), so the step path shards with NO data-path collective:
rank r owns the contiguous global env range [r*per_rank, (r+1)*per_rank) and its
RNG streams are keyed by the GLOBAL env index, which makes results independent of
the number of ranks.  The only exchange is the learner hand-off: one all-gather of
a rollout segment (obs/action/reward/done), issued once per segment, never per step.

One process per GPU, `torch.distributed` (backend "nccl" is RCCL over xGMI on ROCm; "gloo" for
the CPU tests).
"""
from dataclasses import dataclass


@dataclass(frozen=True)
class Shard:
    rank: int
    world_size: int
    num_envs_total: int

    def __post_init__(self):
        if self.num_envs_total % self.world_size:
            raise ValueError("num_envs_total must be divisible by world_size")
        if not (0 <= self.rank < self.world_size):
            raise ValueError("rank out of range")

    @property
    def num_envs(self):
        return self.num_envs_total // self.world_size

    @property
    def env_index_base(self):
        return self.rank * self.num_envs

    def env_kwargs(self):
        """kwargs for RexBatchEnv so that this rank simulates its slice of the global batch."""
        return {"num_envs": self.num_envs, "env_index_base": self.env_index_base}


def shard_from_env(num_envs_total):
    """Shard of this process from torchrun's RANK / WORLD_SIZE."""
    import os
    return Shard(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), num_envs_total)


class RolloutGather:
    """Handle of an all-gather of one rollout segment in flight (gather_rollout(..., async_op=True)): the collectives run on
    RCCL's own stream while the next segment's env steps keep the compute stream busy; `wait()` orders the current stream
    behind them and returns the segment in global env order."""

    def __init__(self, bufs, works, shapes, world):
        self._bufs, self._works, self._shapes, self._world = bufs, works, shapes, world
        self._out = None

    def wait(self):
        if self._out is None:
            for w in self._works:
                w.wait()
            out = {}
            for name, buf in self._bufs.items():
                t, n = self._shapes[name][0], self._shapes[name][1]
                # [world, T, n, ...] -> [T, world * n, ...]: global env index = rank * n + local index
                o = buf.view(self._world, *self._shapes[name]).transpose(0, 1).reshape(t, self._world * n, *self._shapes[name][2:])
                # (T == 1, or one rank: the reshape is a view of the cached receive buffer, which the next gather on this slot
                #  overwrites -- the caller gets its own tensor, as it does whenever the reshape had to copy)
                out[name] = o.clone() if o.data_ptr() == buf.data_ptr() else o
            self._out = out
        return self._out


_GATHER_BUFFERS = {}


def clear_gather_buffers():
    """Drop the cached receive buffers of gather_rollout (e.g. before a process group is destroyed, or when a run changes its
    segment shape for good)."""
    _GATHER_BUFFERS.clear()


def _group_key(group):
    """cache key of a process group: its ranks (an id() can be handed to a new group once the old one is gone)"""
    import torch.distributed as dist
    if group is None:
        return "world"
    try:
        return tuple(dist.get_process_group_ranks(group))
    except Exception:
        return id(group)


def gather_rollout(segment, group=None, always=False, async_op=False, slot=0):
    """All-gather one rollout segment to every rank (the learner hand-off).

    segment: dict name -> tensor [T, N_local, ...] (same T and dtypes on every rank).
    Returns dict name -> tensor [T, N_total, ...] in global env order (async_op=True: a RolloutGather whose wait() does).
    One all-gather per tensor per segment, into receive buffers that are allocated once per (name, shape, slot) and reused:
    at 29 B/env-step (walk-IK) a 25-step segment of 65 536 envs is 47.5 MB in total, i.e. latency-bound on xGMI; never
    call this per step.  `slot` names the receive-buffer set: a caller that keeps two segments in flight (fill B while A
    is gathered) alternates slots 0 and 1.
    `always` issues the collectives even with one rank (exercises the backend's dtype support).
    gloo (CPU tests, or several ranks sharing one GPU in a smoke run) gathers device tensors through the host.
    """
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not always):
        out = dict(segment)
        return _Done(out) if async_op else out
    world = dist.get_world_size(group)
    via_host = dist.get_backend(group) == "gloo"
    bufs, works, shapes = {}, [], {}
    for name, t in segment.items():
        t = t.contiguous()
        src = t.cpu() if (via_host and t.is_cuda) else t
        wire = src.view(torch.uint8) if src.dtype == torch.bool else src       # bool travels as bytes
        key = (name, tuple(wire.shape), wire.dtype, str(wire.device), slot, _group_key(group))
        buf = _GATHER_BUFFERS.get(key)
        if buf is None:
            # rank-major along dim 0 ([world * T, N_local, ...]): the concatenated form every backend's all_gather_into_tensor takes
            buf = _GATHER_BUFFERS[key] = torch.empty((world * wire.shape[0],) + tuple(wire.shape[1:]), dtype=wire.dtype, device=wire.device)
        works.append(dist.all_gather_into_tensor(buf, wire, group=group, async_op=True))
        bufs[name], shapes[name] = (buf, t.dtype, t.device), tuple(t.shape)
    handle = RolloutGather({k: v[0] for k, v in bufs.items()}, works, shapes, world)
    if via_host or any(v[1] == torch.bool for v in bufs.values()):
        inner = handle

        class _Cast:
            def wait(self_inner):
                out = inner.wait()
                return {k: (v.view(torch.bool) if bufs[k][1] == torch.bool else v).to(bufs[k][2]) for k, v in out.items()}
        handle = _Cast()
    return handle if async_op else handle.wait()


class _Done:
    def __init__(self, out):
        self._out = out

    def wait(self):
        return self._out
