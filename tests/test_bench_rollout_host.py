"""Host logic of bench.py's Rollout (no GPU): the timed loop steps an env shard into two alternating rollout segments, draws the
other segment's actions a segment ahead, and -- with segment_launch -- issues the steps of a segment as ONE step_segment call.
A recording stand-in for RexBatchEnv checks that both modes take exactly the same steps with the same actions into the same
slices, for step counts that do not divide by the segment length and across consecutive run() calls."""
import numpy as np
import pytest


class _Box:
    def __init__(self, lo, hi):
        self.low, self.high = np.asarray(lo, np.float32), np.asarray(hi, np.float32)


class _FakeEnv:
    """records (first step index in its segment, number of steps, a checksum of the actions) per launch; writes step markers into the outputs"""
    action_dim, obs_dim = 2, 4

    def __init__(self, n):
        self.num_envs, self.action_space = n, _Box([-0.01, -0.01], [0.01, 0.01])
        self.calls, self.clock = [], 0

    def reset(self, idx=None):
        return None

    def bind_out(self, obs, reward, done):
        return (obs, reward, done)

    def _write(self, a, obs, reward, done):
        self.clock += 1
        obs[:] = float(self.clock)
        reward[:] = float(a.sum())
        done[:] = self.clock % 2

    def step(self, a, out=None):
        self.calls.append(("step", 1, float(a.sum())))
        self._write(a, *out)

    def step_segment(self, a, out=None):
        self.calls.append(("segment", int(a.shape[0]), float(a.sum())))
        for t in range(a.shape[0]):
            self._write(a[t], out[0][t], out[1][t], out[2][t])


@pytest.mark.parametrize("T,chunks", [(5, (3, 5, 9, 1)), (4, (8, 2, 7)), (1, (3, 2))])
def test_segment_launches_take_the_same_steps_as_per_step_launches(T, chunks):
    torch = pytest.importorskip("torch")
    import bench
    n, dev = 6, torch.device("cpu")
    outs = {}
    for seg_launch in (False, True):
        gen = torch.Generator(device=dev); gen.manual_seed(7)
        env = _FakeEnv(n)
        ro = bench.Rollout(env, n, T, dev, gen)
        rec = []
        for steps in chunks:                               # warm-up, timed region, ... : the segments continue across the calls
            before = len(env.calls)
            ro.run(steps, False, seg_launch)
            rec.append((ro.clock, [{k: v.clone() for k, v in s.items()} for s in ro.seg]))
            if seg_launch:      # the env steps behind every launch of THIS call: what bench.py divides each launch's kernel time by
                assert ro.launch_steps == [c[1] for c in env.calls[before:]] and sum(ro.launch_steps) == steps
            else:
                assert ro.launch_steps == []
        outs[seg_launch] = (env.calls, rec, env.clock)
    (c0, r0, k0), (c1, r1, k1) = outs[False], outs[True]
    assert k0 == k1 == sum(chunks)
    assert all(c[0] == "step" for c in c0) and len(c0) == sum(chunks)
    assert all(c[0] == "segment" and 1 <= c[1] <= T for c in c1) and sum(c[1] for c in c1) == sum(chunks)
    # a launch never crosses a segment boundary, and inside a run() call it is as long as the segment and the call allow
    pos = 0
    for steps in chunks:
        end = pos + steps
        while pos < end:
            m = min(T - pos % T, end - pos)
            assert c1.pop(0)[1] == m
            pos += m
    # same actions, same slices: the segments' tensors are identical after every run() call
    for (ka, sa), (kb, sb) in zip(r0, r1):
        assert ka == kb
        for x, y in zip(sa, sb):
            for k in x:
                assert torch.equal(x[k], y[k]), k
