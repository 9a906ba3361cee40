"""Recorded PyBullet rollouts, taken out of the reference's own shipped checkpoint.

`rex_gym/policies/turn/ol/model.ckpt-2000000` was saved in the middle of a training phase: the PPO algorithm's episode memory
(`agents/ppo/algorithm.py:66-78,136-175`: `memory/Variable_1..5` = observ, action, mean, logstd, reward per agent and step;
`memory/Variable` = the episode lengths) still holds the 20 episodes collected since the last update -- 8 091 control steps of the
reference's `RexTurnEnv` on real PyBullet, as the learner saw them through its wrapper stack (`playground/trainer.py:48-52`:
LimitDuration(1000), RangeNormalize, ClipAction, ConvertTo32Bit):

    observ[t]  the observation BEFORE step t (`simulate.py:57-76` prevob), RangeNormalize'd: [roll, pitch, w_x, w_y] / their bounds
    action[t]  the policy's sample for step t in RangeNormalize units (ClipAction clips it to [-1, 1] on the way in)
    reward[t]  0.035 - |x| - |y| after step t (`turn_env.py:362-367`)

The other four checkpoints with a data shard were saved right after an update: `memory/Variable` (the lengths) is zero -- but
`EpisodeMemory.clear()` only zeroes the lengths, the buffers keep the last cycle's episodes.  `standup/ol/model.ckpt-2000000` is used here
too: `RexStandupEnv` has no hidden draws and no wall-clock planner, and every one of its 25 rows holds 400 steps followed by zeros (the
run's LimitDuration), so the episodes' extent is known.  (walk/ik and turn/ik ran `GaitPlanner.loop` on the host's wall clock and are not
replayable; `poses` hides a per-episode target draw -- left alone.)  The standup reward is the base position: |x| + |y| + |0.21 - z|
folded as `standup_env.py:141-155` folds it.  What the file does NOT say -- the env's code at training time, each episode's start / target yaw --
is recovered by tests/pybullet_replay.py from the data (6 substeps per control step: the leg switches of `_open_loop_signal` sit
17 steps apart; direction from the first roll excursion; start yaw from the world-frame rates).

Run in the build container:  python tests/golden/make_pybullet_golden.py   -> tests/golden/pybullet_turn_ol_rollouts.npz, pybullet_standup_ol_rollouts.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rex_gym_amd.agents.tf_checkpoint import Checkpoint  # noqa: E402

REFERENCE = os.environ.get("REX_REFERENCE", "/root/reference")


def main(out=os.path.join(ROOT, "tests", "golden", "pybullet_turn_ol_rollouts.npz")):
    ck = Checkpoint(os.path.join(REFERENCE, "rex_gym", "policies", "turn", "ol", "model.ckpt-2000000"))
    length = ck.tensor("memory/Variable")
    keep = np.nonzero(length)[0]
    tmax = int(length.max())
    cut = lambda name: np.ascontiguousarray(ck.tensor(name)[keep, :tmax])   # noqa: E731
    observ, action, reward = cut("memory/Variable_1"), cut("memory/Variable_2"), cut("memory/Variable_5")
    for k, n in enumerate(length[keep]):          # rows past an episode's end hold older episodes' leftovers: not part of the record
        observ[k, n:], action[k, n:], reward[k, n:] = 0, 0, 0
    np.savez_compressed(out, length=length[keep].astype(np.int32), observ=observ, action=action, reward=reward)
    print(out, "episodes", len(keep), "steps", int(length.sum()), "bytes", os.path.getsize(out))
    standup(os.path.join(os.path.dirname(out), "pybullet_standup_ol_rollouts.npz"))


def standup(out):
    ck = Checkpoint(os.path.join(REFERENCE, "rex_gym", "policies", "standup", "ol", "model.ckpt-2000000"))
    observ, action, reward = ck.tensor("memory/Variable_1"), ck.tensor("memory/Variable_2"), ck.tensor("memory/Variable_5")
    used = np.abs(observ).sum(2) > 0
    length = used.sum(1)
    assert np.all(length == 400) and all(np.all(used[k, :400]) for k in range(len(length))), length      # 25 rows x 400 steps, zeros behind
    np.savez_compressed(out, length=length.astype(np.int32), observ=observ[:, :400], action=action[:, :400], reward=reward[:, :400])
    print(out, "episodes", len(length), "steps", int(length.sum()), "bytes", os.path.getsize(out))


if __name__ == "__main__":
    main()
