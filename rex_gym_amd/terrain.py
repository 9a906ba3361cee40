"""Random heightfield pool (host side) -- rex_gym/model/terrain.py:32-54,80-106.

The reference builds ONE 256 x 256 field per env (2 x 2 blocks, heights U(0, 0.05) m, 5 cm cells, 12.8 m square) and
regenerates it on every reset with Python's `random`.  A batch of N envs would need N x 256 KB; instead a pool of K fields
lives in HBM and every episode of env g uses field (g + 977 * episode) mod K.
"""
import random

import numpy as np

ROWS = COLUMNS = 256
HEIGHT_RANGE = 0.05   # terrain.py:32 height_perturbation_range


def random_terrain_pool(k, seed=10):
    """-> (heights [k, 256*256] float32 in the reference layout data[i + j*rows], mids [k] float32).

    The fields are the reference's own, in the order it would see them: `Terrain.__init__` seeds Python's `random` with
    10 (terrain.py:26), `generate_terrain` draws 128 x 128 block heights `random.uniform(0, 0.05)` with j outer and i
    inner (terrain.py:38-43), and every `reset()` draws the next field the same way (`update_terrain`, terrain.py:87-93).
    Field 0 is the env's first terrain, field m the one after m resets (exactly so for envs with a fixed target: the
    walk / gallop / turn envs draw their random targets from the same stream in between).  tests/test_terrain.py pins the
    first two fields against the reference's code."""
    rnd = random.Random(seed)
    draws = np.fromiter((rnd.random() for _ in range(k * (COLUMNS // 2) * (ROWS // 2))), dtype=np.float64)
    blocks = (0.0 + (HEIGHT_RANGE - 0.0) * draws).reshape(k, COLUMNS // 2, ROWS // 2).astype(np.float32)   # uniform(a, b) = a + (b - a) random(); [j, i]
    data = np.repeat(np.repeat(blocks, 2, axis=1), 2, axis=2)                                  # [k, j', i']
    flat = np.ascontiguousarray(data.reshape(k, ROWS * COLUMNS))
    mids = (0.5 * (flat.min(axis=1) + flat.max(axis=1))).astype(np.float32)                     # Bullet centres the shape
    return flat, mids
