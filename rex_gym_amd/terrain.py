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


# ---- heightfield FILES: terrain_type 'hills' (csv) and 'mounts' / 'maze' (png), rex_gym/model/terrain.py:55-78 ----------------------
# The files themselves (heightmaps/ground0.txt, wm_height_out.png, Maze.png) live in the pip package `pybullet_data`, not in rex-gym
# (`setAdditionalSearchPath(pd.getDataPath())`, terrain.py:33): a caller who has them passes the path.  What the reference asks Bullet
# to do with a file -- createCollisionShape(GEOM_HEIGHTFIELD, meshScale, fileName) then resetBasePositionAndOrientation(terrain, pos) --
# is restated here as the arrays RexBatchEnv(heightfield=..., heightfield_cell=..., heightfield_origin=..., init_height=...) takes.
# Bullet's file readers (PhysicsServerCommandProcessor, GEOM_HEIGHTFIELD; restated from Bullet's published source, UNVERIFIED here as
# everything about pybullet): a text file is one row of comma-separated heights per line, row j of the file = row j of the field
# (data[i + j * width], i along x); an image gives height = first channel / 255 per pixel, image row j = field row j; meshScale z
# multiplies the heights, meshScale x / y are the vertex spacing, and the shape is centred on its height range (rex_set_heightfield).
TERRAIN_FILES = {            # terrain_id -> (kind, file name under pybullet_data, meshScale, body position, ROBOT_INIT_POSITION z)
    "hills": ("csv", "heightmaps/ground0.txt", (0.5, 0.5, 0.5), (1.0, 0.0, 2.0), 1.98),          # terrain.py:55-64, 14-20
    "mounts": ("png", "heightmaps/wm_height_out.png", (0.1, 0.1, 24.0), (0.0, 0.0, 2.0), 0.85),    # terrain.py:67-76
    "maze": ("png", "heightmaps/Maze.png", (0.1, 0.1, 1.0), (0.0, 0.0, 0.0), 0.21),               # terrain.py:67-78
}


def _read_png_first_channel(path):
    """8-bit non-interlaced PNG -> uint8 [height, width] of the first channel (grey / red / palette index's red).  Standard library only
    (zlib): no image-library dependency for one file format."""
    import struct
    import zlib
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError(f"{path}: not a PNG file")
    pos, idat, plte, hdr = 8, [], None, None
    while pos < len(raw):
        (length,), kind = struct.unpack(">I", raw[pos:pos + 4]), raw[pos + 4:pos + 8]
        body = raw[pos + 8:pos + 8 + length]
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"PLTE":
            plte = np.frombuffer(body, dtype=np.uint8).reshape(-1, 3)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
        pos += 12 + length
    width, height, depth, ctype, _, _, interlace = hdr
    if depth != 8 or interlace != 0 or ctype not in (0, 2, 3, 4, 6):
        raise ValueError(f"{path}: only 8-bit non-interlaced PNGs are read (depth {depth}, colour type {ctype}, interlace {interlace})")
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    stride = width * ch
    data = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(height, stride + 1)
    out = np.zeros((height, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    for y in range(height):                      # undo the per-row filters (PNG spec 9.2): None, Sub, Up, Average, Paeth
        ft, line = int(data[y, 0]), data[y, 1:].astype(np.int32)
        if ft in (0, 2):
            cur = (line + (prev if ft == 2 else 0)) & 255
        else:
            cur = np.zeros(stride, dtype=np.int32)
            for x in range(stride):
                a = cur[x - ch] if x >= ch else 0
                b = prev[x]
                c = prev[x - ch] if x >= ch else 0
                if ft == 1:
                    pred = a
                elif ft == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if pa <= pb and pa <= pc else (b if pb <= pc else c)
                cur[x] = (line[x] + pred) & 255
        out[y] = cur
        prev = cur
    first = out[:, ::ch]
    return plte[first, 0] if ctype == 3 and plte is not None else first


def load_heightfield(path, kind, mesh_scale=(1.0, 1.0, 1.0), position=(0.0, 0.0, 0.0)):
    """A heightfield file as the reference hands it to Bullet (model/terrain.py:55-78) -> the keyword arguments of RexBatchEnv:
        dict(heightfield [ny, nx] float32 in metres, heightfield_cell (cx, cy), heightfield_origin (x, y, z))
    kind 'csv': one row of comma-separated heights per line; 'png': first channel / 255.  mesh_scale = Bullet's meshScale (x, y: vertex
    spacing, z: height factor), position = where the reference places the terrain body."""
    if kind == "csv":
        rows = []
        with open(path) as f:
            for line in f:
                vals = [v for v in line.replace(";", ",").split(",") if v.strip()]
                if vals:
                    rows.append([float(v) for v in vals])
        if not rows or any(len(r) != len(rows[0]) for r in rows):
            raise ValueError(f"{path}: rows of different lengths")
        h = np.asarray(rows, dtype=np.float64)
    elif kind == "png":
        h = _read_png_first_channel(path).astype(np.float64) / 255.0
    else:
        raise ValueError("kind must be 'csv' or 'png'")
    if h.shape[0] < 2 or h.shape[1] < 2:
        raise ValueError(f"{path}: a heightfield needs at least 2 x 2 vertices")
    return dict(heightfield=np.ascontiguousarray(h * float(mesh_scale[2]), dtype=np.float32),
                heightfield_cell=(float(mesh_scale[0]), float(mesh_scale[1])), heightfield_origin=tuple(float(v) for v in position))


def load_reference_terrain(terrain_id, data_path):
    """terrain_type 'hills' / 'mounts' / 'maze' of the reference for a caller who has pybullet_data: `data_path` = pybullet_data.getDataPath().
    -> keyword arguments for RexBatchEnv(terrain_type=terrain_id, **kw), including init_height (ROBOT_INIT_POSITION, terrain.py:14-20)."""
    import os
    if terrain_id not in TERRAIN_FILES:
        raise ValueError(f"terrain_id must be one of {sorted(TERRAIN_FILES)}")
    kind, name, scale, pos, z0 = TERRAIN_FILES[terrain_id]
    kw = load_heightfield(os.path.join(data_path, name), kind, scale, pos)
    kw["init_height"] = z0
    return kw
