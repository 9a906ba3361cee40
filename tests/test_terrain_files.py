"""rex_gym_amd.terrain.load_heightfield: the csv / png heightfield files of model/terrain.py:55-78 -> the arrays RexBatchEnv takes.
Synthetic files (the reference's own live in pip pybullet_data); CPU only."""
import struct
import zlib

import numpy as np
import pytest

from rex_gym_amd import terrain


def _png(path, img, ctype, filters):
    """write an 8-bit PNG with the given per-row filter types (so that every unfilter branch is read back)"""
    h, w = img.shape[:2]
    ch = {0: 1, 2: 3, 6: 4}[ctype]
    rows = img.reshape(h, w * ch).astype(np.int32)
    raw = bytearray()
    prev = np.zeros(w * ch, dtype=np.int32)
    for y in range(h):
        ft = filters[y % len(filters)]
        cur = rows[y]
        a = np.concatenate([np.zeros(ch, dtype=np.int32), cur[:-ch]])
        c = np.concatenate([np.zeros(ch, dtype=np.int32), prev[:-ch]])
        if ft == 0: pred = 0
        elif ft == 1: pred = a
        elif ft == 2: pred = prev
        elif ft == 3: pred = (a + prev) >> 1
        else:
            p = a + prev - c
            pa, pb, pc = np.abs(p - a), np.abs(p - prev), np.abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
        raw.append(ft); raw += bytes(((cur - pred) & 255).astype(np.uint8))
        prev = cur
    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(bytes(raw))) + chunk(b"IEND", b""))


@pytest.mark.parametrize("ctype", [0, 2, 6])
def test_png_heightfield_first_channel_over_255_times_mesh_scale(tmp_path, ctype):
    rng = np.random.RandomState(ctype)
    ch = {0: 1, 2: 3, 6: 4}[ctype]
    img = rng.randint(0, 256, (9, 13, ch)).astype(np.uint8)
    p = str(tmp_path / "field.png")
    _png(p, img, ctype, filters=[0, 1, 2, 3, 4])
    kw = terrain.load_heightfield(p, "png", mesh_scale=(0.1, 0.1, 24.0), position=(0.0, 0.0, 2.0))      # 'mounts', terrain.py:67-76
    assert kw["heightfield"].shape == (9, 13) and kw["heightfield"].dtype == np.float32
    np.testing.assert_allclose(kw["heightfield"], img[:, :, 0].astype(np.float64) / 255.0 * 24.0, rtol=1e-6)
    assert kw["heightfield_cell"] == (0.1, 0.1) and kw["heightfield_origin"] == (0.0, 0.0, 2.0)


def test_csv_heightfield_and_the_reference_terrain_table(tmp_path):
    rng = np.random.RandomState(1)
    h = rng.uniform(0, 3, (7, 5))
    (tmp_path / "heightmaps").mkdir()
    with open(tmp_path / "heightmaps" / "ground0.txt", "w") as f:
        for row in h:
            f.write(",".join(f"{v:.6f}" for v in row) + ",\n")              # (Bullet's sample file ends its rows with a comma)
    kw = terrain.load_reference_terrain("hills", str(tmp_path))               # terrain.py:55-64: meshScale .5 .5 .5, body at [1, 0, 2], drop height 1.98
    np.testing.assert_allclose(kw["heightfield"], 0.5 * h, atol=1e-6)
    assert kw["heightfield_cell"] == (0.5, 0.5) and kw["heightfield_origin"] == (1.0, 0.0, 2.0) and kw["init_height"] == 1.98
    assert terrain.TERRAIN_FILES["mounts"][2] == (0.1, 0.1, 24.0) and terrain.TERRAIN_FILES["maze"][3] == (0.0, 0.0, 0.0)
    with pytest.raises(ValueError):
        terrain.load_reference_terrain("plane", str(tmp_path))
    with open(tmp_path / "ragged.txt", "w") as f:
        f.write("1,2,3\n4,5\n")
    with pytest.raises(ValueError):
        terrain.load_heightfield(str(tmp_path / "ragged.txt"), "csv")
    with pytest.raises(ValueError):
        terrain.load_heightfield(str(tmp_path / "heightmaps" / "ground0.txt"), "png")
