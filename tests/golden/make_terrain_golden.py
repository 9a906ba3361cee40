#!/usr/bin/env python3
"""Golden for the random heightfield: the reference's Terrain.generate_terrain / update_terrain (model/terrain.py)
run unmodified against a recording stand-in for pybullet; the first two fields it hands to createCollisionShape are
summarised (leading samples in its own layout, sums, extrema) in tests/golden/terrain_golden.json.

Run in the build container:  PYTHONPATH=/root/reference python tests/golden/make_terrain_golden.py
"""
import json
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.environ.get("REX_REFERENCE", "/root/reference"))
fields = []


def create_collision_shape(**kw):
    assert kw["meshScale"] == [.05, .05, 1] and kw["numHeightfieldRows"] == 256 and kw["numHeightfieldColumns"] == 256
    fields.append(np.asarray(kw["heightfieldData"], np.float64))
    return len(fields)


pb = types.ModuleType("pybullet")
pb.GEOM_HEIGHTFIELD, pb.GEOM_CONCAVE_INTERNAL_EDGE, pb.COV_ENABLE_RENDERING = 9, 2, 7
pb.createCollisionShape = create_collision_shape
sys.modules["pybullet"] = pb
sys.modules["pybullet_data"] = types.ModuleType("pybullet_data")
sys.modules["pybullet_data"].getDataPath = lambda: "/nonexistent"

from rex_gym.model.terrain import Terrain     # noqa: E402

client = types.SimpleNamespace(GEOM_HEIGHTFIELD=9, COV_ENABLE_RENDERING=7, createCollisionShape=create_collision_shape,
                               setAdditionalSearchPath=lambda p: None, configureDebugVisualizer=lambda *a: None,
                               createMultiBody=lambda *a: 1, resetBasePositionAndOrientation=lambda *a: None,
                               changeVisualShape=lambda *a, **k: None)
t = Terrain("random", "random")               # seeds Python's random with 10 (terrain.py:26)
t.generate_terrain(types.SimpleNamespace(pybullet_client=client))
t.update_terrain()                            # what RexGymEnv.reset() does next (rex_gym_env.py:347-348)
out = dict(fields=[])
for f in fields:
    out["fields"].append(dict(n=int(f.size), head=f[:520].tolist(), rows_at=[int(k) for k in (256 * 100, 256 * 255)],
                              row_100=f[256 * 100:256 * 100 + 16].tolist(), row_255=f[256 * 255:256 * 255 + 16].tolist(),
                              sum=float(f.sum()), sum_sq=float((f * f).sum()), min=float(f.min()), max=float(f.max())))
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "terrain_golden.json")
json.dump(out, open(path, "w"))
print("wrote", path, [(d["sum"], d["min"], d["max"]) for d in out["fields"]])
