#!/usr/bin/env python3
"""Census of the contacts the restated stepSimulation does NOT model by default: link boxes vs ground, link vs link.

TEST INFRASTRUCTURE (fp64 oracle + numpy forward kinematics of the compiled model tables); nothing here is product code.

For every task env it plays episodes on the oracle (uniform-random actions, episodes end by the env's own termination
test or at 1 000 steps) with RexConfig.body_contacts = 1 and reports
  * body-vs-ground: the share of substeps in which any link collision box (rex.urdf:15-33,63-108,119-124,151-156,
    170-175) comes below the ground surface (Bullet's box-box detector reports penetrating points only), and the lowest box corner seen;
  * self collision (the reference loads the robot with URDF_USE_SELF_COLLISION, rex.py:276-281: every link pair except
    parent-child collides): the smallest separation of any admissible link-box pair over all visited poses, by the
    separating-axis test (a lower bound of the distance; <= 0 means the boxes overlap and Bullet's box-box detector,
    which reports penetrating points only, would emit contacts).  The toe hull is stood in for by its bounding box.
Writes profiles/r02_contact_census.md.
"""
import argparse
import ctypes
import itertools
import os
import re
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
from orclib import OracleEnv, default_config, S_Q  # noqa: E402


def model_tables():
    """The arrays of rex_gym_amd/csrc/rex_model_gen.h (generated from rex.urdf by tools/compile_model.py)."""
    txt = open(os.path.join(ROOT, "rex_gym_amd", "csrc", "rex_model_gen.h")).read()

    def arr(name):
        m = re.search(name + r"\[[^=]*=\s*\{(.*?)\};", txt, re.S)
        body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
        return np.array([float(v) for v in re.findall(r"-?\d+\.?\d*(?:e-?\d+)?", body)])
    t = dict(parent=arr("REX_PARENT").astype(int), axis=arr("REX_JOINT_AXIS").astype(int),
             jpos=arr("REX_JOINT_POS").reshape(-1, 3), box_body=arr("REX_BOX_BODY").astype(int),
             box_c=arr("REX_BOX_CENTER").reshape(-1, 3), box_h=arr("REX_BOX_HALF").reshape(-1, 3))
    return t


def quat_mat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def fk(t, st):
    """world rotation / origin of the 13 merged bodies from one state column."""
    R = [quat_mat(st[3:7])]
    p = [st[0:3].copy()]
    for i in range(1, 13):
        a = t["axis"][i - 1]
        ang = st[S_Q + i - 1]
        c, s = np.cos(ang), np.sin(ang)
        Rq = np.eye(3)
        j, k = (a + 1) % 3, (a + 2) % 3
        Rq[j, j] = c; Rq[j, k] = -s; Rq[k, j] = s; Rq[k, k] = c
        par = t["parent"][i]
        R.append(R[par] @ Rq)
        p.append(p[par] + R[par] @ t["jpos"][i - 1])
    return R, p


def sat_separation(Ra, ca, ha, Rb, cb, hb):
    """largest separation over the 15 SAT axes of two oriented boxes (> 0: disjoint, lower bound of their distance)."""
    d = cb - ca
    axes = [Ra[:, i] for i in range(3)] + [Rb[:, i] for i in range(3)]
    for i in range(3):
        for j in range(3):
            v = np.cross(Ra[:, i], Rb[:, j])
            n = np.linalg.norm(v)
            if n > 1e-9:
                axes.append(v / n)
    best = -np.inf
    for ax in axes:
        ra = np.sum(ha * np.abs(Ra.T @ ax))
        rb = np.sum(hb * np.abs(Rb.T @ ax))
        best = max(best, abs(d @ ax) - ra - rb)
    return best


def link_boxes(t):
    """(name, body, centre, half, link id, parent link id): the 15 link boxes + 4 toe hull bounding boxes."""
    names = ["base", "chassis_rear", "chassis_front"]
    out = [("base", 0, t["box_c"][0], t["box_h"][0], "base", None),
           ("chassis_rear", 0, t["box_c"][1], t["box_h"][1], "chassis_rear", "base"),
           ("chassis_front", 0, t["box_c"][2], t["box_h"][2], "chassis_front", "base")]
    legs = ["FL", "FR", "RL", "RR"]
    for l, ln in enumerate(legs):
        b = 3 + 3 * l
        out.append((ln + "_shoulder", 1 + 3 * l, t["box_c"][b], t["box_h"][b], ln + "_shoulder", "base"))
        out.append((ln + "_leg", 2 + 3 * l, t["box_c"][b + 1], t["box_h"][b + 1], ln + "_leg", ln + "_shoulder"))
        out.append((ln + "_foot", 3 + 3 * l, t["box_c"][b + 2], t["box_h"][b + 2], ln + "_foot", ln + "_leg"))
        # toe hull (stl/foot.stl after the URDF transform): bounds in the toe-link frame, toe link at z = -0.115
        lo, hi = np.array([-0.0164, -0.01, -0.0194]), np.array([0.0194, 0.01, 0.0058])
        out.append((ln + "_toe", 3 + 3 * l, (lo + hi) / 2 + np.array([0, 0, -0.115]), (hi - lo) / 2, ln + "_toe", ln + "_foot"))
    return out


def census(task, signal, n, steps, seed=0, self_rows=False):
    t = model_tables()
    boxes = link_boxes(t)
    pairs = [(a, b) for a, b in itertools.combinations(range(len(boxes)), 2)
             if boxes[a][5] != boxes[b][4] and boxes[b][5] != boxes[a][4]]
    cfg = default_config(task, signal, num_envs=n, body_contacts=1, seed=seed)
    env = OracleEnv(cfg)
    lib = env.o.lib
    lib.orc_body_points.restype = ctypes.c_long
    lib.orc_substeps_with_body_points.restype = ctypes.c_long
    lib.orc_body_points(1); lib.orc_substeps_with_body_points(1)
    lib.orc_set_probe.argtypes = [ctypes.c_char_p, ctypes.c_double]
    lib.orc_set_probe(b"self_collision", 1.0 if self_rows else 0.0)    # the census asks what happens WITHOUT the link-link rows
    env.reset()
    rng = np.random.default_rng(seed)
    lo = np.minimum(*_bounds(task, signal)); hi = np.maximum(*_bounds(task, signal))
    min_sep = {}
    overlapping = samples = 0
    low_corner = np.inf
    age = np.zeros(n, int)
    total_sub = 0
    for k in range(steps):
        obs, rew, done, cmd = env.step(rng.uniform(lo, hi, (n, env.action_dim)))
        total_sub += n * cfg.action_repeat
        age += 1
        st = env.get_state()
        if k % 5 == 0:
            for i in range(n):
                R, p = fk(t, st[:, i])
                world = [(R[b[1]], p[b[1]] + R[b[1]] @ b[2], b[3]) for b in boxes]
                for bx, (Rw, c, h) in zip(boxes, world):
                    if not bx[0].endswith("_toe"):
                        low_corner = min(low_corner, c[2] - np.sum(h * np.abs(Rw[2, :])))
                samples += 1
                worst = np.inf
                for a, b in pairs:
                    s = sat_separation(*world[a], *world[b])
                    worst = min(worst, s)
                    key = (re.sub(r"^(FL|FR|RL|RR)_", "", boxes[a][0]), re.sub(r"^(FL|FR|RL|RR)_", "", boxes[b][0]),
                           boxes[a][0][:2] == boxes[b][0][:2] and boxes[a][0][2:3] == "_")
                    if s < min_sep.get(key, np.inf):
                        min_sep[key] = s
                overlapping += worst < -1e-4
        idx = np.nonzero(done | (age >= 1000))[0]
        if idx.size:
            env.reset(idx)
            age[idx] = 0
    pts = lib.orc_body_points(1)
    sub = lib.orc_substeps_with_body_points(1)
    lib.orc_set_probe(b"self_collision", 1.0)
    env.close()
    return dict(task=task, signal=signal, substeps=total_sub, substeps_with_body_points=int(sub), body_points=int(pts),
                lowest_box_corner=float(low_corner), min_sep=min_sep,
                overlap_share=overlapping / max(samples, 1))


def _bounds(task, signal):
    if task == "walk":
        b, d = (0.4, 2) if signal == "ik" else (0.01, 8)
    elif task == "gallop":
        b, d = (0.4, 2) if signal == "ik" else (0.3, 4)
    elif task == "turn":
        b, d = 0.01, 2
    else:
        b, d = 0.1, 1
    return -np.full(d, b), np.full(d, b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=8)
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_contact_census.md"))
    a = ap.parse_args()
    lines = ["# Contacts beyond the toes: how often they would act (fp64 oracle, random actions, episodes end by the env's own test)",
             "",
             f"{a.envs} envs x {a.steps} control steps per env type.  Generated by `tools/contact_census.py`.",
             "",
             "The link-link rows (`body_contacts`: leg and foot boxes against the base body's boxes) are switched OFF for the census rows --",
             "it asks where they would be needed; the last row repeats RexPosesEnv with them on.",
             "",
             "| env | substeps | substeps with a kept link-box point (a corner below the ground; in the last row also a link-link candidate) | lowest box corner [m] | admissible link pairs that come within 5 mm (smallest SAT separation, mm; <= 0 = overlap) | sampled poses with a pair overlapping by > 0.1 mm |",
             "|---|---|---|---|---|---|"]
    for task, signal, rows in (("walk", "ik", False), ("walk", "ol", False), ("gallop", "ik", False), ("gallop", "ol", False),
                               ("turn", "ik", False), ("turn", "ol", False), ("poses", "ik", False), ("standup", "ol", False),
                               ("poses", "ik", True)):
        r = census(task, signal, a.envs, a.steps, self_rows=rows)
        ms = sorted(r["min_sep"].items(), key=lambda kv: kv[1])
        near = "; ".join(f"{x}-{y}{' (same leg)' if s else ''} {val * 1000:.2f}" for (x, y, s), val in ms if val < 0.005) or "none"
        line = (f"| {task}-{signal}{' WITH the link-link rows' if rows else ''} | {r['substeps']} | {r['substeps_with_body_points']} "
                f"({100.0 * r['substeps_with_body_points'] / r['substeps']:.2f} %) | {r['lowest_box_corner']:.4f} | {near} | "
                f"{100.0 * r['overlap_share']:.2f} % |")
        print(line, flush=True)
        lines.append(line)
    with open(a.out, "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
