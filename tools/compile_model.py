#!/usr/bin/env python3
"""Offline model compiler: rex.urdf -> flat constant table (C header).

Reads the reference robot description (data, not code):
  rex_gym/util/pybullet_data/assets/urdf/rex.urdf (+ stl/foot.stl for the toe hull)
and emits `rex_gym_amd/csrc/rex_model_gen.h`, which both the CPU oracle and the
HIP kernels include.  The generated header is committed, so neither the GPU box
nor the tests need /root/reference at run time.

What it reproduces (marked UNVERIFIED against a live PyBullet):
  * rex.py:276-287 calls loadURDF WITHOUT URDF_USE_INERTIA_FROM_FILE, so Bullet
    recomputes every link inertia from its collision geometry:
      - one un-offset primitive  -> the primitive's own inertia (box formula)
      - offset / several shapes  -> box inertia of the compound's AABB
      - no collision             -> zero inertia, full mass (point mass at origin)
    and, with no <inertial><origin>, every link COM sits at its link origin.
  * fixed joints are merged into their movable parent (dynamically equivalent
    to Bullet keeping them as 0-DoF links).
  * the toe collision hull (half cylinder, stl/foot.stl) is reduced to an
    analytic cylinder segment about the toe-link y axis.

Usage: python tools/compile_model.py [--urdf PATH] [--out PATH]
"""
import argparse
import math
import os
import struct
import xml.etree.ElementTree as ET

import numpy as np

DEFAULT_URDF = "/root/reference/rex_gym/util/pybullet_data/assets/urdf/rex.urdf"
DEFAULT_OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                           "rex_gym_amd", "csrc", "rex_model_gen.h")
URDF_COLLISION_MARGIN = 0.001  # Bullet's gUrdfDefaultCollisionMargin

# motor order of the reference (model/mark_constants.py:3-8)
BASE_MOTOR_NAMES = [
    "motor_front_left_shoulder", "motor_front_left_leg", "foot_motor_front_left",
    "motor_front_right_shoulder", "motor_front_right_leg", "foot_motor_front_right",
    "motor_rear_left_shoulder", "motor_rear_left_leg", "foot_motor_rear_left",
    "motor_rear_right_shoulder", "motor_rear_right_leg", "foot_motor_rear_right",
]


def rpy_to_mat(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    # URDF: R = Rz(yaw) Ry(pitch) Rx(roll)
    return np.array([
        [cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
        [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
        [-sp, cp * sr, cp * cr]])


def parse_origin(elem):
    xyz = np.zeros(3)
    rpy = np.zeros(3)
    if elem is not None:
        o = elem.find("origin")
        if o is not None:
            if o.get("xyz"):
                xyz = np.array([float(v) for v in o.get("xyz").split()])
            if o.get("rpy"):
                rpy = np.array([float(v) for v in o.get("rpy").split()])
    return xyz, rpy


def read_stl_vertices(path):
    d = open(path, "rb").read()
    n = struct.unpack("<I", d[80:84])[0]
    rec = np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")])
    a = np.frombuffer(d[84:84 + 50 * n], dtype=rec)
    return a["v"].reshape(-1, 3).astype(np.float64)


def box_inertia(mass, size):
    lx, ly, lz = size
    return mass / 12.0 * np.array([ly * ly + lz * lz, lx * lx + lz * lz, lx * lx + ly * ly])


class Link:
    def __init__(self, elem, urdf_dir):
        self.name = elem.get("name")
        inertial = elem.find("inertial")
        self.mass = float(inertial.find("mass").get("value")) if inertial is not None else 0.0
        ixyz, irpy = parse_origin(inertial)
        assert not ixyz.any() and not irpy.any(), "inertial origin not supported (none in rex.urdf)"
        self.collisions = []  # (kind, params, xyz, rpy)
        for c in elem.findall("collision"):
            xyz, rpy = parse_origin(c)
            g = c.find("geometry")
            if g.find("box") is not None:
                size = np.array([float(v) for v in g.find("box").get("size").split()])
                self.collisions.append(("box", size, xyz, rpy))
            elif g.find("cylinder") is not None:
                cy = g.find("cylinder")
                self.collisions.append(("cylinder", (float(cy.get("radius")), float(cy.get("length"))), xyz, rpy))
            elif g.find("mesh") is not None:
                m = g.find("mesh")
                scale = np.array([float(v) for v in m.get("scale", "1 1 1").split()])
                verts = read_stl_vertices(os.path.join(urdf_dir, m.get("filename"))) * scale
                self.collisions.append(("mesh", verts, xyz, rpy))
            else:
                raise ValueError("unsupported collision geometry in " + self.name)

    def bullet_inertia_diag(self):
        """Diagonal inertia (about the link origin == COM) the way Bullet's URDF importer computes it."""
        if self.mass == 0.0 or not self.collisions:
            return np.zeros(3)
        if len(self.collisions) == 1:
            kind, prm, xyz, rpy = self.collisions[0]
            if not xyz.any() and not rpy.any():
                if kind == "box":
                    return box_inertia(self.mass, prm)
                if kind == "cylinder":  # btCylinderShapeZ::calculateLocalInertia
                    radius, length = prm
                    t1 = self.mass / 12.0 * length * length + self.mass / 4.0 * radius * radius
                    t2 = self.mass / 2.0 * radius * radius
                    return np.array([t1, t1, t2])
        # compound: box inertia of the AABB of all children (btCompoundShape::calculateLocalInertia)
        lo = np.full(3, np.inf)
        hi = np.full(3, -np.inf)
        for kind, prm, xyz, rpy in self.collisions:
            R = rpy_to_mat(rpy)
            if kind == "box":
                he = np.asarray(prm) / 2.0
                ctr = np.zeros(3)
            elif kind == "cylinder":
                he = np.array([prm[0], prm[0], prm[1] / 2.0])
                ctr = np.zeros(3)
            else:  # convex hull: cached local AABB incl. margin, then btTransformAabb adds the margin again
                vmin, vmax = prm.min(0), prm.max(0)
                he = (vmax - vmin) / 2.0 + 2.0 * URDF_COLLISION_MARGIN
                ctr = (vmax + vmin) / 2.0
            c = R @ ctr + xyz
            e = np.abs(R) @ he
            lo = np.minimum(lo, c - e)
            hi = np.maximum(hi, c + e)
        return box_inertia(self.mass, hi - lo)


def merge_bodies(links, joints, root):
    """Merge fixed children into their movable ancestor. Returns list of bodies in tree order.

    Each body: dict(name, parent(body idx), joint(name), r(3), E0(3x3), axis(3), lower, upper,
                    mass, com(3), I(3x3 about com), members[(link, xyz, R)])
    """
    children = {}
    for j in joints:
        children.setdefault(j["parent"], []).append(j)
    bodies = []

    def new_body(link_name, parent_idx, joint):
        b = dict(name=link_name, parent=parent_idx, joint=joint, members=[])
        bodies.append(b)
        idx = len(bodies) - 1
        stack = [(link_name, np.zeros(3), np.eye(3))]
        movable = []
        while stack:
            ln, xyz, R = stack.pop()
            b["members"].append((ln, xyz, R))
            for j in children.get(ln, []):
                jx = xyz + R @ j["xyz"]
                jR = R @ rpy_to_mat(j["rpy"])
                if j["type"] == "fixed":
                    stack.append((j["child"], jx, jR))
                else:
                    movable.append((j, jx, jR))
        b["movable"] = movable
        return idx

    def recurse(idx):
        for j, jx, jR in bodies[idx]["movable"]:
            jj = dict(j)
            jj["xyz_in_body"] = jx
            jj["R_in_body"] = jR
            cidx = new_body(j["child"], idx, jj)
            recurse(cidx)

    ridx = new_body(root, -1, None)
    recurse(ridx)
    for b in bodies:
        m = 0.0
        h = np.zeros(3)
        for ln, xyz, R in b["members"]:
            m += links[ln].mass
            h += links[ln].mass * xyz
        com = h / m
        I = np.zeros((3, 3))
        for ln, xyz, R in b["members"]:
            L = links[ln]
            Il = R @ np.diag(L.bullet_inertia_diag()) @ R.T
            d = xyz - com
            I += Il + L.mass * (d @ d * np.eye(3) - np.outer(d, d))
        b["mass"], b["com"], b["I"] = m, com, I
    return bodies


def fmt(x):
    x = float(x)
    if abs(x) < 1e-13:  # drop STL float32 / accumulation noise
        x = 0.0
    return repr(float(f"{x:.13g}"))


ARM_MOTOR_NAMES = ["motor_arm_m1", "motor_arm_m2", "motor_arm_m3", "motor_arm_m4", "motor_arm_m5", "motor_arm_m6"]
DEFAULT_ARM_URDF = "/root/reference/rex_gym/util/pybullet_data/assets/urdf/rex_arm.urdf"
DEFAULT_ARM_OUT = os.path.join(os.path.dirname(DEFAULT_OUT), "rex_arm_model_gen.h")


def load_bodies(urdf, motor_names):
    urdf_dir = os.path.dirname(urdf)
    root = ET.parse(urdf).getroot()
    links = {l.get("name"): Link(l, urdf_dir) for l in root.findall("link")}
    joints = []
    for j in root.findall("joint"):
        xyz, rpy = parse_origin(j)
        ax = j.find("axis")
        lim = j.find("limit")
        joints.append(dict(
            name=j.get("name"), type=j.get("type"), parent=j.find("parent").get("link"),
            child=j.find("child").get("link"), xyz=xyz, rpy=rpy,
            axis=np.array([float(v) for v in ax.get("xyz").split()]) if ax is not None else np.array([1.0, 0, 0]),
            lower=float(lim.get("lower")) if lim is not None else 0.0,
            upper=float(lim.get("upper")) if lim is not None else 0.0))
    child_links = {j["child"] for j in joints}
    root_link = [n for n in links if n not in child_links]
    assert len(root_link) == 1
    bodies = merge_bodies(links, joints, root_link[0])
    by_joint = {b["joint"]["name"]: i for i, b in enumerate(bodies) if b["joint"] is not None}
    order = [0] + [by_joint[n] for n in motor_names]
    assert sorted(order) == list(range(len(bodies))), "unexpected set of movable joints"
    remap = {old: new for new, old in enumerate(order)}
    bodies = [bodies[i] for i in order]
    for b in bodies:
        b["parent"] = remap[b["parent"]] if b["parent"] >= 0 else -1
    return bodies, links


def emit_arm(base_bodies):
    """rex_arm.urdf (mark='arm'): the 13 base-mark bodies must be identical; emit the 6 arm bodies as a general
    serial chain (joint frames carry fixed rotations, axes are +-z)."""
    bodies, _ = load_bodies(DEFAULT_ARM_URDF, BASE_MOTOR_NAMES + ARM_MOTOR_NAMES)
    assert len(bodies) == 19
    for a, b in zip(bodies[:13], base_bodies):
        assert abs(a["mass"] - b["mass"]) < 1e-12 and np.allclose(a["I"], b["I"]) and np.allclose(a["com"], b["com"]), a["name"]
    arm = bodies[13:]
    out = []
    w = out.append
    w("// GENERATED by tools/compile_model.py from the reference's rex_arm.urdf -- do not edit.")
    w("// The 6 arm bodies of mark='arm' (rex_gym/util/pybullet_data/assets/urdf/rex_arm.urdf:610-791); bodies 0..12 are")
    w("// identical to rex.urdf (asserted) and come from rex_model_gen.h.  Same Bullet-style inertias (compound AABB),")
    w("// fixed tips merged.  Body 13+k hangs off REXA_PARENT[k] through joint 12+k: the joint frame sits at REXA_POS[k]")
    w("// in the parent body frame with fixed rotation REXA_E0[k] (joint-frame axes in parent coordinates, row-major),")
    w("// and turns about REXA_AXIS[k] (unit vector in the joint = child frame).")
    w("#ifndef REX_ARM_MODEL_GEN_H")
    w("#define REX_ARM_MODEL_GEN_H")
    w('#include "rex_model_gen.h"')
    w("#define REXA_NJ 6")
    w("REX_CONST int REXA_PARENT[REXA_NJ] = {" + ", ".join(str(b["parent"]) for b in arm) + "};")
    w("REX_CONST double REXA_POS[REXA_NJ][3] = {")
    for b in arm:
        w("  {" + ", ".join(fmt(v) for v in b["joint"]["xyz_in_body"]) + "},  /* " + b["joint"]["name"] + " */")
    w("};")
    w("REX_CONST double REXA_E0[REXA_NJ][9] = {")
    for b in arm:
        w("  {" + ", ".join(fmt(v) for v in b["joint"]["R_in_body"].ravel()) + "},")
    w("};")
    w("REX_CONST double REXA_AXIS[REXA_NJ][3] = {")
    for b in arm:
        a = b["joint"]["axis"] / np.linalg.norm(b["joint"]["axis"])
        w("  {" + ", ".join(fmt(v) for v in a) + "},")
    w("};")
    w("REX_CONST double REXA_LOWER[REXA_NJ] = {" + ", ".join(fmt(b["joint"]["lower"]) for b in arm) + "};")
    w("REX_CONST double REXA_UPPER[REXA_NJ] = {" + ", ".join(fmt(b["joint"]["upper"]) for b in arm) + "};")
    w("REX_CONST double REXA_MASS[REXA_NJ] = {" + ", ".join(fmt(b["mass"]) for b in arm) + "};")
    w("REX_CONST double REXA_COM[REXA_NJ][3] = {")
    for b in arm:
        w("  {" + ", ".join(fmt(v) for v in b["com"]) + "},  /* " + b["name"] + " */")
    w("};")
    w("/* rotational inertia about the COM, body axes: xx, yy, zz, xy, xz, yz */")
    w("REX_CONST double REXA_INERTIA[REXA_NJ][6] = {")
    for b in arm:
        I = b["I"]
        w("  {" + ", ".join(fmt(v) for v in (I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2])) + "},")
    w("};")
    for b in arm:
        a = b["joint"]["axis"]
        assert abs(a[0]) < 1e-12 and abs(a[1]) < 1e-12 and abs(abs(a[2]) - 1) < 1e-12, "arm axes are +-z in rex_arm.urdf"
        assert np.allclose(b["I"], np.diag(np.diag(b["I"]))), "arm inertias are diagonal"
        E = b["joint"]["R_in_body"]
        assert np.allclose(np.abs(E), np.round(np.abs(E))), "arm joint frames are signed permutations"
    w("/* all arm axes are +-z of the joint frame: sign per joint (used by the HIP arm chain) */")
    w("REX_CONST double REXA_AXIS_SIGN[REXA_NJ] = {" + ", ".join(fmt(b["joint"]["axis"][2]) for b in arm) + "};")
    w("/* ARM_POSES['rest'] (rex_gym/model/rex_constants.py:3-8): the command the envs append for the arm motors */")
    w("#ifndef REX_DIAG_ARM_REST_INSIDE")
    w("REX_CONST double REXA_REST[REXA_NJ] = {-1.6, -1.6, 0.0, 0.0, 1.6, 0.0};")
    w("#else")
    w("/* DIAGNOSTIC builds only (test artefacts: the diagnostic twins of the library and of the checker -- never the product): the reference's rest pose")
    w("   commands m1, m2, m5 0.1 rad BEYOND their +-1.5 rad bounds, so their limit rows switch with the last bit of the joint angle and")
    w("   no two float paths keep the same event sequence; with the three targets REX_DIAG_ARM_REST_INSIDE rad INSIDE the bounds the arm's rows")
    w("   stay quiet and the 200-step parity window of the mark-arm workload measures arithmetic, not row flicker (tests/test_gpu_parity.py) */")
    w("REX_CONST double REXA_REST[REXA_NJ] = {-1.5 + (REX_DIAG_ARM_REST_INSIDE), -1.5 + (REX_DIAG_ARM_REST_INSIDE), 0.0, 0.0, 1.5 - (REX_DIAG_ARM_REST_INSIDE), 0.0};")
    w("#endif")
    w("#endif /* REX_ARM_MODEL_GEN_H */")
    with open(DEFAULT_ARM_OUT, "w") as f:
        f.write("\n".join(out) + "\n")
    print(f"wrote {DEFAULT_ARM_OUT}: arm mass {sum(b['mass'] for b in arm):.4f} kg")
    for i, b in enumerate(arm):
        print(13 + i, b["name"], "parent", b["parent"], "m=%.3f" % b["mass"], "com", np.round(b["com"], 5), "I", np.round(b["I"], 7).tolist())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--urdf", default=DEFAULT_URDF)
    ap.add_argument("--out", default=DEFAULT_OUT)
    args = ap.parse_args()
    urdf_dir = os.path.dirname(args.urdf)
    root = ET.parse(args.urdf).getroot()
    links = {l.get("name"): Link(l, urdf_dir) for l in root.findall("link")}
    joints = []
    for j in root.findall("joint"):
        xyz, rpy = parse_origin(j)
        ax = j.find("axis")
        lim = j.find("limit")
        joints.append(dict(
            name=j.get("name"), type=j.get("type"), parent=j.find("parent").get("link"),
            child=j.find("child").get("link"), xyz=xyz, rpy=rpy,
            axis=np.array([float(v) for v in ax.get("xyz").split()]) if ax is not None else np.array([1.0, 0, 0]),
            lower=float(lim.get("lower")) if lim is not None else 0.0,
            upper=float(lim.get("upper")) if lim is not None else 0.0))
    child_links = {j["child"] for j in joints}
    root_link = [n for n in links if n not in child_links]
    assert len(root_link) == 1
    bodies = merge_bodies(links, joints, root_link[0])

    # reorder movable bodies into the reference motor order
    by_joint = {b["joint"]["name"]: i for i, b in enumerate(bodies) if b["joint"] is not None}
    order = [0] + [by_joint[n] for n in BASE_MOTOR_NAMES]
    assert sorted(order) == list(range(len(bodies))), "rex.urdf (base mark) expected: 1 base + 12 motor bodies"
    remap = {old: new for new, old in enumerate(order)}
    bodies = [bodies[i] for i in order]
    for b in bodies:
        b["parent"] = remap[b["parent"]] if b["parent"] >= 0 else -1

    nb = len(bodies)
    total_mass = sum(b["mass"] for b in bodies)

    # toe cylinder: the hull is a half cylinder about the mesh y axis; the URDF collision origin
    # (rpy about y, xyz along y) keeps that axis on the toe-link y axis.
    toes = []
    toe_radius = None
    toe_halflen = None
    for bi, b in enumerate(bodies):
        for ln, xyz, R in b["members"]:
            for kind, prm, cx, crpy in links[ln].collisions:
                if kind != "mesh":
                    continue
                Rc = rpy_to_mat(crpy)
                assert abs(crpy[0]) < 1e-12 and abs(crpy[2]) < 1e-12, "toe hull: rotation about y only"
                rad = float(np.sqrt(prm[:, 0] ** 2 + prm[:, 2] ** 2).max())
                ymin, ymax = prm[:, 1].min(), prm[:, 1].max()
                ctr_mesh = np.array([0.0, 0.5 * (ymin + ymax), 0.0])
                ctr = xyz + R @ (Rc @ ctr_mesh + cx)
                axis = R @ Rc @ np.array([0.0, 1.0, 0.0])
                toes.append((bi, ctr, axis))
                toe_radius = rad
                toe_halflen = 0.5 * (ymax - ymin)
    assert len(toes) == 4
    toes.sort(key=lambda t: t[0])

    out = []
    w = out.append
    w("// GENERATED by tools/compile_model.py from the reference's rex.urdf -- do not edit.")
    w("// Robot DATA (masses, offsets, limits, collision extents) of nicrusso7/rex-gym:")
    w("//   rex_gym/util/pybullet_data/assets/urdf/rex.urdf:15-609, stl/foot.stl.")
    w("// Inertias are recomputed from collision geometry the way Bullet's URDF importer does when")
    w("// URDF_USE_INERTIA_FROM_FILE is absent (rex_gym/model/rex.py:276-287); fixed links are merged.")
    w("#ifndef REX_MODEL_GEN_H")
    w("#define REX_MODEL_GEN_H")
    w("#ifdef __cplusplus")
    w("#define REX_CONST static constexpr")
    w("#else")
    w("#define REX_CONST static const")
    w("#endif")
    w("")
    w(f"#define REX_NB {nb}            /* bodies: base + 12 motor links, motor order of mark_constants.py:3-8 */")
    w(f"#define REX_NJ {nb - 1}            /* actuated revolute joints */")
    w("#define REX_NLEG 4           /* leg order: front_left, front_right, rear_left, rear_right */")
    w(f"#define REX_TOTAL_MASS {fmt(total_mass)}")
    w("")
    w("/* body i>=1 hangs off REX_PARENT[i] through joint i-1; joint frames have no fixed rotation */")
    w("REX_CONST int REX_PARENT[REX_NB] = {" + ", ".join(str(b["parent"]) for b in bodies) + "};")
    axes = []
    for b in bodies[1:]:
        j = b["joint"]
        assert np.allclose(j["R_in_body"], np.eye(3)), "leg joints carry no rpy in rex.urdf"
        a = j["axis"]
        k = int(np.argmax(np.abs(a)))
        assert np.allclose(np.abs(a), np.eye(3)[k]) and a[k] > 0
        axes.append(k)
    w("/* joint axis index in the child frame: 0 = x, 1 = y, 2 = z */")
    w("REX_CONST int REX_JOINT_AXIS[REX_NJ] = {" + ", ".join(str(a) for a in axes) + "};")
    w("/* joint origin in the parent BODY frame [m] */")
    w("REX_CONST double REX_JOINT_POS[REX_NJ][3] = {")
    for b in bodies[1:]:
        w("  {" + ", ".join(fmt(v) for v in b["joint"]["xyz_in_body"]) + "},  /* " + b["joint"]["name"] + " */")
    w("};")
    w("REX_CONST double REX_JOINT_LOWER[REX_NJ] = {" + ", ".join(fmt(b["joint"]["lower"]) for b in bodies[1:]) + "};")
    w("REX_CONST double REX_JOINT_UPPER[REX_NJ] = {" + ", ".join(fmt(b["joint"]["upper"]) for b in bodies[1:]) + "};")
    w("/* merged-body mass [kg], COM in body frame [m], rotational inertia about the COM, body axes")
    w("   (xx, yy, zz, xy, xz, yz) [kg m^2] */")
    w("REX_CONST double REX_MASS[REX_NB] = {" + ", ".join(fmt(b["mass"]) for b in bodies) + "};")
    w("REX_CONST double REX_COM[REX_NB][3] = {")
    for b in bodies:
        w("  {" + ", ".join(fmt(v) for v in b["com"]) + "},  /* " + b["name"] + " */")
    w("};")
    w("REX_CONST double REX_INERTIA[REX_NB][6] = {")
    for b in bodies:
        I = b["I"]
        w("  {" + ", ".join(fmt(v) for v in (I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2])) + "},")
    w("};")
    w("")
    w("/* toe contact geometry: cylinder segment about the toe-link y axis (stl/foot.stl hull), per leg */")
    w(f"#define REX_TOE_RADIUS {fmt(toe_radius)}       /* hull radius [m]; the solver adds the 1 mm Bullet margin */")
    w(f"#define REX_TOE_HALFLEN {fmt(toe_halflen)}")
    w(f"#define REX_COLLISION_MARGIN {fmt(URDF_COLLISION_MARGIN)}")
    w("REX_CONST int REX_TOE_BODY[REX_NLEG] = {" + ", ".join(str(t[0]) for t in toes) + "};")
    w("REX_CONST double REX_TOE_CENTER[REX_NLEG][3] = {")
    for t in toes:
        w("  {" + ", ".join(fmt(v) for v in t[1]) + "},")
    w("};")
    w("REX_CONST double REX_TOE_AXIS[REX_NLEG][3] = {")
    for t in toes:
        w("  {" + ", ".join(fmt(v if abs(v) > 1e-15 else 0.0) for v in t[2]) + "},")
    w("};")
    w("")
    # collision boxes of every link (rex.urdf:15-33,63-108,119-124,151-156,170-175), expressed in the frame of the
    # merged body they ride on: the candidates of the body-vs-ground contact rows.  All <collision> origins in rex.urdf
    # are pure translations, so every box is axis-aligned with its body frame.
    boxes = []
    for bi, b in enumerate(bodies):
        for ln, xyz, R in sorted(b["members"], key=lambda m: m[0] != b["name"]):
            for kind, prm, cx, crpy in links[ln].collisions:
                if kind != "box":
                    continue
                assert not crpy.any() and np.allclose(R, np.eye(3)), "rotated collision box"
                boxes.append((bi, xyz + cx, np.asarray(prm) / 2.0, ln))
    w("/* link collision boxes in body frames: body index, centre [m], half extents [m] (Bullet keeps a box's margin")
    w("   inside its extents, so these are the contact surfaces) */")
    w(f"#define REX_NBOX {len(boxes)}")
    w("REX_CONST int REX_BOX_BODY[REX_NBOX] = {" + ", ".join(str(bx[0]) for bx in boxes) + "};")
    w("REX_CONST double REX_BOX_CENTER[REX_NBOX][3] = {")
    for bx in boxes:
        w("  {" + ", ".join(fmt(v) for v in bx[1]) + "},  /* " + bx[3] + " */")
    w("};")
    w("REX_CONST double REX_BOX_HALF[REX_NBOX][3] = {")
    for bx in boxes:
        w("  {" + ", ".join(fmt(v) for v in bx[2]) + "},")
    w("};")
    w("")
    # leg-structured scalar view (the four legs are mirror images); used by the HIP kernels, which
    # unroll the star topology statically instead of walking the generic table.
    def leg_bodies(l):
        return bodies[1 + 3 * l], bodies[2 + 3 * l], bodies[3 + 3 * l]
    s0, u0, f0 = leg_bodies(0)
    sx = []
    sy = []
    for l in range(4):
        s, u, f = leg_bodies(l)
        hip = s["joint"]["xyz_in_body"]
        upp = u["joint"]["xyz_in_body"]
        kne = f["joint"]["xyz_in_body"]
        h0 = s0["joint"]["xyz_in_body"]
        assert abs(abs(hip[0]) - abs(h0[0])) < 1e-12 and abs(abs(hip[1]) - abs(h0[1])) < 1e-12 and hip[2] == 0
        assert upp[0] == 0 and upp[2] == 0 and abs(abs(upp[1]) - abs(u0["joint"]["xyz_in_body"][1])) < 1e-12
        assert np.allclose(kne, f0["joint"]["xyz_in_body"]) and kne[1] == 0
        assert np.sign(upp[1]) == np.sign(hip[1])
        for a, b in ((s, s0), (u, u0), (f, f0)):
            assert abs(a["mass"] - b["mass"]) < 1e-12 and np.allclose(a["I"], b["I"]) and np.allclose(a["com"], b["com"])
            assert np.allclose(a["I"], np.diag(np.diag(a["I"]))), "leg inertias are diagonal in rex.urdf"
        assert not s["com"].any() and not u["com"].any() and not f["com"][:2].any()
        sx.append(int(np.sign(hip[0])))
        sy.append(int(np.sign(hip[1])))
    assert not bodies[0]["com"].any() and np.allclose(bodies[0]["I"], np.diag(np.diag(bodies[0]["I"])))
    w("/* ---- leg-structured view (asserted mirror-symmetric by the compiler) ---- */")
    w("/* hip joint at (SX*HIP_X, SY*HIP_Y, 0) in base; upper-leg joint at (0, SY*UPPER_Y, 0) in shoulder;")
    w("   knee at (KNEE_X, 0, KNEE_Z) in upper leg; axes x, y, y */")
    w("REX_CONST int REX_LEG_SX[REX_NLEG] = {" + ", ".join(str(v) for v in sx) + "};")
    w("REX_CONST int REX_LEG_SY[REX_NLEG] = {" + ", ".join(str(v) for v in sy) + "};")
    w(f"#define REX_HIP_X {fmt(abs(s0['joint']['xyz_in_body'][0]))}")
    w(f"#define REX_HIP_Y {fmt(abs(s0['joint']['xyz_in_body'][1]))}")
    w(f"#define REX_UPPER_Y {fmt(abs(u0['joint']['xyz_in_body'][1]))}")
    w(f"#define REX_KNEE_X {fmt(f0['joint']['xyz_in_body'][0])}")
    w(f"#define REX_KNEE_Z {fmt(f0['joint']['xyz_in_body'][2])}")
    w(f"#define REX_TOE_Z {fmt(toes[0][1][2])}")
    for nm, b in (("BASE", bodies[0]), ("SHOULDER", s0), ("UPPER", u0), ("LOWER", f0)):
        w(f"#define REX_{nm}_MASS {fmt(b['mass'])}")
        w(f"#define REX_{nm}_IXX {fmt(b['I'][0, 0])}")
        w(f"#define REX_{nm}_IYY {fmt(b['I'][1, 1])}")
        w(f"#define REX_{nm}_IZZ {fmt(b['I'][2, 2])}")
    w(f"#define REX_LOWER_COM_Z {fmt(f0['com'][2])}")
    for l in range(4):
        for k in range(3):
            assert bodies[1 + 3 * l + k]["joint"]["lower"] == bodies[1 + k]["joint"]["lower"]
            assert bodies[1 + 3 * l + k]["joint"]["upper"] == bodies[1 + k]["joint"]["upper"]
    w("/* joint limits, identical on the four legs: shoulder, leg, foot */")
    w("REX_CONST double REX_LEG_LIMIT_LO[3] = {" + ", ".join(fmt(bodies[1 + k]["joint"]["lower"]) for k in range(3)) + "};")
    w("REX_CONST double REX_LEG_LIMIT_HI[3] = {" + ", ".join(fmt(bodies[1 + k]["joint"]["upper"]) for k in range(3)) + "};")
    w("")
    w("#endif /* REX_MODEL_GEN_H */")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        f.write("\n".join(out) + "\n")
    print(f"wrote {args.out}: {nb} bodies, total mass {total_mass:.4f} kg, toe r={toe_radius:.5f} hl={toe_halflen:.5f}")
    for i, b in enumerate(bodies):
        print(i, b["name"], "parent", b["parent"], "m=%.4f" % b["mass"], "com", np.round(b["com"], 6),
              "Idiag", np.diag(b["I"]))


if __name__ == "__main__":
    main()
    if os.path.exists(DEFAULT_ARM_URDF):
        emit_arm(load_bodies(DEFAULT_URDF, BASE_MOTOR_NAMES)[0])
