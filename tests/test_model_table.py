"""The generated robot table (rex_gym_amd/csrc/rex_model_gen.h) against numbers worked out BY HAND from the reference's
rex.urdf -- independent of tools/compile_model.py, which wrote the table, and of the oracle and kernels, which both
include it (so that none of the parity tests could see a mistake of the compiler).

Every expectation below is arithmetic on literals copied from
rex_gym/util/pybullet_data/assets/urdf/rex.urdf (line numbers cited), under the rule Bullet's URDF importer applies when
URDF_USE_INERTIA_FROM_FILE is absent (rex_gym/model/rex.py:276-287; SURVEY.md 9.2, UNVERIFIED against a live PyBullet):
a link's inertia is recomputed from its collision geometry -- the box formula m/12 (ly^2 + lz^2, lx^2 + lz^2, lx^2 + ly^2)
for one un-offset box, the same formula on the AABB of the collision shapes (about the link origin) when a shape is
offset, zero for a link without collision -- and the COM sits at the link origin (no <inertial><origin> in the file).
Fixed children are merged into their parent (mass, first moment, parallel axis).
"""
import os
import re

import numpy as np

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rex_gym_amd", "csrc", "rex_model_gen.h")
SRC = open(HEADER).read()


def define(name):
    return float(re.search(r"#define\s+%s\s+(\S+)" % name, SRC).group(1))


def array(name):
    body = re.search(r"%s\[[^=]*=\s*\{(.*?)\};" % name, SRC, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body)
    return [float(v) for v in re.findall(r"-?\d+\.?\d*(?:e-?\d+)?", body)]


def box(m, lx, ly, lz):
    return m / 12.0 * np.array([ly * ly + lz * lz, lx * lx + lz * lz, lx * lx + ly * ly])


def test_base_body_is_base_link_plus_the_two_chassis_links():
    # base_link: 1.20 kg, box 0.14 0.11 0.07 at the origin (rex.urdf:15-34); chassis_front_link 0.05 kg, box 0.058 0.11 0.07
    # offset x = -0.145 (:63-85); chassis_rear_link 0.05 kg, box 0.04 0.11 0.07 offset x = +0.135 (:86-108); both fixed to
    # base_link with no joint origin.  An offset box -> AABB rule: the box formula on its own size, about the link origin.
    inertia = box(1.20, 0.14, 0.11, 0.07) + box(0.05, 0.058, 0.11, 0.07) + box(0.05, 0.04, 0.11, 0.07)
    assert abs(define("REX_BASE_MASS") - (1.20 + 0.05 + 0.05)) < 1e-12
    np.testing.assert_allclose([define("REX_BASE_IXX"), define("REX_BASE_IYY"), define("REX_BASE_IZZ")], inertia, rtol=1e-9)
    np.testing.assert_allclose(inertia, [1.841666667e-3, 2.511516667e-3, 3.291516667e-3], rtol=1e-9)   # the digits, spelled out
    assert array("REX_COM")[0:3] == [0.0, 0.0, 0.0]
    np.testing.assert_allclose(array("REX_INERTIA")[0:3], inertia, rtol=1e-9)


def test_shoulder_and_upper_leg_links():
    # front_left_shoulder_link: 0.10 kg, one un-offset box 0.044 0.038 0.07 (rex.urdf:111-129) -> the box's own inertia
    shoulder = box(0.10, 0.044, 0.038, 0.07)
    np.testing.assert_allclose([define("REX_SHOULDER_IXX"), define("REX_SHOULDER_IYY"), define("REX_SHOULDER_IZZ")], shoulder, rtol=1e-9)
    np.testing.assert_allclose(shoulder, [5.286666667e-5, 5.696666667e-5, 2.816666667e-5], rtol=1e-9)
    assert define("REX_SHOULDER_MASS") == 0.10
    # front_left_leg_link: 0.1 kg, box 0.028 0.036 0.12 offset z = -0.05 (:143-161) -> AABB rule; its cover
    # front_left_leg_link_cover: 0.5 kg, NO collision (:130-142) -> a point mass at the shared origin (fixed joint, origin 0)
    upper = box(0.1, 0.028, 0.036, 0.12)
    np.testing.assert_allclose([define("REX_UPPER_IXX"), define("REX_UPPER_IYY"), define("REX_UPPER_IZZ")], upper, rtol=1e-9)
    np.testing.assert_allclose(upper, [1.308e-4, 1.265333333e-4, 1.733333333e-5], rtol=1e-9)
    assert abs(define("REX_UPPER_MASS") - (0.1 + 0.5)) < 1e-12
    # foot link 0.1 kg + toe link 0.005 kg fixed at z = -0.115 (:162-200, :230-234): mass and first moment of the merged body
    assert abs(define("REX_LOWER_MASS") - 0.105) < 1e-12
    assert abs(define("REX_LOWER_COM_Z") - (0.005 * -0.115 / 0.105)) < 1e-12


def test_joint_frames_axes_and_limits():
    # motor_front_left_shoulder: origin -0.093 -0.036 0, axis x, limits [-1.0, 1.0] (rex.urdf:201-208)
    # motor_front_left_leg:      origin 0 -0.052 0,      axis y, limits [-2.17, 0.97] (:209-216)
    # foot_motor_front_left:     origin -0.01 0 -0.12,   axis y, limits [-0.1, 2.59]  (:222-229)
    pos = np.array(array("REX_JOINT_POS")).reshape(12, 3)
    np.testing.assert_array_equal(pos[0], [-0.093, -0.036, 0.0])
    np.testing.assert_array_equal(pos[1], [0.0, -0.052, 0.0])
    np.testing.assert_array_equal(pos[2], [-0.01, 0.0, -0.12])
    assert [int(v) for v in array("REX_JOINT_AXIS")[:3]] == [0, 1, 1]
    assert array("REX_JOINT_LOWER")[:3] == [-1.0, -2.17, -0.1] and array("REX_JOINT_UPPER")[:3] == [1.0, 0.97, 2.59]
    # the mirrored legs: front right hip at y = +0.036 (:326-333), rear left at x = +0.093 (:451-458)
    np.testing.assert_array_equal(pos[3], [-0.093, 0.036, 0.0])
    np.testing.assert_array_equal(pos[6], [0.093, -0.036, 0.0])
    assert (define("REX_HIP_X"), define("REX_HIP_Y"), define("REX_UPPER_Y"), define("REX_KNEE_X"), define("REX_KNEE_Z"), define("REX_TOE_Z")) == \
        (0.093, 0.036, 0.052, -0.01, -0.12, -0.115)
    # link collision boxes as contact surfaces: half the URDF sizes, at the URDF offsets
    half = np.array(array("REX_BOX_HALF")).reshape(15, 3)
    ctr = np.array(array("REX_BOX_CENTER")).reshape(15, 3)
    np.testing.assert_allclose(half[:5], [[0.07, 0.055, 0.035], [0.02, 0.055, 0.035], [0.029, 0.055, 0.035], [0.022, 0.019, 0.035], [0.014, 0.018, 0.06]], atol=1e-15)
    np.testing.assert_array_equal(ctr[:5], [[0, 0, 0], [0.135, 0, 0], [-0.145, 0, 0], [0, 0, 0], [0, 0, -0.05]])
    assert abs(define("REX_TOTAL_MASS") - (1.3 + 4 * (0.1 + 0.6 + 0.105))) < 1e-12
