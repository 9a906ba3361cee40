/*
 * rex_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this library.
 * The product path (rex_gym_amd + librexsim_hip.so) never imports, links or calls it.
 *
 * What it restates, function by function, of nicrusso7/rex-gym @ v0.2.7 (/root/reference):
 *   orc_ik_*        rex_gym/model/kinematics.py:28-142
 *   orc_gait_*      rex_gym/model/gait_planner.py:22-134 (phase clock moved from wall time to an
 *                   explicit `now`)
 *   orc_motor_*     rex_gym/model/motor.py:76-143
 *   env logic       rex_gym/envs/gym/walk_env.py:125-154,207-324,326-362,
 *                   rex_gym/envs/gym/gallop_env.py:142-160,212-329,349-356,
 *                   rex_gym/envs/rex_gym_env.py:369-414,490-542, rex_gym/model/rex.py:158-163,
 *                   296-324,568-641,717-733
 *   physics         pybullet==2.8.3 (requirements.txt:2) is a third-party C++ dependency that is
 *                   NOT vendored in the reference and NOT installable here.  Its published
 *                   algorithm is restated: Featherstone articulated-body forward dynamics
 *                   (btMultiBody::computeAccelerationsArticulatedBodyAlgorithmMultiDof), contact rows
 *                   with ABA unit-impulse responses (calcAccelerationDeltasMultiDof) solved by
 *                   projected Gauss-Seidel in Bullet's row order (btMultiBodyConstraintSolver),
 *                   semi-implicit Euler.  SURVEY.md section 9.2 lists the Bullet behaviours assumed.
 *
 * PARITY STATUS: the controller half and the env-level command logic (goal / brake / hold
 * state machines of the five envs) are pinned against golden vectors generated from the
 * reference's own Python code (tests/golden/).  The physics half has no PyBullet to run against here
 * (**parity unpinned** in that sense); its one external pin is the record of 20 PyBullet episodes found in the
 * episode memory of the reference's shipped turn / ol checkpoint (tests/golden/pybullet_turn_ol_rollouts.npz,
 * tests/test_oracle_pybullet_record.py): gait events on the record's control steps, roll / pitch to 2.4e-3 rad
 * RMS over 25 steps -- not the 1e-3 rad of north_star.
 *
 * Build: `make -C oracle` -> oracle/_build/librex_oracle_f64.so (REAL=double) and _f32.so (float).
 * The physics deliberately uses a DIFFERENT formulation (body-coordinate spatial ABA, dense 18-dof
 * delta-velocity PGS) from the HIP kernels (world-aligned CRBA + Cholesky-whitened PGS), so that
 * agreement between the two is evidence for both.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rexsim.h"              /* state-word layout + config struct (interface only) */
#include "../rex_gym_amd/csrc/rex_model_gen.h" /* robot data compiled from rex.urdf */

#ifndef REAL
#define REAL double
#endif
typedef REAL real;

/* The library is built per robot mark: default = mark 'base' (rex.urdf, 13 bodies / 12 motors); -DREX_ARM =
 * mark 'arm' (rex_arm.urdf: the same 13 bodies + the 6-joint arm chain, 19 bodies / 18 motors,
 * model/mark_constants.py:14-27). */
#ifdef REX_ARM
#include "../rex_gym_amd/csrc/rex_arm_model_gen.h"
#define NB (REX_NB + REXA_NJ)
#define NJ (REX_NJ + REXA_NJ)
#else
#define NB REX_NB
#define NJ REX_NJ
#endif
#define NDOF (6 + NJ)
#define HIST_WORDS (3 * NJ + 7)   /* q, qd, observed torque, base quaternion, base angular velocity */
/* persistent-state word layout for NJ motors (rexsim.h spells out the NJ = 12 instance as enum RexStateWord) */
#define SW_Q 13
#define SW_QD (13 + NJ)
#define SW_PHI (13 + 2 * NJ)
#define SW_LASTT (SW_PHI + 1)
#define SW_ALPHA (SW_PHI + 2)
#define SW_TARGET (SW_PHI + 3)
#define SW_ENDTIME (SW_PHI + 4)
#define SW_AUX (SW_PHI + 5)
#define SW_FLAGS (SW_PHI + 6)
#define SW_STEPS (SW_PHI + 7)
#define SW_EPISODE (SW_PHI + 8)
#define SW_MOTOR_EN (SW_PHI + 9)
#define SW_OVERHEAT (SW_PHI + 10)
#define SW_HIST (SW_OVERHEAT + NJ / 2)
#define SW_WORDS (SW_HIST + 1)

/* ---- model table access (body i >= 1 is moved by joint i-1) ---- */
static int m_parent(int i) {
#ifdef REX_ARM
  if (i >= REX_NB) return REXA_PARENT[i - REX_NB];
#endif
  return REX_PARENT[i];
}
static real m_mass0(int i) {
#ifdef REX_ARM
  if (i >= REX_NB) return (real)REXA_MASS[i - REX_NB];
#endif
  return (real)REX_MASS[i];
}
static real m_com(int i, int k) {
#ifdef REX_ARM
  if (i >= REX_NB) return (real)REXA_COM[i - REX_NB][k];
#endif
  return (real)REX_COM[i][k];
}
static real P_INERTIA_SCALE, P_LEG_INERTIA_ADD;
static real m_inertia(int i, int k) {
#ifdef REX_ARM
  if (i >= REX_NB) return (real)REXA_INERTIA[i - REX_NB][k] * P_INERTIA_SCALE;
#endif
  return (real)REX_INERTIA[i][k] * P_INERTIA_SCALE + (i >= 1 && k < 3 ? P_LEG_INERTIA_ADD : 0);
}
static void m_joint_pos(int i, real r[3]) {
#ifdef REX_ARM
  if (i >= REX_NB) { for (int k = 0; k < 3; ++k) r[k] = (real)REXA_POS[i - REX_NB][k]; return; }
#endif
  for (int k = 0; k < 3; ++k) r[k] = (real)REX_JOINT_POS[i - 1][k];
}
static void m_joint_axis(int i, real a[3]) { /* unit axis in the joint (= child) frame */
  a[0] = a[1] = a[2] = 0;
#ifdef REX_ARM
  if (i >= REX_NB) { for (int k = 0; k < 3; ++k) a[k] = (real)REXA_AXIS[i - REX_NB][k]; return; }
#endif
  a[REX_JOINT_AXIS[i - 1]] = 1;
}
static void m_joint_E0(int i, real E0[3][3]) { /* fixed rotation of the joint frame in the parent body frame */
  for (int x = 0; x < 3; ++x) for (int y = 0; y < 3; ++y) E0[x][y] = (x == y);
#ifdef REX_ARM
  if (i >= REX_NB) for (int x = 0; x < 3; ++x) for (int y = 0; y < 3; ++y) E0[x][y] = (real)REXA_E0[i - REX_NB][3 * x + y];
#endif
}
static real m_lower(int j) {
#ifdef REX_ARM
  if (j >= REX_NJ) return (real)REXA_LOWER[j - REX_NJ];
#endif
  return (real)REX_JOINT_LOWER[j];
}
static real m_upper(int j) {
#ifdef REX_ARM
  if (j >= REX_NJ) return (real)REXA_UPPER[j - REX_NJ];
#endif
  return (real)REX_JOINT_UPPER[j];
}

#define ORC_API __attribute__((visibility("default")))

/* ---------- Bullet / PyBullet world parameters ---------- */
#define GRAVITY_Z ((real)-10.0)          /* rex_gym_env.py:314 */
static real MB_LINEAR_DAMPING = (real)0.04;  /* btMultiBody default, 9.2-4 */
static real MB_ANGULAR_DAMPING = (real)0.04; /* (variables only so the conservation tests can zero them) */
#define MB_MAX_COORD_VEL ((real)100.0)   /* btMultiBody::m_maxCoordinateVelocity */
#define CONTACT_ERP ((real)0.2)          /* btContactSolverInfo::m_erp2 */
#define CONTACT_BREAKING ((real)0.02)    /* gContactBreakingThreshold */
static real FRICTION_MU = (real)0.5;      /* toe 0.5 x plane 1.0, 9.2-7 (variable for sensitivity probes) */
#define ROBOT_INIT_Z ((real)0.21)        /* terrain.py:14-20 */

/* solver statistics (tools / tuning).  Counted only while DBG_STATS is set (orc_solver_stats / orc_solver_hist /
 * orc_body_points switch it on): shared counters that every OpenMP thread bumps once per sweep would otherwise drag the
 * cache line they share with the read-only solver parameters from core to core (on the 2-socket GPU hosts that halved
 * the 16-thread throughput of the cpu_baseline). */
static int DBG_STATS = 0;
static long DBG_HIST[64] __attribute__((aligned(128))) = {0};
static long DBG_LEGS[5][5] = {{0}};   /* [legs within breaking distance][legs carrying load] per substep (single-threaded census runs) */
static long DBG_SWEEPS = 0, DBG_SUBSTEPS = 0;
static real DBG_JOINT_FRICTION = 0, DBG_JOINT_VISC = 0; /* sensitivity probe only (off by default) */
/* ---- sensitivity probes over the Bullet-behaviour assumptions of SURVEY.md 9.2 (tools/physics_sensitivity.py).
 * Every probe defaults to the value the product kernels use; changing one changes the oracle only. ---- */
static real DBG_GAIT_CLOCK = 1;          /* wall-clock seconds per simulated second seen by GaitPlanner.loop */
static real P_ERP = CONTACT_ERP;          /* contact / joint-limit error reduction (btContactSolverInfo::m_erp2) */
static real P_SLOP = 0;                   /* btContactSolverInfo::m_linearSlop: penetration = distance + slop */
static real P_INERTIA_SCALE = 1;          /* rotational inertias of all links x this */
static real P_LEG_INERTIA_ADD = 0;        /* kg m^2 added to every leg link's principal inertias (rotor / armature) */
static int P_TOE_MODE = 2;                /* toe manifold stand-in: 2 = both cylinder ends, 1 = the lower end only,
                                             4 = both ends + the arc points 0.3 rad before / behind the lowest line */
static int P_LIMIT_EXACT = 1;             /* 1: joint-limit rows only once the bound is reached (Bullet's literal rule; the default);
                                             0: a predictive row from 0.15 rad before the bound (round 1's formulation) */
static real P_BREAKING = CONTACT_BREAKING;
static real P_MARGIN = (real)REX_COLLISION_MARGIN;
static int P_FRICTION_DIRS = 2;           /* 1: btPlaneSpace1's first tangent only (no SOLVER_USE_2_FRICTION_DIRECTIONS) */
static int P_CONE = 0;                    /* 1: friction pair clamped to the cone instead of the pyramid */
static real P_COVER_COM_Z = 0;            /* z of the 0.5 kg leg covers' centre of mass in the leg-link frame (URDF: 0) */
static real clampr(real x, real lo, real hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* =====================================================================================
 *                         controller: inverse kinematics
 * ===================================================================================== */
#define IK_L ((real)0.23)
#define IK_W ((real)0.075)
#define IK_HIP ((real)0.055)
#define IK_LEG ((real)0.10652)
#define IK_FOOT ((real)0.145)
#define IK_YDIST ((real)0.185)
#define IK_HEIGHT ((real)0.2)

/* kinematics.py:48-68 : RT = Rx(roll) Ry(pitch) Rz(yaw) * Trans(pos); identity rotation when all
 * three angles are exactly zero.  transform(coord) = RT * [coord;1] = R (coord + pos). */
static void ik_transform(const real coord[3], const real orn[3], const real pos[3], real out[3]) {
  real t[3] = {coord[0] + pos[0], coord[1] + pos[1], coord[2] + pos[2]};
  if (orn[0] != 0 || orn[1] != 0 || orn[2] != 0) {
    real cx = cos(orn[0]), sx = sin(orn[0]), cy = cos(orn[1]), sy = sin(orn[1]);
    real cz = cos(orn[2]), sz = sin(orn[2]);
    /* Rz */
    real a0 = cz * t[0] - sz * t[1], a1 = sz * t[0] + cz * t[1], a2 = t[2];
    /* Ry */
    real b0 = cy * a0 + sy * a2, b1 = a1, b2 = -sy * a0 + cy * a2;
    /* Rx */
    out[0] = b0;
    out[1] = cx * b1 - sx * b2;
    out[2] = sx * b1 + cx * b2;
  } else {
    out[0] = t[0]; out[1] = t[1]; out[2] = t[2];
  }
}

/* kinematics.py:80-102 */
static void ik_leg(const real c[3], int right_side, real out[3]) {
  const real hip = IK_HIP, leg = IK_LEG, foot = IK_FOOT;
  real domain = (c[1] * c[1] + c[2] * c[2] - hip * hip + c[0] * c[0] - leg * leg - foot * foot) / (2 * foot * leg);
  if (domain > 1) domain = (real)0.99;
  else if (domain < -1) domain = (real)-0.99;
  real gamma = atan2(-sqrt(1 - domain * domain), domain);
  real sq = c[1] * c[1] + c[2] * c[2] - hip * hip;
  if (sq < 0) sq = 0;
  real alpha = atan2(-c[0], sqrt(sq)) - atan2(foot * sin(gamma), leg + foot * cos(gamma));
  real hip_val = right_side ? -hip : hip;
  real theta = -atan2(c[2], c[1]) - atan2(sqrt(sq), hip_val);
  out[0] = theta; out[1] = -alpha; out[2] = -gamma;
}

/* kinematics.py:104-142.  frames: 4x3 rows FR,FL,RR,RL.  angles out: same leg order. */
static void ik_solve(const real orn[3], const real pos[3], const real frames[12], real angles[12], real tframes[12]) {
  const real hipv[4][3] = {{IK_L / 2, -IK_W / 2, 0}, {IK_L / 2, IK_W / 2, 0}, {-IK_L / 2, -IK_W / 2, 0}, {-IK_L / 2, IK_W / 2, 0}};
  real inv_orn[3] = {-orn[0], -orn[1], -orn[2]}, inv_pos[3] = {-pos[0], -pos[1], -pos[2]};
  for (int l = 0; l < 4; ++l) {
    real hv[3], coord[3], tc[3];
    ik_transform(hipv[l], orn, pos, hv);
    for (int k = 0; k < 3; ++k) coord[k] = frames[3 * l + k] - hv[k];
    ik_transform(coord, inv_orn, inv_pos, tc);
    ik_leg(tc, (l % 2) == 0, &angles[3 * l]);
    if (tframes) for (int k = 0; k < 3; ++k) tframes[3 * l + k] = hv[k] + tc[k];
  }
}

ORC_API void orc_ik_solve(int n, const real* orn, const real* pos, const real* frames, real* angles, real* tframes) {
  for (int i = 0; i < n; ++i)
    ik_solve(orn + 3 * i, pos + 3 * i, frames + 12 * i, angles + 12 * i, tframes ? tframes + 12 * i : 0);
}

/* =====================================================================================
 *                         controller: gait planner
 * ===================================================================================== */
/* Clocks and gait phases are double in BOTH builds (the reference's are Python floats): every discrete decision of the
 * controller -- the 0.99 latch, the phase wrap at 1, stance / swing at 0.5, the ramp and brake windows of the envs --
 * is then taken exactly as the reference takes it, also where the arithmetic around it is float (the product's rule,
 * rex_controller.h).  With a 5 ms control step and periods like 0.5 s those comparisons are exact ties in real numbers. */
typedef double clk;
typedef struct { clk phi, last_time; real alpha; } Gait;

static real binom11(int k) { /* gait_planner.py:22-24 with n = 11 */
  static const real f[12] = {1, 1, 2, 6, 24, 120, 720, 5040, 40320, 362880, 3628800, 39916800};
  return f[11] / (f[k] * f[11 - k]);
}

/* gait_planner.py:30-40 */
static void gait_stance(real phi_st, real v, real angle_deg, real out[3]) {
  real c = cos(angle_deg * (real)(M_PI / 180.0)), s = sin(angle_deg * (real)(M_PI / 180.0));
  const real A = (real)0.001, half_l = (real)0.05;
  real p = half_l * (1 - 2 * phi_st);
  out[0] = c * p * fabs(v);
  out[1] = -s * p * fabs(v);
  out[2] = -A * cos((real)M_PI / (2 * half_l) * p);
}

/* gait_planner.py:42-58: degree-11 Bernstein basis, only control points 0..9 are summed */
static void gait_swing(real phi_sw, real v, real angle_deg, real direction, real out[3]) {
  static const real PX[12] = {-0.04, -0.056, -0.06, -0.06, -0.06, 0., 0., 0., 0.06, 0.06, 0.056, 0.04};
  static const real PZ[12] = {0., 0., 0.0405, 0.0405, 0.0405, 0.0405, 0.0405, 0.0495, 0.0495, 0.0495, 0., 0.};
  real c = cos(angle_deg * (real)(M_PI / 180.0)), s = sin(angle_deg * (real)(M_PI / 180.0));
  real av = fabs(v);
  real sx = 0, sy = 0, sz = 0;
  for (int i = 0; i < 10; ++i) {
    real X = av * c * PX[i] * direction;
    real Y = av * s * (-X);
    real Z = av * PZ[i];
    real b = binom11(i) * pow(phi_sw, (real)i) * pow(1 - phi_sw, (real)(11 - i));
    sx += X * b; sy += Y * b; sz += Z * b;
  }
  out[0] = sx; out[1] = sy; out[2] = sz;
}

/* gait_planner.py:60-94 */
static void gait_step_trajectory(Gait* g, clk phi, real v, real angle, real w_rot, const real ctf[3], real direction, real coord[3]) {
  const clk step_offset = 0.5;
  const real R2D = (real)(180.0 / M_PI);
  if (phi >= 1) phi = phi - 1;
  real r = sqrt(ctf[0] * ctf[0] + ctf[1] * ctf[1]);
  real foot_angle = atan2(ctf[1], ctf[0]);
  real circle;
  if (w_rot >= 0) circle = 90 - (foot_angle - g->alpha) * R2D;
  else circle = 270 - (foot_angle - g->alpha) * R2D;
  real lng[3], rot[3];
  if (phi <= step_offset) {
    real ps = (real)(phi / step_offset);
    gait_stance(ps, v, angle, lng);
    gait_stance(ps, w_rot, circle, rot);
  } else {
    real ps = (real)((phi - step_offset) / (1 - step_offset));
    gait_swing(ps, v, angle, direction, lng);
    gait_swing(ps, w_rot, circle, direction, rot);
  }
  real mag = atan2(sqrt(rot[0] * rot[0] + rot[1] * rot[1]), r);
  if (ctf[1] > 0) g->alpha = (rot[0] < 0) ? -mag : mag;
  else g->alpha = (rot[0] < 0) ? mag : -mag;
  coord[0] = lng[0] + rot[0];
  coord[1] = lng[1] + rot[1];
  coord[2] = lng[2] + rot[2];
}

/* gait_planner.py:96-134; mode 0 = walk offsets, 1 = gallop offsets (gait_planner.py:15-20) */
static void gait_loop(Gait* g, int mode, real v, real angle, real w_rot, clk T, real direction, clk now, real frame[12]) {
  static const clk OFF[2][4] = {{0., 0.5, 0.5, 0.}, {0., 0., 0.8, 0.8}};
  const real base[4][3] = {{IK_L / 2, -IK_YDIST / 2, -IK_HEIGHT}, {IK_L / 2, IK_YDIST / 2, -IK_HEIGHT},
                           {-IK_L / 2, -IK_YDIST / 2, -IK_HEIGHT}, {-IK_L / 2, IK_YDIST / 2, -IK_HEIGHT}};
  if (T <= 0.01) T = 0.01;
  if (g->phi >= 0.99) g->last_time = now;
  g->phi = (now - g->last_time) / T;
  for (int l = 0; l < 4; ++l) {
    real sc[3];
    gait_step_trajectory(g, g->phi + OFF[mode][l], v, angle, w_rot, base[l], direction, sc);
    for (int k = 0; k < 3; ++k) frame[3 * l + k] = base[l][k] + sc[k];
  }
}

/* planner: n x (phi,last_time,alpha) in/out; params: n x (v, angle, w_rot, T, direction, now) -- doubles in both builds
 * (the clock values must arrive unrounded); the trajectory arithmetic runs in `real` */
ORC_API void orc_gait_loop(int n, int mode, double* planner, const double* params, real* frames) {
  for (int i = 0; i < n; ++i) {
    Gait g = {planner[3 * i], planner[3 * i + 1], (real)planner[3 * i + 2]};
    const double* p = params + 6 * i;
    gait_loop(&g, mode, (real)p[0], (real)p[1], (real)p[2], p[3], (real)p[4], p[5], frames + 12 * i);
    planner[3 * i] = g.phi; planner[3 * i + 1] = g.last_time; planner[3 * i + 2] = g.alpha;
  }
}

/* =====================================================================================
 *                         actuator model  (model/motor.py)
 * ===================================================================================== */
#define MOTOR_VOLTAGE ((real)32.0)
#define MOTOR_RESISTANCE ((real)0.186)
#define MOTOR_KT ((real)0.0954)
#define MOTOR_VISCOUS ((real)0.0)
#define MOTOR_VCLIP ((real)50.0)
#define MOTOR_OBS_LIMIT ((real)5.7)
#define OVERHEAT_TORQUE ((real)2.45) /* rex.py:13 */
#define OVERHEAT_TIME ((real)1.0)    /* rex.py:14 */

static void motor_torque(real cmd, real q, real qd, real qd_true, real kp, real kd, real* actual, real* observed) {
  static const real CUR[7] = {0, 10, 20, 30, 40, 50, 60};
  static const real TRQ[7] = {0, 1, 1.9, 2.45, 3.0, 3.25, 3.5};
  real pwm = -1 * kp * (q - cmd) - kd * qd;                              /* motor.py:111 */
  pwm = clampr(pwm, -1, 1);                                              /* :113 */
  *observed = clampr(MOTOR_KT * (pwm * MOTOR_VOLTAGE / MOTOR_RESISTANCE), -MOTOR_OBS_LIMIT, MOTOR_OBS_LIMIT); /* :127-129 */
  real vnet = clampr(pwm * MOTOR_VOLTAGE - (MOTOR_KT + MOTOR_VISCOUS) * qd_true, -MOTOR_VCLIP, MOTOR_VCLIP);  /* :132-135 */
  real cur = vnet / MOTOR_RESISTANCE;
  real sgn = (cur > 0) - (cur < 0);
  real mag = fabs(cur);
  real t;
  if (mag >= CUR[6]) t = TRQ[6];                                          /* np.interp clamps */
  else {
    int k = 0;
    while (k < 5 && mag >= CUR[k + 1]) ++k;
    t = TRQ[k] + (TRQ[k + 1] - TRQ[k]) * (mag - CUR[k]) / (CUR[k + 1] - CUR[k]);
  }
  *actual = sgn * t;                                                      /* strength ratio 1.0 */
}

ORC_API void orc_motor_torque(int n, const real* cmd, const real* q, const real* qd, const real* qd_true,
                              real kp, real kd, real* actual, real* observed) {
  for (int i = 0; i < n; ++i) motor_torque(cmd[i], q[i], qd[i], qd_true[i], kp, kd, &actual[i], &observed[i]);
}

/* =====================================================================================
 *                 quaternion helpers with PyBullet's conventions
 * ===================================================================================== */
static void quat_to_mat(const real q[4], real R[3][3]) { /* btMatrix3x3::setRotation */
  real d = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  real s = 2 / d;
  real xs = q[0] * s, ys = q[1] * s, zs = q[2] * s;
  real wx = q[3] * xs, wy = q[3] * ys, wz = q[3] * zs;
  real xx = q[0] * xs, xy = q[0] * ys, xz = q[0] * zs;
  real yy = q[1] * ys, yz = q[1] * zs, zz = q[2] * zs;
  R[0][0] = 1 - (yy + zz); R[0][1] = xy - wz; R[0][2] = xz + wy;
  R[1][0] = xy + wz; R[1][1] = 1 - (xx + zz); R[1][2] = yz - wx;
  R[2][0] = xz - wy; R[2][1] = yz + wx; R[2][2] = 1 - (xx + yy);
}

static void quat_to_euler(const real q[4], real rpy[3]) { /* pybullet getEulerFromQuaternion */
  real x = q[0], y = q[1], z = q[2], w = q[3];
  real sqx = x * x, sqy = y * y, sqz = z * z, squ = w * w;
  real sarg = -2 * (x * z - w * y);
  if (sarg <= (real)-0.99999) {
    rpy[1] = (real)(-0.5 * M_PI); rpy[0] = 0; rpy[2] = 2 * atan2(x, -y);
  } else if (sarg >= (real)0.99999) {
    rpy[1] = (real)(0.5 * M_PI); rpy[0] = 0; rpy[2] = 2 * atan2(-x, y);
  } else {
    rpy[1] = asin(sarg);
    rpy[0] = atan2(2 * (y * z + w * x), squ - sqx - sqy + sqz);
    rpy[2] = atan2(2 * (x * y + w * z), squ + sqx - sqy - sqz);
  }
}

static void euler_to_quat(const real rpy[3], real q[4]) { /* btQuaternion::setEulerZYX + normalize */
  real hr = rpy[0] * (real)0.5, hp = rpy[1] * (real)0.5, hy = rpy[2] * (real)0.5;
  real cr = cos(hr), sr = sin(hr), cp = cos(hp), sp = sin(hp), cy = cos(hy), sy = sin(hy);
  q[0] = sr * cp * cy - cr * sp * sy;
  q[1] = cr * sp * cy + sr * cp * sy;
  q[2] = cr * cp * sy - sr * sp * cy;
  q[3] = cr * cp * cy + sr * sp * sy;
  real n = 1 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int k = 0; k < 4; ++k) q[k] *= n;
}

/* =====================================================================================
 *       physics: articulated-body algorithm in body coordinates (spatial vectors [ang; lin])
 * ===================================================================================== */
typedef struct {
  real pos[3], quat[4], linvel[3], angvel[3]; /* base, world frame */
  real q[NJ], qd[NJ];
} Phys;

typedef struct {
  real Rw[NB][3][3];   /* body -> world */
  real pw[NB][3];      /* body origin, world */
  real X[NB][6][6];    /* motion transform parent -> body */
  real v[NB][6], c[NB][6];
  real IA[NB][6][6], pA[NB][6];
  real U[NB][6], Dinv[NB], u[NB];
  real S[NB][3];       /* joint axis of body i in its own frame (motion subspace = [S; 0]) */
  real L0[6][6];       /* Cholesky factor of the base articulated inertia */
  real a[NB][6];
  int fixed_base;      /* loadURDF(useFixedBase=True) (on_rack, rex.py:269-287): btMultiBody::m_fixedBase */
} Aba;

static void cross3(const real a[3], const real b[3], real o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
static real dot3(const real a[3], const real b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void matvec3(const real R[3][3], const real v[3], real o[3]) {
  for (int i = 0; i < 3; ++i) o[i] = R[i][0] * v[0] + R[i][1] * v[1] + R[i][2] * v[2];
}
static void matTvec3(const real R[3][3], const real v[3], real o[3]) {
  for (int i = 0; i < 3; ++i) o[i] = R[0][i] * v[0] + R[1][i] * v[1] + R[2][i] * v[2];
}

/* domain randomisation of the current substep (Rex.SetBaseMasses / SetLegMasses change masses only) */
static __thread real MASS_SCALE_BASE = 1, MASS_SCALE_LEG = 1;
static real body_mass(int i) { return m_mass0(i) * (i == 0 ? MASS_SCALE_BASE : MASS_SCALE_LEG); }

/* spatial inertia of body i about its own origin, body axes */
static void body_inertia6(int i, real I[6][6]) {
  real m = body_mass(i);
  real c[3] = {m_com(i, 0), m_com(i, 1), m_com(i, 2)};
  real Ic[3][3] = {{m_inertia(i, 0), m_inertia(i, 3), m_inertia(i, 4)},
                   {m_inertia(i, 3), m_inertia(i, 1), m_inertia(i, 5)},
                   {m_inertia(i, 4), m_inertia(i, 5), m_inertia(i, 2)}};
  real cc = dot3(c, c);
  real hx[3][3] = {{0, -m * c[2], m * c[1]}, {m * c[2], 0, -m * c[0]}, {-m * c[1], m * c[0], 0}};
  memset(I, 0, sizeof(real) * 36);
  for (int r = 0; r < 3; ++r)
    for (int s = 0; s < 3; ++s) {
      I[r][s] = Ic[r][s] + m * ((r == s ? cc : 0) - c[r] * c[s]);
      I[r][3 + s] = hx[r][s];
      I[3 + r][s] = -hx[r][s];
    }
  for (int r = 0; r < 3; ++r) I[3 + r][3 + r] = m;
}

static void crf_apply(const real v[6], const real f[6], real o[6]) { /* v x* f */
  real t1[3], t2[3];
  cross3(v, f, t1); cross3(v + 3, f + 3, t2);
  o[0] = t1[0] + t2[0]; o[1] = t1[1] + t2[1]; o[2] = t1[2] + t2[2];
  cross3(v, f + 3, o + 3);
}
static void crm_apply(const real v[6], const real m[6], real o[6]) { /* v x m */
  real t1[3], t2[3];
  cross3(v, m, o);
  cross3(v, m + 3, t1); cross3(v + 3, m, t2);
  o[3] = t1[0] + t2[0]; o[4] = t1[1] + t2[1]; o[5] = t1[2] + t2[2];
}
static void mat6vec(const real M[6][6], const real v[6], real o[6]) {
  for (int i = 0; i < 6; ++i) { real s = 0; for (int k = 0; k < 6; ++k) s += M[i][k] * v[k]; o[i] = s; }
}
static void mat6Tvec(const real M[6][6], const real v[6], real o[6]) {
  for (int i = 0; i < 6; ++i) { real s = 0; for (int k = 0; k < 6; ++k) s += M[k][i] * v[k]; o[i] = s; }
}

/* Bullet's per-link damping, expressed as a bias force (force needed for zero acceleration) */
static void add_damping(int i, const real v[6], real pA[6]) {
  real m = body_mass(i);
  real c[3] = {m_com(i, 0), m_com(i, 1), m_com(i, 2)};
  real wxc[3], vc[3];
  cross3(v, c, wxc);
  for (int k = 0; k < 3; ++k) vc[k] = v[3 + k] + wxc[k];
  real sv = sqrt(dot3(vc, vc)), sw = sqrt(dot3(v, v));
  real Iw[3] = {m_inertia(i, 0) * v[0] + m_inertia(i, 3) * v[1] + m_inertia(i, 4) * v[2],
                m_inertia(i, 3) * v[0] + m_inertia(i, 1) * v[1] + m_inertia(i, 5) * v[2],
                m_inertia(i, 4) * v[0] + m_inertia(i, 5) * v[1] + m_inertia(i, 2) * v[2]};
  real F[3], N[3], cxF[3];
  for (int k = 0; k < 3; ++k) {
    F[k] = m * vc[k] * (MB_LINEAR_DAMPING + MB_LINEAR_DAMPING * sv);
    N[k] = Iw[k] * (MB_ANGULAR_DAMPING + MB_ANGULAR_DAMPING * sw);
  }
  cross3(c, F, cxF);
  for (int k = 0; k < 3; ++k) { pA[k] += N[k] + cxF[k]; pA[3 + k] += F[k]; }
}

static int chol6(const real A[6][6], real L[6][6]) {
  memset(L, 0, sizeof(real) * 36);
  for (int j = 0; j < 6; ++j) {
    real s = A[j][j];
    for (int k = 0; k < j; ++k) s -= L[j][k] * L[j][k];
    if (!(s > 0)) return -1;
    L[j][j] = sqrt(s);
    for (int i = j + 1; i < 6; ++i) {
      real t = A[i][j];
      for (int k = 0; k < j; ++k) t -= L[i][k] * L[j][k];
      L[i][j] = t / L[j][j];
    }
  }
  return 0;
}
static void chol6_solve(const real L[6][6], const real b[6], real x[6]) {
  real y[6];
  for (int i = 0; i < 6; ++i) { real s = b[i]; for (int k = 0; k < i; ++k) s -= L[i][k] * y[k]; y[i] = s / L[i][i]; }
  for (int i = 5; i >= 0; --i) { real s = y[i]; for (int k = i + 1; k < 6; ++k) s -= L[k][i] * x[k]; x[i] = s / L[i][i]; }
}

/* forward kinematics + ABA; writes joint accelerations and base world accelerations */
static void aba_forward(const Phys* s, const real tau[NJ], Aba* A, real qdd[NJ], real wdot_w[3], real vdot_w[3]) {
  /* base */
  quat_to_mat(s->quat, A->Rw[0]);
  memcpy(A->pw[0], s->pos, sizeof(real) * 3);
  matTvec3(A->Rw[0], s->angvel, A->v[0]);
  matTvec3(A->Rw[0], s->linvel, A->v[0] + 3);
  for (int i = 0; i < NB; ++i) {
    if (i > 0) {
      int p = m_parent(i);
      real r[3], ax[3], E0[3][3];
      m_joint_pos(i, r); m_joint_axis(i, ax); m_joint_E0(i, E0);
      memcpy(A->S[i], ax, sizeof(ax));
      real cq = cos(s->q[i - 1]), sq = sin(s->q[i - 1]);
      /* rotation about the joint axis (Rodrigues), then the fixed frame rotation: child -> parent = E0 * Rq */
      real Rq[3][3], Rrel[3][3];
      const real axx[3][3] = {{0, -ax[2], ax[1]}, {ax[2], 0, -ax[0]}, {-ax[1], ax[0], 0}};
      for (int x = 0; x < 3; ++x) for (int y = 0; y < 3; ++y) Rq[x][y] = cq * (x == y) + sq * axx[x][y] + (1 - cq) * ax[x] * ax[y];
      for (int x = 0; x < 3; ++x) for (int y = 0; y < 3; ++y) { real t = 0; for (int k = 0; k < 3; ++k) t += E0[x][k] * Rq[k][y]; Rrel[x][y] = t; }
      real E[3][3];
      for (int x = 0; x < 3; ++x) for (int y = 0; y < 3; ++y) E[x][y] = Rrel[y][x];
      /* world pose */
      for (int x = 0; x < 3; ++x)
        for (int y = 0; y < 3; ++y) {
          real t = 0; for (int k = 0; k < 3; ++k) t += A->Rw[p][x][k] * Rrel[k][y];
          A->Rw[i][x][y] = t;
        }
      real rw[3]; matvec3(A->Rw[p], r, rw);
      for (int k = 0; k < 3; ++k) A->pw[i][k] = A->pw[p][k] + rw[k];
      /* X = [[E,0],[-E rx, E]] */
      real rx[3][3] = {{0, -r[2], r[1]}, {r[2], 0, -r[0]}, {-r[1], r[0], 0}};
      memset(A->X[i], 0, sizeof(real) * 36);
      for (int x = 0; x < 3; ++x)
        for (int y = 0; y < 3; ++y) {
          A->X[i][x][y] = E[x][y];
          A->X[i][3 + x][3 + y] = E[x][y];
          real t = 0; for (int k = 0; k < 3; ++k) t += E[x][k] * rx[k][y];
          A->X[i][3 + x][y] = -t;
        }
      mat6vec(A->X[i], A->v[p], A->v[i]);
      real vJ[6] = {ax[0] * s->qd[i - 1], ax[1] * s->qd[i - 1], ax[2] * s->qd[i - 1], 0, 0, 0};
      for (int k = 0; k < 3; ++k) A->v[i][k] += vJ[k];
      crm_apply(A->v[i], vJ, A->c[i]);
    }
    body_inertia6(i, A->IA[i]);
    real Iv[6];
    mat6vec(A->IA[i], A->v[i], Iv);
    crf_apply(A->v[i], Iv, A->pA[i]);
    add_damping(i, A->v[i], A->pA[i]);
  }
  for (int i = NB - 1; i >= 1; --i) {
    int p = m_parent(i);
    const real* S = A->S[i];
    for (int k = 0; k < 6; ++k) A->U[i][k] = A->IA[i][k][0] * S[0] + A->IA[i][k][1] * S[1] + A->IA[i][k][2] * S[2];
    real D = dot3(S, A->U[i]);
    A->Dinv[i] = 1 / D;
    A->u[i] = tau[i - 1] - dot3(S, A->pA[i]);
    real Ia[6][6], pa[6], Iac[6];
    for (int x = 0; x < 6; ++x) for (int y = 0; y < 6; ++y) Ia[x][y] = A->IA[i][x][y] - A->U[i][x] * A->U[i][y] * A->Dinv[i];
    mat6vec(Ia, A->c[i], Iac);
    for (int k = 0; k < 6; ++k) pa[k] = A->pA[i][k] + Iac[k] + A->U[i][k] * (A->u[i] * A->Dinv[i]);
    /* IA_p += X^T Ia X ; pA_p += X^T pa */
    real T[6][6];
    for (int x = 0; x < 6; ++x) for (int y = 0; y < 6; ++y) { real t = 0; for (int k = 0; k < 6; ++k) t += Ia[x][k] * A->X[i][k][y]; T[x][y] = t; }
    for (int x = 0; x < 6; ++x) for (int y = 0; y < 6; ++y) { real t = 0; for (int k = 0; k < 6; ++k) t += A->X[i][k][x] * T[k][y]; A->IA[p][x][y] += t; }
    real pp[6]; mat6Tvec(A->X[i], pa, pp);
    for (int k = 0; k < 6; ++k) A->pA[p][k] += pp[k];
  }
  chol6(A->IA[0], A->L0);
  real nb[6];
  for (int k = 0; k < 6; ++k) nb[k] = -A->pA[0][k];
  if (A->fixed_base) {   /* the base does not accelerate: relative to free fall it moves up at g */
    const real gup[3] = {0, 0, -GRAVITY_Z};
    A->a[0][0] = A->a[0][1] = A->a[0][2] = 0;
    matTvec3(A->Rw[0], gup, A->a[0] + 3);
  } else
    chol6_solve(A->L0, nb, A->a[0]); /* acceleration relative to free fall (gravity handled as a field) */
  for (int i = 1; i < NB; ++i) {
    int p = m_parent(i);
    mat6vec(A->X[i], A->a[p], A->a[i]);
    for (int k = 0; k < 6; ++k) A->a[i][k] += A->c[i][k];
    real Ua = 0; for (int k = 0; k < 6; ++k) Ua += A->U[i][k] * A->a[i][k];
    qdd[i - 1] = A->Dinv[i] * (A->u[i] - Ua);
    for (int k = 0; k < 3; ++k) A->a[i][k] += A->S[i][k] * qdd[i - 1];
  }
  /* world-frame classical accelerations of the base */
  real wxv[3], al[3];
  cross3(A->v[0], A->v[0] + 3, wxv);
  for (int k = 0; k < 3; ++k) al[k] = A->a[0][3 + k] + wxv[k];
  matvec3(A->Rw[0], A->a[0], wdot_w);
  matvec3(A->Rw[0], al, vdot_w);
  vdot_w[2] += GRAVITY_Z;
  if (A->fixed_base) for (int k = 0; k < 3; ++k) wdot_w[k] = vdot_w[k] = 0;
}

/* unit torque impulse on joint jbody-1 (between body jbody and its parent) -> generalized velocity change */
static void impulse_response_joint(const Aba* A, int jbody, real out[NDOF]) {
  real pA[NB][6], u[NB], da[NB][6];
  memset(pA, 0, sizeof(pA));
  for (int i = NB - 1; i >= 1; --i) {
    int p = m_parent(i);
    u[i] = (i == jbody ? (real)1 : (real)0) - dot3(A->S[i], pA[i]);
    real pa[6], pp[6];
    for (int k = 0; k < 6; ++k) pa[k] = pA[i][k] + A->U[i][k] * (u[i] * A->Dinv[i]);
    mat6Tvec(A->X[i], pa, pp);
    for (int k = 0; k < 6; ++k) pA[p][k] += pp[k];
  }
  real nb[6];
  for (int k = 0; k < 6; ++k) nb[k] = -pA[0][k];
  if (A->fixed_base) memset(da[0], 0, sizeof(da[0]));
  else chol6_solve(A->L0, nb, da[0]);
  for (int k = 0; k < 6; ++k) out[k] = da[0][k];
  for (int i = 1; i < NB; ++i) {
    int p = m_parent(i);
    mat6vec(A->X[i], da[p], da[i]);
    real Ua = 0; for (int k = 0; k < 6; ++k) Ua += A->U[i][k] * da[i][k];
    real dq = A->Dinv[i] * (u[i] - Ua);
    out[6 + i - 1] = dq;
    for (int k = 0; k < 3; ++k) da[i][k] += A->S[i][k] * dq;
  }
}

/* impulse f_k (6, body-k coords) applied to body k -> generalized velocity change */
static void impulse_response_at(const Aba* A, int kbody, const real fk[6], real out[NDOF]) {
  real pA[NB][6], u[NB], da[NB][6];
  memset(pA, 0, sizeof(pA));
  for (int k = 0; k < 6; ++k) pA[kbody][k] = -fk[k];
  for (int i = NB - 1; i >= 1; --i) {
    int p = m_parent(i);
    u[i] = -dot3(A->S[i], pA[i]);
    real pa[6], pp[6];
    for (int k = 0; k < 6; ++k) pa[k] = pA[i][k] + A->U[i][k] * (u[i] * A->Dinv[i]);
    mat6Tvec(A->X[i], pa, pp);
    for (int k = 0; k < 6; ++k) pA[p][k] += pp[k];
  }
  real nb[6];
  for (int k = 0; k < 6; ++k) nb[k] = -pA[0][k];
  if (A->fixed_base) memset(da[0], 0, sizeof(da[0]));
  else chol6_solve(A->L0, nb, da[0]);
  for (int k = 0; k < 6; ++k) out[k] = da[0][k];
  for (int i = 1; i < NB; ++i) {
    int p = m_parent(i);
    mat6vec(A->X[i], da[p], da[i]);
    real Ua = 0; for (int k = 0; k < 6; ++k) Ua += A->U[i][k] * da[i][k];
    real dq = A->Dinv[i] * (u[i] - Ua);
    out[6 + i - 1] = dq;
    for (int k = 0; k < 3; ++k) da[i][k] += A->S[i][k] * dq;
  }
}

/* ---- ground: the z = 0 box (plane.urdf) united with an optional random heightfield (model/terrain.py:32-54):
 * 256 x 256 vertex heights, 5 cm cells, centred on the origin; Bullet centres a heightfield shape on the middle of
 * its height range, so the field sits at raw - mid and the plane shows wherever that is below 0.
 * Cell triangulation: Bullet's default (diagonal from vertex (i,j+1) to (i+1,j)). ---- */
#define HF_N 256
#define HF_CELL ((real)0.05)
#define HF_INV_CELL ((real)20.0)    /* 1 / HF_CELL: written as a product on both sides (oracle and kernels) */
typedef struct { int nx, ny; real inv_cx, inv_cy, off_x, off_y, max_x, max_y; } HfGeom;   /* grid geometry of a heightfield pool */
static const HfGeom HF_RANDOM = {HF_N, HF_N, (real)20.0, (real)20.0, (real)127.5, (real)127.5, (real)254.999, (real)254.999};   /* model/terrain.py:32-54 */
typedef struct { const float* h; real mid; real base_mass_scale, leg_mass_scale, mu; int has_params; int body_contacts; HfGeom geo; int fixed_base; } Ground;

/* fid (event trace): the facet the point stands on -- 1 + 2 (cell index) + (upper triangle); 0 where the plane is on top */
static void ground_query_f(const Ground* g, real x, real y, real* height, real n[3], uint32_t* fid) {
  n[0] = 0; n[1] = 0; n[2] = 1; *height = 0;
  *fid = 0;
  if (!g || !g->h) return;
  const HfGeom* q = &g->geo;
  real fx = x * q->inv_cx + q->off_x, fy = y * q->inv_cy + q->off_y;
  fx = clampr(fx, 0, q->max_x); fy = clampr(fy, 0, q->max_y);
  int i = (int)fx, j = (int)fy;
  real u = fx - i, v = fy - j;
  real h00 = g->h[j * q->nx + i], h10 = g->h[j * q->nx + i + 1], h01 = g->h[(j + 1) * q->nx + i], h11 = g->h[(j + 1) * q->nx + i + 1];
  real hh, gx, gy;
  if (u + v <= 1) { hh = h00 + u * (h10 - h00) + v * (h01 - h00); gx = (h10 - h00) * q->inv_cx; gy = (h01 - h00) * q->inv_cy; }
  else { hh = h11 + (1 - u) * (h01 - h11) + (1 - v) * (h10 - h11); gx = (h11 - h01) * q->inv_cx; gy = (h11 - h10) * q->inv_cy; }
  hh -= g->mid;
  if (hh <= 0) return;                      /* the plane is on top here */
  real inv = 1 / sqrt(gx * gx + gy * gy + 1);
  *height = hh; n[0] = -gx * inv; n[1] = -gy * inv; n[2] = inv;
  *fid = 1u + 2u * (uint32_t)(j * q->nx + i) + (u + v <= 1 ? 0u : 1u);
}
static void ground_query(const Ground* g, real x, real y, real* height, real n[3]) { uint32_t fid; ground_query_f(g, x, y, height, n, &fid); }

/* ---- event trace (orc_set_event_trace; the kernels fold the same words the same way, rex_set_event_trace in include/rexsim.h):
 * per substep, which toe points are within the breaking distance, the heightfield facet under each of them (under the end
 * centre and under the contact point), which joint / arm bounds are reached -> one chained hash per env; the solver sweep
 * counts -> a second one.  Thread-local: set by env_step for the env it is stepping. ---- */
static __thread uint32_t* TR_EVENTS = 0;
static __thread uint32_t* TR_SWEEPS = 0;
static __thread uint32_t* TR_LEG_EVENTS = 0;   /* the events without the arm's bounds */
static __thread uint32_t TR_STEP_SWEEPS = 0;   /* sweeps of the current env.step() so far */
static uint32_t trace_point(int p, uint32_t f0, uint32_t f1) {
  uint32_t m = f0 * 0x9E3779B1u + f1 * 0x85EBCA77u + (uint32_t)(p + 1) * 0xC2B2AE3Du;
  m ^= m >> 15; m *= 0x2C1B3C6Du; m ^= m >> 12;
  return m;
}
static uint32_t trace_mix(uint32_t h, uint32_t w) { h = (h ^ w) * 0x01000193u; return h ^ (h >> 13); }

/* btPlaneSpace1 */
static void plane_space(const real n[3], real p[3], real q[3]) {
  if (fabs(n[2]) > (real)0.7071067811865475244) {
    real a = n[1] * n[1] + n[2] * n[2], k = 1 / sqrt(a);
    p[0] = 0; p[1] = -n[2] * k; p[2] = n[1] * k;
    q[0] = a * k; q[1] = -n[0] * p[2]; q[2] = n[0] * p[1];
  } else {
    real a = n[0] * n[0] + n[1] * n[1], k = 1 / sqrt(a);
    p[0] = -n[1] * k; p[1] = n[0] * k; p[2] = 0;
    q[0] = -n[2] * p[1]; q[1] = n[2] * p[0]; q[2] = a * k;
  }
}

#define MAX_TOE_POINTS 16
#define MAX_POINTS (MAX_TOE_POINTS + 4 + 2 * REX_NLEG)   /* toe manifold points + the kept link-box points (4 base, 2 per leg) */
#define MAX_ROWS (3 * MAX_POINTS + NJ)
#define LIMIT_ACTIVATION ((real)0.15)  /* a limit row further away than this cannot act: |qd| dt <= 100 * 1e-3 */
typedef struct {
  real J[NDOF], resp[NDOF];
  real rhs, invdiag, lo, hi, lambda;
  int normal_row; /* for friction rows: index of the normal row that bounds them; -1 for normals */
  int pair_row;   /* friction rows: the other tangent row of the same point (cone probe), else -1 */
  real mu;        /* friction rows: the coefficient of their pair of surfaces */
} Row;

/* Jacobian row of world direction d at world point P on body kbody, in (base body coords, joints) */
static void contact_jacobian(const Aba* A, int kbody, const real P[3], const real d[3], real J[NDOF], real fk[6]) {
  real rel[3], pk[3], dk[3], phi[6];
  for (int k = 0; k < 3; ++k) rel[k] = P[k] - A->pw[kbody][k];
  matTvec3(A->Rw[kbody], rel, pk);
  matTvec3(A->Rw[kbody], d, dk);
  cross3(pk, dk, phi);
  phi[3] = dk[0]; phi[4] = dk[1]; phi[5] = dk[2];
  memcpy(fk, phi, sizeof(phi));
  for (int k = 0; k < NDOF; ++k) J[k] = 0;
  int i = kbody;
  while (i > 0) {
    J[6 + i - 1] = dot3(A->S[i], phi);
    real pp[6];
    mat6Tvec(A->X[i], phi, pp);
    memcpy(phi, pp, sizeof(pp));
    i = m_parent(i);
  }
  for (int k = 0; k < 6; ++k) J[k] = phi[k];
}

/* body-vs-ground contact rows (config: RexConfig.body_contacts): the link collision boxes of rex.urdf:15-33,63-108,
 * 119-124,151-156,170-175 against the ground.  Bullet's box-box detector clips the box face that looks at the ground
 * and keeps at most four points per pair, PENETRATING ones only (a toe hull goes through GJK, which reports the
 * closest points of separated shapes as well); here the candidates of a box are the four corners of its face most
 * aligned with the ground normal under its centre, each against the ground under ITSELF, active once below it. */
#define BODY_ACTIVATION ((real)0.0)
/* A corner becomes a candidate 1 mm before it enters the other box (its row then only bounds the closing speed by
 * distance / dt, as every contact row at a positive distance does): Bullet keeps a manifold point alive after the boxes
 * have separated again (until contactBreakingThreshold = 0.02 m), so a resting link-link contact does not switch on and
 * off around zero depth there either. */
#define SELF_MARGIN ((real)0.001)
#define SELF_MU ((real)0.25)   /* btManifoldResult::calculateCombinedFriction: the product of the two links' URDF defaults, 0.5 x 0.5 */
static int BODY_CONTACTS = 0;
static int P_SELF_COLLISION = 1;          /* probe: leg boxes against the base body's boxes (only with body_contacts) */
static int P_SELF_DIAG_BULLET = 1;        /* a link-link row's diagonal as btMultiBodyConstraintSolver::setupMultiBodyContactConstraint builds it:
                                             denom0 + denom1, the two links' own terms without their cross term; 0 (probe): the exact diagonal */
static long DBG_SELF_POINTS = 0;

/* Jacobian row and impulse response of direction d at point P of body kbody -- against the world, or (self) against the
 * base body: the row of the relative velocity, whose base part vanishes (the pair's forces are internal to the robot) */
static real point_row(const Aba* A, int kbody, int self, const real P[3], const real d[3], Row* r) {
  real fk[6];
  contact_jacobian(A, kbody, P, d, r->J, fk);
  impulse_response_at(A, kbody, fk, r->resp);
  real diag_separate = 0;   /* the two links' own terms only, as btMultiBodyConstraintSolver sums them */
  if (self) {
    real J0[NDOF], r0[NDOF], f0[6];
    contact_jacobian(A, 0, P, d, J0, f0);
    impulse_response_at(A, 0, f0, r0);
    for (int k = 0; k < NDOF; ++k) diag_separate += r->J[k] * r->resp[k] + J0[k] * r0[k];
    for (int k = 0; k < NDOF; ++k) { r->J[k] -= J0[k]; r->resp[k] -= r0[k]; }
    for (int k = 0; k < 6; ++k) r->J[k] = 0;
  }
  return diag_separate;
}
static long DBG_BODY_POINTS = 0, DBG_BODY_SUBSTEPS = 0;   /* statistics (single-threaded census runs): active body points, substeps with any */

/* one 1 ms world step: the restated stepSimulation */
static void physics_substep(Phys* s, const real tau[NJ], real dt, int iterations, real residual_threshold, const Ground* ground) {
  static __thread Aba A;
  real qdd[NJ], wdot[3], vdot[3];
  real mu = FRICTION_MU;
  MASS_SCALE_BASE = 1; MASS_SCALE_LEG = 1;
  if (ground && ground->has_params) { MASS_SCALE_BASE = ground->base_mass_scale; MASS_SCALE_LEG = ground->leg_mass_scale; mu = ground->mu; }
  A.fixed_base = ground && ground->fixed_base;
  aba_forward(s, tau, &A, qdd, wdot, vdot);
  /* v <- v + dt a  (btMultiBodyDynamicsWorld::solveConstraints, before the constraint solve) */
  for (int k = 0; k < 3; ++k) { s->angvel[k] += dt * wdot[k]; s->linvel[k] += dt * vdot[k]; }
  for (int j = 0; j < NJ; ++j) s->qd[j] += dt * qdd[j];

  /* generalized velocity in solver coordinates: base in body axes, then joints */
  real nu[NDOF];
  matTvec3(A.Rw[0], s->angvel, nu);
  matTvec3(A.Rw[0], s->linvel, nu + 3);
  for (int j = 0; j < NJ; ++j) nu[6 + j] = s->qd[j];

  /* contact detection at the start-of-step pose: the two end points of each toe cylinder against the
   * ground.  Per end: ground normal n0 under the end centre -> the point of the end circle that is lowest
   * along n0 -> ground height / normal under THAT point -> signed distance to the local ground plane. */
  Row rows[MAX_ROWS];
  int nrow = 0, npoint = 0;
  int normal_of_point[MAX_POINTS];
  real PtP[MAX_POINTS][3], PtN[MAX_POINTS][3];
  int PtBody[MAX_POINTS];
  int PtSelf[MAX_POINTS];    /* 0: against the ground; 1: against a box of the base body (row of the RELATIVE velocity) */
  real PtDist[MAX_POINTS];
  memset(PtSelf, 0, sizeof(PtSelf));
  const real rad = (real)REX_TOE_RADIUS + P_MARGIN;
  uint32_t tr_active = 0, tr_facets = 0, tr_arm = 0;
  for (int l = 0; l < REX_NLEG; ++l) {
    int kb = REX_TOE_BODY[l];
    real ctr[3] = {(real)REX_TOE_CENTER[l][0], (real)REX_TOE_CENTER[l][1], (real)REX_TOE_CENTER[l][2]};
    real axl[3] = {(real)REX_TOE_AXIS[l][0], (real)REX_TOE_AXIS[l][1], (real)REX_TOE_AXIS[l][2]};
    real cw[3], aw[3];
    matvec3(A.Rw[kb], ctr, cw);
    matvec3(A.Rw[kb], axl, aw);
    for (int k = 0; k < 3; ++k) cw[k] += A.pw[kb][k];
    int first = npoint;
    for (int e = 0; e < 2; ++e) {
      real sgn = e == 0 ? (real)-1 : (real)1;
      real ce[3], n0[3], h0;
      uint32_t fid0, fid1;
      for (int k = 0; k < 3; ++k) ce[k] = cw[k] + sgn * (real)REX_TOE_HALFLEN * aw[k];
      ground_query_f(ground, ce[0], ce[1], &h0, n0, &fid0);
      real na = dot3(n0, aw);
      real dv[3] = {n0[0] - na * aw[0], n0[1] - na * aw[1], n0[2] - na * aw[2]};
      real dn = sqrt(dot3(dv, dv));
      real inv = dn > (real)1e-9 ? 1 / dn : 0;
      int narc = P_TOE_MODE == 4 ? 3 : 1;
      for (int a = 0; a < narc; ++a) {
        /* probe: extra manifold points on the arc, 0.3 rad either side of the lowest line (rotation about the axis) */
        real ang = a == 0 ? 0 : (a == 1 ? (real)0.3 : (real)-0.3);
        real u[3] = {-inv * dv[0], -inv * dv[1], -inv * dv[2]}, t[3], dir[3];
        cross3(aw, u, t);
        for (int k = 0; k < 3; ++k) dir[k] = cos(ang) * u[k] + sin(ang) * t[k];
        real P[3], n[3], h;
        for (int k = 0; k < 3; ++k) P[k] = a == 0 ? ce[k] - rad * inv * dv[k] : ce[k] + rad * dir[k];   /* a == 0: the lowest line itself */
        ground_query_f(ground, P[0], P[1], &h, n, &fid1);
        real dist = (P[2] - h) * n[2];
        if (dist < P_BREAKING) {
          if (a == 0) { tr_active |= 1u << (2 * l + e); if (ground && ground->h) tr_facets ^= trace_point(2 * l + e, fid0, fid1 | (fabs(n[2]) > (real)0.7071067811865475244 ? 0x80000000u : 0u)); }   /* bit 31: the branch plane_space takes */
          memcpy(PtP[npoint], P, sizeof(P));
          memcpy(PtN[npoint], n, sizeof(n));
          PtBody[npoint] = kb;
          PtDist[npoint] = dist;
          ++npoint;
        }
      }
    }
    if (P_TOE_MODE == 1 && npoint - first == 2) {   /* probe: a single manifold point, the lower end */
      int keep = PtDist[first] <= PtDist[first + 1] ? first : first + 1;
      if (keep != first) { memcpy(PtP[first], PtP[keep], sizeof(PtP[0])); memcpy(PtN[first], PtN[keep], sizeof(PtN[0])); PtDist[first] = PtDist[keep]; }
      npoint = first + 1;
    }
  }
  const int toe_points = npoint;
  if (BODY_CONTACTS || (ground && ground->body_contacts)) {
    /* candidates: the ground-facing face of every link box.  Kept: per GROUP of boxes -- the base with its two chassis
     * boxes, and each leg with its shoulder / leg / foot boxes -- the REX_BODY_SLOTS deepest penetrating corners (4 for the
     * base group, 2 per leg; Bullet keeps at most 4 points per box pair, deepest first).  Ties go to the earlier
     * candidate (box order of rex_model_gen.h, corner order below). */
    for (int grp = 0; grp < 1 + REX_NLEG; ++grp) {
      const int slots = grp == 0 ? 4 : 2;
      real bestD[4]; real bestP[4][3], bestN[4][3]; int bestB[4], bestS[4]; int nbest = 0;
      for (int b = 0; b < REX_NBOX; ++b) {
        int kb = REX_BOX_BODY[b];
        int g_of = kb == 0 ? 0 : 1 + (kb - 1) / 3;
        if (g_of != grp) continue;
        real cl[3] = {(real)REX_BOX_CENTER[b][0], (real)REX_BOX_CENTER[b][1], (real)REX_BOX_CENTER[b][2]};
        real cw[3];
        matvec3(A.Rw[kb], cl, cw);
        for (int k = 0; k < 3; ++k) cw[k] += A.pw[kb][k];
        real n0[3], h0;
        ground_query(ground, cw[0], cw[1], &h0, n0);
        /* reach of the box along the ground normal, per box axis: the face axis is the largest */
        real reach[3], sg[3];
        for (int ax = 0; ax < 3; ++ax) {
          real e_n = A.Rw[kb][0][ax] * n0[0] + A.Rw[kb][1][ax] * n0[1] + A.Rw[kb][2][ax] * n0[2];
          reach[ax] = (real)REX_BOX_HALF[b][ax] * fabs(e_n);
          sg[ax] = e_n > 0 ? (real)-1 : (real)1;      /* the side of the box that looks at the ground */
        }
        int fa = 0;
        if (reach[1] > reach[fa]) fa = 1;
        if (reach[2] > reach[fa]) fa = 2;
        int a1 = (fa + 1) % 3, a2 = (fa + 2) % 3;
        for (int c = 0; c < 4; ++c) {
          real loc[3];
          loc[fa] = sg[fa] * (real)REX_BOX_HALF[b][fa];
          loc[a1] = ((c & 1) ? (real)1 : (real)-1) * (real)REX_BOX_HALF[b][a1];
          loc[a2] = ((c & 2) ? (real)1 : (real)-1) * (real)REX_BOX_HALF[b][a2];
          real P[3], n[3], h;
          matvec3(A.Rw[kb], loc, P);
          for (int k = 0; k < 3; ++k) P[k] += cw[k];
          ground_query(ground, P[0], P[1], &h, n);
          real dist = (P[2] - h) * n[2];
          if (!(dist < BODY_ACTIVATION)) continue;
          /* insertion into the group's deepest-first list */
          int pos = nbest;
          while (pos > 0 && dist < bestD[pos - 1]) --pos;
          if (pos >= slots) continue;
          int last = nbest < slots ? nbest : slots - 1;
          for (int k = last; k > pos; --k) { bestD[k] = bestD[k - 1]; bestB[k] = bestB[k - 1]; bestS[k] = bestS[k - 1]; memcpy(bestP[k], bestP[k - 1], sizeof(bestP[0])); memcpy(bestN[k], bestN[k - 1], sizeof(bestN[0])); }
          bestD[pos] = dist; bestB[pos] = kb; bestS[pos] = 0; memcpy(bestP[pos], P, sizeof(P)); memcpy(bestN[pos], n, sizeof(n));
          if (nbest < slots) ++nbest;
        }
      }
      /* self collision (loadURDF(flags=URDF_USE_SELF_COLLISION), rex.py:276-281): the leg-link and foot-link boxes of this
       * leg against the three boxes of the base body (base_link, chassis_front, chassis_rear) -- the only admissible pairs
       * that ever overlap (profiles/r02_contact_census.md: the upper-leg box against the edge of the rolled base in
       * RexPosesEnv).  The face cases of Bullet's box-box detector reduced to their penetrating vertices: a corner of one
       * box inside (or within SELF_MARGIN of) the other, pushed out through the face of least penetration of the box that contains it; edge-edge
       * crossings without a contained corner generate nothing.  The candidates compete with the leg's ground candidates
       * for its two slots, deepest first.  Row: relative velocity of the leg link against the base body at the corner. */
      if (grp >= 1 && P_SELF_COLLISION) {
        for (int a = 0; a < 3; ++a)                /* boxes 0..2 of rex_model_gen.h: the base body's */
          for (int bb = 1; bb <= 2; ++bb) {        /* this leg's leg-link and foot-link boxes (the shoulder is the base's child) */
            const int b = 3 + 3 * (grp - 1) + bb, kb = REX_BOX_BODY[b];
            real cA[3], cB[3], t[3];
            for (int k = 0; k < 3; ++k) t[k] = (real)REX_BOX_CENTER[a][k];
            matvec3(A.Rw[0], t, cA);
            for (int k = 0; k < 3; ++k) { cA[k] += A.pw[0][k]; t[k] = (real)REX_BOX_CENTER[b][k]; }
            matvec3(A.Rw[kb], t, cB);
            for (int k = 0; k < 3; ++k) cB[k] += A.pw[kb][k];
            for (int side = 0; side < 2; ++side) {   /* 0: corners of the leg box inside the base box; 1: the other way round */
              const real (*RX)[3] = side == 0 ? A.Rw[0] : A.Rw[kb];      /* the containing box */
              const real (*RY)[3] = side == 0 ? A.Rw[kb] : A.Rw[0];      /* the box whose corners are tested */
              const real* cX = side == 0 ? cA : cB; const real* cY = side == 0 ? cB : cA;
              const double* hX = side == 0 ? REX_BOX_HALF[a] : REX_BOX_HALF[b];
              const double* hY = side == 0 ? REX_BOX_HALF[b] : REX_BOX_HALF[a];
              for (int c = 0; c < 8; ++c) {
                real loc[3] = {((c & 1) ? (real)1 : (real)-1) * (real)hY[0], ((c & 2) ? (real)1 : (real)-1) * (real)hY[1],
                               ((c & 4) ? (real)1 : (real)-1) * (real)hY[2]};
                real P[3], rel[3], inX[3];
                matvec3(RY, loc, P);
                for (int k = 0; k < 3; ++k) { P[k] += cY[k]; rel[k] = P[k] - cX[k]; }
                matTvec3(RX, rel, inX);
                int ax = -1; real depth = 0;
                int inside = 1;
                for (int k = 0; k < 3; ++k) {
                  real d = (real)hX[k] - fabs(inX[k]);
                  if (!(d + SELF_MARGIN > 0)) { inside = 0; break; }
                  if (ax < 0 || d < depth) { ax = k; depth = d; }
                }
                if (!inside) continue;
                /* direction of the push on the LEG link: out of the base box, or into the leg box's own face */
                real sg = inX[ax] >= 0 ? (real)1 : (real)-1;
                if (side == 1) sg = -sg;
                real n[3] = {sg * RX[0][ax], sg * RX[1][ax], sg * RX[2][ax]};
                real dist = -depth;
                int pos = nbest;
                while (pos > 0 && dist < bestD[pos - 1]) --pos;
                if (pos >= slots) continue;
                int last = nbest < slots ? nbest : slots - 1;
                for (int k = last; k > pos; --k) { bestD[k] = bestD[k - 1]; bestB[k] = bestB[k - 1]; bestS[k] = bestS[k - 1]; memcpy(bestP[k], bestP[k - 1], sizeof(bestP[0])); memcpy(bestN[k], bestN[k - 1], sizeof(bestN[0])); }
                bestD[pos] = dist; bestB[pos] = kb; bestS[pos] = 1; memcpy(bestP[pos], P, sizeof(P)); memcpy(bestN[pos], n, sizeof(n));
                if (nbest < slots) ++nbest;
                if (DBG_STATS) ++DBG_SELF_POINTS;
              }
            }
          }
      }
      for (int k = 0; k < nbest; ++k) {
        memcpy(PtP[npoint], bestP[k], sizeof(bestP[0]));
        memcpy(PtN[npoint], bestN[k], sizeof(bestN[0]));
        PtBody[npoint] = bestB[k];
        PtSelf[npoint] = bestS[k];
        PtDist[npoint] = bestD[k];
        ++npoint;
        if (DBG_STATS) ++DBG_BODY_POINTS;
      }
    }
    if (DBG_STATS && npoint > toe_points) ++DBG_BODY_SUBSTEPS;
  }
  /* non-contact rows come first in every sweep (btMultiBodyConstraintSolver::solveSingleIteration): the URDF joint
   * limits (btMultiBodyJointLimitConstraint, one unilateral row per bound).  btMultiBodyJointLimitConstraint::
   * createConstraintRows instantiates a bound's row only once the bound is reached (`if (penetration > 0) continue;`)
   * and then pushes back through the ERP term; only the near bound of a joint can be. */
  for (int j = 0; j < NJ; ++j) {
    real lo_gap = s->q[j] - m_lower(j), hi_gap = m_upper(j) - s->q[j];
    int lower = lo_gap < hi_gap;
    real gap = lower ? lo_gap : hi_gap;
    if (gap >= LIMIT_ACTIVATION) continue;
    if (P_LIMIT_EXACT && gap > 0) continue;   /* btMultiBodyJointLimitConstraint skips a row while its bound is not reached */
    if (j < 12) tr_active |= 1u << (8 + j); else tr_arm |= 1u << (j - 12);
    Row* r = &rows[nrow++];
    real sgn = lower ? (real)1 : (real)-1;
    for (int k = 0; k < NDOF; ++k) r->J[k] = 0;
    r->J[6 + j] = sgn;
    impulse_response_joint(&A, j + 1, r->resp);
    for (int k = 0; k < NDOF; ++k) r->resp[k] *= sgn;
    real diag = r->resp[6 + j] * sgn, vel = sgn * nu[6 + j];
    r->invdiag = 1 / diag;
    real poserr = 0, velerr = -vel;
    if (gap > 0) velerr -= gap / dt; else poserr = -gap * P_ERP / dt;
    r->rhs = (poserr + velerr) * r->invdiag;
    r->lo = 0; r->hi = (real)1e10; r->lambda = 0; r->normal_row = -1; r->pair_row = -1;
  }
  /* then all contact normals, then the friction pairs (Bullet's per-iteration order).  Among the normals the link-box
   * points come before the toe points, among the friction pairs after them (Bullet's own order among manifolds is that
   * of its broadphase pairs, i.e. arbitrary; this one keeps the toe rows contiguous for the kernels' row pipeline). */
  int order[MAX_POINTS], nord = 0;
  for (int p = toe_points; p < npoint; ++p) order[nord++] = p;
  for (int p = 0; p < toe_points; ++p) order[nord++] = p;
  for (int q = 0; q < npoint; ++q) {
    int p = order[q];
    Row* r = &rows[nrow];
    real dsep = point_row(&A, PtBody[p], PtSelf[p], PtP[p], PtN[p], r);
    real diag = 0, vel = 0;
    for (int k = 0; k < NDOF; ++k) { diag += r->J[k] * r->resp[k]; vel += r->J[k] * nu[k]; }
    if (PtSelf[p] && P_SELF_DIAG_BULLET) diag = dsep;
    r->invdiag = 1 / diag;
    real pen = PtDist[p] + P_SLOP;
    real poserr = 0, velerr = -vel;
    if (pen > 0) velerr -= pen / dt; else poserr = -pen * P_ERP / dt;
    r->rhs = (poserr + velerr) * r->invdiag;
    r->lo = 0; r->hi = (real)1e10; r->lambda = 0; r->normal_row = -1; r->pair_row = -1;
    normal_of_point[p] = nrow++;
  }
  nord = 0;
  for (int p = 0; p < toe_points; ++p) order[nord++] = p;
  for (int p = toe_points; p < npoint; ++p) order[nord++] = p;
  for (int q = 0; q < npoint; ++q)
    for (int d = 0; d < P_FRICTION_DIRS; ++d) {
      int p = order[q];
      Row* r = &rows[nrow];
      real t1[3], t2[3];
      plane_space(PtN[p], t1, t2);
      real dsep = point_row(&A, PtBody[p], PtSelf[p], PtP[p], d == 0 ? t1 : t2, r);
      r->mu = PtSelf[p] ? SELF_MU : mu;
      real diag = 0, vel = 0;
      for (int k = 0; k < NDOF; ++k) { diag += r->J[k] * r->resp[k]; vel += r->J[k] * nu[k]; }
      if (PtSelf[p] && P_SELF_DIAG_BULLET) diag = dsep;
      r->invdiag = 1 / diag;
      r->rhs = -vel * r->invdiag;
      r->lambda = 0; r->normal_row = normal_of_point[p];
      r->pair_row = P_FRICTION_DIRS == 2 ? (d == 0 ? nrow + 1 : nrow - 1) : -1;
      ++nrow;
    }
  /* projected Gauss-Seidel (btMultiBodyConstraintSolver::resolveSingleConstraintRowGeneric) */
  real dv[NDOF];
  for (int k = 0; k < NDOF; ++k) dv[k] = 0;
  for (int it = 0; it < iterations; ++it) {
    real worst = 0; /* btMultiBodyConstraintSolver::solveSingleIteration: max squared velocity residual */
    for (int i = 0; i < nrow; ++i) {
      Row* r = &rows[i];
      real lo = r->lo, hi = r->hi;
      if (r->normal_row >= 0) { hi = r->mu * rows[r->normal_row].lambda; lo = -hi; }
      real dvel = 0;
      for (int k = 0; k < NDOF; ++k) dvel += r->J[k] * dv[k];
      real dl = r->rhs - dvel * r->invdiag;
      real sum = r->lambda + dl;
      if (sum < lo) { dl = lo - r->lambda; sum = lo; }
      else if (sum > hi) { dl = hi - r->lambda; sum = hi; }
      if (P_CONE && r->normal_row >= 0 && r->pair_row >= 0) {   /* probe: implicit cone, |(l1, l2)| <= mu ln */
        real other = rows[r->pair_row].lambda, lim = r->mu * rows[r->normal_row].lambda;
        real un = r->lambda + (r->rhs - dvel * r->invdiag);
        real mag = sqrt(un * un + other * other);
        real want = mag > lim && mag > 0 ? un * lim / mag : un;
        sum = want; dl = want - r->lambda;
      }
      r->lambda = sum;
      for (int k = 0; k < NDOF; ++k) dv[k] += r->resp[k] * dl;
      real resid = dl / r->invdiag;
      if (resid * resid > worst) worst = resid * resid;
    }
    /* btSequentialImpulseConstraintSolver::solveGroupCacheFriendlyIterations: leave the sweep loop once
     * the residual is below m_leastSquaresResidualThreshold (PyBullet default 1e-7) */
    if (DBG_STATS) ++DBG_SWEEPS;
    ++TR_STEP_SWEEPS;
    if (worst <= residual_threshold) { if (DBG_STATS) DBG_HIST[it < 63 ? it : 63]++; break; }
    if (DBG_STATS && it == iterations - 1) DBG_HIST[63]++;
  }
  if (DBG_STATS) ++DBG_SUBSTEPS;
  if (TR_EVENTS) {
    *TR_EVENTS = trace_mix(trace_mix(trace_mix(*TR_EVENTS, tr_active), tr_facets), tr_arm);
    *TR_SWEEPS = trace_mix(*TR_SWEEPS, TR_STEP_SWEEPS);
    *TR_LEG_EVENTS = trace_mix(trace_mix(*TR_LEG_EVENTS, tr_active), tr_facets);
  }
  if (DBG_STATS) {   /* census: legs with a toe point within the breaking distance / legs that carry a normal impulse */
    int act[REX_NLEG] = {0}, load[REX_NLEG] = {0};
    for (int p = 0; p < toe_points; ++p) {
      int l = (PtBody[p] - 1) / 3;
      act[l] = 1;
      if (rows[normal_of_point[p]].lambda > 0) load[l] = 1;
    }
    int na = 0, nl = 0;
    for (int l = 0; l < REX_NLEG; ++l) { na += act[l]; nl += load[l]; }
    DBG_LEGS[na][nl]++;
  }
  /* apply, clamp (btMultiBody::applyDeltaVeeMultiDof), integrate positions with the NEW velocities */
  real dw[3], dl[3];
  matvec3(A.Rw[0], dv, dw);
  matvec3(A.Rw[0], dv + 3, dl);
  for (int k = 0; k < 3; ++k) {
    s->angvel[k] = clampr(s->angvel[k] + dw[k], -MB_MAX_COORD_VEL, MB_MAX_COORD_VEL);
    s->linvel[k] = clampr(s->linvel[k] + dl[k], -MB_MAX_COORD_VEL, MB_MAX_COORD_VEL);
  }
  for (int j = 0; j < NJ; ++j) s->qd[j] = clampr(s->qd[j] + dv[6 + j], -MB_MAX_COORD_VEL, MB_MAX_COORD_VEL);
  for (int k = 0; k < 3; ++k) s->pos[k] += dt * s->linvel[k];
  for (int j = 0; j < NJ; ++j) s->q[j] += dt * s->qd[j];
  /* orientation: q <- exp(w dt) * q  (world-frame angular velocity) */
  real w = sqrt(dot3(s->angvel, s->angvel));
  real ang = w * dt;
  real sc;
  if (w < (real)0.001) sc = (real)0.5 * dt - dt * dt * dt * (real)0.020833333333 * w * w;
  else sc = sin((real)0.5 * ang) / w;
  real dq[4] = {s->angvel[0] * sc, s->angvel[1] * sc, s->angvel[2] * sc, cos((real)0.5 * ang)};
  real* q = s->quat;
  real nq[4] = {dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1],
                dq[3] * q[1] - dq[0] * q[2] + dq[1] * q[3] + dq[2] * q[0],
                dq[3] * q[2] + dq[0] * q[1] - dq[1] * q[0] + dq[2] * q[3],
                dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2]};
  real nn = 1 / sqrt(nq[0] * nq[0] + nq[1] * nq[1] + nq[2] * nq[2] + nq[3] * nq[3]);
  for (int k = 0; k < 4; ++k) q[k] = nq[k] * nn;
}

/* =====================================================================================
 *                                   Philox4x32-10
 * ===================================================================================== */
static void philox4x32(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
static float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

ORC_API void orc_philox(uint32_t* ctr4, uint32_t k0, uint32_t k1) { philox4x32(ctr4, k0, k1); }

/* sensor noise (Rex._AddSensorNoise, model/rex.py:765-769): four standard normals per Philox block, keyed by
 * (seed; episode, global env, 16 + block, step); block numbering as in csrc/rexsim.hip */
enum { NZ_GOAL = 0, NZ_REWARD_RPY = 1, NZ_FALLEN_RPY = 2, NZ_OBS_RPY = 3, NZ_OBS_RATE = 4, NZ_TORQUE = 8, NZ_VELOCITY = 16, NZ_ANGLE = 24 };
static void gauss4(const RexConfig* c, int gidx, int episode, int step, int block, real* z) {
  uint32_t ctr[4] = {(uint32_t)episode, (uint32_t)gidx, 16u + (uint32_t)block, (uint32_t)step};
  philox4x32(ctr, (uint32_t)c->seed, (uint32_t)(c->seed >> 32));
  for (int p = 0; p < 2; ++p) {
    float u1 = (float)((ctr[2 * p] >> 8) + 1u) * (1.0f / 16777216.0f), u2 = u01(ctr[2 * p + 1]);
    real r = sqrt(-2 * log((real)u1)), a = (real)6.28318530717958648 * (real)u2;
    z[2 * p] = r * cos(a); z[2 * p + 1] = r * sin(a);
  }
}
static int noise_on(const RexConfig* c) { for (int k = 0; k < 5; ++k) if (c->noise_stdev[k] > 0) return 1; return 0; }

/* =====================================================================================
 *                                   environments
 * ===================================================================================== */
static const real POSE_STAND[12] = {0., -0.88643435, 1.30197369, 0., -0.88643435, 1.30197369,
                                    0., -0.88643435, 1.30197369, 0., -0.88643435, 1.30197369};
static const real POSE_STAND_OL[12] = {0.15192765, -0.90412283, 1.48156545, -0.15192765, -0.90412283, 1.48156545,
                                       0.15192765, -0.90412283, 1.48156545, -0.15192765, -0.90412283, 1.48156545};

typedef struct {
  Phys ph;
  /* GaitPlanner state: phase (informational: the decision `_phi >= 0.99` is flag REX_F_PHASE_WRAP), the env step at
   * which _last_time was latched, the arc angle; end_step = the env step at which the goal was reached */
  real phi, alpha;
  int32_t last_step, end_step;
  real target, aux;
  uint32_t flags;
  int32_t steps, episode;
  uint32_t motor_enabled;
  uint16_t overheat[NJ];
  real tau_obs[NJ]; /* Rex._observed_motor_torques (transient; not part of the persistent state) */
  /* Rex._observation_history (deque(maxlen=100) of 43-vectors, rex.py:122): ring, hist_head = newest slot */
  int hist_head, hist_len;
  real (*hist)[HIST_WORDS];   /* [REX_HISTORY_LEN][43], allocated only when a latency is configured */
  real ctrl_obs[HIST_WORDS];  /* Rex._control_observation */
  int nz_gidx, nz_episode, nz_step; /* keys of the current step's sensor-noise draws (transient) */
} Env;

typedef struct {
  RexConfig cfg;
  Env* envs;
  Env snapshot[5]; /* settled reset state (rex.py:296-324), one per task of a REX_TASK_MIXED batch */
  int n_mix, mix_task[5];
  /* terrain pool (terrain_type='random'): K heightfields, one settled snapshot per terrain */
  int n_terrain;
  float* heights; /* [K][nx*ny] */
  HfGeom geo;
  float* mids;    /* [K] */
  Env* terrain_snapshot;
  float* body_params; /* [3][N] or NULL */
  uint32_t* trace;    /* orc_set_event_trace: caller-owned [3][N] or NULL */
} Orc;

/* ---- REX_TASK_MIXED: env g runs one task of the mix for its whole life, drawn from its own Philox stream ---- */
static int task_action_repeat(int task) { return (task == REX_TASK_GALLOP || task == REX_TASK_POSES) ? 6 : 5; }
static int mixed_task_of(const Orc* o, int idx) {
  uint32_t ctr[4] = {0xFFFFFFFFu, (uint32_t)(o->cfg.env_index_base + idx), 2u, 0u};
  philox4x32(ctr, (uint32_t)o->cfg.seed, (uint32_t)(o->cfg.seed >> 32));
  return o->mix_task[ctr[0] % (uint32_t)o->n_mix];
}
static RexConfig task_cfg(const Orc* o, int task) {   /* the per-task constants of the reference env classes */
  RexConfig c = o->cfg;
  if (o->cfg.task != REX_TASK_MIXED) return c;
  c.task = task;
  c.action_repeat = task_action_repeat(task);
  c.solver_iterations = 300 / c.action_repeat;
  c.energy_weight = task == REX_TASK_GALLOP ? 0.005f : 0.0005f;
  return c;
}
static RexConfig env_cfg(const Orc* o, int idx) { return o->cfg.task == REX_TASK_MIXED ? task_cfg(o, mixed_task_of(o, idx)) : o->cfg; }
static int mix_slot(const Orc* o, int task) { for (int k = 0; k < o->n_mix; ++k) if (o->mix_task[k] == task) return k; return 0; }

/* terrain of (global env index, episode): the reference regenerates the field on every reset
 * (rex_gym_env.py:347-348); here each episode picks one of the K pool entries */
static int terrain_index(const Orc* o, int idx, int episode) {
  if (!o->n_terrain) return -1;
  return (int)(((uint32_t)(o->cfg.env_index_base + idx) + 977u * (uint32_t)episode) % (uint32_t)o->n_terrain);
}
static Ground env_ground(const Orc* o, int idx, int episode) {
  Ground g = {0, 0, 1, 1, FRICTION_MU, 0, o->cfg.body_contacts, o->geo, o->cfg.on_rack};
  if (o->body_params) {
    int n = o->cfg.num_envs;
    g.has_params = 1; g.base_mass_scale = (real)o->body_params[idx]; g.leg_mass_scale = (real)o->body_params[n + idx];
    g.mu = (real)o->body_params[2 * n + idx];
  }
  if (o->cfg.mass_scale_hi > 0 || o->cfg.friction_hi > 0) {   /* per-reset draws (env_randomizer hook, rex_gym_env.py:345-346) */
    uint32_t ctr[4] = {(uint32_t)episode, (uint32_t)(o->cfg.env_index_base + idx), 1u, 0u};
    philox4x32(ctr, (uint32_t)o->cfg.seed, (uint32_t)(o->cfg.seed >> 32));
    g.has_params = 1;
    if (o->cfg.mass_scale_hi > 0) {
      g.base_mass_scale = (real)fmaf(o->cfg.mass_scale_hi - o->cfg.mass_scale_lo, u01(ctr[0]), o->cfg.mass_scale_lo);
      g.leg_mass_scale = (real)fmaf(o->cfg.mass_scale_hi - o->cfg.mass_scale_lo, u01(ctr[1]), o->cfg.mass_scale_lo);
    }
    if (o->cfg.friction_hi > 0) g.mu = (real)fmaf(o->cfg.friction_hi - o->cfg.friction_lo, u01(ctr[2]), o->cfg.friction_lo);
  }
  int t = terrain_index(o, idx, episode);
  if (t >= 0) { g.h = o->heights + (size_t)t * o->geo.nx * o->geo.ny; g.mid = (real)o->mids[t]; }
  return g;
}

ORC_API int orc_obs_dim(const RexConfig* c) {
  if (c->gallop_no_angles) return 4;                       /* use_angle_in_observation=False, gallop_env.py:344-356 */
  if (c->task == REX_TASK_MIXED) return ((c->task_mix >> REX_TASK_GALLOP) & 1) ? 4 + NJ : 4;
  return c->task == REX_TASK_GALLOP ? 4 + NJ : 4;
}
ORC_API int orc_num_motors(void) { return NJ; }
ORC_API int orc_state_words(void) { return SW_WORDS; }
ORC_API int orc_action_dim(const RexConfig* c) {
  if (c->task == REX_TASK_MIXED) {
    int best = 0;
    for (int t = 0; t < 5; ++t) if ((c->task_mix >> t) & 1) { RexConfig one = *c; one.task = t; int d = orc_action_dim(&one); if (d > best) best = d; }
    return best;
  }
  if (c->task == REX_TASK_WALK) return c->signal == REX_SIGNAL_IK ? 2 : 8;
  if (c->task == REX_TASK_GALLOP) return c->signal == REX_SIGNAL_IK ? 2 : 4;
  if (c->task == REX_TASK_POSES || c->task == REX_TASK_STANDUP) return 1;         /* standup_env.py:99-101 */
  return 2;
}

/* INIT_POSES['rest_position'] (rex_constants.py:41-46); the foot target 6 rad lies beyond the URDF bound 2.59 */
static const real POSE_REST[12] = {-0.4, -1.5, 6, 0.4, -1.5, 6, -0.4, -1.5, 6, 0.4, -1.5, 6};
static const real* init_pose(const RexConfig* c) {
  if (c->task == REX_TASK_STANDUP) return POSE_REST;                   /* standup_env.py:108-110 */
  return c->signal == REX_SIGNAL_OL ? POSE_STAND_OL : POSE_STAND;
}
/* Rex.GetTimeSinceReset (rex.py:155-156): step counter x time step, the env clock every ramp / brake / phase threshold
 * is compared against.  RexConfig carries the time step as a float32 (0.001f = 0.001 (1 + 4.7e-8)); the reference
 * multiplies by the Python float 0.001, and thresholds such as `t <= 0.8` sit exactly on multiples of the control
 * step.  The fp64 build therefore runs its clock on the time step rounded to 1 ns, which reproduces the reference's
 * comparisons exactly (tests/test_oracle_env_commands.py); the fp32 build multiplies in float like the HIP kernels,
 * where the product rounds back onto the same float as the threshold. */
/* RexConfig carries its real-valued settings as float32; the reference holds them as the Python floats the caller
 * wrote (0.001, 0.02, 0.3 ...).  The fp64 build reads each setting back as the shortest decimal (7 significant digits)
 * that float came from, so that its arithmetic is the reference's to the last bits. */
static double as_written(float x) {
  static __thread struct { uint32_t key; int used; double val; } memo[64];
  uint32_t bits; memcpy(&bits, &x, 4);
  unsigned slot = (bits * 2654435761u) >> 26;
  if (memo[slot].used && memo[slot].key == bits) return memo[slot].val;
  char buf[40];
  snprintf(buf, sizeof buf, "%.7g", (double)x);
  double v = strtod(buf, 0);
  memo[slot].key = bits; memo[slot].used = 1; memo[slot].val = v;
  return v;
}
/* the env clock `step_counter * time_step` (rex.py:155-156) at env step k: a correctly rounded double product on the
 * decimal the caller wrote, in both builds */
static clk step_time(const RexConfig* c, int32_t k) { return (double)(k * c->action_repeat) * as_written(c->sim_time_step); }
static clk env_time(const RexConfig* c, const Env* e) { return step_time(c, e->steps); }
/* the simulation time step and the latencies as the physics / history code sees them: the reference holds them as
 * Python floats (0.001, 0.02 ...), RexConfig as float32; the fp64 build undoes the float32 rounding so that its
 * trajectories are the reference's to the last bits (tests/test_oracle_rollouts.py), the fp32 build computes in float */
static real sim_dt(const RexConfig* c) { return sizeof(real) == sizeof(double) ? (real)as_written(c->sim_time_step) : (real)c->sim_time_step; }
static real cfgf(float x) { return sizeof(real) == sizeof(double) ? (real)as_written(x) : (real)x; }
/* substep counts are integers of the Python-float quotient (int(0.5 / 0.001) = 500, rex.py:319; int(latency /
 * time_step), rex.py:747): always taken in double on the decimal values, a float quotient lands just below (499) */
static int whole_steps(float duration, float dt) { return (int)(as_written(duration) / as_written(dt)); }

/* RexGymEnv._transform_action_to_motor_command (rex_gym_env.py:363-367): the 12 leg targets, then ARM_POSES['rest'] */
static void full_command(const real leg12[12], real cmd[NJ]) {
  memcpy(cmd, leg12, sizeof(real) * 12);
#ifdef REX_ARM
  for (int k = 0; k < REXA_NJ; ++k) cmd[12 + k] = (real)REXA_REST[k];
#endif
}

/* Rex.GetTrueObservation (rex.py:717-724) */
static void true_observation(const Env* e, real o[HIST_WORDS]) {
  for (int j = 0; j < NJ; ++j) { o[j] = e->ph.q[j]; o[NJ + j] = e->ph.qd[j]; o[2 * NJ + j] = e->tau_obs[j]; }
  for (int k = 0; k < 4; ++k) o[3 * NJ + k] = e->ph.quat[k];
  for (int k = 0; k < 3; ++k) o[3 * NJ + 4 + k] = e->ph.angvel[k];
}
/* Rex._GetDelayedObservation (rex.py:735-753) */
static void delayed_observation(const RexConfig* c, const Env* e, real latency, real o[HIST_WORDS]) {
  if (!e->hist) { true_observation(e, o); return; }
#define SLOT(k) e->hist[(e->hist_head - (k) + 2 * REX_HISTORY_LEN) % REX_HISTORY_LEN]
  if (latency <= 0 || e->hist_len == 1) { memcpy(o, SLOT(0), sizeof(real) * HIST_WORDS); return; }
  real dt = sim_dt(c);
  int n = (int)((double)latency / (double)dt);
  if (n + 1 >= e->hist_len) { memcpy(o, SLOT(e->hist_len - 1), sizeof(real) * HIST_WORDS); return; }
  real alpha = (latency - n * dt) / dt;
  for (int k = 0; k < HIST_WORDS; ++k) o[k] = (1 - alpha) * SLOT(n)[k] + alpha * SLOT(n + 1)[k];
#undef SLOT
}
/* Rex.ReceiveObservation (rex.py:726-733) */
static void receive_observation(const RexConfig* c, Env* e) {
  if (!e->hist) { true_observation(e, e->ctrl_obs); return; }   /* no latency anywhere: the deque's newest entry is all that is read */
  e->hist_head = (e->hist_head + 1) % REX_HISTORY_LEN;
  if (e->hist_len < REX_HISTORY_LEN) e->hist_len++;
  true_observation(e, e->hist[e->hist_head]);
  delayed_observation(c, e, cfgf(c->control_latency), e->ctrl_obs);
}

/* Rex.ApplyAction + stepSimulation + ReceiveObservation (rex.py:158-163,568-641) */
static void rex_substep(const RexConfig* c, Env* e, const real cmd[NJ], const Ground* ground) {
  real tau[NJ];
  real pd[HIST_WORDS];
  delayed_observation(c, e, cfgf(c->pd_latency), pd);                  /* _GetPDObservation, rex.py:755-759 */
  for (int j = 0; j < NJ; ++j) {
    real act, obs;
    motor_torque(cmd[j], pd[j], pd[NJ + j], e->ph.qd[j], cfgf(c->motor_kp), cfgf(c->motor_kd), &act, &obs);
    if (fabs(act) > OVERHEAT_TORQUE) { if (e->overheat[j] < 65535) e->overheat[j]++; } else e->overheat[j] = 0;
    if ((real)e->overheat[j] > OVERHEAT_TIME / sim_dt(c)) e->motor_enabled &= ~(1u << j);
    e->tau_obs[j] = obs;
    tau[j] = ((e->motor_enabled >> j) & 1u) ? act : 0;
    if (DBG_JOINT_FRICTION > 0 && (j % 3) != 1) tau[j] -= clampr(DBG_JOINT_VISC * e->ph.qd[j], -DBG_JOINT_FRICTION, DBG_JOINT_FRICTION);
  }
  physics_substep(&e->ph, tau, sim_dt(c), c->solver_iterations, (real)c->solver_residual_threshold, ground);
  receive_observation(c, e);
}

/* reset height: INIT_RACK_POSITION (rex.py:11) on the rack, else ROBOT_INIT_POSITION[terrain] (terrain.py:14-20) */
static real init_z(const RexConfig* c) { return c->on_rack ? (real)1 : c->init_height > 0 ? cfgf(c->init_height) : ROBOT_INIT_Z; }
static void settle(Orc* o, Env* e, const Ground* ground, const RexConfig* cfgp) {
  (void)o;
  real (*hist)[HIST_WORDS] = e->hist;     /* the snapshot keeps its own ring when a latency is configured */
  memset(e, 0, sizeof(*e));
  if (!hist && (cfgp->pd_latency > 0 || cfgp->control_latency > 0)) hist = calloc(REX_HISTORY_LEN, sizeof(real[HIST_WORDS]));
  e->hist = hist;
  e->ph.pos[2] = init_z(cfgp);
  e->ph.quat[3] = 1;
  full_command(POSE_STAND, e->ph.q); /* ResetPose: INIT_POSES[pose_id='stand'] (+ _ResetArmMotors: ARM_POSES['rest'], rex.py:395-400) */
  e->motor_enabled = (1u << NJ) - 1;
  /* RexPosesEnv.reset() calls the base reset with initial_motor_angles=None: the reset motion is
   * skipped (rex.py:308), the robot starts at the drop height in the 'stand' pose */
  /* Rex.Reset: `_observation_history.clear()` (rex.py:309), then -- only when a reset motion follows -- one
   * ReceiveObservation of the dropped robot (rex.py:314) */
  e->hist_head = REX_HISTORY_LEN - 1; e->hist_len = 0;
  if (cfgp->task == REX_TASK_POSES) { receive_observation(cfgp, e); return; }   /* rex.py:323 alone */
  receive_observation(cfgp, e);
  real cmd[NJ];
  full_command(POSE_STAND, cmd);
  for (int k = 0; k < 100; ++k) rex_substep(cfgp, e, cmd, ground);      /* rex.py:315-318 */
  int nreset = whole_steps(0.5f, cfgp->sim_time_step);             /* rex.py:319 */
  full_command(init_pose(cfgp), cmd);
  for (int k = 0; k < nreset; ++k) rex_substep(cfgp, e, cmd, ground);
  receive_observation(cfgp, e);                                          /* rex.py:323: the last state once more */
}

/* the observation the controller-facing getters read (Rex._control_observation): delayed when a latency is set */
static void control_observation(const Env* e, real o[HIST_WORDS]) { memcpy(o, e->ctrl_obs, sizeof(real) * HIST_WORDS); }

static void env_observation(const RexConfig* c, const Env* e, real* obs) {
  real co[HIST_WORDS], rpy[3], z[4];
  control_observation(e, co);
  quat_to_euler(co + 3 * NJ, rpy);
  real wx = co[3 * NJ + 4], wy = co[3 * NJ + 5];
  if (c->noise_stdev[3] > 0) { gauss4(c, e->nz_gidx, e->nz_episode, e->nz_step, NZ_OBS_RPY, z); rpy[0] += cfgf(c->noise_stdev[3]) * z[0]; rpy[1] += cfgf(c->noise_stdev[3]) * z[1]; }
  if (c->noise_stdev[4] > 0) { gauss4(c, e->nz_gidx, e->nz_episode, e->nz_step, NZ_OBS_RATE, z); wx += cfgf(c->noise_stdev[4]) * z[0]; wy += cfgf(c->noise_stdev[4]) * z[1]; }
  obs[0] = rpy[0]; obs[1] = rpy[1]; obs[2] = wx; obs[3] = wy;
  if (c->task == REX_TASK_GALLOP && !c->gallop_no_angles) {   /* `if self._use_angle_in_observation:` gallop_env.py:353 */
    real nz[20] = {0};
    if (c->noise_stdev[0] > 0) for (int b = 0; b < (NJ + 3) / 4; ++b) gauss4(c, e->nz_gidx, e->nz_episode, e->nz_step, NZ_ANGLE + b, nz + 4 * b);
    for (int j = 0; j < NJ; ++j) { /* GetMotorAngles: noise, then MapToMinusPiToPi (rex.py:26-41,457-468) */
      real a = fmod(co[j] + cfgf(c->noise_stdev[0]) * nz[j], (real)(2 * M_PI));
      if (a >= (real)M_PI) a -= (real)(2 * M_PI); else if (a < -(real)M_PI) a += (real)(2 * M_PI);
      obs[4 + j] = a;
    }
  }
}

static void env_reset(Orc* o, int idx) {
  const RexConfig cfg_env = env_cfg(o, idx);
  const RexConfig* c = &cfg_env;
  const int slot = mix_slot(o, c->task);
  Env* e = &o->envs[idx];
  int32_t episode = e->episode;
  real (*hist)[HIST_WORDS] = e->hist;
  /* the env keeps ONE GaitPlanner for its lifetime (walk_env.py:99, turn_env.py:117): its turning-arc angle `_alpha`
   * (gait_planner.py:76-85) survives reset() and enters the first step of the next episode */
  real alpha = e->alpha;
  {
    int t = terrain_index(o, idx, episode + 1);
    *e = t >= 0 ? o->terrain_snapshot[t * o->n_mix + slot] : o->snapshot[slot];   /* settled on this episode's terrain (under this env's task) */
  }
  e->episode = episode + 1;
  e->hist = hist;
  e->phi = 0; e->last_step = 0; e->alpha = alpha;
  uint32_t ctr[4] = {(uint32_t)e->episode, (uint32_t)(c->env_index_base + idx), 0, 0};   /* key = seed, counter = (episode, global env) */
  philox4x32(ctr, (uint32_t)c->seed, (uint32_t)(c->seed >> 32));
  e->flags = 0;
  if (c->task == REX_TASK_WALK) {
    int backwards = c->backwards < 0 ? (int)(ctr[0] >> 31) : c->backwards;     /* walk_env.py:133-136 */
    if (backwards) e->flags |= REX_F_BACKWARDS;
    if (c->target_position != 0.0f) e->target = cfgf(c->target_position);
    else {                                                                      /* walk_env.py:143-147 */
      float u = u01(ctr[1]);
      e->target = backwards ? (real)(-2.0f - u) : (real)(1.0f + 2.0f * u);
    }
  } else if (c->task == REX_TASK_GALLOP) {
    if (c->target_position != 0.0f) e->target = cfgf(c->target_position);
    else e->target = (real)(1.0f + 2.0f * u01(ctr[1]));                         /* gallop_env.py:150-152 */
  }
  e->end_step = 0; e->aux = 0; e->steps = 0;
  if (c->task == REX_TASK_POSES) {                                              /* poses_env.py:153-192 */
    /* _ranges (rex_gym_env.py:258-265): base_y, base_z, roll, pitch, yaw */
    static const float LO[5] = {-0.007f, -0.048f, -0.78539816339744830962f, -0.78539816339744830962f, -0.78539816339744830962f};
    static const float HI[5] = {0.007f, 0.021f, 0.78539816339744830962f, 0.78539816339744830962f, 0.78539816339744830962f};
    if (c->pose_index >= 0) { e->aux = (real)c->pose_index; e->target = cfgf(c->pose_value); }
    else {
      int k = e->episode % 5;               /* deque rotation: one pop per reset() */
      e->aux = (real)k;
      e->target = (real)fmaf(HI[k] - LO[k], u01(ctr[1]), LO[k]);
    }
  }
  if (c->task == REX_TASK_TURN) {                                               /* turn_env.py:129-160 */
    real tgt = (c->orient_fixed & 1) ? cfgf(c->target_orient) : (real)fmaf(5.8f, u01(ctr[1]), 0.2f);
    real ini = c->on_rack ? cfgf(2.1f) : (c->orient_fixed & 2) ? cfgf(c->init_orient) : (real)fmaf(5.8f, u01(ctr[2]), 0.2f);   /* turn_env.py:140-143 */
    e->target = tgt; e->aux = ini;
    real rpy[3] = {0, 0, ini};
    euler_to_quat(rpy, e->ph.quat);                                             /* resetBasePositionAndOrientation */
    e->ph.pos[0] = 0; e->ph.pos[1] = 0; e->ph.pos[2] = init_z(c);
  }
  /* the reference's deque is not touched by reset() after Rex.Reset: it holds the last 100 observations of the reset
   * motion (the newest twice, rex.py:323), and the turn env's teleport (turn_env.py:158-159) happens behind its back --
   * the observation reset() returns and the first step's yaw reading are those of the settled, not yet turned robot */
  if (e->hist) {
    const Env* snap = o->n_terrain ? &o->terrain_snapshot[terrain_index(o, idx, e->episode) * o->n_mix + slot] : &o->snapshot[slot];
    memcpy(e->hist, snap->hist, sizeof(real[HIST_WORDS]) * REX_HISTORY_LEN);
  }
}

/* the time GaitPlanner.loop reads: wall-clock seconds (gait_planner.py:108-110) = simulated time x the host's
 * wall-seconds-per-simulated-second (RexConfig.gait_clock_scale; DBG_GAIT_CLOCK is the sensitivity tool's multiplier) */
static clk gait_now(const RexConfig* c, clk t) { return t * (c->gait_clock_scale > 0 ? as_written(c->gait_clock_scale) : 1.0) * (clk)DBG_GAIT_CLOCK; }
/* GaitPlanner.loop on the env's planner state.  _last_time is `now` of the env step that latched it and `_phi >= 0.99`
 * is a flag: both are exact, whatever `real` is. */
static void env_gait_loop(const RexConfig* c, Env* e, int mode, real v, real angle, real w_rot, clk T, real direction, real frame[12]) {
  Gait g = {(e->flags & REX_F_PHASE_WRAP) ? 1.0 : 0.0, gait_now(c, step_time(c, e->last_step)), e->alpha};
  if (e->flags & REX_F_PHASE_WRAP) e->last_step = e->steps;
  gait_loop(&g, mode, v, angle, w_rot, T, direction, gait_now(c, env_time(c, e)), frame);
  e->phi = (real)g.phi; e->alpha = g.alpha;
  if (g.phi >= 0.99) e->flags |= REX_F_PHASE_WRAP; else e->flags &= ~REX_F_PHASE_WRAP;
}

/* walk_env.py:229-244 */
static real walk_gait_coeff(clk t, real a0) { clk p = 0.8 + (clk)a0; return (0 <= t && t <= p) ? (real)t : (real)1.0; }
static real walk_brake_coeff(clk t, real a1, clk end_t) {
  clk p = 0.8 + (clk)a1;
  return (end_t <= t && t <= p + end_t) ? (real)(1 - (t - end_t)) : (real)0.0;
}

static void order_signal(const real ang[12], real cmd[12]) { /* FR,FL,RR,RL -> FL,FR,RL,RR (walk_env.py:284-289) */
  for (int k = 0; k < 3; ++k) { cmd[k] = ang[3 + k]; cmd[3 + k] = ang[k]; cmd[6 + k] = ang[9 + k]; cmd[9 + k] = ang[6 + k]; }
}

static void walk_command(const RexConfig* c, Env* e, const real* action, real cmd[12]) {
  const real* ip = init_pose(c);
  if (e->flags & REX_F_STAY_STILL) { memcpy(cmd, ip, sizeof(real) * 12); return; }       /* walk_env.py:318-319 */
  const clk t = env_time(c, e);          /* rex.py:155-156 */
  if (e->target != 0) {                                                                 /* walk_env.py:207-215 */
    if (fabs(e->ph.pos[0]) >= fabs(e->target) - (real)0.15) {
      e->flags |= REX_F_GOAL_REACHED;
      if (!(e->flags & REX_F_TERMINATING)) { e->end_step = e->steps; e->flags |= REX_F_TERMINATING; }
    }
  }
  const clk end_t = step_time(c, e->end_step);
  int backwards = (e->flags & REX_F_BACKWARDS) != 0;
  if (c->signal == REX_SIGNAL_IK) {                                                     /* walk_env.py:252-290 */
    real gait_coeff = walk_gait_coeff(t, action[0]);
    real step = backwards ? (real)-0.3 : (real)0.6;
    clk period = backwards ? 0.5 : 0.65;
    real base_x = backwards ? (real)0.0 : (real)0.01;
    real pos[3] = {base_x, 0, 0}, orn[3] = {0, 0, 0};
    real step_length = step * gait_coeff;
    if (e->flags & REX_F_GOAL_REACHED) {
      real b = walk_brake_coeff(t, action[1], end_t);
      step_length *= b;
      if (b == 0) e->flags |= REX_F_STAY_STILL;
    }
    real direction = step_length < 0 ? (real)-1.0 : (real)1.0;
    real frames[12], ang[12];
    env_gait_loop(c, e, 0, step_length, 0, 0, period, direction, frames);
    ik_solve(orn, pos, frames, ang, 0);
    order_signal(ang, cmd);
  } else {                                                                              /* walk_env.py:292-315 */
    real l_a = (real)0.1, f_a = (real)0.2;
    const real period = (real)(1.0 / 8);
    if (e->flags & REX_F_GOAL_REACHED) {
      real b = walk_brake_coeff(t, 0, end_t);
      l_a *= b; f_a *= b;
      /* `if coeff is 0.0: self._stay_still = True` (walk_env.py:300) is an IDENTITY test: it holds exactly when
       * _evaluate_brakes_stage_coeff hands back its `end_value=0.0` argument (the same constant object), i.e. when t
       * lies outside the brake window -- never for the computed 1 - (t - end_t), even where that is 0
       * (tests/golden/make_env_golden.py runs the reference's code: `stay` turns on one step past the window) */
      if (!(end_t <= t && t <= 0.8 + end_t)) e->flags |= REX_F_STAY_STILL;
    }
    real sc = walk_gait_coeff(t, 0);
    l_a *= sc; f_a *= sc;
    real l_ext = l_a * cos(2 * (real)M_PI / period * (real)t), f_ext = f_a * cos(2 * (real)M_PI / period * (real)t);
    real pose[12] = {0, l_ext + action[0], f_ext + action[1], 0, -l_ext + action[2], -f_ext + action[3],
                     0, -l_ext + action[4], -f_ext + action[5], 0, l_ext + action[6], f_ext + action[7]};
    for (int j = 0; j < 12; ++j) cmd[j] = ip[j] + pose[j];
  }
}

/* gallop_env.py:234-249 */
static real gallop_brake_coeff(clk t, real a0, clk end_t) {
  clk p = 1.0 + (clk)a0;
  return (end_t <= t && t <= p + end_t) ? (real)(1 - (t - end_t)) : (real)0.0;
}
static real gallop_gait_coeff(clk t, real a1) { clk p = 1.0 + (clk)a1; return (0 <= t && t <= p) ? (real)t : (real)1.0; }

static void gallop_command(const RexConfig* c, Env* e, const real* action, real cmd[12]) {
  if (e->flags & REX_F_STAY_STILL) { memcpy(cmd, POSE_STAND, sizeof(real) * 12); return; } /* rex.initial_pose */
  const clk t = env_time(c, e);
  if (e->target != 0) {                                                                 /* gallop_env.py:212-220 */
    if (fabs(e->ph.pos[0]) >= fabs(e->target)) {
      e->flags |= REX_F_GOAL_REACHED;
      if (!(e->flags & REX_F_TERMINATING)) { e->end_step = e->steps; e->flags |= REX_F_TERMINATING; }
    }
  }
  const clk end_t = step_time(c, e->end_step);
  if (c->signal == REX_SIGNAL_IK) {                                                     /* gallop_env.py:257-285 */
    real gait_coeff = gallop_gait_coeff(t, action[1]);
    real pos[3] = {(real)0.01, 0, (real)-0.007}, orn[3] = {0, 0, 0};
    real step_length = (real)1.3 * gait_coeff;
    if (e->flags & REX_F_GOAL_REACHED) step_length *= gallop_brake_coeff(t, action[0], end_t);
    real frames[12], ang[12];
    env_gait_loop(c, e, 1, step_length, 0, 0, 0.3, (real)1.0, frames);
    ik_solve(orn, pos, frames, ang, 0);
    order_signal(ang, cmd);
  } else {                                                                              /* gallop_env.py:287-304 */
    real lp[4] = {action[0], action[1], action[2], action[3]};
    if (e->flags & REX_F_GOAL_REACHED) {
      real b = gallop_brake_coeff(t, 0, end_t);
      for (int k = 0; k < 4; ++k) lp[k] *= b;
      /* gallop_env.py:291: the same identity test as in walk_env.py:300 -- true outside the brake window */
      if (!(end_t <= t && t <= 1.0 + end_t)) e->flags |= REX_F_STAY_STILL;
    }
    const real* ip = init_pose(c);
    for (int l = 0; l < 4; ++l) {
      cmd[3 * l] = ip[3 * l];
      cmd[3 * l + 1] = ip[3 * l + 1] + (l < 2 ? lp[0] : lp[2]);
      cmd[3 * l + 2] = ip[3 * l + 2] + (l < 2 ? lp[1] : lp[3]);
    }
  }
}

/* turn_env.py:313-322 */
static int turn_clockwise(const Env* e) {
  real diff = fabs(e->aux - e->target);
  if (e->aux < e->target) return diff > (real)3.14;
  return diff < (real)3.14;
}

/* RexTurnEnv._transform_action_to_motor_command (turn_env.py:239-347) */
static void turn_command(const RexConfig* c, Env* e, const real* action, real cmd[12]) {
  const real* ip = init_pose(c);
  const clk t = env_time(c, e);
  if (e->flags & REX_F_STAY_STILL) {
    if (t - step_time(c, e->end_step) >= 1) e->flags |= REX_F_ENV_GOAL;                          /* _terminate_with_delay */
    memcpy(cmd, ip, sizeof(real) * 12);
    return;
  }
  {                                                                                 /* _check_target_position */
    real rpy[3], co[HIST_WORDS];
    control_observation(e, co);
    quat_to_euler(co + 3 * NJ, rpy);
    if (c->noise_stdev[3] > 0) {   /* GetBaseOrientation = quaternion of the NOISY roll / pitch / yaw (rex.py:430-442,530-537) */
      real z[4], qq[4];
      gauss4(c, e->nz_gidx, e->nz_episode, e->nz_step, NZ_GOAL, z);
      for (int k = 0; k < 3; ++k) rpy[k] += cfgf(c->noise_stdev[3]) * z[k];
      euler_to_quat(rpy, qq);
      quat_to_euler(qq, rpy);
    }
    real cz = rpy[2];
    if (cz < 0) cz += (real)6.28;
    if (fabs(e->target - cz) <= (real)0.01) {
      e->flags |= REX_F_GOAL_REACHED;
      if (!(e->flags & REX_F_TERMINATING)) { e->end_step = e->steps; e->flags |= REX_F_TERMINATING; }
    }
  }
  int clockwise = turn_clockwise(e);
  if (e->flags & REX_F_GOAL_REACHED) e->flags |= REX_F_STAY_STILL;                  /* turn_env.py:259-260,272-273 */
  if (c->signal == REX_SIGNAL_IK) {
    real coeff = (0 <= t && t <= 0.8) ? (real)t : (real)1.0;
    real dirv = (real)-0.5 * coeff;
    if (clockwise) dirv = -dirv;
    real pos[3] = {(real)0.009, 0, 0}, orn[3] = {0, 0, 0};
    real frames[12], ang[12];
    env_gait_loop(c, e, 0, (real)0.02, 0, dirv + action[0], 0.75 + (clk)action[1], (real)1.0, frames);
    ik_solve(orn, pos, frames, ang, 0);
    order_signal(ang, cmd);
  } else {
    const real ext = (real)0.1, swing = (real)0.03 + action[0], swipe = (real)0.05 + action[1];
    int ith = ((int)(t / 0.1)) % 2;
    real m = clockwise ? (real)1 : (real)-1;   /* right_* poses (clockwise) = left_* with the swing sign flipped, turn_env.py:280-297 */
    real ms = m * swing;
    real first[12] = {swipe, ext, ms, -swipe, ext, -ms, swipe, -ext, -ms, -swipe, -ext, ms};
    real second[12] = {-swipe, 0, -ms, swipe, 0, ms, -swipe, 0, ms, swipe, 0, -ms};
    for (int j = 0; j < 12; ++j) cmd[j] = POSE_STAND_OL[j] + (ith ? second[j] : first[j]);
  }
}

/* RexPosesEnv._signal (poses_env.py:186-225): one body-pose component ramps to its target, IK on the
 * default foot frames */
static void poses_command(const RexConfig* c, Env* e, const real* action, real cmd[12]) {
  const clk t = env_time(c, e), p = 0.8 + (clk)action[0];
  real coeff = (0 <= t && t <= p) ? (real)t : (real)1.0;
  real staged = e->target * coeff;
  real pos[3] = {(real)0.01, 0, 0}, orn[3] = {0, 0, 0};
  int k = (int)e->aux;
  if (k == 0) pos[1] = staged; else if (k == 1) pos[2] = staged; else orn[k - 2] = staged;
  const real frames[12] = {IK_L / 2, -IK_YDIST / 2, -IK_HEIGHT, IK_L / 2, IK_YDIST / 2, -IK_HEIGHT,
                           -IK_L / 2, -IK_YDIST / 2, -IK_HEIGHT, -IK_L / 2, IK_YDIST / 2, -IK_HEIGHT};
  real ang[12];
  ik_solve(orn, pos, frames, ang, 0);
  order_signal(ang, cmd);
}

/* rex_gym_env.py:501-542 */
static real base_reward(const RexConfig* c, Env* e) {
  real x = -e->ph.pos[0];
  if (c->backwards > 0) x = -x; /* `if self._backwards:` tests the constructor argument, not the episode draw */
  real fwd;
  e->target = fabs(e->target); /* rex_gym_env.py:510; _target_position is never None after reset() */
  real T = e->target;
  if (x > T + (real)0.15) fwd = T - x;
  else if (T <= x && x <= T + (real)0.15) fwd = 1;
  else if (x <= (real)0.05) fwd = 0;
  else fwd = x / T;
  if (fwd > (real)c->forward_reward_cap) fwd = (real)c->forward_reward_cap;   /* min(forward_reward, cap), rex_gym_env.py:525 (+inf: none) */
  real drift = -fabs(e->ph.pos[1]);
  real co[HIST_WORDS], rpy[3], qq[4], R[3][3];
  control_observation(e, co);
  quat_to_euler(co + 3 * NJ, rpy);  /* GetBaseOrientation: delayed quat -> RPY -> quat (rex.py:530-537) */
  if (noise_on(c)) {
    real z[4], nt[20], nv[20];
    if (c->noise_stdev[3] > 0) { gauss4(c, e->nz_gidx, e->nz_episode, e->nz_step, NZ_REWARD_RPY, z); for (int k = 0; k < 3; ++k) rpy[k] += cfgf(c->noise_stdev[3]) * z[k]; }
    for (int b = 0; b < (NJ + 3) / 4; ++b) {
      gauss4(c, e->nz_gidx, e->nz_episode, e->nz_step, NZ_TORQUE + b, nt + 4 * b);
      gauss4(c, e->nz_gidx, e->nz_episode, e->nz_step, NZ_VELOCITY + b, nv + 4 * b);
    }
    for (int j = 0; j < NJ; ++j) { co[2 * NJ + j] += cfgf(c->noise_stdev[2]) * nt[j]; co[NJ + j] += cfgf(c->noise_stdev[1]) * nv[j]; }
  }
  euler_to_quat(rpy, qq);
  quat_to_mat(qq, R);
  real shake = -fabs(R[2][0] + R[2][1]);
  real dotp = 0;
  for (int j = 0; j < NJ; ++j) dotp += co[2 * NJ + j] * co[NJ + j];   /* GetMotorTorques . GetMotorVelocities */
  real energy = -fabs(dotp) * sim_dt(c);
  return cfgf(c->distance_weight) * fwd + cfgf(c->energy_weight) * energy + cfgf(c->drift_weight) * drift + cfgf(c->shake_weight) * shake;
}

static int env_fallen(const RexConfig* c, const Env* e) {
  real rpy[3];
  quat_to_euler(e->ph.quat, rpy);
  if (c->task == REX_TASK_GALLOP || c->task == REX_TASK_STANDUP)        /* gallop_env.py:319-329, standup_env.py:138-148 (true RPY) */
    return fabs(rpy[0]) > (real)0.3 || fabs(rpy[1]) > (real)0.5;
  real co[HIST_WORDS];
  control_observation(e, co);
  quat_to_euler(co + 3 * NJ, rpy);                                      /* GetBaseOrientation (delayed) */
  if (c->noise_stdev[3] > 0) { real z[4]; gauss4(c, e->nz_gidx, e->nz_episode, e->nz_step, NZ_FALLEN_RPY, z); for (int k = 0; k < 3; ++k) rpy[k] += cfgf(c->noise_stdev[3]) * z[k]; }
  real qq[4], R[3][3];                                                  /* walk_env.py:326-338 */
  euler_to_quat(rpy, qq);
  quat_to_mat(qq, R);
  return R[2][2] < (real)0.85;
}

/* Box bounds of the reference envs (walk_env.py:104-114,364-378; gallop_env.py:119-130,358-376;
 * turn_env.py:100-110; poses_env.py:115-117; rex_gym_env.py:277-278) for the folded wrappers */
static void action_bounds(const RexConfig* c, real* lo, real* hi) {
  real b;
  if (c->task == REX_TASK_WALK) b = c->signal == REX_SIGNAL_IK ? (real)0.4 : (real)0.01;
  else if (c->task == REX_TASK_GALLOP) b = c->signal == REX_SIGNAL_IK ? (real)-0.4 : (real)-0.3; /* low=+b, high=-b */
  else if (c->task == REX_TASK_TURN) b = (real)0.01;
  else b = (real)0.1;
  *lo = -b; *hi = b;
}
static real obs_bound(const RexConfig* c, int k) {
  real two_pi = (real)(2 * M_PI);
  if (k == 2 || k == 3) return two_pi / sim_dt(c) + (real)0.01;
  return two_pi + (real)0.01;
}
static void normalize_obs(const RexConfig* c, real* obs, int n) {
  for (int k = 0; k < n; ++k) { real hi = obs_bound(c, k), lo = -hi; obs[k] = 2 * (obs[k] - lo) / (hi - lo) - 1; }
}

/* <Env>._transform_action_to_motor_command of the five task envs: the 12 leg targets from the env action and the
 * env's controller state (gait phase, goal / brake flags, end time) at t = steps * action_repeat * dt */
static void env_command(const RexConfig* c, Env* e, const real* action, real* leg_cmd) {
  if (c->task == REX_TASK_GALLOP) gallop_command(c, e, action, leg_cmd);
  else if (c->task == REX_TASK_TURN) turn_command(c, e, action, leg_cmd);
  else if (c->task == REX_TASK_POSES) poses_command(c, e, action, leg_cmd);
  else if (c->task == REX_TASK_STANDUP) {                                           /* RexStandupEnv._signal, standup_env.py:113-120 */
    const clk t = env_time(c, e);    /* GetTimeSinceReset, rex.py:155-156 */
    real f = t > 0.1 ? 1 : ((real)0.1 + action[0]) / ((real)t + 1) + (real)1.5;    /* the 'brake' function */
    for (int j = 0; j < 12; ++j) leg_cmd[j] = POSE_STAND[j] * f;
  }
  else walk_command(c, e, action, leg_cmd);
}

static void env_step(Orc* o, int idx, const real* action_in, real* obs, real* reward, uint8_t* done, real* motor_cmd) {
  const RexConfig cfg_env = env_cfg(o, idx);
  const RexConfig* c = &cfg_env;
  Env* e = &o->envs[idx];
  e->nz_gidx = c->env_index_base + idx; e->nz_episode = e->episode; e->nz_step = e->steps;
  TR_EVENTS = o->trace ? o->trace + idx : 0; TR_SWEEPS = o->trace ? o->trace + o->cfg.num_envs + idx : 0; TR_STEP_SWEEPS = 0;
  TR_LEG_EVENTS = o->trace ? o->trace + 2 * (size_t)o->cfg.num_envs + idx : 0;
  real cmd[NJ], leg_cmd[12];
  real action[8];
  {
    int ad = orc_action_dim(c);   /* this env's own task; the batch's rows may be wider (REX_TASK_MIXED) */
    real lo, hi;
    action_bounds(c, &lo, &hi);
    for (int k = 0; k < ad; ++k) {
      real a = action_in[k];
      if (c->range_normalize) { a = clampr(a, -1, 1); a = (a + 1) / 2 * (hi - lo) + lo; }   /* ClipAction, RangeNormalize */
      action[k] = a;
    }
  }
  env_command(c, e, action, leg_cmd);
  full_command(leg_cmd, cmd);
  if (TR_EVENTS) {   /* the controller's discrete decisions of this step: goal / brake / hold flags, gait latches */
    const uint32_t w = trace_mix(e->flags, (uint32_t)e->last_step * 65537u + (uint32_t)e->end_step);
    *TR_EVENTS = trace_mix(*TR_EVENTS, w);
    *TR_LEG_EVENTS = trace_mix(*TR_LEG_EVENTS, w);
  }
  Ground ground = env_ground(o, idx, e->episode);
  for (int k = 0; k < c->action_repeat; ++k) rex_substep(c, e, cmd, &ground);    /* Rex.Step, rex.py:158-163 */
  TR_EVENTS = 0; TR_SWEEPS = 0; TR_LEG_EVENTS = 0;   /* (the reset motion of an auto-reset below is not part of the trace: the kernels restore a snapshot) */
  if (c->task == REX_TASK_TURN) *reward = (real)0.035 - fabs(e->ph.pos[0]) - fabs(e->ph.pos[1]);  /* turn_env.py:362-367 */
  else if (c->task == REX_TASK_POSES) { *reward = 1; for (int j = 0; j < NJ; ++j) (void)e->tau_obs[j]; } /* poses_env.py:267-269 */
  else if (c->task == REX_TASK_STANDUP) {                                /* standup_env.py:150-166, target (0, 0, 0.21) */
    real pr = fabs(e->ph.pos[0]) + fabs(e->ph.pos[1]) + fabs((real)0.21 - e->ph.pos[2]);
    pr = pr < (real)0.1 ? 1 - pr : -pr;
    if (e->ph.pos[2] > (real)0.21) pr = -1 - pr;
    *reward = pr;
  }
  else *reward = base_reward(c, e);
  int d = env_fallen(c, e);
  if (c->task == REX_TASK_POSES) d = 0;                                 /* is_fallen() returns False, poses_env.py:265 */
  if ((e->flags & REX_F_ENV_GOAL) && c->task != REX_TASK_STANDUP) d = 1; /* rex_gym_env.py:495; standup overrides _termination */
  if (c->task == REX_TASK_GALLOP && e->ph.pos[1] > (real)0.3) d = 1;    /* gallop_env.py:315-317 */
  e->steps += 1;
  if (c->max_episode_steps > 0 && e->steps >= c->max_episode_steps) d = 1;
  *done = (uint8_t)d;
  if (motor_cmd) memcpy(motor_cmd, cmd, sizeof(real) * NJ);
  if (d) e->flags |= REX_F_DONE;
  if (d && c->auto_reset) { int g = e->nz_gidx, ep = e->nz_episode, st = e->nz_step; env_reset(o, idx); e->nz_gidx = g; e->nz_episode = ep; e->nz_step = st; }
  for (int k = 4; k < orc_obs_dim(&o->cfg); ++k) obs[k] = 0;   /* a narrower task leaves the tail of a mixed batch's row 0 */
  env_observation(c, e, obs);
  if (c->range_normalize) normalize_obs(c, obs, orc_obs_dim(&o->cfg));
}

/* ---------------------------------- batch API (mirrors include/rexsim.h) ---------------------------------- */
ORC_API int orc_sizeof_real(void) { return (int)sizeof(real); }
/* thread count of the OpenMP loops over envs (the environment variable is read once, when the first OpenMP runtime of
 * the process initialises -- in a torch process long before this library is loaded) */
#ifdef _OPENMP
#include <omp.h>
ORC_API int orc_set_threads(int n) { if (n > 0) omp_set_num_threads(n); return omp_get_max_threads(); }
#else
ORC_API int orc_set_threads(int n) { (void)n; return 1; }
#endif

ORC_API void* orc_create(const RexConfig* cfg) {
  Orc* o = (Orc*)calloc(1, sizeof(Orc));
  o->cfg = *cfg;
  o->envs = (Env*)calloc((size_t)cfg->num_envs, sizeof(Env));
  if (cfg->pd_latency > 0 || cfg->control_latency > 0)
    for (int i = 0; i < cfg->num_envs; ++i) o->envs[i].hist = calloc(REX_HISTORY_LEN, sizeof(real[HIST_WORDS]));
  o->n_mix = 1; o->mix_task[0] = cfg->task;
  if (cfg->task == REX_TASK_MIXED) {
    o->n_mix = 0;
    for (int t = 0; t < 5; ++t) if ((cfg->task_mix >> t) & 1) o->mix_task[o->n_mix++] = t;
  }
  o->geo = HF_RANDOM;
  Ground g0 = {0, 0, 1, 1, FRICTION_MU, 0, cfg->body_contacts, HF_RANDOM, cfg->on_rack};
  for (int k = 0; k < o->n_mix; ++k) { RexConfig ct = task_cfg(o, o->mix_task[k]); settle(o, &o->snapshot[k], &g0, &ct); }
  return o;
}
/* event trace: trace = caller-owned uint32 [3][num_envs] (NULL: off), updated by orc_step as the kernels update theirs */
ORC_API void orc_set_event_trace(void* h, uint32_t* trace) { ((Orc*)h)->trace = trace; }
ORC_API void orc_set_body_params(void* h, const float* params) {
  Orc* o = (Orc*)h;
  free(o->body_params); o->body_params = 0;
  if (params) {
    o->body_params = (float*)malloc(sizeof(float) * 3 * (size_t)o->cfg.num_envs);
    memcpy(o->body_params, params, sizeof(float) * 3 * (size_t)o->cfg.num_envs);
  }
}
ORC_API void orc_destroy(void* h) { Orc* o = (Orc*)h; for (int i = 0; i < o->cfg.num_envs; ++i) free(o->envs[i].hist); for (int k = 0; k < o->n_mix; ++k) free(o->snapshot[k].hist); for (int t = 0; t < o->n_terrain * o->n_mix; ++t) free(o->terrain_snapshot[t].hist); free(o->body_params); free(o->envs); free(o->heights); free(o->mids); free(o->terrain_snapshot); free(o); }

/* terrain pool: heights [k][256*256] raw vertex heights (terrain.py:36-43 layout: data[i + j*rows], i along x),
 * mids [k] = (min+max)/2 of each field */
static void install_terrain(Orc* o, const float* heights, const float* mids, int k) {
  for (int t = 0; t < o->n_terrain * o->n_mix; ++t) free(o->terrain_snapshot[t].hist);
  free(o->heights); free(o->mids); free(o->terrain_snapshot);
  size_t per = (size_t)o->geo.nx * o->geo.ny;
  o->n_terrain = k;
  o->heights = (float*)malloc(sizeof(float) * (size_t)k * per);
  o->mids = (float*)malloc(sizeof(float) * (size_t)k);
  o->terrain_snapshot = (Env*)calloc((size_t)k * o->n_mix, sizeof(Env));
  memcpy(o->heights, heights, sizeof(float) * (size_t)k * per);
  memcpy(o->mids, mids, sizeof(float) * (size_t)k);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic)
#endif
  for (int r = 0; r < k * o->n_mix; ++r) {
    int t = r / o->n_mix;
    Ground g = {o->heights + (size_t)t * per, (real)o->mids[t], 1, 1, FRICTION_MU, 0, o->cfg.body_contacts, o->geo, o->cfg.on_rack};
    RexConfig ct = task_cfg(o, o->mix_task[r % o->n_mix]);
    settle(o, &o->terrain_snapshot[r], &g, &ct);
  }
}
ORC_API void orc_set_terrain(void* h, const float* heights, const float* mids, int k) {
  Orc* o = (Orc*)h;
  o->geo = HF_RANDOM;
  install_terrain(o, heights, mids, k);
}
/* any other heightfield pool (rex_set_heightfield in include/rexsim.h): heights in metres, [k][ny][nx] */
ORC_API void orc_set_heightfield(void* h, const float* heights, const float* mids, int k, int nx, int ny, float cell_x, float cell_y,
                                 float origin_x, float origin_y) {
  Orc* o = (Orc*)h;
  /* the same float32 arithmetic as the product's host code, so that both sides hold identical grid constants */
  float icx = 1.0f / cell_x, icy = 1.0f / cell_y;
  HfGeom g = {nx, ny, (real)icx, (real)icy, (real)(0.5f * (float)(nx - 1) - origin_x * icx), (real)(0.5f * (float)(ny - 1) - origin_y * icy),
              (real)((float)(nx - 1) - 0.001f), (real)((float)(ny - 1) - 0.001f)};
  o->geo = g;
  install_terrain(o, heights, mids, k);
}

ORC_API void orc_reset(void* h, const int32_t* indices, int n, real* obs) {
  Orc* o = (Orc*)h;
  int od = orc_obs_dim(&o->cfg);
  int cnt = indices ? n : o->cfg.num_envs;
  for (int r = 0; r < cnt; ++r) {
    int idx = indices ? indices[r] : r;
    env_reset(o, idx);
    RexConfig ce = env_cfg(o, idx);
    o->envs[idx].nz_gidx = ce.env_index_base + idx; o->envs[idx].nz_episode = o->envs[idx].episode; o->envs[idx].nz_step = -1;
    for (int k = 4; k < od; ++k) obs[(size_t)r * od + k] = 0;
    env_observation(&ce, &o->envs[idx], obs + (size_t)r * od);
    if (o->cfg.range_normalize) normalize_obs(&o->cfg, obs + (size_t)r * od, od);
  }
}

ORC_API void orc_step(void* h, const real* action, real* obs, real* reward, uint8_t* done, real* motor_cmd) {
  Orc* o = (Orc*)h;
  int od = orc_obs_dim(&o->cfg), ad = orc_action_dim(&o->cfg);
  int n = o->cfg.num_envs;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int i = 0; i < n; ++i)
    env_step(o, i, action + (size_t)i * ad, obs + (size_t)i * od, reward + i, done + i, motor_cmd ? motor_cmd + (size_t)i * NJ : 0);
}

/* The command half of env.step() alone (no physics, no step counter): what the env would send to the motors for
 * `action` from env idx's CURRENT state.  Lets tests replay scripted (time, base pose, action) sequences against
 * vectors produced by the reference's own _transform_action_to_motor_command (tests/golden/make_env_golden.py). */
ORC_API void orc_env_command(void* h, int idx, const real* action, real* cmd_out) {
  Orc* o = (Orc*)h;
  real leg_cmd[12];
  RexConfig ce = env_cfg(o, idx);
  env_command(&ce, &o->envs[idx], action, leg_cmd);
  full_command(leg_cmd, cmd_out);
}

/* state exchange: numeric values, word-major [REX_STATE_WORDS][N] as in rexsim.h (ints as numbers;
 * overheat counters unpacked are NOT used here: words SW_OVERHEAT+k hold lo + 65536*hi) */
ORC_API void orc_get_state(void* h, double* out) {
  Orc* o = (Orc*)h;
  int n = o->cfg.num_envs;
  for (int i = 0; i < n; ++i) {
    const Env* e = &o->envs[i];
#define W(w) out[(size_t)(w) * n + i]
    for (int k = 0; k < 3; ++k) { W(REX_S_POS + k) = e->ph.pos[k]; W(REX_S_LINVEL + k) = e->ph.linvel[k]; W(REX_S_ANGVEL + k) = e->ph.angvel[k]; }
    for (int k = 0; k < 4; ++k) W(REX_S_QUAT + k) = e->ph.quat[k];
    for (int j = 0; j < NJ; ++j) { W(SW_Q + j) = e->ph.q[j]; W(SW_QD + j) = e->ph.qd[j]; }
    W(SW_PHI) = e->phi; W(SW_LASTT) = e->last_step; W(SW_ALPHA) = e->alpha;   /* LASTT / ENDTIME: env step counts, as in rexsim.h */
    W(SW_TARGET) = e->target; W(SW_ENDTIME) = e->end_step; W(SW_AUX) = e->aux;
    W(SW_FLAGS) = e->flags; W(SW_STEPS) = e->steps; W(SW_EPISODE) = e->episode;
    W(SW_MOTOR_EN) = e->motor_enabled;
    W(SW_HIST) = e->hist ? (double)(e->hist_head + 256 * e->hist_len) : 0.0;
    for (int k = 0; k < NJ / 2; ++k) W(SW_OVERHEAT + k) = (double)e->overheat[2 * k] + 65536.0 * (double)e->overheat[2 * k + 1];
#undef W
  }
}

ORC_API void orc_set_state(void* h, const double* in) {
  Orc* o = (Orc*)h;
  int n = o->cfg.num_envs;
  for (int i = 0; i < n; ++i) {
    Env* e = &o->envs[i];
#define W(w) in[(size_t)(w) * n + i]
    for (int k = 0; k < 3; ++k) { e->ph.pos[k] = (real)W(REX_S_POS + k); e->ph.linvel[k] = (real)W(REX_S_LINVEL + k); e->ph.angvel[k] = (real)W(REX_S_ANGVEL + k); }
    for (int k = 0; k < 4; ++k) e->ph.quat[k] = (real)W(REX_S_QUAT + k);
    for (int j = 0; j < NJ; ++j) { e->ph.q[j] = (real)W(SW_Q + j); e->ph.qd[j] = (real)W(SW_QD + j); }
    e->phi = (real)W(SW_PHI); e->last_step = (int32_t)W(SW_LASTT); e->alpha = (real)W(SW_ALPHA);
    e->target = (real)W(SW_TARGET); e->end_step = (int32_t)W(SW_ENDTIME); e->aux = (real)W(SW_AUX);
    e->flags = (uint32_t)W(SW_FLAGS); e->steps = (int32_t)W(SW_STEPS); e->episode = (int32_t)W(SW_EPISODE);
    e->motor_enabled = (uint32_t)W(SW_MOTOR_EN);
    if (!e->hist) true_observation(e, e->ctrl_obs);   /* an injected state is also what the robot last observed */
    for (int k = 0; k < NJ / 2; ++k) {
      uint32_t v = (uint32_t)W(SW_OVERHEAT + k);
      e->overheat[2 * k] = (uint16_t)(v & 0xFFFFu); e->overheat[2 * k + 1] = (uint16_t)(v >> 16);
    }
#undef W
  }
}

ORC_API void orc_set_joint_friction(real f, real visc) { DBG_JOINT_FRICTION = f; DBG_JOINT_VISC = visc; }
ORC_API void orc_leg_census(long* out, int reset) { DBG_STATS = 1; memcpy(out, DBG_LEGS, sizeof(DBG_LEGS)); if (reset) memset(DBG_LEGS, 0, sizeof(DBG_LEGS)); }
ORC_API void orc_solver_hist(long* h) { DBG_STATS = 1; for (int i = 0; i < 64; ++i) { h[i] = DBG_HIST[i]; DBG_HIST[i] = 0; } }
ORC_API void orc_solver_stats(long* sweeps, long* substeps, int reset) { DBG_STATS = 1; *sweeps = DBG_SWEEPS; *substeps = DBG_SUBSTEPS; if (reset) { DBG_SWEEPS = 0; DBG_SUBSTEPS = 0; } }
ORC_API void orc_set_friction(real mu) { FRICTION_MU = mu; }
ORC_API void orc_set_gait_clock(real s) { DBG_GAIT_CLOCK = s; }
ORC_API void orc_set_body_contacts(int on) { BODY_CONTACTS = on; }
ORC_API long orc_body_points(int reset) { DBG_STATS = 1; long v = DBG_BODY_POINTS; if (reset) DBG_BODY_POINTS = 0; return v; }
ORC_API long orc_self_points(int reset) { DBG_STATS = 1; long v = DBG_SELF_POINTS; if (reset) DBG_SELF_POINTS = 0; return v; }
ORC_API long orc_substeps_with_body_points(int reset) { DBG_STATS = 1; long v = DBG_BODY_SUBSTEPS; if (reset) DBG_BODY_SUBSTEPS = 0; return v; }
/* sensitivity probes by name (tools/physics_sensitivity.py); returns 0 when the name is known */
ORC_API int orc_set_probe(const char* name, double v) {
  if (!strcmp(name, "erp")) P_ERP = (real)v;
  else if (!strcmp(name, "slop")) P_SLOP = (real)v;
  else if (!strcmp(name, "inertia_scale")) P_INERTIA_SCALE = (real)v;
  else if (!strcmp(name, "leg_inertia_add")) P_LEG_INERTIA_ADD = (real)v;
  else if (!strcmp(name, "toe_mode")) P_TOE_MODE = (int)v;
  else if (!strcmp(name, "limit_exact")) P_LIMIT_EXACT = (int)v;
  else if (!strcmp(name, "breaking")) P_BREAKING = (real)v;
  else if (!strcmp(name, "margin")) P_MARGIN = (real)v;
  else if (!strcmp(name, "friction_dirs")) P_FRICTION_DIRS = (int)v;
  else if (!strcmp(name, "cone")) P_CONE = (int)v;
  else if (!strcmp(name, "cover_com_z")) P_COVER_COM_Z = (real)v;
  else if (!strcmp(name, "gait_clock")) DBG_GAIT_CLOCK = (real)v;
  else if (!strcmp(name, "mu")) FRICTION_MU = (real)v;
  else if (!strcmp(name, "lin_damping")) MB_LINEAR_DAMPING = (real)v;
  else if (!strcmp(name, "ang_damping")) MB_ANGULAR_DAMPING = (real)v;
  else if (!strcmp(name, "joint_friction")) { DBG_JOINT_FRICTION = (real)v; DBG_JOINT_VISC = (real)1e3; }
  else if (!strcmp(name, "body_contacts")) BODY_CONTACTS = (int)v;
  else if (!strcmp(name, "self_collision")) P_SELF_COLLISION = (int)v;
  else if (!strcmp(name, "self_diag_bullet")) P_SELF_DIAG_BULLET = (int)v;
  else return -1;
  return 0;
}
ORC_API void orc_set_damping(real lin, real ang) { MB_LINEAR_DAMPING = lin; MB_ANGULAR_DAMPING = ang; }

/* ---- physics-only probes used by the oracle's own unit tests (tests/test_oracle_physics.py) ---- */
/* state: pos3 quat4 linvel3 angvel3 q12 qd12 = 37 reals, in/out; tau 12 */
static int PROBE_FIXED_BASE = 0;   /* orc_physics_substep only: the body hangs on the rack (loadURDF(useFixedBase=True), rex.py:269-287) */
ORC_API void orc_physics_fixed_base(int on) { PROBE_FIXED_BASE = on; }
ORC_API void orc_physics_substep(real* st, const real* tau, real dt, int iterations, int nsteps, real residual_threshold) {
  Phys p;
  Ground rack = {0, 0, 1, 1, FRICTION_MU, 0, 0, HF_RANDOM, 1};
  memcpy(p.pos, st, sizeof(real) * 3); memcpy(p.quat, st + 3, sizeof(real) * 4);
  memcpy(p.linvel, st + 7, sizeof(real) * 3); memcpy(p.angvel, st + 10, sizeof(real) * 3);
  memcpy(p.q, st + 13, sizeof(real) * NJ); memcpy(p.qd, st + 13 + NJ, sizeof(real) * NJ);
  for (int k = 0; k < nsteps; ++k) physics_substep(&p, tau, dt, iterations, residual_threshold, PROBE_FIXED_BASE ? &rack : 0);
  memcpy(st, p.pos, sizeof(real) * 3); memcpy(st + 3, p.quat, sizeof(real) * 4);
  memcpy(st + 7, p.linvel, sizeof(real) * 3); memcpy(st + 10, p.angvel, sizeof(real) * 3);
  memcpy(st + 13, p.q, sizeof(real) * NJ); memcpy(st + 13 + NJ, p.qd, sizeof(real) * NJ);
}

/* unconstrained accelerations (ABA) for a state: out = wdot_w3, vdot_w3, qdd12 */
ORC_API void orc_forward_dynamics(const real* st, const real* tau, real* out) {
  Phys p;
  static __thread Aba A;
  memcpy(p.pos, st, sizeof(real) * 3); memcpy(p.quat, st + 3, sizeof(real) * 4);
  memcpy(p.linvel, st + 7, sizeof(real) * 3); memcpy(p.angvel, st + 10, sizeof(real) * 3);
  memcpy(p.q, st + 13, sizeof(real) * NJ); memcpy(p.qd, st + 13 + NJ, sizeof(real) * NJ);
  aba_forward(&p, tau, &A, out + 6, out, out + 3);
}

/* total mechanical energy (kinetic + m g h) and world linear momentum, for conservation tests */
ORC_API void orc_energy_momentum(const real* st, real* out /* E, px, py, pz */) {
  Phys p;
  static __thread Aba A;
  real tau[NJ] = {0}, qdd[NJ], wd[3], vd[3];
  memcpy(p.pos, st, sizeof(real) * 3); memcpy(p.quat, st + 3, sizeof(real) * 4);
  memcpy(p.linvel, st + 7, sizeof(real) * 3); memcpy(p.angvel, st + 10, sizeof(real) * 3);
  memcpy(p.q, st + 13, sizeof(real) * NJ); memcpy(p.qd, st + 13 + NJ, sizeof(real) * NJ);
  aba_forward(&p, tau, &A, qdd, wd, vd); /* fills Rw, pw, v (body spatial velocities) */
  real E = 0, P[3] = {0, 0, 0};
  for (int i = 0; i < NB; ++i) {
    real I[6][6], Iv[6];
    body_inertia6(i, I);
    mat6vec(I, A.v[i], Iv);
    real ke = 0; for (int k = 0; k < 6; ++k) ke += A.v[i][k] * Iv[k];
    E += (real)0.5 * ke;
    real c[3] = {m_com(i, 0), m_com(i, 1), m_com(i, 2)}, cw[3];
    matvec3(A.Rw[i], c, cw);
    E += m_mass0(i) * (-GRAVITY_Z) * (A.pw[i][2] + cw[2]);
    real pl[3]; matvec3(A.Rw[i], Iv + 3, pl);
    for (int k = 0; k < 3; ++k) P[k] += pl[k];
  }
  out[0] = E; out[1] = P[0]; out[2] = P[1]; out[3] = P[2];
}
