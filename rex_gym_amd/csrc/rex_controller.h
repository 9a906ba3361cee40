// rex_controller.h -- device restatement (fp32) of the reference controller and actuator:
//   model/gait_planner.py:22-134, model/kinematics.py:28-142, model/motor.py:76-143.
// Same maths as the numpy code; Bezier powers are built by repeated multiplication instead of
// np.power + factorial, and the swing y component uses Y_i = -|v| s X_i (gait_planner.py:48).
#pragma once
#include <hip/hip_runtime.h>

#include "rex_device.h"

namespace rex {

// kinematics.py:6-13
constexpr float kIkL = 0.23f, kIkW = 0.075f, kIkHip = 0.055f, kIkLeg = 0.10652f, kIkFoot = 0.145f;
constexpr float kIkYDist = 0.185f, kIkHeight = 0.2f;
constexpr float kPi = 3.14159265358979323846f;

// kinematics.py:48-78 : R (coord + pos), R = Rx Ry Rz, identity when all angles are exactly 0
__device__ __forceinline__ void ik_transform(const float* coord, const float* orn, const float* pos, float* out) {
  const float t0 = coord[0] + pos[0], t1 = coord[1] + pos[1], t2 = coord[2] + pos[2];
  if (orn[0] != 0.0f || orn[1] != 0.0f || orn[2] != 0.0f) {
    float sx, cx, sy, cy, sz, cz;
    sincos_fast(orn[0], sx, cx); sincos_fast(orn[1], sy, cy); sincos_fast(orn[2], sz, cz);
    const float a0 = cz * t0 - sz * t1, a1 = sz * t0 + cz * t1, a2 = t2;
    const float b0 = cy * a0 + sy * a2, b1 = a1, b2 = -sy * a0 + cy * a2;
    out[0] = b0; out[1] = cx * b1 - sx * b2; out[2] = sx * b1 + cx * b2;
  } else {
    out[0] = t0; out[1] = t1; out[2] = t2;
  }
}

// kinematics.py:80-102
__device__ __forceinline__ void ik_leg(const float* c, bool right_side, float* out) {
  const float hip = kIkHip, leg = kIkLeg, foot = kIkFoot;
  float domain = (c[1] * c[1] + c[2] * c[2] - hip * hip + c[0] * c[0] - leg * leg - foot * foot) / (2.0f * foot * leg);
  if (domain > 1.0f) domain = 0.99f;
  else if (domain < -1.0f) domain = -0.99f;
  const float gamma = atan2_fast(-sqrtf(1.0f - domain * domain), domain);
  float sq = c[1] * c[1] + c[2] * c[2] - hip * hip;
  if (sq < 0.0f) sq = 0.0f;
  const float rs = sqrtf(sq);
  float sg, cg;
  sincos_fast(gamma, sg, cg);
  const float alpha = atan2_fast(-c[0], rs) - atan2_fast(foot * sg, leg + foot * cg);
  const float theta = -atan2_fast(c[2], c[1]) - atan2_fast(rs, right_side ? -hip : hip);
  out[0] = theta; out[1] = -alpha; out[2] = -gamma;
}

// kinematics.py:104-142 for ONE leg l of the planner's order FR, FL, RR, RL: frame (3) -> angles (3)
__device__ __forceinline__ void ik_solve_leg(const float* orn, const float* pos, const float* frame, int l, float* angles) {
  const float hv[3] = {l < 2 ? kIkL / 2 : -kIkL / 2, (l & 1) ? kIkW / 2 : -kIkW / 2, 0.f};
  const float iorn[3] = {-orn[0], -orn[1], -orn[2]}, ipos[3] = {-pos[0], -pos[1], -pos[2]};
  float hvx[3], coord[3], tc[3];
  ik_transform(hv, orn, pos, hvx);
  coord[0] = frame[0] - hvx[0]; coord[1] = frame[1] - hvx[1]; coord[2] = frame[2] - hvx[2];
  ik_transform(coord, iorn, ipos, tc);
  ik_leg(tc, (l & 1) == 0, angles);
}
// ... and all four: frames/angles rows FR, FL, RR, RL
__device__ __forceinline__ void ik_solve(const float* orn, const float* pos, const float* frames, float* angles) {
#pragma unroll
  for (int l = 0; l < 4; ++l) ik_solve_leg(orn, pos, frames + 3 * l, l, angles + 3 * l);
}

// Sum_{i<10} P_i C(11,i) t^i (1-t)^(11-i) for the x and z control polygons (gait_planner.py:42-58)
__device__ __forceinline__ void bezier_xz(float t, float& bx, float& bz) {
  const float PX[10] = {-0.04f, -0.056f, -0.06f, -0.06f, -0.06f, 0.f, 0.f, 0.f, 0.06f, 0.06f};
  const float PZ[10] = {0.f, 0.f, 0.0405f, 0.0405f, 0.0405f, 0.0405f, 0.0405f, 0.0495f, 0.0495f, 0.0495f};
  const float BIN[10] = {1.f, 11.f, 55.f, 165.f, 330.f, 462.f, 462.f, 330.f, 165.f, 55.f};
  const float u = 1.0f - t;
  float up[12];
  up[0] = 1.0f;
#pragma unroll
  for (int k = 1; k < 12; ++k) up[k] = up[k - 1] * u;
  float tp = 1.0f;
  bx = 0.0f; bz = 0.0f;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const float b = BIN[i] * tp * up[11 - i];
    bx += PX[i] * b; bz += PZ[i] * b;
    tp *= t;
  }
}

// (The compiler's default for device code is to contract a * b - c into one fused operation: with last_time = now = t * clock
// just latched, fma(t, clock, -last_time) is the rounding error of the product instead of 0, the phase of the legs half a
// cycle away becomes 0.5 + 1 ulp, and they swing where the reference's stand.  Every function that touches the double
// clock therefore switches contraction off.)
// GaitPlanner state.  phi and last_time are doubles, as in the reference: with a 5 ms control step the tests on them are
// exact ties in real numbers (0.495 / 0.5 against 0.99, 0.25 / 0.5 against 0.5), decided by double rounding.  The step
// kernel does not store them: it keeps the env step that latched last_time and the outcome of `phi >= 0.99` (rexsim.h,
// "Clocks") and rebuilds both.  A few dozen DP operations per env.step(), far off the solver loops.
struct GaitState { double phi, last_time; float alpha; };

// gait_planner.py:30-40 and 42-58 for one (v, angle) pair at a given phase
__device__ __forceinline__ void gait_component(bool stance, float ph, float bx, float bz, float v, float angle_deg,
                                               float direction, float* out) {
  float s, c;
  sincos_fast(angle_deg * (kPi / 180.0f), s, c);
  const float av = fabsf(v);
  if (stance) {
    const float p = 0.05f * (1.0f - 2.0f * ph);
    out[0] = c * p * av;
    out[1] = -s * p * av;
    float sp, cp;
    sincos_fast(kPi / (2.0f * 0.05f) * p, sp, cp);
    out[2] = -0.001f * cp;
  } else {
    const float X = av * c * direction * bx;
    out[0] = X;
    out[1] = av * s * (-X);
    out[2] = av * bz;
  }
}

// default stance foot of leg l (FR, FL, RR, RL): (+-L/2, +-Ydist/2, -height), gait_planner.py:112-119
__device__ __forceinline__ float gait_bx0(int l) { return l < 2 ? kIkL / 2 : -kIkL / 2; }
__device__ __forceinline__ float gait_by0(int l) { return (l & 1) ? kIkYDist / 2 : -kIkYDist / 2; }

// where leg l is in its cycle (step_trajectory, gait_planner.py:60-70): phase offset, wrap at 1, stance up to 0.5 -- all
// in double --, the normalised phase inside the stance or swing, and the Bezier basis sums of a swing
struct LegPhase { bool stance; float ph, bx, bz; };
__device__ __forceinline__ LegPhase leg_phase(double gphi, int mode, int l) {
#pragma clang fp contract(off)   // exact double clock arithmetic: no fused multiply-add (rexsim.h, "Clocks")
  const double off = mode == 0 ? ((l == 1 || l == 2) ? 0.5 : 0.0) : (l >= 2 ? 0.8 : 0.0);
  double phi = gphi + off;
  if (phi >= 1.0) phi -= 1.0;
  LegPhase lp;
  lp.stance = phi <= 0.5;
  lp.ph = (float)(lp.stance ? phi / 0.5 : (phi - 0.5) / (1.0 - 0.5));
  lp.bx = 0.f; lp.bz = 0.f;
  if (!lp.stance) bezier_xz(lp.ph, lp.bx, lp.bz);
  return lp;
}
// the rotation part of leg l's step and what it does to the planner's arc angle (gait_planner.py:71-85): `alpha` is
// carried from leg to leg (FR, FL, RR, RL) and from call to call
__device__ __forceinline__ void gait_rot(const LegPhase& lp, int l, float w_rot, float direction, float& alpha, float* rot) {
  const float bx0 = gait_bx0(l), by0 = gait_by0(l);
  const float r = sqrtf(bx0 * bx0 + by0 * by0);
  const float foot_angle = atan2_fast(by0, bx0);
  const float circle = (w_rot >= 0.0f ? 90.0f : 270.0f) - (foot_angle - alpha) * (180.0f / kPi);
  gait_component(lp.stance, lp.ph, lp.bx, lp.bz, w_rot, circle, direction, rot);
  const float mag = atan2_fast(sqrtf(rot[0] * rot[0] + rot[1] * rot[1]), r);
  if (by0 > 0.0f) alpha = rot[0] < 0.0f ? -mag : mag;
  else alpha = rot[0] < 0.0f ? mag : -mag;
}
// translation part + default stance = the foot frame of leg l
__device__ __forceinline__ void gait_frame(const LegPhase& lp, int l, float v, float angle, float direction, const float* rot, float* frame) {
  float lng[3];
  gait_component(lp.stance, lp.ph, lp.bx, lp.bz, v, angle, direction, lng);
  frame[0] = gait_bx0(l) + lng[0] + rot[0];
  frame[1] = gait_by0(l) + lng[1] + rot[1];
  frame[2] = -kIkHeight + lng[2] + rot[2];
}
// GaitPlanner.loop's clock part (gait_planner.py:104-110): the 0.99 latch and the new phase, in double
__device__ __forceinline__ void gait_clock(GaitState& g, double& T, double now) {
#pragma clang fp contract(off)   // exact double clock arithmetic: no fused multiply-add (rexsim.h, "Clocks")
  if (T <= 0.01) T = 0.01;
  if (g.phi >= 0.99) g.last_time = now;
  g.phi = (now - g.last_time) / T;
}

// gait_planner.py:96-134 with the phase clock on `now`; mode 0 walk, 1 gallop (offsets :15-20): all four foot frames
__device__ __forceinline__ void gait_loop(GaitState& g, int mode, float v, float angle, float w_rot, double T,
                                          float direction, double now, float* frame) {
  gait_clock(g, T, now);
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const LegPhase lp = leg_phase(g.phi, mode, l);
    float rot[3];
    gait_rot(lp, l, w_rot, direction, g.alpha, rot);
    gait_frame(lp, l, v, angle, direction, rot, frame + 3 * l);
  }
}

// The same call seen from ONE leg (lane groups of the step kernel: a lane owns one leg, rex_kernels.h): the foot frame of
// leg `own` only.  The arc angle alpha is the one quantity that runs through the legs in sequence; it moves only when the
// gait turns (w_rot != 0) or a turn has left it non-zero -- `chain` (wave-uniform; the caller ballots it): then every lane
// walks the rotation parts of all four legs (a quarter of a leg's work each) and keeps its own; otherwise alpha stays 0
// (mag = atan2(0, r) = 0) and the lane computes its own leg alone.
__device__ __forceinline__ void gait_loop_leg(GaitState& g, int mode, float v, float angle, float w_rot, double T,
                                              float direction, double now, int own, bool chain, float* frame) {
  gait_clock(g, T, now);
  float rot[3] = {0.f, 0.f, 0.f};
  LegPhase mine{true, 0.f, 0.f, 0.f};
  if (chain) {
#pragma unroll 1
    for (int l = 0; l < 4; ++l) {
      const LegPhase lp = leg_phase(g.phi, mode, l);
      float r3[3];
      gait_rot(lp, l, w_rot, direction, g.alpha, r3);
      if (l == own) { mine = lp; rot[0] = r3[0]; rot[1] = r3[1]; rot[2] = r3[2]; }
    }
  } else {
    mine = leg_phase(g.phi, mode, own);
    float a = g.alpha;                 // 0: what the other legs leave behind as well
    gait_rot(mine, own, w_rot, direction, a, rot);
    g.alpha = 0.0f;
  }
  gait_frame(mine, own, v, angle, direction, rot, frame);
}

// model/motor.py:76-143 (position-control branch, strength ratio 1).  Divisions by the constants R and 10
// are written as multiplications by their reciprocals (1 ulp from the numpy result, far inside the 1e-5 N m
// parity tolerance) -- an IEEE fp32 division is a ~15-instruction sequence on gfx950.
__device__ __forceinline__ void motor_torque(float cmd, float q, float qd, float qd_true, float kp, float kd,
                                             float& actual, float& observed) {
  constexpr float V = 32.0f, RINV = 1.0f / 0.186f, KT = 0.0954f;
  float pwm = -1.0f * kp * (q - cmd) - kd * qd;
  pwm = __builtin_amdgcn_fmed3f(pwm, -1.0f, 1.0f);
  observed = __builtin_amdgcn_fmed3f(KT * (pwm * V * RINV), -5.7f, 5.7f);
  const float vnet = __builtin_amdgcn_fmed3f(pwm * V - (KT + 0.0f) * qd_true, -50.0f, 50.0f);
  const float cur = vnet * RINV;
  const float mag = fabsf(cur);
  // np.interp(|I|, [0,10,...,60], [0,1,1.9,2.45,3.0,3.25,3.5]), clamped at the ends.  The table is concave (slopes
  // 0.1, 0.09, 0.055, 0.055, 0.025, 0.025), so the interpolant is the minimum of its segment lines and the end
  // clamp: branch-free, where a per-segment select compiles to a switch tree per motor.
  const float t = fminf(fminf(fminf(0.1f * mag, fmaf(0.09f, mag, 0.1f)), fminf(fmaf(0.055f, mag, 0.8f), fmaf(0.025f, mag, 2.0f))), 3.5f);
  actual = copysignf(t, cur);
}

}  // namespace rex
