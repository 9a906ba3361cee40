// rex_device.h -- gfx950 device code of the batched Rex simulator: one physics substep of one environment, carried
// by one lane (64 envs per wave) or by a group of 4 / 8 adjacent lanes that split the legs, the constraint rows and
// the velocity components between them (<= 16 envs per wave; "lanes-per-env helpers" and pgs_dv below).
//
// Physics formulation (differs on purpose from the CPU oracle's body-coordinate ABA; the two must
// agree to rounding -- tests/test_gpu_parity.py):
//   * every spatial quantity is expressed in world-aligned axes about ONE reference point, the base
//     origin, so composite inertias and subtree wrenches are plain sums (no per-joint transforms);
//   * the star topology (floating base + 4 three-joint legs) is eliminated leg by leg:
//       H_f  = leg joint-space inertia (3x3)        = G_f G_f^T        (Cholesky)
//       B_f  = base/leg coupling (6x3),  Bw_f = G_f^-1 B_f^T
//       A    = composite base inertia - sum_f Bw_f^T Bw_f = Lc Lc^T    (6x6 Cholesky)
//     which is the articulated-body algorithm written as a sparse Cholesky factorisation of M(q);
//   * in the whitened coordinates x = (y, z_0..z_3),  y = Lc^T nu_base,  z_f = G_f^T qd_f + Bw_f nu_base
//     the mass matrix is the identity, so a constraint row is ONE 9-vector Jt (6 base + 3 own-leg
//     entries) that serves as Jacobian AND unit-impulse response: per PGS row 9 FMAs to read the
//     velocity, 9 to apply the impulse (Bullet's delta-velocity PGS costs 18 + 18 on 18 dofs).
// Reference semantics restated here: SURVEY.md section 8(a) rows a1-a23 (file:line cited inline).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rexsim.h"
#include "rex_model_gen.h"
#include "rex_arm_model_gen.h"

#define REX_WAVE 64
#define REX_NPOINT 8                  /* 2 toe-cylinder end points per foot */
#define REX_NCROW (3 * REX_NPOINT)    /* contact rows: normal + 2 pyramid friction rows per point */
#define REX_NLROW (3 * REX_NLEG)      /* joint-limit rows: the near bound of each of the 12 joints */
#define REX_NROW (REX_NCROW + REX_NLROW)
#define REX_ROW_F4 3                  /* float4 chunks per row in LDS */
#define REX_LEG_F4 7                  /* float4 chunks per leg parked in LDS (Bw 18 + G 6 + z 3) when one lane carries an env */
/* Leg chunks of one env in LDS.  One lane per env (EPW = 64): a lane parks all four leg factors around the sweep loop
   (7 chunks per leg).  Lane groups: a lane needs the factor of ITS leg only; it stays in registers (1 chunk per leg: the
   whitened leg velocity z, which every lane reads) where registers are cheaper than LDS -- 8 lanes per env (200 VGPRs)
   and mark 'arm' (LDS-bound: 3 -> 4 workgroups per CU at 16 envs per wave) -- and is parked at 16 envs per wave of mark
   'base', whose 4 lanes per env already run into the AGPRs
   -- and in the link-box kernels of mark 'base': with the factor held in registers across the sweep loop those kernels
   (140 KB of code, 450 registers) came out of hipcc 7.2 with the joint velocities of the back-substitution wrong although
   no link-box row was in reach (same source minus the never-executed candidate search: correct), round 3. */
#ifndef REX_LEG_F4_OF
#define REX_LEG_F4_OF(EPW, ARM, BODY) ((((EPW) <= 8 && !((BODY) && !(ARM))) || ((EPW) <= 16 && (ARM))) ? 1 : REX_LEG_F4)
#endif
#define REX_ROWS_F4_OF(LEGF4) (REX_NROW * REX_ROW_F4 + REX_NLEG * (LEGF4))   /* rows + leg chunks: 136 (2.2 KB) or 112 */
#define REX_LDS_F4_PER_ENV REX_ROWS_F4_OF(REX_LEG_F4)
/* small-batch waves (EPW <= 16) only: chunks 0..1 the whitened base velocity y on its way to / from the lanes that own
   its components (pgs_dv), 2..7 the couplings A(r, r-1) of consecutive contact rows */
#define REX_PARK_F4 8
/* 16 envs per wave, mark 'base' ("pair layout", physics_substep / pgs_dv): 6 more chunks, the inverse diagonals of the 24
   contact rows, whose word in the row makes room for a zero */
#define REX_PAIR_LAYOUT(EPW, ARM) ((EPW) == 16 && !(ARM))
#define REX_PARK_F4_OF(EPW, ARM) (REX_PAIR_LAYOUT(EPW, ARM) ? 14 : REX_PARK_F4)
#define REX_PARK_KI 8
/* link-box contact rows (RexConfig.body_contacts): 12 point slots -- 0..3 the base group (base + chassis boxes), 4 + 2 L + k
   leg L's boxes -- with a normal row (index slot) and two friction rows (12 + 2 slot + d) each, 3 chunks per row like the
   toe rows, in their own LDS region behind the hand-over chunks; lane groups only */
#define REX_NBSLOT 12
#define REX_NBROW (3 * REX_NBSLOT)
// behind the rows: the base Jacobian (2 chunks) of each of the 24 leg-group rows when its point is held against the base
// body -- what the Bullet diagonal of a link-link row needs besides the relative row (physics_substep, row finishing)
#define REX_NBJAC (3 * (REX_NBSLOT - 4))
#define REX_BODY_F4 (REX_NBROW * REX_ROW_F4 + 2 * REX_NBJAC)
#define REX_PARK_XY 0
#define REX_PARK_CPL 2

namespace rex {

// ---- world / solver constants (same values as oracle/rex_oracle.c) ----
constexpr float kGravity = 10.0f;          // rex_gym_env.py:314 (0,0,-10)
constexpr float kLinDamp = 0.04f;          // btMultiBody default damping
constexpr float kAngDamp = 0.04f;
constexpr float kMaxCoordVel = 100.0f;     // btMultiBody::m_maxCoordinateVelocity
constexpr float kErp = 0.2f;               // btContactSolverInfo::m_erp2
constexpr float kBreaking = 0.02f;         // contact breaking threshold
// btMultiBodyJointLimitConstraint::createConstraintRows instantiates a bound's row only once the bound is reached
// (`if (penetration > 0) continue;`): a joint inside its range has no limit row, one at or beyond a bound is pushed back
// through the ERP term
constexpr float kLimitActivation = 0.0f;
constexpr float kMu = 0.5f;                // toe 0.5 x plane 1.0
constexpr float kInitZ = 0.21f;            // terrain.py:14-20 (default drop height; RexConfig.init_height overrides)
constexpr float kToeRad = (float)(REX_TOE_RADIUS + REX_COLLISION_MARGIN);
constexpr float kToeHalf = (float)REX_TOE_HALFLEN;

// LDS view of one workgroup.  EPW = envs per wave (compile-time power of two <= 64): lane l works on env
// slot l & (EPW-1); lanes beyond EPW mirror a live lane (see rex_step_kernel).  Layout float4[chunk][slot]:
// lanes of different slots hit consecutive 16 B -> conflict-free ds_read_b128.
template <int EPW, int LEGF4 = REX_LEG_F4, bool BODY = false>
struct Lds {
  static constexpr int kEpw = EPW;
  static constexpr bool kBody = BODY;
  float4* p; int slot;
  float4* pk;   // hand-over region of pgs_dv (REX_PARK_F4 chunks per env, behind the rows of all marks); null when EPW = 64
  float4* pb;   // link-box contact rows (BODY only): REX_BODY_F4 chunks per env behind the hand-over region
  __device__ __forceinline__ float4& brow(int r, int c) const { return pb[(r * REX_ROW_F4 + c) * EPW + slot]; }
  __device__ __forceinline__ float4& bjac(int i, int c) const { return pb[(REX_NBROW * REX_ROW_F4 + 2 * i + c) * EPW + slot]; }   // i = 3 (slot - 4) + direction
  __device__ __forceinline__ float4& park(int c) const { return pk[c * EPW + slot]; }
  // scalar views for the lanes that own single components (pgs_dv): float f of a chunk sequence starting at chunk c0
  __device__ __forceinline__ float& parkf(int c0, int f) const { return reinterpret_cast<float*>(&pk[(c0 + (f >> 2)) * EPW + slot])[f & 3]; }
  __device__ __forceinline__ float& zf(int l, int f) const { return reinterpret_cast<float*>(&zc(l))[f]; }
  __device__ __forceinline__ float& rowf(int r, int f) const { return reinterpret_cast<float*>(&row(r, f >> 2))[f & 3]; }
  __device__ __forceinline__ float4& row(int r, int c) const { return p[(r * REX_ROW_F4 + c) * EPW + slot]; }
  static constexpr int kLegF4 = LEGF4;
  __device__ __forceinline__ float4& leg(int l, int c) const { return p[(REX_NROW * REX_ROW_F4 + l * kLegF4 + c) * EPW + slot]; }
  __device__ __forceinline__ float4& zc(int l) const { return leg(l, kLegF4 - 1); }   // (z0, z1, z2, 0) of leg l
};

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return f3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
  return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

typedef float v2 __attribute__((ext_vector_type(2)));

// sin and cos to ~1 ulp for |x| < ~1e4: Cody-Waite reduction by pi/2 (3-term split) + minimax
// polynomials on [-pi/4, pi/4].  Joint angles never leave a few radians, so the Payne-Hanek path of
// the libm sincosf (hundreds of instructions, divergent) is dead weight here.
__device__ __forceinline__ void sincos_fast(float x, float& s, float& c) {
  const float k = rintf(x * 0.636619772367581343f);      // x * 2/pi
  float r = fmaf(k, -1.57079601287841796875f, x);        // pi/2 = hi + mid + lo
  r = fmaf(k, -3.1391647326017846353352069854736328125e-07f, r);
  r = fmaf(k, -5.390302529957764765e-15f, r);
  const float r2 = r * r;
  // least-squares minimax fits on r^2 in [0, (pi/4)^2] (tools: numpy Chebyshev nodes; max abs error 9e-8)
  float ps = fmaf(r2, 2.7243820799024522e-06f, -1.9840039244367917e-04f);
  ps = fmaf(ps, r2, 8.33333178609531e-03f);
  ps = fmaf(ps, r2, -1.6666666663625898e-01f);
  const float sr = fmaf(ps * r2, r, r);
  float pc = fmaf(r2, 2.4542922330261397e-05f, -1.3888279652923955e-03f);
  pc = fmaf(pc, r2, 4.16666645388584e-02f);
  const float cr = fmaf(pc * r2, r2, fmaf(r2, -0.5f, 1.0f));
  const int q = (int)k;
  const float s0 = (q & 1) ? cr : sr;
  const float c0 = (q & 1) ? sr : cr;
  s = (q & 2) ? -s0 : s0;
  c = ((q + 1) & 2) ? -c0 : c0;
}

__device__ __forceinline__ float cos_half(float a) { float s, c; sincos_fast(a, s, c); return c; }

// atan2 to 3e-7 rad (what a float32 libm gives) in ~25 branch-free instructions: octant reduction to t = min/max in
// [0, 1], Cephes' atanf split at tan(pi/8) with its degree-4 polynomial in t^2, hardware reciprocals.  A step calls it
// ~30 times (leg IK, gait planner, Euler angles); the libm version is several times that size.
__device__ __forceinline__ float atan2_fast(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  const float t = mx > 0.0f ? mn * __builtin_amdgcn_rcpf(mx) : 0.0f;
  const bool big = t > 0.41421356237f;
  const float u = big ? (t - 1.0f) * __builtin_amdgcn_rcpf(t + 1.0f) : t;
  const float z = u * u;
  float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
  p = fmaf(p, z, 1.99777106478e-1f);
  p = fmaf(p, z, -3.33329491539e-1f);
  float a = (big ? 0.78539816339744831f : 0.0f) + fmaf(p * z, u, u);
  a = ay > ax ? 1.57079632679489662f - a : a;
  a = x < 0.0f ? 3.14159265358979324f - a : a;
  return copysignf(a, y);
}
__device__ __forceinline__ float asin_fast(float s) { return atan2_fast(s, sqrtf(fmaxf(fmaf(-s, s, 1.0f), 0.0f))); }

// symmetric 3x3: xx yy zz xy xz yz
struct s33 { float xx, yy, zz, xy, xz, yz; };
__device__ __forceinline__ f3 mul(const s33& m, f3 v) {
  return f3{m.xx * v.x + m.xy * v.y + m.xz * v.z, m.xy * v.x + m.yy * v.y + m.yz * v.z,
            m.xz * v.x + m.yz * v.y + m.zz * v.z};
}
// R diag(ix,iy,iz) R^T with R = [ex ey ez] columns
__device__ __forceinline__ s33 rot_inertia(f3 ex, f3 ey, f3 ez, float ix, float iy, float iz) {
  s33 m;
  m.xx = ix * ex.x * ex.x + iy * ey.x * ey.x + iz * ez.x * ez.x;
  m.yy = ix * ex.y * ex.y + iy * ey.y * ey.y + iz * ez.y * ez.y;
  m.zz = ix * ex.z * ex.z + iy * ey.z * ey.z + iz * ez.z * ez.z;
  m.xy = ix * ex.x * ex.y + iy * ey.x * ey.y + iz * ez.x * ez.y;
  m.xz = ix * ex.x * ex.z + iy * ey.x * ey.z + iz * ez.x * ez.z;
  m.yz = ix * ex.y * ex.z + iy * ey.y * ey.z + iz * ez.y * ez.z;
  return m;
}
// point-mass inertia about the origin: m (|c|^2 1 - c c^T)
__device__ __forceinline__ void add_point(s33& m, float mass, f3 c) {
  float cc = dot(c, c);
  m.xx += mass * (cc - c.x * c.x); m.yy += mass * (cc - c.y * c.y); m.zz += mass * (cc - c.z * c.z);
  m.xy -= mass * c.x * c.y; m.xz -= mass * c.x * c.z; m.yz -= mass * c.y * c.z;
}
__device__ __forceinline__ void add(s33& a, const s33& b) {
  a.xx += b.xx; a.yy += b.yy; a.zz += b.zz; a.xy += b.xy; a.xz += b.xz; a.yz += b.yz;
}

// ---------------------------------------------------------------------------------------------
// per-substep shared quantities of the base
struct BaseKin {
  f3 ex, ey, ez;   // columns of the base rotation (body axes in world)
  f3 w, v;         // base angular / linear velocity, world
  float px, py;    // base origin x, y (world): only the heightfield lookup needs them
  float height;    // base origin z
};

// Ground = the z = 0 box (plane.urdf) united with an optional random heightfield (model/terrain.py:32-54):
// 256 x 256 vertex heights, 5 cm cells, centred on the origin, shifted down by `mid` (Bullet centres a
// heightfield on the middle of its height range) -- the plane shows wherever the field is
// below 0.  Triangulation: Bullet's default diagonal (i,j+1)-(i+1,j).  h == nullptr: plane only.
// Grid geometry of the heightfield pool (wave-uniform): nx vertices per row, index scale and offset per axis
// (vertex coordinate = x * inv_cx + off_x, clamped to [0, max_x]).  The reference's random terrain: 256, 20, 127.5, 254.999.
struct HfGeom { int nx; float inv_cx, inv_cy, off_x, off_y, max_x, max_y; };
struct Ground {
  // the heightfield pool (wave-uniform; nullptr: plane only) and this env's field in it as a 32-bit element offset: one
  // register instead of an address pair carried through the whole step, and "is there a field" is a scalar test
  const float* h; unsigned off; float mid;
  // per-env domain randomisation (Rex.SetBaseMasses / SetLegMasses, rex.py:659-692: masses only -- Bullet keeps the
  // inertia tensors computed at load time; plus the foot friction coefficient)
  float base_mass_scale, leg_mass_scale, mu;
  HfGeom geo;
  // on_rack (loadURDF(useFixedBase=True), rex.py:269-287): kRackAnchor added to the mass and the principal inertias of the
  // base block -- the base's share of every velocity change becomes 1e-9 of what it was, and the integrator drops it; 0 = free base
  float anchor = 0.0f;
};
constexpr float kRackAnchor = 1.0e9f;
constexpr float kSelfMargin = 0.001f;   // a corner is a candidate from 1 mm before it enters the other box (oracle: SELF_MARGIN)
constexpr float kSelfMu = 0.25f;   // btManifoldResult::calculateCombinedFriction: 0.5 x 0.5, the URDF default of both links
// (every division by the cell size is a multiplication by its inverse, here and in the oracle)
// `fid` (event trace only): the facet the point stands on -- 1 + 2 (cell index) + (upper triangle), 0 where the plane is on top
__device__ __forceinline__ void ground_query(const Ground& g, float x, float y, float& height, f3& n, unsigned& fid) {
  n = f3{0.f, 0.f, 1.f}; height = 0.0f;
  fid = 0u;
  const HfGeom& q = g.geo;
  float fx = fminf(fmaxf(x * q.inv_cx + q.off_x, 0.0f), q.max_x), fy = fminf(fmaxf(y * q.inv_cy + q.off_y, 0.0f), q.max_y);
  const int i = (int)fx, j = (int)fy;
  const float u = fx - (float)i, v = fy - (float)j;
  const unsigned o = g.off + (unsigned)(j * q.nx + i);
  const float h00 = g.h[o], h10 = g.h[o + 1u], h01 = g.h[o + (unsigned)q.nx], h11 = g.h[o + (unsigned)q.nx + 1u];
  float hh, gx, gy;
  if (u + v <= 1.0f) { hh = h00 + u * (h10 - h00) + v * (h01 - h00); gx = (h10 - h00) * q.inv_cx; gy = (h01 - h00) * q.inv_cy; }
  else { hh = h11 + (1.0f - u) * (h01 - h11) + (1.0f - v) * (h10 - h11); gx = (h11 - h01) * q.inv_cx; gy = (h11 - h10) * q.inv_cy; }
  hh -= g.mid;
  if (hh > 0.0f) {
    const float inv = rsqrtf(gx * gx + gy * gy + 1.0f);
    height = hh; n = f3{-gx * inv, -gy * inv, inv};
    fid = 1u + 2u * (unsigned)(j * q.nx + i) + (u + v <= 1.0f ? 0u : 1u);
  }
}
// ---- event trace (debug; rex_set_event_trace): the discrete events of a substep folded into one word per env, the same way
// in oracle/rex_oracle.c -- which toe points are in reach, which facet of the heightfield each of them stands on (under the
// end centre and under the contact point), which joint / arm bounds are reached.  Two runs whose words agree took every
// discrete decision of the contact set-up alike; what is left between them is continuous round-off.
__device__ __forceinline__ void ground_query(const Ground& g, float x, float y, float& height, f3& n) {
  unsigned fid;
  ground_query(g, x, y, height, n, fid);
}
__device__ __forceinline__ unsigned trace_point(int p, unsigned f0, unsigned f1) {
  unsigned m = f0 * 0x9E3779B1u + f1 * 0x85EBCA77u + (unsigned)(p + 1) * 0xC2B2AE3Du;
  m ^= m >> 15; m *= 0x2C1B3C6Du; m ^= m >> 12;
  return m;
}
__device__ __forceinline__ unsigned trace_mix(unsigned h, unsigned w) { h = (h ^ w) * 0x01000193u; return h ^ (h >> 13); }
// btPlaneSpace1
__device__ __forceinline__ void plane_space(f3 n, f3& p, f3& q) {
  if (fabsf(n.z) > 0.7071067811865475244f) {
    const float a = n.y * n.y + n.z * n.z, k = rsqrtf(a);
    p = f3{0.f, -n.z * k, n.y * k};
    q = f3{a * k, -n.x * p.z, n.x * p.y};
  } else {
    const float a = n.x * n.x + n.y * n.y, k = rsqrtf(a);
    p = f3{-n.y * k, n.x * k, 0.f};
    q = f3{-n.z * p.y, n.z * p.x, a * k};
  }
}

// what a leg leaves behind for the back-substitution after the constraint solve
struct LegFactor {
  float gi1, gi2, gi3;     // 1 / diag(G)
  float g21, g31, g32;     // strict lower triangle of G
  float Bw[3][6];          // G^-1 B^T
  float z[3];              // whitened predicted velocity of the leg
};
// the part of LegFactor that is only needed again after the constraint solve is parked in LDS

template <class SM>
__device__ __forceinline__ void leg_park(const SM& sm, int leg, const LegFactor& L) {
  sm.leg(leg, 0) = make_float4(L.Bw[0][0], L.Bw[0][1], L.Bw[0][2], L.Bw[0][3]);
  sm.leg(leg, 1) = make_float4(L.Bw[0][4], L.Bw[0][5], L.Bw[1][0], L.Bw[1][1]);
  sm.leg(leg, 2) = make_float4(L.Bw[1][2], L.Bw[1][3], L.Bw[1][4], L.Bw[1][5]);
  sm.leg(leg, 3) = make_float4(L.Bw[2][0], L.Bw[2][1], L.Bw[2][2], L.Bw[2][3]);
  sm.leg(leg, 4) = make_float4(L.Bw[2][4], L.Bw[2][5], L.gi1, L.gi2);
  sm.leg(leg, 5) = make_float4(L.gi3, L.g21, L.g31, L.g32);
  sm.zc(leg) = make_float4(L.z[0], L.z[1], L.z[2], 0.0f);
}
template <class SM>
__device__ __forceinline__ void leg_unpark(const SM& sm, int leg, LegFactor& L) {
  const float4 a = sm.leg(leg, 0), b = sm.leg(leg, 1), c = sm.leg(leg, 2);
  const float4 d = sm.leg(leg, 3), e = sm.leg(leg, 4), f = sm.leg(leg, 5);
  L.Bw[0][0] = a.x; L.Bw[0][1] = a.y; L.Bw[0][2] = a.z; L.Bw[0][3] = a.w; L.Bw[0][4] = b.x; L.Bw[0][5] = b.y;
  L.Bw[1][0] = b.z; L.Bw[1][1] = b.w; L.Bw[1][2] = c.x; L.Bw[1][3] = c.y; L.Bw[1][4] = c.z; L.Bw[1][5] = c.w;
  L.Bw[2][0] = d.x; L.Bw[2][1] = d.y; L.Bw[2][2] = d.z; L.Bw[2][3] = d.w; L.Bw[2][4] = e.x; L.Bw[2][5] = e.y;
  L.gi1 = e.z; L.gi2 = e.w; L.gi3 = f.x; L.g21 = f.y; L.g31 = f.z; L.g32 = f.w;
}

// accumulators of the base block
struct BaseAccum {
  s33 Io;          // composite rotational inertia about the base origin
  f3 h;            // composite first moment  sum m c
  float m;         // composite mass
  float S[21];     // sum_f Bw_f^T Bw_f, lower triangle row-major (i*(i+1)/2 + j)
  f3 N, F;         // total bias wrench about the base origin
  float bz[6];     // sum_f Bw_f^T zdot_f
};

__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }



// One leg: forward kinematics, Newton-Euler bias, composite inertia, leg Cholesky, Schur
// contributions to the base, and the (unwhitened in the base part) contact rows of its toe.
// `leg` is a run-time index: the four legs share ONE copy of this code (the caller's leg loop is kept
// rolled so that a substep's instruction stream stays inside the instruction cache); only the mirror
// signs of the hip offsets differ between legs.
// `esel` >= 0 (8 lanes per env: two lanes share a leg): emit the rows of toe end point `esel` only, the lane's
// partner emits the other one; -1: both.
template <class SM>
__device__ __forceinline__ void leg_pass(int leg, const BaseKin& bk, const float* __restrict__ q, const float* __restrict__ qd,
                                         const float* __restrict__ tau, float dt, LegFactor& L, BaseAccum& acc,
                                         const SM& sm, unsigned& active_mask, const Ground& ground, int esel, bool tracing, unsigned& facets) {
  // facets (event trace, `tracing`: wave-uniform): ^= trace_point of every toe point in reach
  static_assert(REX_LEG_SX[0] == -1 && REX_LEG_SX[1] == -1 && REX_LEG_SX[2] == 1 && REX_LEG_SX[3] == 1, "leg mirror table");
  static_assert(REX_LEG_SY[0] == -1 && REX_LEG_SY[1] == 1 && REX_LEG_SY[2] == -1 && REX_LEG_SY[3] == 1, "leg mirror table");
  const float SX = leg < 2 ? -1.0f : 1.0f, SY = (leg & 1) ? 1.0f : -1.0f;
  const float HX = SX * (float)REX_HIP_X, HY = SY * (float)REX_HIP_Y, UY = SY * (float)REX_UPPER_Y;
  constexpr float KX = (float)REX_KNEE_X, KZ = (float)REX_KNEE_Z, CZ = (float)REX_LOWER_COM_Z, TZ = (float)REX_TOE_Z;
  const float M1 = (float)REX_SHOULDER_MASS * ground.leg_mass_scale, M2 = (float)REX_UPPER_MASS * ground.leg_mass_scale,
              M3 = (float)REX_LOWER_MASS * ground.leg_mass_scale;

  const float q1 = q[0], q2 = q[1], q3 = q[2];
  const float qd1 = qd[0], qd2 = qd[1], qd3 = qd[2];
  float s1, c1, s2, c2, s3, c3;
  sincos_fast(q1, s1, c1);
  sincos_fast(q2, s2, c2);
  sincos_fast(q3, s3, c3);
  const float s23 = s2 * c3 + c2 * s3, c23 = c2 * c3 - s2 * s3;

  // --- forward kinematics (rotations as world columns; positions relative to the base origin) ---
  const f3 x1 = bk.ex;                                  // R1 = R0 Rx(q1)
  const f3 y1 = c1 * bk.ey + s1 * bk.ez;
  const f3 z1 = c1 * bk.ez - s1 * bk.ey;
  const f3 x2 = c2 * x1 - s2 * z1, z2 = s2 * x1 + c2 * z1;       // R2 = R1 Ry(q2)
  const f3 x3 = c23 * x1 - s23 * z1, z3 = s23 * x1 + c23 * z1;   // R3 = R1 Ry(q2+q3)
  const f3 a1 = x1, a2 = y1;                            // joint axes (a3 == a2)
  const f3 o1 = HX * bk.ex + HY * bk.ey;
  const f3 o2 = o1 + UY * y1;
  const f3 d23 = KX * x2 + KZ * z2;
  const f3 o3 = o2 + d23;
  const f3 e3 = CZ * z3;
  const f3 cm3 = o3 + e3;
  const f3 d12 = UY * y1;

  // --- velocities ---
  const f3 w0 = bk.w;
  const f3 w1 = w0 + qd1 * a1;
  const f3 w2 = w1 + qd2 * a2;
  const f3 w3 = w2 + qd3 * a2;
  const f3 vo1 = bk.v + cross(w0, o1);
  const f3 vo2 = vo1 + cross(w1, d12);
  const f3 vo3 = vo2 + cross(w2, d23);
  const f3 vc3 = vo3 + cross(w3, e3);

  // --- Newton-Euler bias with zero generalized acceleration; gravity as +g on the base ---
  const f3 al1 = cross(w0, qd1 * a1);
  const f3 al2 = al1 + cross(w1, qd2 * a2);
  const f3 al3 = al2 + cross(w2, qd3 * a2);
  const f3 ao1 = mk(0.f, 0.f, kGravity) + cross(w0, cross(w0, o1));
  const f3 ao2 = ao1 + cross(al1, d12) + cross(w1, cross(w1, d12));
  const f3 ao3 = ao2 + cross(al2, d23) + cross(w2, cross(w2, d23));
  const f3 ac3 = ao3 + cross(al3, e3) + cross(w3, cross(w3, e3));

  const s33 I1 = rot_inertia(x1, y1, z1, (float)REX_SHOULDER_IXX, (float)REX_SHOULDER_IYY, (float)REX_SHOULDER_IZZ);
  const s33 I2 = rot_inertia(x2, y1, z2, (float)REX_UPPER_IXX, (float)REX_UPPER_IYY, (float)REX_UPPER_IZZ);
  const s33 I3 = rot_inertia(x3, y1, z3, (float)REX_LOWER_IXX, (float)REX_LOWER_IYY, (float)REX_LOWER_IZZ);

  const f3 Iw1 = mul(I1, w1), Iw2 = mul(I2, w2), Iw3 = mul(I3, w3);
  const float dl1 = kLinDamp + kLinDamp * sqrtf(dot(vo1, vo1));
  const float dl2 = kLinDamp + kLinDamp * sqrtf(dot(vo2, vo2));
  const float dl3 = kLinDamp + kLinDamp * sqrtf(dot(vc3, vc3));
  const float da1 = kAngDamp + kAngDamp * sqrtf(dot(w1, w1));
  const float da2 = kAngDamp + kAngDamp * sqrtf(dot(w2, w2));
  const float da3 = kAngDamp + kAngDamp * sqrtf(dot(w3, w3));
  const f3 f1 = M1 * ao1 + (M1 * dl1) * vo1;
  const f3 f2 = M2 * ao2 + (M2 * dl2) * vo2;
  const f3 f3_ = M3 * ac3 + (M3 * dl3) * vc3;
  const f3 n1 = mul(I1, al1) + cross(w1, Iw1) + da1 * Iw1;
  const f3 n2 = mul(I2, al2) + cross(w2, Iw2) + da2 * Iw2;
  const f3 n3 = mul(I3, al3) + cross(w3, Iw3) + da3 * Iw3;
  // subtree wrenches about the base origin
  const f3 F3 = f3_, N3 = n3 + cross(cm3, f3_);
  const f3 F2 = F3 + f2, N2 = N3 + n2 + cross(o2, f2);
  const f3 F1 = F2 + f1, N1 = N2 + n1 + cross(o1, f1);
  const float C1 = dot(a1, N1 - cross(o1, F1));
  const float C2 = dot(a2, N2 - cross(o2, F2));
  const float C3 = dot(a2, N3 - cross(o3, F3));
  acc.N = acc.N + N1;
  acc.F = acc.F + F1;

  // --- composite inertias about the base origin ---
  s33 Io3 = I3; add_point(Io3, M3, cm3);
  const f3 h3 = M3 * cm3;
  s33 Io2 = I2; add_point(Io2, M2, o2); add(Io2, Io3);
  const f3 h2 = h3 + M2 * o2;
  s33 Io1 = I1; add_point(Io1, M1, o1); add(Io1, Io2);
  const f3 h1 = h2 + M1 * o1;
  const float m3c = M3, m2c = M3 + M2, m1c = M3 + M2 + M1;
  add(acc.Io, Io1);
  acc.h = acc.h + h1;
  acc.m += m1c;

  // --- joint columns F_j = I^c_j S_j,  S_j = [a_j ; o_j x a_j] ---
  const f3 v1 = cross(o1, a1), v2 = cross(o2, a2), v3 = cross(o3, a2);
  const f3 Fl1 = m1c * v1 + cross(a1, h1), Fa1 = mul(Io1, a1) + cross(h1, v1);
  const f3 Fl2 = m2c * v2 + cross(a2, h2), Fa2 = mul(Io2, a2) + cross(h2, v2);
  const f3 Fl3 = m3c * v3 + cross(a2, h3), Fa3 = mul(Io3, a2) + cross(h3, v3);
  const float H11 = dot(a1, Fa1) + dot(v1, Fl1);
  const float H21 = dot(a1, Fa2) + dot(v1, Fl2);
  const float H31 = dot(a1, Fa3) + dot(v1, Fl3);
  const float H22 = dot(a2, Fa2) + dot(v2, Fl2);
  const float H32 = dot(a2, Fa3) + dot(v2, Fl3);
  const float H33 = dot(a2, Fa3) + dot(v3, Fl3);

  // --- G = chol(H) ---
  const float gi1 = rsqrtf(H11);
  const float g21 = H21 * gi1, g31 = H31 * gi1;
  const float p22 = H22 - g21 * g21;
  const float gi2 = rsqrtf(p22);
  const float g32 = (H32 - g31 * g21) * gi2;
  const float p33 = H33 - g31 * g31 - g32 * g32;
  const float gi3 = rsqrtf(p33);
  const float g11 = H11 * gi1, g22 = p22 * gi2, g33 = p33 * gi3;   // diag(G)
  L.gi1 = gi1; L.gi2 = gi2; L.gi3 = gi3; L.g21 = g21; L.g31 = g31; L.g32 = g32;

  // --- Bw = G^-1 B^T, columns k = (ang xyz, lin xyz) ---
  const float B1[6] = {Fa1.x, Fa1.y, Fa1.z, Fl1.x, Fl1.y, Fl1.z};
  const float B2[6] = {Fa2.x, Fa2.y, Fa2.z, Fl2.x, Fl2.y, Fl2.z};
  const float B3[6] = {Fa3.x, Fa3.y, Fa3.z, Fl3.x, Fl3.y, Fl3.z};
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float u1 = B1[k] * gi1;
    const float u2 = (B2[k] - g21 * u1) * gi2;
    const float u3 = (B3[k] - g31 * u1 - g32 * u2) * gi3;
    L.Bw[0][k] = u1; L.Bw[1][k] = u2; L.Bw[2][k] = u3;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j)
      acc.S[tri(i, j)] += L.Bw[0][i] * L.Bw[0][j] + L.Bw[1][i] * L.Bw[1][j] + L.Bw[2][i] * L.Bw[2][j];

  // --- free whitened acceleration of the leg and its predicted whitened velocity ---
  const float r1 = tau[0] - C1, r2 = tau[1] - C2, r3 = tau[2] - C3;
  const float zd1 = r1 * gi1;
  const float zd2 = (r2 - g21 * zd1) * gi2;
  const float zd3 = (r3 - g31 * zd1 - g32 * zd2) * gi3;
#pragma unroll
  for (int k = 0; k < 6; ++k) acc.bz[k] += L.Bw[0][k] * zd1 + L.Bw[1][k] * zd2 + L.Bw[2][k] * zd3;
  const float nu0[6] = {bk.w.x, bk.w.y, bk.w.z, bk.v.x, bk.v.y, bk.v.z};
  // z = G^T qd + Bw nu0
  float zc1 = g11 * qd1 + g21 * qd2 + g31 * qd3;
  float zc2 = g22 * qd2 + g32 * qd3;
  float zc3 = g33 * qd3;
#pragma unroll
  for (int k = 0; k < 6; ++k) { zc1 += L.Bw[0][k] * nu0[k]; zc2 += L.Bw[1][k] * nu0[k]; zc3 += L.Bw[2][k] * nu0[k]; }
  L.z[0] = zc1 + dt * zd1; L.z[1] = zc2 + dt * zd2; L.z[2] = zc3 + dt * zd3;

  // --- toe contact points: both ends of the toe cylinder against the ground ---
  const f3 tc = o3 + TZ * z3;
  const f3 aw = y1;
  // rows of one contact point P (relative to the base origin) with signed distance `dist` and frame (n, t1, t2)
  auto emit_rows = [&](int e, f3 P, float dist, f3 nrm, f3 t1, f3 t2) {
    const bool act = dist < kBreaking;
    const int p = 2 * leg + e;
    if (act) active_mask |= 1u << p;
    const f3 r1v = P - o1, r2v = P - o2, r3v = P - o3;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const f3 dir = d == 0 ? nrm : (d == 1 ? t1 : t2);
      const f3 Jw = cross(P, dir);
      const float Jq1 = dot(a1, cross(r1v, dir));
      const float Jq2 = dot(a2, cross(r2v, dir));
      const float Jq3 = dot(a2, cross(r3v, dir));
      const float j1 = Jq1 * gi1;
      const float j2 = (Jq2 - g21 * j1) * gi2;
      const float j3 = (Jq3 - g31 * j1 - g32 * j2) * gi3;
      const float Jb[6] = {Jw.x, Jw.y, Jw.z, dir.x, dir.y, dir.z};
      float g[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) g[k] = Jb[k] - (L.Bw[0][k] * j1 + L.Bw[1][k] * j2 + L.Bw[2][k] * j3);
      float target = 0.0f;
      if (d == 0) target = dist > 0.0f ? -dist / dt : -dist * (kErp / dt);
      const int r = d == 0 ? p : (REX_NPOINT + 2 * p + (d - 1));
      sm.row(r, 0) = make_float4(g[0], g[1], g[2], g[3]);
      sm.row(r, 1) = make_float4(g[4], g[5], j1, j2);
      sm.row(r, 2) = make_float4(j3, target, act ? 1.0f : 0.0f, 0.0f);
    }
  };
  if (ground.h == nullptr) {   // flat plane: constant frame, the compiler folds the crosses
    const f3 dv = mk(-aw.z * aw.x, -aw.z * aw.y, 1.0f - aw.z * aw.z);   // n - (n.a) a, n = +z
    const float dn2 = dot(dv, dv);
    const float inv = dn2 > 1e-18f ? rsqrtf(dn2) : 0.0f;
    if (esel >= 0) {           // wave-uniform: all lanes of a wave run the same lanes-per-env mode
      const float sg = esel == 0 ? -kToeHalf : kToeHalf;
      const f3 P = tc + sg * aw - (kToeRad * inv) * dv;
      emit_rows(esel, P, bk.height + P.z, mk(0.f, 0.f, 1.f), mk(0.f, -1.f, 0.f), mk(1.f, 0.f, 0.f));
    } else {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float sg = e == 0 ? -kToeHalf : kToeHalf;
        const f3 P = tc + sg * aw - (kToeRad * inv) * dv;
        emit_rows(e, P, bk.height + P.z, mk(0.f, 0.f, 1.f), mk(0.f, -1.f, 0.f), mk(1.f, 0.f, 0.f));   // btPlaneSpace1(+z)
      }
    }
  } else {                     // heightfield: normal under the end centre -> lowest point along it -> local plane
#pragma unroll 1
    for (int t = 0; t < (esel >= 0 ? 1 : 2); ++t) {
      const int e = esel >= 0 ? esel : t;    // esel >= 0: one trip, every lane with its own end point
      const float sg = e == 0 ? -kToeHalf : kToeHalf;
      const f3 ce = tc + sg * aw;
      f3 n0, nrm, t1, t2;
      float h0, h;
      unsigned fid0 = 0u, fid1 = 0u;
      ground_query(ground, bk.px + ce.x, bk.py + ce.y, h0, n0, fid0);
      const float na = dot(n0, aw);
      const f3 dv = n0 - na * aw;
      const float dn2 = dot(dv, dv);
      const float inv = dn2 > 1e-18f ? rsqrtf(dn2) : 0.0f;
      const f3 P = ce - (kToeRad * inv) * dv;
      ground_query(ground, bk.px + P.x, bk.py + P.y, h, nrm, fid1);
      plane_space(nrm, t1, t2);
      const float dist = (bk.height + P.z - h) * nrm.z;
      emit_rows(e, P, dist, nrm, t1, t2);
      // (bit 31: which branch btPlaneSpace1 took for the friction directions of this normal -- with a friction pyramid the
      //  tangent frame is part of the problem, and |n.z| crosses 0.7071 on a 45-degree facet)
      if (tracing && dist < kBreaking) facets ^= trace_point(2 * leg + e, fid0, fid1 | (fabsf(nrm.z) > 0.7071067811865475244f ? 0x80000000u : 0u));
    }
  }

  // --- link-box contact rows of this leg (RexConfig.body_contacts): shoulder, leg and foot boxes of rex.urdf:119-124,
  //     151-156,170-175 against the ground.  Candidates: the four corners of each box's ground-facing face; kept: the two
  //     deepest penetrating ones (Bullet's box-box detector reports penetrating points only, deepest first).  Same
  //     selection as oracle/rex_oracle.c.  A point on the shoulder moves with joint 1 only, on the leg link with joints
  //     1-2, on the foot link with all three.
  if constexpr (SM::kBody) {
    constexpr int B0 = 3;   // boxes 3..5 of rex_model_gen.h: the first leg's (all legs carry identical boxes)
    static_assert(REX_BOX_BODY[B0] == 1 && REX_BOX_BODY[B0 + 1] == 2 && REX_BOX_BODY[B0 + 2] == 3, "leg box table");
    const f3 bo[3] = {o1, o2, o3};
    const f3 bx[3] = {x1, x2, x3}, bz[3] = {z1, z2, z3};
    float lowest = 1e9f;
    float rch[3][3], sgn[3][3];
    f3 ctr[3], nb[3];
    float hb[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      ctr[b] = bo[b] + (float)REX_BOX_CENTER[B0 + b][2] * bz[b];
      nb[b] = mk(0.f, 0.f, 1.f); hb[b] = 0.0f;
      if (ground.h != nullptr) ground_query(ground, bk.px + ctr[b].x, bk.py + ctr[b].y, hb[b], nb[b]);
      const float en[3] = {dot(bx[b], nb[b]), dot(y1, nb[b]), dot(bz[b], nb[b])};
      float sum = 0.0f;
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        rch[b][ax] = (float)REX_BOX_HALF[B0 + b][ax] * fabsf(en[ax]);
        sgn[b][ax] = en[ax] > 0.0f ? -1.0f : 1.0f;
        sum += rch[b][ax];
      }
      lowest = fminf(lowest, (bk.height + ctr[b].z - hb[b]) * nb[b].z - sum);
    }
    // a box corner can only be below the ground if the box's lowest point along the normal under its centre is (on the
    // heightfield the ground under a corner may stand up to the field's roughness above that under the centre)
    const bool near = lowest < (ground.h != nullptr ? 0.06f : 0.0f);
    // self collision (URDF_USE_SELF_COLLISION, rex.py:276-281): this leg's leg-link and foot-link boxes against the three
    // boxes of the base body (same y / z extents, strung along the base's x axis).  Cheap reject first: the six face axes
    // of a pair as separating axes.
    constexpr float kAy = (float)REX_BOX_HALF[0][1], kAz = (float)REX_BOX_HALF[0][2];
    static_assert(REX_BOX_HALF[1][1] == REX_BOX_HALF[0][1] && REX_BOX_HALF[2][1] == REX_BOX_HALF[0][1] &&
                  REX_BOX_HALF[1][2] == REX_BOX_HALF[0][2] && REX_BOX_HALF[2][2] == REX_BOX_HALF[0][2] &&
                  REX_BOX_CENTER[1][1] == 0.0 && REX_BOX_CENTER[1][2] == 0.0 && REX_BOX_CENTER[2][1] == 0.0 && REX_BOX_CENTER[2][2] == 0.0,
                  "base body boxes: one y-z section, centres on the base's x axis");
    float Cm[2][3][3], dA0[2][3], dB0[2][3];   // per leg box: C[i][j] = base axis i . box axis j; centre offset in both frames
    unsigned pairnear = 0;   // bit 3 bb + a: leg box bb and base box a are not separated by a face axis
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
      const f3 ea[3] = {bk.ex, bk.ey, bk.ez}, eb[3] = {bx[1 + bb], y1, bz[1 + bb]};
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) Cm[bb][i][j] = dot(ea[i], eb[j]);
        dA0[bb][i] = dot(ctr[1 + bb], ea[i]);      // the base box centres sit at (cx, 0, 0): subtracted per box below
        dB0[bb][i] = dot(ctr[1 + bb], eb[i]);
      }
      const float hB[3] = {(float)REX_BOX_HALF[B0 + 1 + bb][0], (float)REX_BOX_HALF[B0 + 1 + bb][1], (float)REX_BOX_HALF[B0 + 1 + bb][2]};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float cx = (float)REX_BOX_CENTER[a][0], hA[3] = {(float)REX_BOX_HALF[a][0], kAy, kAz};
        float sep = -1e9f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const float di = dA0[bb][i] - (i == 0 ? cx : 0.0f);
          sep = fmaxf(sep, fabsf(di) - hA[i] - (hB[0] * fabsf(Cm[bb][i][0]) + hB[1] * fabsf(Cm[bb][i][1]) + hB[2] * fabsf(Cm[bb][i][2])));
          const float dj = dB0[bb][i] - cx * Cm[bb][0][i];
          sep = fmaxf(sep, fabsf(dj) - hB[i] - (hA[0] * fabsf(Cm[bb][0][i]) + hA[1] * fabsf(Cm[bb][1][i]) + hA[2] * fabsf(Cm[bb][2][i])));
        }
        if (sep < 2.0f * kSelfMargin) pairnear |= 1u << (3 * bb + a);   // (a corner within the margin of a box can be up to sqrt(3) margins away along the other box's axes)
      }
    }
    unsigned pairs = 0;      // wave-uniform: the pairs some env of the wave has to look at corner by corner
#pragma unroll
    for (int b6 = 0; b6 < 6; ++b6) if (__builtin_amdgcn_ballot_w64((pairnear >> b6) & 1u) != 0) pairs |= 1u << b6;
    const bool anyself = pairs != 0;
    if (__builtin_amdgcn_ballot_w64(near) != 0 || anyself) {
      float bestD[2] = {0.0f, 0.0f};
      f3 bestP[2] = {mk(0.f, 0.f, 0.f), mk(0.f, 0.f, 0.f)}, bestN[2] = {mk(0.f, 0.f, 1.f), mk(0.f, 0.f, 1.f)};
      int bestL[2] = {0, 0};   // 0 = empty slot, else 1 + box (= number of joints that move the point)
      bool bestS[2] = {false, false};   // the point is held against the base body (row of the relative velocity), not the ground
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        int fa = 0;
        if (rch[b][1] > rch[b][fa]) fa = 1;
        if (rch[b][2] > rch[b][fa]) fa = 2;
        const f3 ax3[3] = {bx[b], y1, bz[b]};
        const float hx = (float)REX_BOX_HALF[B0 + b][0], hy = (float)REX_BOX_HALF[B0 + b][1], hz = (float)REX_BOX_HALF[B0 + b][2];
#pragma unroll
        for (int cnr = 0; cnr < 4; ++cnr) {
          const float s1 = (cnr & 1) ? 1.0f : -1.0f, s2 = (cnr & 2) ? 1.0f : -1.0f;
          // local corner: face axis at its ground side, the other two axes (cyclic order) at +-1
          const float lx = fa == 0 ? sgn[b][0] : (fa == 1 ? s2 : s1);
          const float ly = fa == 1 ? sgn[b][1] : (fa == 2 ? s2 : s1);
          const float lz = fa == 2 ? sgn[b][2] : (fa == 0 ? s2 : s1);
          const f3 P = ctr[b] + (lx * hx) * ax3[0] + (ly * hy) * ax3[1] + (lz * hz) * ax3[2];
          f3 n = mk(0.f, 0.f, 1.f);
          float h = 0.0f;
          if (ground.h != nullptr) ground_query(ground, bk.px + P.x, bk.py + P.y, h, n);
          const float dist = (bk.height + P.z - h) * n.z;
          if (dist < 0.0f) {   // deepest-first list of two; ties keep the earlier candidate
            if (bestL[0] == 0 || dist < bestD[0]) {
              bestD[1] = bestD[0]; bestP[1] = bestP[0]; bestN[1] = bestN[0]; bestL[1] = bestL[0]; bestS[1] = bestS[0];
              bestD[0] = dist; bestP[0] = P; bestN[0] = n; bestL[0] = 1 + b; bestS[0] = false;
            } else if (bestL[1] == 0 || dist < bestD[1]) {
              bestD[1] = dist; bestP[1] = P; bestN[1] = n; bestL[1] = 1 + b; bestS[1] = false;
            }
          }
        }
      }
      if (anyself) {
        // The face cases of Bullet's box-box detector reduced to their penetrating vertices: a corner of one box inside
        // the other, pushed out through the face of least penetration of the box that contains it (oracle/rex_oracle.c,
        // same candidate order: base box, leg box, side, corner).  The candidates compete with the ground candidates above.
#pragma unroll 1
        for (int a = 0; a < 3; ++a) {
          const float cx = a == 0 ? (float)REX_BOX_CENTER[0][0] : (a == 1 ? (float)REX_BOX_CENTER[1][0] : (float)REX_BOX_CENTER[2][0]);
          const float hAx = a == 0 ? (float)REX_BOX_HALF[0][0] : (a == 1 ? (float)REX_BOX_HALF[1][0] : (float)REX_BOX_HALF[2][0]);
          const float hA[3] = {hAx, kAy, kAz};
          const f3 cA = cx * bk.ex;
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) {
            if (!((pairs >> (3 * bb + a)) & 1u)) continue;   // wave-uniform
            const float hB[3] = {(float)REX_BOX_HALF[B0 + 1 + bb][0], (float)REX_BOX_HALF[B0 + 1 + bb][1], (float)REX_BOX_HALF[B0 + 1 + bb][2]};
            const f3 ea[3] = {bk.ex, bk.ey, bk.ez}, eb[3] = {bx[1 + bb], y1, bz[1 + bb]};
            const float dA[3] = {dA0[bb][0] - cx, dA0[bb][1], dA0[bb][2]};
            const float dB[3] = {dB0[bb][0] - cx * Cm[bb][0][0], dB0[bb][1] - cx * Cm[bb][0][1], dB0[bb][2] - cx * Cm[bb][0][2]};
#pragma unroll
            for (int side = 0; side < 2; ++side) {
#pragma unroll 1
              for (int c = 0; c < 8; ++c) {
                const float sg[3] = {(c & 1) ? 1.0f : -1.0f, (c & 2) ? 1.0f : -1.0f, (c & 4) ? 1.0f : -1.0f};
                float in[3];
#pragma unroll
                for (int k = 0; k < 3; ++k)   // the corner in the frame of the box that may contain it
                  in[k] = side == 0 ? dA[k] + (sg[0] * hB[0] * Cm[bb][k][0] + sg[1] * hB[1] * Cm[bb][k][1] + sg[2] * hB[2] * Cm[bb][k][2])
                                    : -dB[k] + (sg[0] * hA[0] * Cm[bb][0][k] + sg[1] * hA[1] * Cm[bb][1][k] + sg[2] * hA[2] * Cm[bb][2][k]);
                const float d0 = (side == 0 ? hA[0] : hB[0]) - fabsf(in[0]), d1 = (side == 0 ? hA[1] : hB[1]) - fabsf(in[1]),
                            d2 = (side == 0 ? hA[2] : hB[2]) - fabsf(in[2]);
                if (!(d0 + kSelfMargin > 0.0f && d1 + kSelfMargin > 0.0f && d2 + kSelfMargin > 0.0f)) continue;
                int ax = 0; float depth = d0;
                if (d1 < depth) { ax = 1; depth = d1; }
                if (d2 < depth) { ax = 2; depth = d2; }
                const float inax = ax == 0 ? in[0] : (ax == 1 ? in[1] : in[2]);
                const float sgn = (inax >= 0.0f ? 1.0f : -1.0f) * (side == 0 ? 1.0f : -1.0f);   // the push on the LEG link
                const f3 axv = side == 0 ? (ax == 0 ? ea[0] : (ax == 1 ? ea[1] : ea[2])) : (ax == 0 ? eb[0] : (ax == 1 ? eb[1] : eb[2]));
                const f3 n = sgn * axv;
                const f3 P = side == 0 ? ctr[1 + bb] + (sg[0] * hB[0]) * eb[0] + (sg[1] * hB[1]) * eb[1] + (sg[2] * hB[2]) * eb[2]
                                       : cA + (sg[0] * hA[0]) * ea[0] + (sg[1] * hA[1]) * ea[1] + (sg[2] * hA[2]) * ea[2];
                const float dist = -depth;
                if (bestL[0] == 0 || dist < bestD[0]) {
                  bestD[1] = bestD[0]; bestP[1] = bestP[0]; bestN[1] = bestN[0]; bestL[1] = bestL[0]; bestS[1] = bestS[0];
                  bestD[0] = dist; bestP[0] = P; bestN[0] = n; bestL[0] = 2 + bb; bestS[0] = true;
                } else if (bestL[1] == 0 || dist < bestD[1]) {
                  bestD[1] = dist; bestP[1] = P; bestN[1] = n; bestL[1] = 2 + bb; bestS[1] = true;
                }
              }
            }
          }
        }
      }
      // bits 21 + leg / 25 + leg: the leg's first / second link-box slot holds a point (the solver skips empty slots wave by wave)
      if (bestL[0] != 0) active_mask |= 1u << (21 + leg);
      if (bestL[1] != 0) active_mask |= 1u << (25 + leg);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (esel >= 0 && k != esel) continue;   // 8 lanes per env: the two lanes of a leg take one slot each
        const int slot = 4 + 2 * leg + k;
        const bool act = bestL[k] != 0;
        const f3 P = bestP[k], nrm = bestN[k];
        const bool self = bestS[k];
        f3 t1 = mk(0.f, -1.f, 0.f), t2 = mk(1.f, 0.f, 0.f);
        if (ground.h != nullptr || anyself) plane_space(nrm, t1, t2);   // (+z gives exactly the constant pair)
        const f3 r1v = P - o1, r2v = P - o2, r3v = P - o3;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const f3 dir = d == 0 ? nrm : (d == 1 ? t1 : t2);
          const f3 Jb0 = cross(P, dir);
          const f3 Jw = self ? mk(0.f, 0.f, 0.f) : Jb0;                    // against the base body: the base part of the relative row vanishes
          // ... and the base's own Jacobian at the point goes next to the row, for the diagonal Bullet gives such a row
          sm.bjac(3 * (slot - 4) + d, 0) = make_float4(Jb0.x, Jb0.y, Jb0.z, dir.x);
          sm.bjac(3 * (slot - 4) + d, 1) = make_float4(dir.y, dir.z, act && self ? 1.0f : 0.0f, 0.0f);
          const float Jq1 = dot(a1, cross(r1v, dir));
          const float Jq2 = bestL[k] >= 2 ? dot(a2, cross(r2v, dir)) : 0.0f;
          const float Jq3 = bestL[k] >= 3 ? dot(a2, cross(r3v, dir)) : 0.0f;
          const float j1 = Jq1 * gi1;
          const float j2 = (Jq2 - g21 * j1) * gi2;
          const float j3 = (Jq3 - g31 * j1 - g32 * j2) * gi3;
          const float Jb[6] = {Jw.x, Jw.y, Jw.z, self ? 0.0f : dir.x, self ? 0.0f : dir.y, self ? 0.0f : dir.z};
          float g[6];
#pragma unroll
          for (int m = 0; m < 6; ++m) g[m] = act ? Jb[m] - (L.Bw[0][m] * j1 + L.Bw[1][m] * j2 + L.Bw[2][m] * j3) : 0.0f;
          // penetrating: position error through ERP; a self-collision candidate still on its way in: the closing speed bound
          const float target = d == 0 ? (bestD[k] > 0.0f ? -bestD[k] / dt : -bestD[k] * (kErp / dt)) : 0.0f;
          const int r = d == 0 ? slot : (REX_NBSLOT + 2 * slot + (d - 1));
          sm.brow(r, 0) = make_float4(g[0], g[1], g[2], g[3]);
          sm.brow(r, 1) = make_float4(g[4], g[5], act ? j1 : 0.0f, act ? j2 : 0.0f);
          // a friction row has no target: its .y carries the friction coefficient of the pair of surfaces instead (link against
          // link: the product of the URDF defaults); .w stays 0 -- the element lanes without a component of the row read
          const float c2y = d == 0 ? (act ? target : 0.0f) : (self ? kSelfMu : ground.mu);
          sm.brow(r, 2) = make_float4(act ? j3 : 0.0f, c2y, act ? 1.0f : 0.0f, 0.0f);
        }
      }
    }
  }

  // --- joint-limit rows (btMultiBodyJointLimitConstraint): the near bound of each joint, once reached
  //     (kLimitActivation); J = +-e_k on the leg's joints, so the whitened row is a column of G^-1 ---
  const float qv[3] = {q1, q2, q3};
  bool near_any = false;
#pragma unroll
  for (int k = 0; k < 3; ++k)
    near_any |= fminf(qv[k] - (float)REX_LEG_LIMIT_LO[k], (float)REX_LEG_LIMIT_HI[k] - qv[k]) <= kLimitActivation;
  if (__builtin_amdgcn_ballot_w64(near_any) == 0) return;   // no env of this wave has this leg near a bound
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float lo_gap = qv[k] - (float)REX_LEG_LIMIT_LO[k], hi_gap = (float)REX_LEG_LIMIT_HI[k] - qv[k];
    const bool lower = lo_gap < hi_gap;
    const float gap = lower ? lo_gap : hi_gap;
    const bool act = gap <= kLimitActivation;
    if (act) active_mask |= 1u << (REX_NPOINT + 3 * leg + k);
    const float sgn = lower ? 1.0f : -1.0f;
    const float j1 = k == 0 ? sgn * gi1 : 0.0f;
    const float j2 = k == 0 ? (-g21 * j1) * gi2 : (k == 1 ? sgn * gi2 : 0.0f);
    const float j3 = k == 2 ? sgn * gi3 : (-g31 * j1 - g32 * j2) * gi3;
    float g[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) g[m] = -(L.Bw[0][m] * j1 + L.Bw[1][m] * j2 + L.Bw[2][m] * j3);
    const float target = gap > 0.0f ? -gap / dt : -gap * (kErp / dt);
    const int r = REX_NCROW + 3 * leg + k;
    sm.row(r, 0) = make_float4(g[0], g[1], g[2], g[3]);
    sm.row(r, 1) = make_float4(g[4], g[5], j1, j2);
    sm.row(r, 2) = make_float4(j3, target, act ? 1.0f : 0.0f, 0.0f);
  }
}

// Lower-triangular 6x6 Cholesky factor, packed: off-diagonals + inverse diagonal
struct Chol6 {
  float l[15];   // strict lower triangle, row-major: (1,0) (2,0) (2,1) (3,0) ...
  float di[6];   // 1 / L_ii
  float d[6];    // L_ii
};
__device__ __forceinline__ int tri_s(int i, int j) { return i * (i - 1) / 2 + j; }  // i > j

__device__ __forceinline__ void chol6(const float* A /*21 lower*/, Chol6& C) {
  float Lf[6][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float s = A[tri(j, j)];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= Lf[j][k] * Lf[j][k];
    const float di = rsqrtf(s);
    C.di[j] = di;
    C.d[j] = s * di;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      float t = A[tri(i, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= Lf[i][k] * Lf[j][k];
      Lf[i][j] = t * di;
      C.l[tri_s(i, j)] = Lf[i][j];
    }
  }
}
// x = Lc^-1 b
__device__ __forceinline__ void fwd6(const Chol6& C, const float* b, float* x) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float s = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= C.l[tri_s(i, k)] * x[k];
    x[i] = s * C.di[i];
  }
}
// x = Lc^-T b
__device__ __forceinline__ void bwd6(const Chol6& C, const float* b, float* x) {
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    float s = b[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) s -= C.l[tri_s(k, i)] * x[k];
    x[i] = s * C.di[i];
  }
}
// x = Lc^T b
__device__ __forceinline__ void mulT6(const Chol6& C, const float* b, float* x) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float s = b[i] * C.d[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) s += C.l[tri_s(k, i)] * b[k];
    x[i] = s;
  }
}

struct PhysState {
  float pos[3], quat[4], lin[3], ang[3];
  // one env per lane: the 12 leg motors in motor order (+ 6 arm motors for mark='arm'; untouched and optimised away otherwise).
  // Lane groups: slots 0..2 hold the joints of the lane's OWN leg, 12..17 the arm's; 3..11 are never touched.
  float q[18], qd[18];
};

struct BaseAccum;
struct Chol6;
struct PgsX;
// Hook for an extra branch on the base (the arm of mark='arm', rex_arm_device.h).  NoArm = mark 'base'.
struct NoArm {
  static constexpr int NM = 12;
  static constexpr bool kHasRows = false;
  __device__ __forceinline__ void pass(const BaseKin&, PhysState&, const float*, float, BaseAccum&, const Ground&) {}
  __device__ __forceinline__ void finish(const Chol6&) {}
  __device__ __forceinline__ void sweep(PgsX&, float&) {}
  __device__ __forceinline__ void back(const float*, PhysState&) {}
  template <int LPE> __device__ __forceinline__ void dv_begin(int) {}
  template <int LPE, int NY> __device__ __forceinline__ void dv_sweep(float*, float&, float) {}
  template <int LPE> __device__ __forceinline__ void dv_end(int) {}
  __device__ __forceinline__ void dv_gather() {}
};

// whitened solver state: y = 3 packed pairs, per leg z = (pair, scalar)
struct PgsX {
  v2 y01, y23, y45;
  v2 z01[4];
  float z2[4];
};

// one constraint row: vel = Jt . x ; impulse step ; x += Jt dl.   Packed fp32 (v_pk_fma_f32) on the pairs.
// Row chunk 2 = (j2, invd * target, invd, diag): nl = lam + invd (target - vel) is evaluated as
// fma(-invd, vel, lam + invd*target) so that only one fma sits between vel and the clamp, and the
// 9-term dot runs as two independent packed accumulators (shorter dependent chain for a lone wave).
template <int LEG, bool FRICTION>
__device__ __forceinline__ void pgs_row(const float4 c0, const float4 c1, const float4 c2, PgsX& x, float& lam, float lim, float& worst) {
  const v2 a = {c0.x, c0.y}, b = {c0.z, c0.w}, c = {c1.x, c1.y}, d = {c1.z, c1.w};
  const v2 acc0 = __builtin_elementwise_fma(c, x.y45, a * x.y01);
  const v2 acc1 = __builtin_elementwise_fma(d, x.z01[LEG], b * x.y23);
  const float base = lam + c2.y;
  const v2 acc = acc0 + acc1;
  const float vel = fmaf(c2.x, x.z2[LEG], acc.x + acc.y);
  float nl = fmaf(-c2.z, vel, base);
  if (FRICTION) nl = __builtin_amdgcn_fmed3f(nl, -lim, lim);
  else nl = fmaxf(nl, 0.0f);
  const float dl = nl - lam;
  lam = nl;
  worst = fmaxf(worst, fabsf(dl * c2.w));   // velocity residual |dl / invdiag|
  const v2 dl2 = {dl, dl};
  x.y01 = __builtin_elementwise_fma(a, dl2, x.y01);
  x.y23 = __builtin_elementwise_fma(b, dl2, x.y23);
  x.y45 = __builtin_elementwise_fma(c, dl2, x.y45);
  x.z01[LEG] = __builtin_elementwise_fma(d, dl2, x.z01[LEG]);
  x.z2[LEG] = fmaf(c2.x, dl, x.z2[LEG]);
}

template <int LEG, class SM>
__device__ __forceinline__ void pgs_leg_normals(const SM& sm, PgsX& x, float* lam, float& worst) {
  const int p0 = 2 * LEG, p1 = 2 * LEG + 1;
  const float4 a0 = sm.row(p0, 0), a1 = sm.row(p0, 1), a2 = sm.row(p0, 2);
  const float4 b0 = sm.row(p1, 0), b1 = sm.row(p1, 1), b2 = sm.row(p1, 2);
  pgs_row<LEG, false>(a0, a1, a2, x, lam[p0], 0.0f, worst);
  pgs_row<LEG, false>(b0, b1, b2, x, lam[p1], 0.0f, worst);
}

template <int LEG, class SM>
__device__ __forceinline__ void pgs_leg_limits(const SM& sm, PgsX& x, float* lam, float& worst) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int r = REX_NCROW + 3 * LEG + k;
    const float4 a0 = sm.row(r, 0), a1 = sm.row(r, 1), a2 = sm.row(r, 2);
    pgs_row<LEG, false>(a0, a1, a2, x, lam[r], 0.0f, worst);
  }
}

template <int LEG, class SM>
__device__ __forceinline__ void pgs_leg_friction(const SM& sm, PgsX& x, float* lam, float& worst, float mu) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int p = 2 * LEG + e;
    const float lim = mu * lam[p];
    const int r0 = REX_NPOINT + 2 * p, r1 = r0 + 1;
    // a point that carries no normal impulse (and no friction impulse left from an earlier sweep) in ANY
    // lane can only produce zero friction steps this sweep: skip its two rows for the wavefront (exact)
    if (__builtin_amdgcn_ballot_w64(lim > 0.0f || lam[r0] != 0.0f || lam[r1] != 0.0f) == 0) continue;
    const float4 a0 = sm.row(r0, 0), a1 = sm.row(r0, 1), a2 = sm.row(r0, 2);
    const float4 b0 = sm.row(r1, 0), b1 = sm.row(r1, 1), b2 = sm.row(r1, 2);
    pgs_row<LEG, true>(a0, a1, a2, x, lam[r0], lim, worst);
    pgs_row<LEG, true>(b0, b1, b2, x, lam[r1], lim, worst);
  }
}

// ---- lanes-per-env helpers (EPW < 64) ----
// A wave that carries EPW <= 16 envs gives every env a GROUP of LPE adjacent lanes (LPE = 8 for EPW <= 8, 4 for
// EPW = 16; lane = LPE * slot + p; for EPW = 4 the upper 32 lanes repeat the lower 32) that run the same arithmetic on
// the same state (see rex_step_kernel).  The four legs of an env are independent until the base Cholesky, so a lane
// factorises ONE leg (leg p for LPE = 4, leg p / 2 for LPE = 8) and the per-leg partial sums meet in an xor butterfly
// inside the group (DPP, no LDS); the row finishing, the sweep loop (pgs_dv) and the back-substitution are split
// over the lanes as well.  Each butterfly step adds the same two numbers in both lanes, so the lanes of a group stay
// bit-identical.
__device__ __forceinline__ constexpr int lanes_per_env(int epw) { return epw <= 8 ? 8 : 4; }
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
constexpr int kDppXor1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;  // row_half_mirror: lane i of 8 reads lane 7 - i (the other quad)
constexpr int kDppShr1 = 0x111;        // row_shr:1: lane i of a row of 16 reads lane i - 1 (lane 0: 0)
// sum over the lanes of a group, every lane holding a distinct addend
template <int LPE>
__device__ __forceinline__ float group_sum(float v) {
  v += dpp_f<kDppXor1>(v);
  v += dpp_f<kDppXor2>(v);
  if (LPE == 8) v += dpp_f<kDppHalfMirror>(v);   // all lanes of a quad hold its total: any lane of the other quad will do
  return v;
}
// sum over the four LEGS of a group: with LPE = 8 lanes 2L and 2L+1 both hold leg L's addend
template <int LPE>
__device__ __forceinline__ float leg_sum(float v) {
  if (LPE == 4) v += dpp_f<kDppXor1>(v);
  v += dpp_f<kDppXor2>(v);
  if (LPE == 8) v += dpp_f<kDppHalfMirror>(v);
  return v;
}
// lane `own` (0..3, a constant after unrolling) of every quad hands its value to the quad: quad_perm [own, own, own, own]
__device__ __forceinline__ float quad_bcast(float v, int own) {
  return own == 0 ? dpp_f<0x00>(v) : (own == 1 ? dpp_f<0x55>(v) : (own == 2 ? dpp_f<0xAA>(v) : dpp_f<0xFF>(v)));
}
template <int LPE>
__device__ __forceinline__ unsigned leg_or(unsigned v) {   // over ALL lanes of the group (with LPE = 8 each lane of a leg sets its own point)
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, kDppXor1, 0xF, 0xF, true);
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, kDppXor2, 0xF, 0xF, true);
  if (LPE == 8) v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, kDppHalfMirror, 0xF, 0xF, true);
  return v;
}
template <int LPE>
__device__ __forceinline__ unsigned group_xor(unsigned v) {  // xor over the lanes of the group (event trace)
  v ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, kDppXor1, 0xF, 0xF, true);
  v ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, kDppXor2, 0xF, 0xF, true);
  if (LPE == 8) v ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, kDppHalfMirror, 0xF, 0xF, true);
  return v;
}
__device__ __forceinline__ float pick_leg(const float* a, int m, int j) {   // a[3 m + j] with a per-lane m, no scratch
  const float lo = m & 1 ? a[3 + j] : a[j], hi = m & 1 ? a[9 + j] : a[6 + j];
  return m & 2 ? hi : lo;
}
// A value held in an accumulation register across a region that needs every VGPR (the sweep loop): the compiler spills
// to AGPRs by itself, but by its own weights -- at 16 envs per wave it kept the substep's bystanders (the base Cholesky
// factor, the body state) in VGPRs and put a quarter of the loop's row slices into AGPRs, one v_accvgpr_read per use.
__device__ __forceinline__ float to_agpr(float v) { float a; asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v)); return a; }
__device__ __forceinline__ float from_agpr(float a) { float v; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a)); return v; }
__device__ __forceinline__ void hold(float& x) { x = to_agpr(x); }
__device__ __forceinline__ void take(float& x) { x = from_agpr(x); }
__device__ __forceinline__ void hold(uint32_t& x) { x = __builtin_bit_cast(uint32_t, to_agpr(__builtin_bit_cast(float, x))); }
__device__ __forceinline__ void take(uint32_t& x) { x = __builtin_bit_cast(uint32_t, from_agpr(__builtin_bit_cast(float, x))); }
__device__ __forceinline__ void hold(int32_t& x) { x = __builtin_bit_cast(int32_t, to_agpr(__builtin_bit_cast(float, x))); }
__device__ __forceinline__ void take(int32_t& x) { x = __builtin_bit_cast(int32_t, from_agpr(__builtin_bit_cast(float, x))); }
template <int N> __device__ __forceinline__ void hold(uint32_t (&x)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) hold(x[k]);
}
template <int N> __device__ __forceinline__ void take(uint32_t (&x)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) take(x[k]);
}
template <int N> __device__ __forceinline__ void hold(float (&x)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) x[k] = to_agpr(x[k]);
}
template <int N> __device__ __forceinline__ void take(float (&x)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) x[k] = from_agpr(x[k]);
}
// LDS rows / parked factors written by one lane of a group are read by the others: order the accesses of the wave
__device__ __forceinline__ void mirror_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }

// ---- the sweep loop of a lane group (EPW <= 16): velocity form with the whitened velocity DISTRIBUTED over the lanes ----
// Lane p of a group owns component p (+ LPE, ...) of the base part y and component p of every leg's part z_L (lanes
// that own none read a zero word of the row and never change).  A row is then: one or two FMAs over the lane's components
// of J~, a DPP sum over the group, the clamp (every lane, redundantly) and that many FMAs for x += J~ dl -- 13 instructions
// in the contact loop against 21 + three 16-byte LDS reads per row when every lane carries the whole of x, and it takes
// any row (contact, joint limit, arm limit) alike.  A lane's slice of the 24 contact rows is read from LDS once per substep.
// The 24 contact rows run software-pipelined: the group sum of row r+1 is taken over x as it stands BEFORE row r is
// solved and corrected by A(r+1, r) dl_r (cpl[]: one 9-term inner product per row and substep, out of the registers of
// the lane that finishes the rows, physics_substep), which leaves fma - fma - clamp - subtract on the dependent chain.
// Joint-limit rows come first, as in Bullet, behind one wave-uniform test (out of line: no env of a wave has a bound
// in reach in 97-100 % of the substeps).  Contact rows out of reach have invd = 0, produce zero impulses and are not
// skipped: a static row sequence is what lets the pipeline run.  What an instruction costs a lone wave here, and the
// formulations that were measured and dropped: DESIGN.md section 5, profiles/r03_microbench.md.
#ifdef REX_PROF
__device__ long long g_prof[10 * 1024];  // per block: cycle counters of the sections of physics_substep (+ [8] whole kernel, [9] launches)
__device__ long long g_prof2[16 * 1024]; // per block: inside pgs_dv: [0] set-up, [1] sweep loop, [2] hand-back; of the kernel: [3] load + command,
                                         // [4] substeps, [5] epilogue, [6] the 100 MHz counter over the kernel, [7] start of the last launch;
                                         // inside the sweeps: [8] link-box rows run, [9] joint-limit rows run, cycles of [10] the limit rows,
                                         // [11] link-box normals, [12] toe rows, [13] link-box friction rows, [14] sweeps of thread 0's env
                                         // (what [8]-[13] cover)
__device__ unsigned g_legmask[65536];   // per env: bits 0-7 the toe points in reach in the last substep, bits 8-15 their OR since the last read-out
#define REX_STAMP(var) const long long var = clock64()
#else
#define REX_STAMP(var)
#endif
// per-variant code shape of the sweep loop (pgs_dv), chosen by measurement (tools/ab_libs.sh)
/* mark 'arm' at 16 envs per wave (every register taken, 300-600 B of scratch): the row keeps invd * target and the sweep
   set-up divides it out again row by row (a branch and an LDS round trip each) -- slower set-up, fewer values in flight */
#ifndef REX_TARGET_BY_DIVISION
#define REX_TARGET_BY_DIVISION(EPW, ARM) ((EPW) == 16 && (ARM))
#endif
#ifndef REX_FINISH_UNROLL
#define REX_FINISH_UNROLL(EPW, ARM, BLOCK) ((EPW) == 16 && (ARM) ? 1 : (BLOCK))   /* mark 'arm' at 16 envs per wave spills: rolled there */
#endif
/* the substep's bystanders (base Cholesky factor, body state, the env's words) parked in AGPRs around the sweeps by hand
   (to_agpr): where the loop's row slices need every VGPR -- 4 lanes per env, two base components per lane */
#ifndef REX_HOLD_ACROSS_SWEEPS
#define REX_HOLD_ACROSS_SWEEPS(EPW, ARM, BODY, MIXED) ((EPW) == 16 && !(ARM))
#endif
/* contact impulses alternating between two register sets, sweeps in pairs: not where it measured slower (link-box rows,
   mixed tasks with their per-lane sweep cap, mark 'arm' at 16 envs per wave) */
/* the base Cholesky factor parked in the free target words of the friction rows during the sweeps (physics_substep) */
#ifndef REX_PARK_IN_ROWS
#define REX_PARK_IN_ROWS(EPW, ARM) ((EPW) == 16 && (ARM) ? 7 : 0)   /* bit 0: the factor's triangle, 1: its diagonal, 2: the body state */
#endif
/* joint-limit impulses held by one lane of the group each (pgs_dv): where every register counts */
#ifndef REX_LIMIT_IMPULSE_BY_LANE
#define REX_LIMIT_IMPULSE_BY_LANE(EPW, ARM) ((EPW) == 16 && (ARM))
#endif
#ifndef REX_PAIRED_SWEEPS
#define REX_PAIRED_SWEEPS(EPW, ARM, BODY, MIXED) (!(BODY) && !(MIXED) && ((EPW) <= 8 || !(ARM)))
#endif
__device__ __forceinline__ constexpr int crow_leg(int r) { return r < REX_NPOINT ? r / 2 : (r - REX_NPOINT) / 4; }
__device__ __forceinline__ int crow_leg_rt(int r) { return r < REX_NPOINT ? r >> 1 : (r - REX_NPOINT) >> 2; }   // (a lane's own row index)

template <int NY, int EPW>
struct DvLane {
  int oy[NY], oz;   // byte offsets (from the LDS base) of this lane's components in row 0
  int oti;          // (LDS-resident rows: of the row's (tgt, invd) pair)
  template <class SM>
  __device__ __forceinline__ static float ld(const SM& sm, int off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(sm.p) + off);
  }
};

// The LDS-resident rows of the sweep -- link-box contact rows, joint-limit rows -- are plain Gauss-Seidel steps on a lane's slice of
// the row: its component(s) of the base part, its component of the leg part, (invd * target | friction coefficient, invd).  Word
// order of such a row (physics_substep, row finishing): g0..g5, z0 z1 z2, 0, tgt, invd -- the pair (tgt, invd) is one 8-byte read,
// word 9 is the zero the lanes without a component of their own read (the toe rows keep theirs in word 11).
// What such a row costs a lone wave is its INSTRUCTION COUNT, ~5 cycles each, as long as no read is waited for:
//  * round 4: the three reads right in front of their use 170 cycles a row, one row ahead 91 (tools/microbench/body_rows.hip);
//  * round 5, measured in the step kernel with per-section counters (tools/prof_sections.py, profiles/r05_sections_poses.txt): the
//    rows of a held roll pose still cost 160-230 cycles each, because every GROUP of rows (a leg's three limit rows, a slot pair's
//    rows) began with a read nobody had issued ahead -- one LDS latency per group and sweep; running the rows software-pipelined like
//    the toe rows (sum of row k + 1 before row k's step, couplings A(k + 1, k) next to the rows) made them SLOWER (two more
//    instructions a row: the chain was never the cost), and so did running all 48 rows as static zero-padded runs (110 cycles a row,
//    but 48 rows in every wave that has one).  So: the groups stay conditional, and the first two rows of group g + 1 are read before
//    group g is solved (RunHead, unconditionally: a read of rows nobody wrote is harmless and cheaper than a branch).
template <int NY> struct RowSlice { float jy[NY]; float jz; float tgt, invd; };   // tgt: invd * target (normal, limit) or the slot's friction coefficient
template <int NY> struct RunHead { RowSlice<NY> a, b; };                          // the first two rows of a group, read ahead
// The reads of a later row are issued, then this: nothing may be scheduled across it.  Without it the machine scheduler sinks the
// reads back to their first use, behind the current row's group sum (round 5: the ISA of the round-4 form had every row's
// ds_reads right in front of its own DPP adds again, and the step kernel measured no faster than with the reads at their use).
__device__ __forceinline__ void prefetch_fence() { __builtin_amdgcn_sched_barrier(0); }
// this lane's slice of the row whose chunk 0 sits `rowoff` bytes into LDS; ln: the lane's word offsets (DvLane, zero word 9)
template <int NY, class SM, class LN>
__device__ __forceinline__ RowSlice<NY> slice_load(const SM& sm, const LN& ln, int rowoff, bool leg_part) {
  RowSlice<NY> s;
#pragma unroll
  for (int i = 0; i < NY; ++i) s.jy[i] = ln.ld(sm, rowoff + ln.oy[i]);
  s.jz = leg_part ? ln.ld(sm, rowoff + ln.oz) : 0.0f;
  const float2 ti = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(sm.p) + rowoff + ln.oti);
  s.tgt = ti.x; s.invd = ti.y;
  return s;
}
template <int LPE, int NY>
__device__ __forceinline__ float slice_step(const RowSlice<NY>& s, int leg, float* ys, float* zs, float& lam, float lim, bool friction, float& worst,
                                            float thr) {
  float part = leg >= 0 ? s.jz * zs[leg >= 0 ? leg : 0] : 0.0f;
#pragma unroll
  for (int i = 0; i < NY; ++i) part = fmaf(s.jy[i], ys[i], part);
  const float vel = group_sum<LPE>(part);
  float nl = fmaf(-s.invd, vel, friction ? lam : lam + s.tgt);
  nl = friction ? __builtin_amdgcn_fmed3f(nl, -lim, lim) : fmaxf(nl, 0.0f);
  const float dl = nl - lam;
  lam = nl;
  worst = fmaxf(worst, fmaf(-thr, s.invd, fabsf(dl)));
#pragma unroll
  for (int i = 0; i < NY; ++i) ys[i] = fmaf(s.jy[i], dl, ys[i]);
  if (leg >= 0) zs[leg >= 0 ? leg : 0] = fmaf(s.jz, dl, zs[leg >= 0 ? leg : 0]);
  return dl;
}
// rows 0 .. N-1 of a group, straight-line; rows 0 and 1 were read ahead (`head`), row k + 2 is read while row k is solved.
// FRICTION: rows 2 j, 2 j + 1 are the friction pair of the point whose normal impulse is lamn[j]; the pair's coefficient sits in .tgt
// of its first row.  leg < 0: base-group rows (no leg part).
template <int LPE, int NY, int N, bool FRICTION, class LOAD>
__device__ __forceinline__ void solve_run(const RunHead<NY>& head, const LOAD& load, int leg, float* ys, float* zs, float* lam, const float* lamn,
                                          float& worst, float thr) {
  RowSlice<NY> s0 = head.a, s1 = head.b;
  float coef = 0.0f;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    RowSlice<NY> s2 = s1;
    if (k + 2 < N) { s2 = load(k + 2); prefetch_fence(); }
    if (FRICTION && (k & 1) == 0) coef = s0.tgt;
    slice_step<LPE, NY>(s0, leg, ys, zs, lam[k], FRICTION ? coef * lamn[k >> 1] : 0.0f, FRICTION, worst, thr);
    s0 = s1; s1 = s2;
  }
}

// `cpl_free`: called once the couplings have been read into registers -- their six LDS chunks are idle until the next
// substep's row finishing, and physics_substep parks bystanders of the sweep loop there where registers are short
template <int LPE, bool LANECAP, class SM, class ARMP, class F>
__device__ __forceinline__ void pgs_dv(const SM& sm, ARMP& armp, PgsX& x, int p, const bool (&lim)[4], bool any_contact,
                                       unsigned bgroups, float mu, int iterations, int lane_iterations, float thr, int& nsweeps, int& lane_sweeps,
                                       const F& cpl_free) {
  constexpr int EPW = SM::kEpw, NY = (6 + LPE - 1) / LPE;
  constexpr int kRow = REX_ROW_F4 * EPW * 16;   // bytes from a row to the next
  constexpr bool kPairedSweeps = REX_PAIRED_SWEEPS(EPW, ARMP::NM > 12, SM::kBody, LANECAP);
  REX_STAMP(t_dv0);
  DvLane<NY, EPW> ln;   // toe rows: the lanes without a component read the zero in word 11
  DvLane<NY, EPW> lr;   // LDS-resident rows (joint limits, link boxes): zero word 9, (tgt, invd) in words 10-11
  int okt;                  // byte offset of the lane's target word in row 0
  int opair = 0, ozc = 0;   // pair layout: of the lane's (y_p, y_(p+4)) word and of its leg component in a contact row
  constexpr bool kPairLayout = REX_PAIR_LAYOUT(EPW, ARMP::NM > 12);
  static_assert(!kPairLayout || (LPE == 4 && NY == 2), "pair layout: 4 lanes per env");
  float ys[NY], zs[REX_NLEG];
  {
    const float yv[6] = {x.y01.x, x.y01.y, x.y23.x, x.y23.y, x.y45.x, x.y45.y};
    sm.park(REX_PARK_XY) = make_float4(yv[0], yv[1], yv[2], yv[3]);
    sm.park(REX_PARK_XY + 1) = make_float4(yv[4], yv[5], 0.0f, 0.0f);
    mirror_sync();
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const int k = p + i * LPE;
      const int f = k < 6 ? k : 11;
      ln.oy[i] = ((f >> 2) * EPW + sm.slot) * 16 + (f & 3) * 4;
      lr.oy[i] = k < 6 ? ln.oy[i] : (2 * EPW + sm.slot) * 16 + 4;
      ys[i] = sm.parkf(REX_PARK_XY, k < 6 ? k : 6);
    }
    const int f = p < 3 ? 6 + p : 11;
    ln.oz = ((f >> 2) * EPW + sm.slot) * 16 + (f & 3) * 4;
    lr.oz = p < 3 ? ln.oz : (2 * EPW + sm.slot) * 16 + 4;
    ln.oti = lr.oti = (2 * EPW + sm.slot) * 16 + 8;
    okt = (2 * EPW + sm.slot) * 16 + (p == 0 ? 1 : 3) * 4;
    if constexpr (kPairLayout) {   // contact rows: (g0 g4 g1 g5) (g2 0 g3 0) (z0 z1 z2 -target); word 5 is a zero
      opair = ((p >> 1) * EPW + sm.slot) * 16 + (p & 1) * 8;
      ozc = ((p < 3 ? 2 : 1) * EPW + sm.slot) * 16 + (p < 3 ? p : 1) * 4;
      okt = ((p == 0 ? 2 : 1) * EPW + sm.slot) * 16 + (p == 0 ? 3 : 1) * 4;
    }
#pragma unroll
    for (int l = 0; l < REX_NLEG; ++l) zs[l] = sm.zf(l, p < 3 ? p : 3);
  }
  armp.template dv_begin<LPE>(p);
  float cpl[REX_NCROW];
#pragma unroll
  for (int c = 0; c < REX_NCROW / 4; ++c) {
    const float4 v = sm.park(REX_PARK_CPL + c);
    cpl[4 * c] = v.x; cpl[4 * c + 1] = v.y; cpl[4 * c + 2] = v.z; cpl[4 * c + 3] = v.w;
  }
  cpl_free();
  constexpr bool kLimLamByLane = REX_LIMIT_IMPULSE_BY_LANE(EPW, ARMP::NM > 12);
  static_assert(!kLimLamByLane || LPE == 4, "limit impulses by lane: 4 lanes per env");
  float lam[kLimLamByLane ? REX_NCROW : REX_NROW];
#pragma unroll
  for (int r = 0; r < (kLimLamByLane ? REX_NCROW : REX_NROW); ++r) lam[r] = 0.0f;
  float laml[3] = {0.0f, 0.0f, 0.0f};
  float lamb[SM::kBody ? REX_NBROW : 1];
  int bodyoff = 0;
  if constexpr (SM::kBody) {
#pragma unroll
    for (int r = 0; r < REX_NBROW; ++r) lamb[r] = 0.0f;
    bodyoff = (int)(reinterpret_cast<const char*>(sm.pb) - reinterpret_cast<const char*>(sm.p));
  }
  // this lane's slice of the 24 contact rows, read once per substep: its components of J~, invd, and -target in lane 0
  // of the group (0 elsewhere): the addend of the lane's first product, so that the group sum is vel - target and the
  // impulse step is one fma, nl = lam - invd (vel - target)
  // (only the 8 normal rows have a target -- emit_rows gives the friction rows none: 16 registers and LDS reads less)
  float Jy[REX_NCROW][NY], Jz[REX_NCROW], Kt[REX_NPOINT], Ki[REX_NCROW];
#pragma unroll
  for (int r = 0; r < REX_NCROW; ++r) {
    if constexpr (kPairLayout) {
      const float2 v = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(sm.p) + r * kRow + opair);
      Jy[r][0] = v.x; Jy[r][NY - 1] = v.y;
      Jz[r] = ln.ld(sm, r * kRow + ozc);
    } else {
#pragma unroll
      for (int i = 0; i < NY; ++i) Jy[r][i] = ln.ld(sm, r * kRow + ln.oy[i]);
      Jz[r] = ln.ld(sm, r * kRow + ln.oz);
      Ki[r] = sm.rowf(r, 10);
    }
    if (r < REX_NPOINT) {
      if constexpr (REX_TARGET_BY_DIVISION(EPW, ARMP::NM > 12)) Kt[r] = (p == 0 && Ki[r] > 0.0f) ? -sm.rowf(r, 9) * __builtin_amdgcn_rcpf(Ki[r]) : 0.0f;
      else Kt[r] = ln.ld(sm, r * kRow + okt);   // -target (0 for a row out of reach): lane 0; the others read a zero word of the row
    }
  }
  if constexpr (kPairLayout) {
#pragma unroll
    for (int c = 0; c < REX_NCROW / 4; ++c) {
      const float4 v = sm.park(REX_PARK_KI + c);
      Ki[4 * c] = v.x; Ki[4 * c + 1] = v.y; Ki[4 * c + 2] = v.z; Ki[4 * c + 3] = v.w;
    }
  }
  REX_STAMP(t_dv1);
  const bool any_lim = lim[0] || lim[1] || lim[2] || lim[3];
  bool running = true;
  auto lim_load = [&](int r) __attribute__((always_inline)) { return slice_load<NY>(sm, lr, r * kRow, true); };   // joint-limit row r (rows REX_NCROW.. of the toe region)
  auto body_load = [&](int r, bool leg_part) __attribute__((always_inline)) { return slice_load<NY>(sm, lr, bodyoff + r * kRow, leg_part); };
  (void)body_load;
  auto sweep = [&](const auto& li, auto& lo, int it) __attribute__((always_inline)) {
    ++nsweeps;
    // (`worst` lives outside the lane mask: a lane that has stopped keeps 0, and `running` is then one compare under the full EXEC --
    //  the wave's exit test reads that very mask instead of rebuilding a full-wave vote from a partial one)
    float worst = 0.0f;
    if (running) {
      ++lane_sweeps;
      REX_STAMP(t_s0);
      // joint-limit rows (non-contact rows come first in Bullet's sweep): plain Gauss-Seidel steps.  One test for all four
      // legs first: a sweep without a bound in reach (nearly all of them) then takes one branch instead of four
      if constexpr (SM::kBody) {
        // the link-box kernels (a held pose keeps joints on their bounds sweep after sweep): the first two rows of leg l + 1 are read
        // before leg l's rows are solved
        if (__builtin_expect(any_lim, 0)) {
          RunHead<NY> h[2];
          h[0].a = lim_load(REX_NCROW); h[0].b = lim_load(REX_NCROW + 1);
#pragma unroll
          for (int l = 0; l < REX_NLEG; ++l) {
            if (l + 1 < REX_NLEG) { h[(l + 1) & 1].a = lim_load(REX_NCROW + 3 * l + 3); h[(l + 1) & 1].b = lim_load(REX_NCROW + 3 * l + 4); }
            prefetch_fence();
            if (!lim[l]) continue;                   // wave-uniform
            solve_run<LPE, NY, 3, false>(h[l & 1], [&](int k) __attribute__((always_inline)) { return lim_load(REX_NCROW + 3 * l + k); }, l, ys, zs,
                                         lam + REX_NCROW + 3 * l, lam, worst, thr);
          }
        }
      } else {
#pragma unroll
        for (int l = 0; l < REX_NLEG; ++l) {
          if (__builtin_expect(!any_lim, 1)) break;  // wave-uniform
          if (!lim[l]) continue;                     // wave-uniform
          // (the three rows of a leg: the slice of row k + 1 is read from LDS while row k is solved)
          RowSlice<NY> cur = lim_load(REX_NCROW + 3 * l);
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int r = REX_NCROW + 3 * l + k;
            RowSlice<NY> nxt = cur;
            if (k + 1 < 3) { nxt = lim_load(r + 1); prefetch_fence(); }
            if constexpr (kLimLamByLane) {
              // the impulse of limit row 3 l + k lives in lane (3 l + k) % 4 of the group only (three registers instead of
              // twelve): every lane evaluates the step on the register of that index, the owner's result goes round (one DPP
              // broadcast inside the quad), the owner keeps the new impulse.  Two more instructions on a row that is rarely run.
              const int own = (3 * l + k) & 3, idx = (3 * l + k) >> 2;   // (constants once the loops are unrolled)
              float part = cur.jz * zs[l];
#pragma unroll
              for (int i = 0; i < NY; ++i) part = fmaf(cur.jy[i], ys[i], part);
              const float vel = group_sum<LPE>(part);
              const float nlo = fmaxf(fmaf(-cur.invd, vel, laml[idx] + cur.tgt), 0.0f);
              const float dl = quad_bcast(nlo - laml[idx], own);
              laml[idx] = p == own ? nlo : laml[idx];
              worst = fmaxf(worst, fmaf(-thr, cur.invd, fabsf(dl)));
#pragma unroll
              for (int i = 0; i < NY; ++i) ys[i] = fmaf(cur.jy[i], dl, ys[i]);
              zs[l] = fmaf(cur.jz, dl, zs[l]);
            } else {
              slice_step<LPE, NY>(cur, l, ys, zs, lam[r], 0.0f, false, worst, thr);   // (in place, also when the contact rows alternate)
            }
            cur = nxt;
          }
        }
      }
      armp.template dv_sweep<LPE, NY>(ys, worst, thr);
      REX_STAMP(t_s1);
      if constexpr (SM::kBody) {
        // link-box normals (among the normals they come before the toe points: the toe rows stay one pipelined block): one branch for
        // the lot, one per GROUP (wave-uniform) -- base group (slots 0-3), then leg L's slots 4 + 2 L, 5 + 2 L -- and the first two rows of
        // the next group read before a group is solved
        if (bgroups != 0) {
          RunHead<NY> h[2];
          h[0].a = body_load(0, false); h[0].b = body_load(1, false);
          h[1].a = body_load(4, true); h[1].b = body_load(5, true);
          prefetch_fence();
          if (bgroups & 1u)
            solve_run<LPE, NY, 4, false>(h[0], [&](int k) __attribute__((always_inline)) { return body_load(k, false); }, -1, ys, zs, lamb, lamb, worst, thr);
#pragma unroll
          for (int l = 0; l < REX_NLEG; ++l) {
            if (l + 1 < REX_NLEG) { h[l & 1].a = body_load(6 + 2 * l, true); h[l & 1].b = body_load(7 + 2 * l, true); }
            prefetch_fence();
            if (!((bgroups >> (1 + l)) & 1u)) continue;       // (a second slot is never filled before the first)
            // (one branch for the second slot, each side straight-line)
            if ((bgroups >> (5 + l)) & 1u)
              solve_run<LPE, NY, 2, false>(h[(l + 1) & 1], [&](int k) __attribute__((always_inline)) { return body_load(4 + 2 * l + k, true); }, l, ys, zs,
                                           lamb + 4 + 2 * l, lamb, worst, thr);
            else
              solve_run<LPE, NY, 1, false>(h[(l + 1) & 1], [&](int k) __attribute__((always_inline)) { return body_load(4 + 2 * l + k, true); }, l, ys, zs,
                                           lamb + 4 + 2 * l, lamb, worst, thr);
          }
        }
      }
      REX_STAMP(t_s2);
      RunHead<NY> hf;   // link-box friction: the first two rows of the base group, read behind the toe rows
      if constexpr (SM::kBody) {
        if (bgroups != 0) { hf.a = body_load(REX_NBSLOT, false); hf.b = body_load(REX_NBSLOT + 1, false); prefetch_fence(); }
      }
      if (any_contact) {
        // contact rows, pipelined; this lane's slice of the rows sits in registers (Jy, Jz, Kc, Ki)
        float S, dlp = 0.0f;
        {
          float part = fmaf(Jz[0], zs[0], Kt[0]);
#pragma unroll
          for (int i = 0; i < NY; ++i) part = fmaf(Jy[0][i], ys[i], part);
          S = group_sum<LPE>(part);
        }
#pragma unroll
        for (int r = 0; r < REX_NCROW; ++r) {
          const int L = crow_leg(r);
          const float sum = fmaf(cpl[r], dlp, S);
          float nl = fmaf(-Ki[r], sum, li[r]);
          if (r < REX_NPOINT) nl = fmaxf(nl, 0.0f);
          else {
            const float lm = mu * lo[(r - REX_NPOINT) / 2];
            nl = __builtin_amdgcn_fmed3f(nl, -lm, lm);
          }
          const float dl = nl - li[r];
          if (r + 1 < REX_NCROW) {   // group sum of the next row over x as it stands before this row's step
            float part = r + 1 < REX_NPOINT ? fmaf(Jz[r + 1], zs[crow_leg(r + 1)], Kt[r + 1]) : Jz[r + 1] * zs[crow_leg(r + 1)];
#pragma unroll
            for (int i = 0; i < NY; ++i) part = fmaf(Jy[r + 1][i], ys[i], part);
            S = group_sum<LPE>(part);
          }
          worst = fmaxf(worst, fmaf(-thr, Ki[r], fabsf(dl)));   // |dl| / invd > thr: Bullet's velocity residual
          lo[r] = nl;
          dlp = dl;
#pragma unroll
          for (int i = 0; i < NY; ++i) ys[i] = fmaf(Jy[r][i], dl, ys[i]);
          zs[L] = fmaf(Jz[r], dl, zs[L]);
        }
      }
      REX_STAMP(t_s3);
      if constexpr (SM::kBody) {
        // link-box friction pairs: after the toe friction rows; the coefficient of a slot sits in its first friction row
        // (ground: the env's foot friction; link against link: kSelfMu); the friction rows of a group are contiguous: rows
        // 12 + 2 slot and 13 + 2 slot
        if (bgroups != 0) {
          RunHead<NY> h[2];
          h[0] = hf;
          h[1].a = body_load(REX_NBSLOT + 8, true); h[1].b = body_load(REX_NBSLOT + 9, true);
          prefetch_fence();
          if (bgroups & 1u)
            solve_run<LPE, NY, 8, true>(h[0], [&](int k) __attribute__((always_inline)) { return body_load(REX_NBSLOT + k, false); }, -1, ys, zs,
                                        lamb + REX_NBSLOT, lamb, worst, thr);
#pragma unroll
          for (int l = 0; l < REX_NLEG; ++l) {
            const int r0 = REX_NBSLOT + 2 * (4 + 2 * l);                                   // first friction row of the leg's first slot
            if (l + 1 < REX_NLEG) { h[l & 1].a = body_load(r0 + 4, true); h[l & 1].b = body_load(r0 + 5, true); }
            prefetch_fence();
            if (!((bgroups >> (1 + l)) & 1u)) continue;
            if ((bgroups >> (5 + l)) & 1u)
              solve_run<LPE, NY, 4, true>(h[(l + 1) & 1], [&](int k) __attribute__((always_inline)) { return body_load(r0 + k, true); }, l, ys, zs, lamb + r0,
                                          lamb + 4 + 2 * l, worst, thr);
            else
              solve_run<LPE, NY, 2, true>(h[(l + 1) & 1], [&](int k) __attribute__((always_inline)) { return body_load(r0 + k, true); }, l, ys, zs, lamb + r0,
                                          lamb + 4 + 2 * l, worst, thr);
          }
        }
      }
#ifdef REX_PROF
      if (threadIdx.x == 0 && blockIdx.x < 1024) {
        long long* p2 = g_prof2 + 16 * blockIdx.x;
        int nb = 0, nl = 0;
        if constexpr (SM::kBody) {
          nb = (bgroups & 1u) ? 12 : 0;
          for (int l = 0; l < REX_NLEG; ++l) nb += ((bgroups >> (1 + l)) & 1u) ? (((bgroups >> (5 + l)) & 1u) ? 6 : 3) : 0;
        }
        if (any_lim) for (int l = 0; l < REX_NLEG; ++l) nl += lim[l] ? 3 : 0;
        p2[8] += nb; p2[9] += nl;
        const long long t_s4 = clock64();
        p2[10] += t_s1 - t_s0; p2[11] += t_s2 - t_s1; p2[12] += t_s3 - t_s2; p2[13] += t_s4 - t_s3; p2[14] += 1;
      }
#endif
    }             // (a lane that has stopped never sweeps again: what it leaves in `lo` is never read)
    // (the per-lane cap is compiled in for mixed-task batches only)
    running = LANECAP ? (worst > 0.0f && it + 1 < lane_iterations) : worst > 0.0f;
  };
  if constexpr (kPairedSweeps) {
    float lamB[REX_NCROW];
    for (int it = 0; it < iterations; it += 2) {
      sweep(lam, lamB, it);
      if (__builtin_amdgcn_ballot_w64(running) == 0 || it + 1 >= iterations) break;
      sweep(lamB, lam, it + 1);
      if (__builtin_amdgcn_ballot_w64(running) == 0) break;
    }
  } else {
    for (int it = 0; it < iterations; ++it) {
      sweep(lam, lam, it);
      if (__builtin_amdgcn_ballot_w64(running) == 0) break;
    }
  }
  REX_STAMP(t_dv2);
  // hand the components back: every lane needs the whole of x for the back-substitution
#pragma unroll
  for (int i = 0; i < NY; ++i) { const int k = p + i * LPE; sm.parkf(REX_PARK_XY, k < 6 ? k : 7) = ys[i]; }
#pragma unroll
  for (int l = 0; l < REX_NLEG; ++l) sm.zf(l, p < 3 ? p : 3) = zs[l];
  armp.template dv_end<LPE>(p);
  mirror_sync();
  {
    const float4 a = sm.park(REX_PARK_XY), b = sm.park(REX_PARK_XY + 1);
    x.y01 = v2{a.x, a.y}; x.y23 = v2{a.z, a.w}; x.y45 = v2{b.x, b.y};
#pragma unroll
    for (int l = 0; l < REX_NLEG; ++l) { const float4 z = sm.zc(l); x.z01[l] = v2{z.x, z.y}; x.z2[l] = z.z; }
  }
  armp.dv_gather();
#ifdef REX_PROF
  if (threadIdx.x == 0 && blockIdx.x < 1024) {
    long long* p2 = g_prof2 + 16 * blockIdx.x;
    p2[0] += t_dv1 - t_dv0; p2[1] += t_dv2 - t_dv1; p2[2] += clock64() - t_dv2;
  }
#endif
}

// The restated pybullet.stepSimulation for one env: tau is held for this substep (one env per lane: the 12 leg torques in
// motor order; lane groups: the 3 torques of the lane's own leg, then the arm's 6 for mark 'arm').
template <class T>
__device__ __forceinline__ void rotate_leg(T* a) {   // 12-entry per-joint array: leg k+1 moves into leg k's slots
  const T t0 = a[0], t1 = a[1], t2 = a[2];
#pragma unroll
  for (int j = 0; j < 9; ++j) a[j] = a[j + 3];
  a[9] = t0; a[10] = t1; a[11] = t2;
}

template <bool LANECAP, bool TRACE, class SM, class ARMP>
__device__ __forceinline__ void physics_substep(PhysState& s, float* tau, float dt, int iterations, int lane_iterations,
                                                float sqrt_res_thr, const SM& sm, const Ground& ground, ARMP& armp, int& lane_sweeps,
                                                unsigned* trace = nullptr, int trace_n = 0, int env = 0, bool live = false) {
  // TRACE (the kernel instantiations launched while rex_set_event_trace is on; the product kernels carry none of it): `trace` =
  // the event trace buffer [3][trace_n]; `live`: this lane stores env's words
  // lane_sweeps: += the solver sweeps THIS env ran (the host regroups large batches by it, rex_regroup_kernel)
  // `iterations`: wave-uniform sweep cap; `lane_iterations` <= iterations: this env's own cap (they differ only in a batch
  // that mixes tasks with different numSolverIterations, REX_TASK_MIXED)
  REX_STAMP(t_begin);
  // base rotation (btMatrix3x3::setRotation)
  BaseKin bk;
  {
    const float x = s.quat[0], y = s.quat[1], z = s.quat[2], w = s.quat[3];
    const float d = x * x + y * y + z * z + w * w, sc = 2.0f / d;
    const float xs = x * sc, ys = y * sc, zs = z * sc;
    const float wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
    bk.ex = mk(1.0f - (yy + zz), xy + wz, xz - wy);
    bk.ey = mk(xy - wz, 1.0f - (xx + zz), yz + wx);
    bk.ez = mk(xz + wy, yz - wx, 1.0f - (xx + yy));
  }
  bk.w = mk(s.ang[0], s.ang[1], s.ang[2]);
  bk.v = mk(s.lin[0], s.lin[1], s.lin[2]);
  bk.px = s.pos[0]; bk.py = s.pos[1];
  bk.height = s.pos[2];

  BaseAccum acc;
  acc.Io = rot_inertia(bk.ex, bk.ey, bk.ez, (float)REX_BASE_IXX, (float)REX_BASE_IYY, (float)REX_BASE_IZZ);
  acc.h = mk(0.f, 0.f, 0.f);
  const float mbase = (float)REX_BASE_MASS * ground.base_mass_scale;
  acc.m = mbase;
#pragma unroll
  for (int k = 0; k < 21; ++k) acc.S[k] = 0.0f;
#pragma unroll
  for (int k = 0; k < 6; ++k) acc.bz[k] = 0.0f;
  {
    const f3 Iw = mul(acc.Io, bk.w);
    const float dl = kLinDamp + kLinDamp * sqrtf(dot(bk.v, bk.v));
    const float da = kAngDamp + kAngDamp * sqrtf(dot(bk.w, bk.w));
    acc.N = cross(bk.w, Iw) + da * Iw;
    acc.F = mk(0.f, 0.f, mbase * kGravity) + (mbase * dl) * bk.v;
  }
  acc.m += ground.anchor; acc.Io.xx += ground.anchor; acc.Io.yy += ground.anchor; acc.Io.zz += ground.anchor;
  // the extra branch on the base (mark='arm'; its 6 torques; no-op otherwise) goes FIRST: its recursion is the point of
  // highest register pressure of the substep, and before the legs neither their factors nor their velocities are live yet
  armp.pass(bk, s, tau + (SM::kEpw <= 16 ? 3 : 12), dt, acc, ground);

  unsigned active = 0, facets = 0;
  PgsX x;
  constexpr int EPW = SM::kEpw;
  constexpr bool kSplitLegs = EPW <= 16;          // group layout: LPE lanes per env
  constexpr int LPE = lanes_per_env(EPW);
  const int pl = kSplitLegs ? (int)(threadIdx.x & (unsigned)(LPE - 1)) : 0;   // lane of the group
  const int mleg = LPE == 8 ? pl >> 1 : pl;                                   // the leg this lane factorises
  LegFactor Lown;   // lane group: the factor of this lane's leg (in registers or parked: REX_LEG_F4_OF)
  if constexpr (kSplitLegs) {
    // lane groups: a lane carries the joint state of ITS leg only (slots 0..2 of q / qd; PhysState)
    const float ql[3] = {s.q[0], s.q[1], s.q[2]};
    const float qdl[3] = {s.qd[0], s.qd[1], s.qd[2]};
    const float tl[3] = {tau[0], tau[1], tau[2]};   // lane groups: `tau` holds the torques of the lane's own leg (then the arm's)
    BaseAccum part;
    part.Io = s33{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    part.h = part.N = part.F = mk(0.f, 0.f, 0.f);
    part.m = 0.0f;
#pragma unroll
    for (int k = 0; k < 21; ++k) part.S[k] = 0.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) part.bz[k] = 0.0f;
    leg_pass(mleg, bk, ql, qdl, tl, dt, Lown, part, sm, active, ground, LPE == 8 ? (pl & 1) : -1, TRACE, facets);
    if constexpr (SM::kLegF4 == 1) sm.zc(mleg) = make_float4(Lown.z[0], Lown.z[1], Lown.z[2], 0.0f);
    else leg_park(sm, mleg, Lown);
    active = leg_or<LPE>(active);
    acc.Io.xx += leg_sum<LPE>(part.Io.xx); acc.Io.yy += leg_sum<LPE>(part.Io.yy); acc.Io.zz += leg_sum<LPE>(part.Io.zz);
    acc.Io.xy += leg_sum<LPE>(part.Io.xy); acc.Io.xz += leg_sum<LPE>(part.Io.xz); acc.Io.yz += leg_sum<LPE>(part.Io.yz);
    acc.h = acc.h + mk(leg_sum<LPE>(part.h.x), leg_sum<LPE>(part.h.y), leg_sum<LPE>(part.h.z));
    acc.N = acc.N + mk(leg_sum<LPE>(part.N.x), leg_sum<LPE>(part.N.y), leg_sum<LPE>(part.N.z));
    acc.F = acc.F + mk(leg_sum<LPE>(part.F.x), leg_sum<LPE>(part.F.y), leg_sum<LPE>(part.F.z));
    acc.m += leg_sum<LPE>(part.m);
#pragma unroll
    for (int k = 0; k < 21; ++k) acc.S[k] += leg_sum<LPE>(part.S[k]);
#pragma unroll
    for (int k = 0; k < 6; ++k) acc.bz[k] += leg_sum<LPE>(part.bz[k]);
    mirror_sync();
  } else {
#pragma unroll 1
    for (int leg = 0; leg < REX_NLEG; ++leg) {
      // the current leg always sits in slots 0..2: q / qd / tau are rotated by one leg per iteration,
      // which keeps every register index static inside the rolled loop
      LegFactor L;
      leg_pass(leg, bk, s.q, s.qd, tau, dt, L, acc, sm, active, ground, -1, TRACE, facets);
      leg_park(sm, leg, L);
      rotate_leg(s.q); rotate_leg(s.qd); rotate_leg(tau);
    }
  }
#pragma unroll
  for (int k = 0; k < REX_NLEG; ++k) {
    const float4 zc = sm.zc(k);
    x.z01[k] = v2{zc.x, zc.y}; x.z2[k] = zc.z;
  }
  REX_STAMP(t_legs);
  if constexpr (TRACE) {   // the substep's discrete events (rex_set_event_trace): one read-modify-write of the env's word, debug runs only
    if constexpr (kSplitLegs) facets = group_xor<LPE>(facets);
    unsigned arm_mask = 0;
    if constexpr (ARMP::NM > 12) {
#pragma unroll
      for (int k = 0; k < 6; ++k)
        if (fminf(s.q[12 + k] - (float)REXA_LOWER[k], (float)REXA_UPPER[k] - s.q[12 + k]) <= kLimitActivation) arm_mask |= 1u << k;
    }
    if (live) {
      trace[env] = trace_mix(trace_mix(trace_mix(trace[env], active & 0xFFFFFu), facets), arm_mask);
      trace[2 * trace_n + env] = trace_mix(trace_mix(trace[2 * trace_n + env], active & 0xFFFFFu), facets);   // the same without the arm's bounds
    }
  }
  if constexpr (SM::kBody) {
    // link-box contact rows of the base group: base_link and the two chassis boxes (rex.urdf:15-33,63-108), the four
    // deepest penetrating corners of their ground-facing faces.  Every lane of the group computes (and writes) the same rows.
    static_assert(kSplitLegs, "link-box contact rows need the lane-group layout (their rows do not fit 64 envs per workgroup in LDS)");
    static_assert(REX_BOX_BODY[0] == 0 && REX_BOX_BODY[1] == 0 && REX_BOX_BODY[2] == 0 && REX_BOX_BODY[3] == 1, "base box table");
    float lowest = 1e9f;
    float rch[3][3], sgn[3][3];
    f3 ctr[3], nb[3];
    float hb[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      ctr[b] = (float)REX_BOX_CENTER[b][0] * bk.ex + (float)REX_BOX_CENTER[b][1] * bk.ey + (float)REX_BOX_CENTER[b][2] * bk.ez;
      nb[b] = mk(0.f, 0.f, 1.f); hb[b] = 0.0f;
      if (ground.h != nullptr) ground_query(ground, bk.px + ctr[b].x, bk.py + ctr[b].y, hb[b], nb[b]);
      const float en[3] = {dot(bk.ex, nb[b]), dot(bk.ey, nb[b]), dot(bk.ez, nb[b])};
      float sum = 0.0f;
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        rch[b][ax] = (float)REX_BOX_HALF[b][ax] * fabsf(en[ax]);
        sgn[b][ax] = en[ax] > 0.0f ? -1.0f : 1.0f;
        sum += rch[b][ax];
      }
      lowest = fminf(lowest, (bk.height + ctr[b].z - hb[b]) * nb[b].z - sum);
    }
    const bool near = lowest < (ground.h != nullptr ? 0.06f : 0.0f);
    if (__builtin_amdgcn_ballot_w64(near) != 0) {
      active |= 1u << 20;
      float bestD[4] = {0.f, 0.f, 0.f, 0.f};
      f3 bestP[4], bestN[4];
      bool bestA[4] = {false, false, false, false};
#pragma unroll
      for (int k = 0; k < 4; ++k) { bestP[k] = mk(0.f, 0.f, 0.f); bestN[k] = mk(0.f, 0.f, 1.f); }
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        int fa = 0;
        if (rch[b][1] > rch[b][fa]) fa = 1;
        if (rch[b][2] > rch[b][fa]) fa = 2;
        const float hx = (float)REX_BOX_HALF[b][0], hy = (float)REX_BOX_HALF[b][1], hz = (float)REX_BOX_HALF[b][2];
#pragma unroll
        for (int cnr = 0; cnr < 4; ++cnr) {
          const float s1 = (cnr & 1) ? 1.0f : -1.0f, s2 = (cnr & 2) ? 1.0f : -1.0f;
          const float lx = fa == 0 ? sgn[b][0] : (fa == 1 ? s2 : s1);
          const float ly = fa == 1 ? sgn[b][1] : (fa == 2 ? s2 : s1);
          const float lz = fa == 2 ? sgn[b][2] : (fa == 0 ? s2 : s1);
          const f3 P = ctr[b] + (lx * hx) * bk.ex + (ly * hy) * bk.ey + (lz * hz) * bk.ez;
          f3 n = mk(0.f, 0.f, 1.f);
          float h = 0.0f;
          if (ground.h != nullptr) ground_query(ground, bk.px + P.x, bk.py + P.y, h, n);
          const float dist = (bk.height + P.z - h) * n.z;
          if (dist < 0.0f) {   // insertion into the deepest-first list of four (ties keep the earlier candidate)
            float dcur = dist; f3 pcur = P, ncur = n; bool acur = true;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const bool take = acur && (!bestA[k] || dcur < bestD[k]);
              const float dt_ = bestD[k]; const f3 pt = bestP[k], nt = bestN[k]; const bool at = bestA[k];
              if (take) { bestD[k] = dcur; bestP[k] = pcur; bestN[k] = ncur; bestA[k] = true; dcur = dt_; pcur = pt; ncur = nt; acur = at; }
            }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool act = bestA[k];
        const f3 P = bestP[k], nrm = bestN[k];
        f3 t1 = mk(0.f, -1.f, 0.f), t2 = mk(1.f, 0.f, 0.f);
        if (ground.h != nullptr) plane_space(nrm, t1, t2);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const f3 dir = d == 0 ? nrm : (d == 1 ? t1 : t2);
          const f3 Jw = cross(P, dir);
          const float target = d == 0 ? -bestD[k] * (kErp / dt) : 0.0f;
          const int r = d == 0 ? k : (REX_NBSLOT + 2 * k + (d - 1));
          sm.brow(r, 0) = act ? make_float4(Jw.x, Jw.y, Jw.z, dir.x) : make_float4(0.f, 0.f, 0.f, 0.f);
          sm.brow(r, 1) = act ? make_float4(dir.y, dir.z, 0.0f, 0.0f) : make_float4(0.f, 0.f, 0.f, 0.f);
          sm.brow(r, 2) = make_float4(0.0f, d == 0 ? (act ? target : 0.0f) : ground.mu, act ? 1.0f : 0.0f, 0.0f);   // friction rows: .y = the coefficient
        }
      }
    }
  }

  // base articulated inertia A = [[Io, hx],[hx^T, m]] - S, then A = Lc Lc^T
  float A[21];
  {
    const s33& I = acc.Io; const f3 h = acc.h;
    const float full[6][6] = {
        {I.xx, I.xy, I.xz, 0.f, -h.z, h.y},
        {I.xy, I.yy, I.yz, h.z, 0.f, -h.x},
        {I.xz, I.yz, I.zz, -h.y, h.x, 0.f},
        {0.f, h.z, -h.y, acc.m, 0.f, 0.f},
        {-h.z, 0.f, h.x, 0.f, acc.m, 0.f},
        {h.y, -h.x, 0.f, 0.f, 0.f, acc.m}};
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) A[tri(i, j)] = full[i][j] - acc.S[tri(i, j)];
  }
  Chol6 Lc;
  chol6(A, Lc);

  // predicted whitened base velocity  y = Lc^T nu0 + dt Lc^-1 (F_b - sum Bw^T zdot)
  float y[6];
  {
    const float nu0[6] = {bk.w.x, bk.w.y, bk.w.z, bk.v.x, bk.v.y, bk.v.z};
    const float rhs[6] = {-acc.N.x - acc.bz[0], -acc.N.y - acc.bz[1], -acc.N.z - acc.bz[2],
                          -acc.F.x - acc.bz[3], -acc.F.y - acc.bz[4], -acc.F.z - acc.bz[5]};
    float yd[6];
    fwd6(Lc, rhs, yd);
    mulT6(Lc, nu0, y);
#pragma unroll
    for (int k = 0; k < 6; ++k) y[k] += dt * yd[k];
  }

  REX_STAMP(t_chol);
  // finish the rows: whiten the base part, inverse diagonal (0 disables an inactive point)
  armp.finish(Lc);
  const bool any_limit = __builtin_amdgcn_ballot_w64(((active >> REX_NPOINT) & 0xFFFu) != 0) != 0;   // bits 8..19: the 12 leg joints
  constexpr bool kPairLayout = REX_PAIR_LAYOUT(SM::kEpw, ARMP::NM > 12);
  auto finish_row = [&](int r, float (&gw)[6], float4& c1, float4& c2, bool contact_dv) __attribute__((always_inline)) {
    const float4 c0 = sm.row(r, 0);
    c1 = sm.row(r, 1); c2 = sm.row(r, 2);
    const float g[6] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y};
    fwd6(Lc, g, gw);
    const float diag = gw[0] * gw[0] + gw[1] * gw[1] + gw[2] * gw[2] + gw[3] * gw[3] + gw[4] * gw[4] + gw[5] * gw[5] +
                       c1.z * c1.z + c1.w * c1.w + c2.x * c2.x;
    const float invd = c2.z != 0.0f ? __builtin_amdgcn_rcpf(diag) : 0.0f;
    if (kPairLayout && contact_dv) {
      // 4 lanes per env, lane p owns y_p and y_(p+4): the two meet in one 8-byte word of the row -- a lane's slice of the row
      // is born as a register pair (what the packed multiply-adds of the sweep take; read as two words the compiler pairs
      // the loads by row and copies every value into a second register).  (g0 g4 g1 g5) (g2 0 g3 0) (z0 z1 z2 -target)
      sm.row(r, 0) = make_float4(gw[0], gw[4], gw[1], gw[5]);
      sm.row(r, 1) = make_float4(gw[2], 0.0f, gw[3], 0.0f);
      sm.row(r, 2) = make_float4(c1.z, c1.w, c2.x, c2.z != 0.0f ? -c2.y : 0.0f);
      sm.parkf(REX_PARK_KI, r) = invd;
      return;
    }
    sm.row(r, 0) = make_float4(gw[0], gw[1], gw[2], gw[3]);
    sm.row(r, 1) = make_float4(gw[4], gw[5], c1.z, c1.w);
    // .y: what the sweep adds to a row's velocity: invd * target for the rows solved one by one (joint limits; all rows of
    // the one-env-per-lane layout), the plain -target (0 when the row is out of reach) for the pipelined contact rows of a
    // lane group (pgs_dv: the group sum is then vel - target); .w = 0 there: the word of the lanes that own no component
    c2.y = contact_dv && !REX_TARGET_BY_DIVISION(SM::kEpw, ARMP::NM > 12) ? (c2.z != 0.0f ? -c2.y : 0.0f) : c2.y * invd;
    // (lane groups, joint-limit rows: read from LDS row by row in the sweep -- (tgt, invd) as one aligned pair, the zero in word 9)
    if (kSplitLegs && !contact_dv) sm.row(r, 2) = make_float4(c2.x, 0.0f, c2.y, invd);
    else sm.row(r, 2) = make_float4(c2.x, c2.y, invd, kSplitLegs ? 0.0f : diag);
  };
  if constexpr (kSplitLegs) {
    // lane p finishes the contact rows [kBlock p, kBlock (p + 1)): consecutive rows meet in one lane, and the couplings of
    // consecutive rows A(r, r-1) = J~_r . J~_(r-1) the pipelined sweep needs (pgs_dv) come out of its registers; the row
    // before a block's first comes from the lane below (DPP row_shr:1; row 0 has no predecessor: lane 0 gets zeros or
    // another env's row, and writes 0)
    constexpr int kBlock = REX_NCROW / LPE;
    float fg[6], fz[3];       // the block's first row: whitened base part, leg part
    float pg[6], pz[3];       // the row before the current one
    int pleg = -1;
    constexpr int kUnroll = REX_FINISH_UNROLL(EPW, ARMP::NM > 12, kBlock);
#pragma unroll kUnroll
    for (int k = 0; k < kBlock; ++k) {
      const int r = kBlock * pl + k;
      float gw[6]; float4 c1, c2;
      finish_row(r, gw, c1, c2, true);
      const float z[3] = {c1.z, c1.w, c2.x};
      if (k > 0) {
        float cp = crow_leg_rt(r) == pleg ? z[0] * pz[0] + z[1] * pz[1] + z[2] * pz[2] : 0.0f;
#pragma unroll
        for (int q = 0; q < 6; ++q) cp = fmaf(gw[q], pg[q], cp);
        sm.parkf(REX_PARK_CPL, r) = cp;
      } else {
#pragma unroll
        for (int q = 0; q < 6; ++q) fg[q] = gw[q];
        fz[0] = z[0]; fz[1] = z[1]; fz[2] = z[2];
      }
#pragma unroll
      for (int q = 0; q < 6; ++q) pg[q] = gw[q];
      pz[0] = z[0]; pz[1] = z[1]; pz[2] = z[2];
      pleg = crow_leg_rt(r);
    }
    {
      // the block's last row goes to the lane above (the DPP moves are wave-wide: every lane shifts)
      const int r0 = kBlock * pl;
      // (every shift outside any condition, the leg test as a factor: inside `same leg ? ... : 0` the compiler branches, and
      // a DPP move executed under a lane mask reads 0 from the lanes the mask has switched off)
      const float nz0 = dpp_f<kDppShr1>(pz[0]), nz1 = dpp_f<kDppShr1>(pz[1]), nz2 = dpp_f<kDppShr1>(pz[2]);
      const float same = crow_leg_rt(r0) == dpp_i<kDppShr1>(pleg) ? 1.0f : 0.0f;
      float cp = same * (fz[0] * nz0 + fz[1] * nz1 + fz[2] * nz2);
#pragma unroll
      for (int q = 0; q < 6; ++q) cp = fmaf(fg[q], dpp_f<kDppShr1>(pg[q]), cp);
      sm.parkf(REX_PARK_CPL, r0) = pl == 0 ? 0.0f : cp;
    }
    if (any_limit) {
      for (int r = REX_NCROW + pl; r < REX_NROW; r += LPE) { float gw[6]; float4 c1, c2; finish_row(r, gw, c1, c2, false); }
    }
  } else {
    for (int r = 0; r < (any_limit ? REX_NROW : REX_NCROW); ++r) { float gw[6]; float4 c1, c2; finish_row(r, gw, c1, c2, false); }
  }

  // bit 0: the base group has rows in reach of some env of the wave; bit 1 + leg / 5 + leg: the leg's first / second slot holds a
  // point in some env of the wave
  unsigned bgroups = 0;
  if constexpr (SM::kBody) {
#pragma unroll
    for (int g = 0; g < 9; ++g) if (__builtin_amdgcn_ballot_w64((active >> (20 + g)) & 1u) != 0) bgroups |= 1u << g;
    if (bgroups != 0) {
      mirror_sync();
      for (int r = pl; r < REX_NBROW; r += LPE) {   // lane p: rows p, p + LPE, ...
        const int slot = r < REX_NBSLOT ? r : (r - REX_NBSLOT) >> 1;
        const int g = slot < 4 ? 0 : 1 + ((slot - 4) >> 1) + 4 * ((slot - 4) & 1);
        if (!((bgroups >> g) & 1u)) continue;
        float4 c0 = sm.brow(r, 0), c1 = sm.brow(r, 1), c2 = sm.brow(r, 2);
        const float gq[6] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y};
        float gw[6];
        fwd6(Lc, gq, gw);
        float diag = gw[0] * gw[0] + gw[1] * gw[1] + gw[2] * gw[2] + gw[3] * gw[3] + gw[4] * gw[4] + gw[5] * gw[5] +
                     c1.z * c1.z + c1.w * c1.w + c2.x * c2.x;
        if (slot >= 4) {
          // a point held against the base body: btMultiBodyConstraintSolver::setupMultiBodyContactConstraint sums the two
          // links' own terms J_A M^-1 J_A^T + J_B M^-1 J_B^T and leaves their cross term out, also when both links belong to
          // one multibody.  With a = whitened link row, b = whitened base row and the relative row g = a - b stored:
          // |a|^2 + |b|^2 = |g|^2 + 2 g.b + 2 |b|^2
          const int i3 = 3 * (slot - 4) + (r < REX_NBSLOT ? 0 : 1 + ((r - REX_NBSLOT) & 1));
          const float4 j0 = sm.bjac(i3, 0), j1 = sm.bjac(i3, 1);
          const float jb[6] = {j0.x, j0.y, j0.z, j0.w, j1.x, j1.y};
          float bw[6];
          fwd6(Lc, jb, bw);
          float gb = 0.0f, bb = 0.0f;
#pragma unroll
          for (int k = 0; k < 6; ++k) { gb = fmaf(gw[k], bw[k], gb); bb = fmaf(bw[k], bw[k], bb); }
          if (j1.z != 0.0f) diag += 2.0f * (gb + bb);
        }
        const float invd = c2.z != 0.0f ? __builtin_amdgcn_rcpf(diag) : 0.0f;
        sm.brow(r, 0) = make_float4(gw[0], gw[1], gw[2], gw[3]);
        sm.brow(r, 1) = make_float4(gw[4], gw[5], c1.z, c1.w);
        sm.brow(r, 2) = make_float4(c2.x, 0.0f, r < REX_NBSLOT ? c2.y * invd : c2.y, invd);   // (z2, 0, tgt, invd); friction rows keep their coefficient as tgt
      }
    }
  }
  if constexpr (kSplitLegs) mirror_sync();
  // projected Gauss-Seidel in Bullet's order: all normals, then all friction rows; a point that no
  // lane of the wavefront has within the breaking distance is skipped for the whole wavefront
  // (its rows could only ever produce zero impulses).
  float lam[REX_NROW];
#pragma unroll
  for (int r = 0; r < REX_NROW; ++r) lam[r] = 0.0f;
#ifdef REX_PROF
  // (census: the env's own toe points in reach this substep, one bit per point, and the wave's union per leg -- tools/leg_census.py)
  if (live && env < 65536) { const unsigned g = g_legmask[env]; g_legmask[env] = (active & 0xFFu) | ((((g >> 8) | active) & 0xFFu) << 8); }
#endif
  const bool any0 = __builtin_amdgcn_ballot_w64((active & 0x03u) != 0) != 0;
  const bool any1 = __builtin_amdgcn_ballot_w64((active & 0x0Cu) != 0) != 0;
  const bool any2 = __builtin_amdgcn_ballot_w64((active & 0x30u) != 0) != 0;
  const bool any3 = __builtin_amdgcn_ballot_w64((active & 0xC0u) != 0) != 0;
  const bool lim0 = __builtin_amdgcn_ballot_w64((active & (7u << (REX_NPOINT + 0))) != 0) != 0;
  const bool lim1 = __builtin_amdgcn_ballot_w64((active & (7u << (REX_NPOINT + 3))) != 0) != 0;
  const bool lim2 = __builtin_amdgcn_ballot_w64((active & (7u << (REX_NPOINT + 6))) != 0) != 0;
  const bool lim3 = __builtin_amdgcn_ballot_w64((active & (7u << (REX_NPOINT + 9))) != 0) != 0;
  x.y01 = v2{y[0], y[1]}; x.y23 = v2{y[2], y[3]}; x.y45 = v2{y[4], y[5]};
  // Bullet leaves the sweep loop as soon as the largest velocity residual of a sweep is below
  // m_leastSquaresResidualThreshold; each lane (env) stops on its own sweep, the wavefront leaves the
  // loop when its last lane has stopped.
  REX_STAMP(t_pgs0);
  int nsweeps = 0;
  if constexpr (kSplitLegs) {
    const bool lim[4] = {lim0, lim1, lim2, lim3};
    constexpr bool kHold = REX_HOLD_ACROSS_SWEEPS(EPW, ARMP::NM > 12, SM::kBody, LANECAP);
    if constexpr (kHold) {
      hold(Lc.l); hold(Lc.di);
      hold(s.pos); hold(s.quat); hold(s.lin); hold(s.ang);
#pragma unroll
      for (int k = 0; k < 3; ++k) { hold(s.q[k]); hold(s.qd[k]); }
    }
    // mark 'arm' at 16 envs per wave has no register left (512 taken, LDS full at four workgroups per CU) -- but a friction
    // row has no target: word 9 of its 16 rows is free once the rows are finished (pgs_dv reads the targets of the 8 normal
    // rows only).  The base Cholesky factor -- identical in the lanes of a group, needed again by the back-substitution --
    // waits there during the sweeps.
    constexpr int kRowParkBits = REX_PARK_IN_ROWS(EPW, ARMP::NM > 12);
    constexpr bool kRowPark = (kRowParkBits & 1) != 0, kRowParkDi = (kRowParkBits & 2) != 0, kRowParkBody = (kRowParkBits & 4) != 0;
    if constexpr (kRowPark) {
#pragma unroll
      for (int k = 0; k < 15; ++k) sm.rowf(REX_NPOINT + k, 9) = Lc.l[k];
      sm.rowf(REX_NPOINT + 15, 9) = Lc.di[0];
    }
    if constexpr (kRowParkDi) {
#pragma unroll
      for (int k = 1; k < 6; ++k) armp.spare(k) = Lc.di[k];   // (and the diagonal words of the arm limit rows, which the lane groups' sweep does not read)
    }
    // ... and what the integrator still needs of the body state (base position and orientation, the arm's joint angles:
    // identical in the lanes of a group; the velocities are rewritten by the back-substitution) in the chunks of the row
    // couplings, which pgs_dv holds in registers during the sweeps (cpl_free)
    auto park_body = [&]() __attribute__((always_inline)) {
      if constexpr (kRowParkBody) {
        sm.park(REX_PARK_CPL + 0) = make_float4(s.pos[0], s.pos[1], s.pos[2], s.quat[0]);
        sm.park(REX_PARK_CPL + 1) = make_float4(s.quat[1], s.quat[2], s.quat[3], s.q[12]);
        sm.park(REX_PARK_CPL + 2) = make_float4(s.q[13], s.q[14], s.q[15], s.q[16]);
        sm.parkf(REX_PARK_CPL + 3, 0) = s.q[17];
      }
    };
    pgs_dv<LPE, LANECAP>(sm, armp, x, pl, lim, any0 || any1 || any2 || any3, bgroups, ground.mu, iterations, lane_iterations, sqrt_res_thr, nsweeps, lane_sweeps, park_body);
    if constexpr (kRowParkBody) {
      const float4 a = sm.park(REX_PARK_CPL + 0), b = sm.park(REX_PARK_CPL + 1), c = sm.park(REX_PARK_CPL + 2);
      s.pos[0] = a.x; s.pos[1] = a.y; s.pos[2] = a.z; s.quat[0] = a.w; s.quat[1] = b.x; s.quat[2] = b.y; s.quat[3] = b.z; s.q[12] = b.w;
      s.q[13] = c.x; s.q[14] = c.y; s.q[15] = c.z; s.q[16] = c.w; s.q[17] = sm.parkf(REX_PARK_CPL + 3, 0);
    }
    if constexpr (kRowPark) {
#pragma unroll
      for (int k = 0; k < 15; ++k) Lc.l[k] = sm.rowf(REX_NPOINT + k, 9);
      Lc.di[0] = sm.rowf(REX_NPOINT + 15, 9);
    }
    if constexpr (kRowParkDi) {
#pragma unroll
      for (int k = 1; k < 6; ++k) Lc.di[k] = armp.spare(k);
    }
    if constexpr (kHold) {
      take(Lc.l); take(Lc.di);
      take(s.pos); take(s.quat); take(s.lin); take(s.ang);
#pragma unroll
      for (int k = 0; k < 3; ++k) { take(s.q[k]); take(s.qd[k]); }
    }
  } else {
    // one env per lane (EPW = 64): every lane carries the whole of x, row by row from LDS
    bool running = true;
    for (int it = 0; it < iterations; ++it) {
      ++nsweeps;
      if (running) {
        ++lane_sweeps;
        float worst = 0.0f;
        if (lim0) pgs_leg_limits<0>(sm, x, lam, worst);   // non-contact rows first (Bullet's sweep order)
        if (lim1) pgs_leg_limits<1>(sm, x, lam, worst);
        if (lim2) pgs_leg_limits<2>(sm, x, lam, worst);
        if (lim3) pgs_leg_limits<3>(sm, x, lam, worst);
        armp.sweep(x, worst);
        if (any0) pgs_leg_normals<0>(sm, x, lam, worst);
        if (any1) pgs_leg_normals<1>(sm, x, lam, worst);
        if (any2) pgs_leg_normals<2>(sm, x, lam, worst);
        if (any3) pgs_leg_normals<3>(sm, x, lam, worst);
        if (any0) pgs_leg_friction<0>(sm, x, lam, worst, ground.mu);
        if (any1) pgs_leg_friction<1>(sm, x, lam, worst, ground.mu);
        if (any2) pgs_leg_friction<2>(sm, x, lam, worst, ground.mu);
        if (any3) pgs_leg_friction<3>(sm, x, lam, worst, ground.mu);
        running = LANECAP ? (worst > sqrt_res_thr && it + 1 < lane_iterations) : worst > sqrt_res_thr;
      }
      if (__builtin_amdgcn_ballot_w64(running) == 0) break;
    }
  }
  REX_STAMP(t_pgs1);
  if constexpr (TRACE) { if (live) trace[trace_n + env] = trace_mix(trace[trace_n + env], (unsigned)lane_sweeps); }   // (cumulative over the env.step())
  y[0] = x.y01.x; y[1] = x.y01.y; y[2] = x.y23.x; y[3] = x.y23.y; y[4] = x.y45.x; y[5] = x.y45.y;

  // back to generalized velocities: nu0 = Lc^-T y ;  qd_f = G^-T (z_f - Bw_f nu0)
  float nu[6];
  bwd6(Lc, y, nu);
  float zt[12];
#pragma unroll
  for (int k = 0; k < REX_NLEG; ++k) { zt[3 * k] = x.z01[k].x; zt[3 * k + 1] = x.z01[k].y; zt[3 * k + 2] = x.z2[k]; }
  if constexpr (kSplitLegs) {
    if constexpr (SM::kLegF4 != 1) leg_unpark(sm, mleg, Lown);
    const LegFactor& L = Lown;
    float t1 = pick_leg(zt, mleg, 0), t2 = pick_leg(zt, mleg, 1), t3 = pick_leg(zt, mleg, 2);
#pragma unroll
    for (int k = 0; k < 6; ++k) { t1 -= L.Bw[0][k] * nu[k]; t2 -= L.Bw[1][k] * nu[k]; t3 -= L.Bw[2][k] * nu[k]; }
    const float u3 = t3 * L.gi3;
    const float u2 = (t2 - L.g32 * u3) * L.gi2;
    const float u1 = (t1 - L.g21 * u2 - L.g31 * u3) * L.gi1;
    s.qd[0] = clampf(u1, -kMaxCoordVel, kMaxCoordVel); s.qd[1] = clampf(u2, -kMaxCoordVel, kMaxCoordVel);
    s.qd[2] = clampf(u3, -kMaxCoordVel, kMaxCoordVel);
    mirror_sync();   // the next substep's leg pass overwrites the z chunks the lanes have just read
  } else {
#pragma unroll 1
    for (int leg = 0; leg < REX_NLEG; ++leg) {
      LegFactor L;
      leg_unpark(sm, leg, L);
      float t1 = zt[0], t2 = zt[1], t3 = zt[2];
#pragma unroll
      for (int k = 0; k < 6; ++k) { t1 -= L.Bw[0][k] * nu[k]; t2 -= L.Bw[1][k] * nu[k]; t3 -= L.Bw[2][k] * nu[k]; }
      const float u3 = t3 * L.gi3;
      const float u2 = (t2 - L.g32 * u3) * L.gi2;
      const float u1 = (t1 - L.g21 * u2 - L.g31 * u3) * L.gi1;
      s.qd[0] = clampf(u1, -kMaxCoordVel, kMaxCoordVel);
      s.qd[1] = clampf(u2, -kMaxCoordVel, kMaxCoordVel);
      s.qd[2] = clampf(u3, -kMaxCoordVel, kMaxCoordVel);
      rotate_leg(s.qd); rotate_leg(zt);
    }
  }
  armp.back(nu, s);
  const float free_base = ground.anchor == 0.0f ? 1.0f : 0.0f;   // a fixed base keeps zero velocity exactly
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    s.ang[k] = free_base * clampf(nu[k], -kMaxCoordVel, kMaxCoordVel);
    s.lin[k] = free_base * clampf(nu[3 + k], -kMaxCoordVel, kMaxCoordVel);
  }
  // semi-implicit Euler with the NEW velocities
#pragma unroll
  for (int k = 0; k < 3; ++k) s.pos[k] += dt * s.lin[k];
#pragma unroll
  for (int j = 0; j < ARMP::NM; ++j) if (!kSplitLegs || j < 3 || j >= 12) s.q[j] += dt * s.qd[j];   // lane groups: own leg (+ arm)
  {
    const float wn = sqrtf(s.ang[0] * s.ang[0] + s.ang[1] * s.ang[1] + s.ang[2] * s.ang[2]);
    const float angle = wn * dt;
    float sc;
    if (wn < 0.001f) sc = 0.5f * dt - dt * dt * dt * 0.020833333333f * wn * wn;
    else { float sh, ch; sincos_fast(0.5f * angle, sh, ch); sc = sh / wn; }
    const float dx = s.ang[0] * sc, dy = s.ang[1] * sc, dz = s.ang[2] * sc, dw = cos_half(0.5f * angle);
    const float qx = s.quat[0], qy = s.quat[1], qz = s.quat[2], qw = s.quat[3];
    const float nx = dw * qx + dx * qw + dy * qz - dz * qy;
    const float ny = dw * qy - dx * qz + dy * qw + dz * qx;
    const float nz = dw * qz + dx * qy - dy * qx + dz * qw;
    const float nw = dw * qw - dx * qx - dy * qy - dz * qz;
    const float nn = rsqrtf(nx * nx + ny * ny + nz * nz + nw * nw);
    s.quat[0] = nx * nn; s.quat[1] = ny * nn; s.quat[2] = nz * nn; s.quat[3] = nw * nn;
#ifdef REX_PROF
    if (threadIdx.x == 0 && blockIdx.x < 1024) {
      long long* p = g_prof + 10 * blockIdx.x;
      p[0] += t_pgs1 - t_pgs0; p[1] += nsweeps; p[2] += clock64() - t_begin; p[4] += 1;
      p[3] += t_pgs0 - t_chol; p[6] += t_legs - t_begin; p[7] += t_chol - t_legs;
      p[5] += any_limit ? 1 : 0;      // substeps with joint-limit rows in reach of some env of the wave
    }
#endif
  }
}

}  // namespace rex
