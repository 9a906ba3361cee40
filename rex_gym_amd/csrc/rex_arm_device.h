// rex_arm_device.h -- the 6-joint arm chain of mark='arm' (rex_arm.urdf:610-791) in the same world-aligned,
// Cholesky-whitened formulation as the legs (rex_device.h): a fifth branch on the base with
//     H_a = G_a G_a^T (6x6),  Bw_a = G_a^-1 B_a^T (6x6),  A -= Bw_a^T Bw_a,  z_a = G_a^T qd_a + Bw_a nu_base.
// The arm never touches the ground; its only constraint rows are the URDF joint limits (the rest pose the envs
// command, ARM_POSES['rest'] = (-1.6, -1.6, 0, 0, 1.6, 0), sits beyond the +-1.5 rad bounds of m1, m2, m5, so three
// limit rows are permanently active).  A limit row is a 12-vector (6 whitened base + 6 whitened arm entries).
#pragma once
#include "rex_arm_model_gen.h"
#include "rex_device.h"

/* the arm kernels run 4..16 envs per wave with the leg factors in registers: its chunks start behind 112 row + z chunks */
#define REX_ARM_BASE_F4 REX_ROWS_F4_OF(1)
#define REX_ARM_NJ 6
#define REX_ARM_PARK_F4 16   /* Bw 36 + G 21 + z 6 = 63 floats */
#define REX_ARM_ROW_F4 4     /* g' 6 + j' 6 + (invd*target, invd, diag, active) */
#define REX_LDS_F4_PER_ENV_ARM_OF(EPW) (REX_ARM_BASE_F4 + REX_ARM_PARK_F4 + REX_ARM_NJ * REX_ARM_ROW_F4)

namespace rex {

struct m33 { f3 x, y, z; };   // columns

template <int EPW>
struct LdsArm {
  static constexpr int kEpw = EPW;
  float4* p; int slot;
  __device__ __forceinline__ float4& park(int c) const { return p[(REX_ARM_BASE_F4 + c) * EPW + slot]; }
  __device__ __forceinline__ float4& row(int k, int c) const {
    return p[(REX_ARM_BASE_F4 + REX_ARM_PARK_F4 + k * REX_ARM_ROW_F4 + c) * EPW + slot];
  }
};

struct ArmFactor {
  Chol6 G;          // Cholesky factor of the arm joint-space inertia
  float Bw[6][6];   // G^-1 B^T: row m = whitened arm coordinate, column k = base coordinate
  float z[6];       // whitened predicted arm velocity
};

// forward kinematics, Newton-Euler bias, composite inertias, arm Cholesky, Schur contributions to the base and the
// limit rows.  q/qd/tau: the 6 arm joints.  Returns bits 0..5: limit row k active.
template <class SMA>
__device__ __forceinline__ unsigned arm_pass(const BaseKin& bk, const float* __restrict__ q, const float* __restrict__ qd,
                                             const float* __restrict__ tau, float dt, ArmFactor& L, BaseAccum& acc,
                                             const SMA& sma, const Ground& ground) {
  // Outward pass.  What the inward pass needs of joint k (axis, origin, COM, inertia, force, moment: 21 floats) is
  // stashed in LDS -- in the chunks that will hold this env's arm limit rows and parked factors afterwards (every lane
  // of the group writes the same numbers) -- instead of 126 registers staying live between the two passes.
  auto stash = [&](int k, int c) -> float4& { return sma.p[(REX_ARM_BASE_F4 + 6 * k + c) * SMA::kEpw + sma.slot]; };
  static_assert(6 * 6 <= REX_ARM_PARK_F4 + REX_ARM_NJ * REX_ARM_ROW_F4, "arm stash must fit the arm rows + park chunks");
  {
    m33 Rp{bk.ex, bk.ey, bk.ez};
    f3 op = mk(0.f, 0.f, 0.f), wp = bk.w, alp = mk(0.f, 0.f, 0.f);
    f3 vop = bk.v;                                    // velocity of the parent origin
    f3 aop = mk(0.f, 0.f, kGravity);                  // acceleration of the parent origin (zero gen. acc., gravity as +g)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      f3 a_k, o_k, c_k, w_k, al_k, fo_k, no_k;
      s33 Ib_k;
      // joint frame in parent coordinates (signed permutation, constants fold), then the turn about +-z
      const float e00 = (float)REXA_E0[k][0], e01 = (float)REXA_E0[k][1], e02 = (float)REXA_E0[k][2];
      const float e10 = (float)REXA_E0[k][3], e11 = (float)REXA_E0[k][4], e12 = (float)REXA_E0[k][5];
      const float e20 = (float)REXA_E0[k][6], e21 = (float)REXA_E0[k][7], e22 = (float)REXA_E0[k][8];
      const f3 jx = e00 * Rp.x + e10 * Rp.y + e20 * Rp.z;
      const f3 jy = e01 * Rp.x + e11 * Rp.y + e21 * Rp.z;
      const f3 jz = e02 * Rp.x + e12 * Rp.y + e22 * Rp.z;
      float sq, cq;
      sincos_fast((float)REXA_AXIS_SIGN[k] * q[k], sq, cq);
      m33 R;
      R.x = cq * jx + sq * jy;
      R.y = cq * jy - sq * jx;
      R.z = jz;
      a_k = (float)REXA_AXIS_SIGN[k] * jz;
      const f3 d = (float)REXA_POS[k][0] * Rp.x + (float)REXA_POS[k][1] * Rp.y + (float)REXA_POS[k][2] * Rp.z;
      o_k = op + d;
      const f3 e = (float)REXA_COM[k][0] * R.x + (float)REXA_COM[k][1] * R.y + (float)REXA_COM[k][2] * R.z;
      c_k = o_k + e;
      // kinematics (same recursion as the legs)
      const f3 vo = vop + cross(wp, d);
      const f3 ao = aop + cross(alp, d) + cross(wp, cross(wp, d));
      w_k = wp + qd[k] * a_k;
      al_k = alp + cross(wp, qd[k] * a_k);
      const f3 vc = vo + cross(w_k, e);
      const f3 ac = ao + cross(al_k, e) + cross(w_k, cross(w_k, e));
      const float m_k = (float)REXA_MASS[k] * ground.leg_mass_scale;
      Ib_k = rot_inertia(R.x, R.y, R.z, (float)REXA_INERTIA[k][0], (float)REXA_INERTIA[k][1], (float)REXA_INERTIA[k][2]);
      const f3 Iw = mul(Ib_k, w_k);
      const float dl = kLinDamp + kLinDamp * sqrtf(dot(vc, vc));
      const float da = kAngDamp + kAngDamp * sqrtf(dot(w_k, w_k));
      fo_k = m_k * ac + (m_k * dl) * vc;
      no_k = mul(Ib_k, al_k) + cross(w_k, Iw) + da * Iw + cross(c_k, fo_k);   // moment about the base origin
      stash(k, 0) = make_float4(a_k.x, a_k.y, a_k.z, o_k.x);
      stash(k, 1) = make_float4(o_k.y, o_k.z, c_k.x, c_k.y);
      stash(k, 2) = make_float4(c_k.z, fo_k.x, fo_k.y, fo_k.z);
      stash(k, 3) = make_float4(no_k.x, no_k.y, no_k.z, Ib_k.xx);
      stash(k, 4) = make_float4(Ib_k.yy, Ib_k.zz, Ib_k.xy, Ib_k.xz);
      stash(k, 5) = make_float4(Ib_k.yz, 0.0f, 0.0f, 0.0f);
      Rp = R; op = o_k; wp = w_k; alp = al_k; vop = vo; aop = ao;
    }
  }
  // inward: subtree wrenches, joint bias, composite inertias, joint columns
  float C[6];
  f3 a[6], Fl[6], Fa[6], v[6];
  {
    f3 F = mk(0.f, 0.f, 0.f), N = F, h = F;
    s33 Io{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float mc = 0.0f;
#pragma unroll
    for (int k = 5; k >= 0; --k) {
      const float4 s0 = stash(k, 0), s1 = stash(k, 1), s2 = stash(k, 2), s3 = stash(k, 3), s4 = stash(k, 4), s5 = stash(k, 5);
      a[k] = mk(s0.x, s0.y, s0.z);
      const f3 o_k = mk(s0.w, s1.x, s1.y), c_k = mk(s1.z, s1.w, s2.x);
      const s33 Ib_k{s3.w, s4.x, s4.y, s4.z, s4.w, s5.x};
      const float m_k = (float)REXA_MASS[k] * ground.leg_mass_scale;
      F = F + mk(s2.y, s2.z, s2.w); N = N + mk(s3.x, s3.y, s3.z);
      C[k] = dot(a[k], N - cross(o_k, F));
      add(Io, Ib_k); add_point(Io, m_k, c_k);
      h = h + m_k * c_k;
      mc += m_k;
      v[k] = cross(o_k, a[k]);
      Fl[k] = mc * v[k] + cross(a[k], h);
      Fa[k] = mul(Io, a[k]) + cross(h, v[k]);
    }
    acc.N = acc.N + N; acc.F = acc.F + F;
    add(acc.Io, Io); acc.h = acc.h + h; acc.m += mc;
  }
  // H (lower triangle): H[j][i] (j >= i) = S_i . F_j
  float H[21];
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int i = 0; i <= j; ++i) H[tri(j, i)] = dot(a[i], Fa[j]) + dot(v[i], Fl[j]);
  chol6(H, L.G);
  // Bw columns: solve G u = B^T[:, k] where B^T[j][k] = component k of (Fa_j, Fl_j)
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    float b[6], u[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const f3 src = k < 3 ? Fa[j] : Fl[j];
      b[j] = (k % 3) == 0 ? src.x : ((k % 3) == 1 ? src.y : src.z);
    }
    fwd6(L.G, b, u);
#pragma unroll
    for (int j = 0; j < 6; ++j) L.Bw[j][k] = u[j];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      float t = 0.0f;
#pragma unroll
      for (int r = 0; r < 6; ++r) t += L.Bw[r][i] * L.Bw[r][j];
      acc.S[tri(i, j)] += t;
    }
  // free whitened acceleration and predicted whitened velocity
  float rhs[6], zd[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) rhs[k] = tau[k] - C[k];
  fwd6(L.G, rhs, zd);
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    float t = 0.0f;
#pragma unroll
    for (int r = 0; r < 6; ++r) t += L.Bw[r][k] * zd[r];
    acc.bz[k] += t;
  }
  {
    const float nu0[6] = {bk.w.x, bk.w.y, bk.w.z, bk.v.x, bk.v.y, bk.v.z};
    float zc[6];
    mulT6(L.G, qd, zc);     // G^T qd
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      float t = zc[r];
#pragma unroll
      for (int k = 0; k < 6; ++k) t += L.Bw[r][k] * nu0[k];
      L.z[r] = t + dt * zd[r];
    }
  }
  // limit rows: near bound of each arm joint
  unsigned active = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float lo_gap = q[k] - (float)REXA_LOWER[k], hi_gap = (float)REXA_UPPER[k] - q[k];
    const bool lower = lo_gap < hi_gap;
    const float gap = lower ? lo_gap : hi_gap;
    const bool act = gap <= kLimitActivation;
    if (act) active |= 1u << k;
    float e[6], j[6], g[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) e[r] = r == k ? (lower ? 1.0f : -1.0f) : 0.0f;
    fwd6(L.G, e, j);
#pragma unroll
    for (int c2 = 0; c2 < 6; ++c2) {
      float t = 0.0f;
#pragma unroll
      for (int r = 0; r < 6; ++r) t += L.Bw[r][c2] * j[r];
      g[c2] = -t;
    }
    const float target = gap > 0.0f ? -gap / dt : -gap * (kErp / dt);
    sma.row(k, 0) = make_float4(g[0], g[1], g[2], g[3]);
    sma.row(k, 1) = make_float4(g[4], g[5], j[0], j[1]);
    sma.row(k, 2) = make_float4(j[2], j[3], j[4], j[5]);
    sma.row(k, 3) = make_float4(target, act ? 1.0f : 0.0f, 0.0f, 0.0f);
  }
  return active;
}

template <class SMA>
__device__ __forceinline__ void arm_park(const SMA& sma, const ArmFactor& L) {
  float buf[64];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int k = 0; k < 6; ++k) buf[6 * r + k] = L.Bw[r][k];
#pragma unroll
  for (int k = 0; k < 15; ++k) buf[36 + k] = L.G.l[k];
#pragma unroll
  for (int k = 0; k < 6; ++k) { buf[51 + k] = L.G.di[k]; buf[57 + k] = L.z[k]; }
  buf[63] = 0.0f;
#pragma unroll
  for (int c = 0; c < REX_ARM_PARK_F4; ++c) sma.park(c) = make_float4(buf[4 * c], buf[4 * c + 1], buf[4 * c + 2], buf[4 * c + 3]);
}
template <class SMA>
__device__ __forceinline__ void arm_unpark(const SMA& sma, ArmFactor& L) {
  float buf[64];
#pragma unroll
  for (int c = 0; c < REX_ARM_PARK_F4; ++c) {
    const float4 t = sma.park(c);
    buf[4 * c] = t.x; buf[4 * c + 1] = t.y; buf[4 * c + 2] = t.z; buf[4 * c + 3] = t.w;
  }
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int k = 0; k < 6; ++k) L.Bw[r][k] = buf[6 * r + k];
#pragma unroll
  for (int k = 0; k < 15; ++k) L.G.l[k] = buf[36 + k];
#pragma unroll
  for (int k = 0; k < 6; ++k) { L.G.di[k] = buf[51 + k]; L.z[k] = buf[57 + k]; }
}

// whiten the base part of the arm limit rows and finish them (after the base Cholesky)
template <class SMA>
__device__ __forceinline__ void arm_rows_finish(const SMA& sma, const Chol6& Lc) {
#pragma unroll 1
  for (int k = 0; k < 6; ++k) {
    const float4 c0 = sma.row(k, 0), c1 = sma.row(k, 1), c2 = sma.row(k, 2), c3 = sma.row(k, 3);
    const float g[6] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y};
    float gw[6];
    fwd6(Lc, g, gw);
    const float diag = gw[0] * gw[0] + gw[1] * gw[1] + gw[2] * gw[2] + gw[3] * gw[3] + gw[4] * gw[4] + gw[5] * gw[5] +
                       c1.z * c1.z + c1.w * c1.w + c2.x * c2.x + c2.y * c2.y + c2.z * c2.z + c2.w * c2.w;
    const float invd = c3.y != 0.0f ? __builtin_amdgcn_rcpf(diag) : 0.0f;
    sma.row(k, 0) = make_float4(gw[0], gw[1], gw[2], gw[3]);
    sma.row(k, 1) = make_float4(gw[4], gw[5], c1.z, c1.w);
    sma.row(k, 3) = make_float4(c3.x * invd, invd, diag, 0.0f);
  }
}

// one sweep over the arm limit rows (they are non-contact rows: first in Bullet's sweep order)
template <class SMA>
__device__ __forceinline__ void pgs_arm_limits(const SMA& sma, PgsX& x, float* za, float* lam_a, unsigned active_any, float& worst) {
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    if (!((active_any >> k) & 1u)) continue;   // wave-uniform: no env of the wave has this bound in reach
    const float4 c0 = sma.row(k, 0), c1 = sma.row(k, 1), c2 = sma.row(k, 2), c3 = sma.row(k, 3);
    const float vel = c0.x * x.y01.x + c0.y * x.y01.y + c0.z * x.y23.x + c0.w * x.y23.y + c1.x * x.y45.x + c1.y * x.y45.y +
                      c1.z * za[0] + c1.w * za[1] + c2.x * za[2] + c2.y * za[3] + c2.z * za[4] + c2.w * za[5];
    const float nl = fmaxf(fmaf(-c3.y, vel, lam_a[k] + c3.x), 0.0f);
    const float dl = nl - lam_a[k];
    lam_a[k] = nl;
    worst = fmaxf(worst, fabsf(dl * c3.z));
    x.y01.x += c0.x * dl; x.y01.y += c0.y * dl; x.y23.x += c0.z * dl; x.y23.y += c0.w * dl; x.y45.x += c1.x * dl; x.y45.y += c1.y * dl;
    za[0] += c1.z * dl; za[1] += c1.w * dl; za[2] += c2.x * dl; za[3] += c2.y * dl; za[4] += c2.z * dl; za[5] += c2.w * dl;
  }
}

// qd_a = G^-T (z_a - Bw nu)
template <class SMA>
__device__ __forceinline__ void arm_back(const SMA& sma, const float* za, const float* nu, float* qd) {
  ArmFactor L;
  arm_unpark(sma, L);
  float t[6], u[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    float s = za[r];
#pragma unroll
    for (int k = 0; k < 6; ++k) s -= L.Bw[r][k] * nu[k];
    t[r] = s;
  }
  bwd6(L.G, t, u);
#pragma unroll
  for (int r = 0; r < 6; ++r) qd[r] = clampf(u[r], -kMaxCoordVel, kMaxCoordVel);
}

// the ARMP hook of physics_substep for mark='arm'
template <int EPW>
struct ArmChain {
  static constexpr int NM = 18;
  static constexpr bool kHasRows = true;   // its joint-limit rows are always in the sweep
  LdsArm<EPW> sma;
  float za[6], lam_a[6];
  unsigned active_any;   // bit k: some env of the wave has arm limit row k in reach

  __device__ __forceinline__ void pass(const BaseKin& bk, PhysState& s, const float* tau, float dt, BaseAccum& acc, const Ground& ground) {
    ArmFactor L;
    const unsigned act = arm_pass(bk, s.q + 12, s.qd + 12, tau, dt, L, acc, sma, ground);   // tau: the arm's 6 torques
    arm_park(sma, L);
    active_any = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      za[k] = L.z[k]; lam_a[k] = 0.0f;
      if (__builtin_amdgcn_ballot_w64((act >> k) & 1u) != 0) active_any |= 1u << k;
    }
  }
  __device__ __forceinline__ void finish(const Chol6& Lc) { arm_rows_finish(sma, Lc); }
  __device__ __forceinline__ void sweep(PgsX& x, float& worst) { pgs_arm_limits(sma, x, za, lam_a, active_any, worst); }
  __device__ __forceinline__ void back(const float* nu, PhysState& s) { arm_back(sma, za, nu, s.qd + 12); }

  // ---- lane-distributed sweep (rex_device.h: pgs_dv): lane p owns arm component p (+ LPE) next to its share of y ----
  // An arm limit row is 16 floats: g' 0..5, j' 6..11, (invd target, invd, diag, 0); float 15 is the zero that the
  // lanes owning no component read.  The parked whitened arm velocity sits in floats 57..62 of the park chunks.
  // This lane's slice of the six arm limit rows, read from LDS once per substep (as pgs_dv does for the contact rows):
  // its components of the base part and of the arm part, the inverse diagonal, and -target in lane 0 of the group (the
  // addend that makes the group sum vel - target).  Rows out of reach have invd = 0 and never move.
  float rjy[6][2], rja[6][2], rki[6], rkt[6];
  float as_[2];
  static constexpr int kRowBytes = REX_ARM_ROW_F4 * EPW * 16;
  __device__ __forceinline__ int foff(int f) const { return ((REX_ARM_BASE_F4 + REX_ARM_PARK_F4 + (f >> 2)) * EPW + sma.slot) * 16 + (f & 3) * 4; }
  __device__ __forceinline__ float ldb(int off) const { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(sma.p) + off); }
  __device__ __forceinline__ float& parkf(int f) const { return reinterpret_cast<float*>(&sma.park(f >> 2))[f & 3]; }
  // the diagonal word of limit row k: written by finish(), read by the one-env-per-lane sweep only -- a free word per row
  // for the lane groups once the rows are finished (physics_substep parks bystanders of the sweep loop there)
  __device__ __forceinline__ float& spare(int k) const { return sma.row(k, 3).z; }
  template <int LPE>
  __device__ __forceinline__ void dv_begin(int p) {
    constexpr int NY = (6 + LPE - 1) / LPE;
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const int k = p + i * LPE;
      const int oy = foff(k < 6 ? k : 15), oa = foff(k < 6 ? 6 + k : 15);
      as_[i] = parkf(k < 6 ? 57 + k : 63);
#pragma unroll
      for (int r = 0; r < 6; ++r) { rjy[r][i] = ldb(r * kRowBytes + oy); rja[r][i] = ldb(r * kRowBytes + oa); }
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const float4 c3 = sma.row(r, 3);                 // (invd * target, invd, diag, 0)
      rki[r] = c3.y;
      rkt[r] = (p == 0 && c3.y > 0.0f) ? -c3.x * __builtin_amdgcn_rcpf(c3.y) : 0.0f;
      lam_a[r] = 0.0f;
    }
  }
  template <int LPE, int NY>
  __device__ __forceinline__ void dv_sweep(float* ys, float& worst, float thr) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      if (!((active_any >> k) & 1u)) continue;   // wave-uniform
      float part = rkt[k];
#pragma unroll
      for (int i = 0; i < NY; ++i) part = fmaf(rjy[k][i], ys[i], fmaf(rja[k][i], as_[i], part));
      const float sum = group_sum<LPE>(part);    // vel - target
      const float nl = fmaxf(fmaf(-rki[k], sum, lam_a[k]), 0.0f);
      const float dl = nl - lam_a[k];
      lam_a[k] = nl;
      worst = fmaxf(worst, fmaf(-thr, rki[k], fabsf(dl)));
#pragma unroll
      for (int i = 0; i < NY; ++i) { ys[i] = fmaf(rjy[k][i], dl, ys[i]); as_[i] = fmaf(rja[k][i], dl, as_[i]); }
    }
  }
  template <int LPE>
  __device__ __forceinline__ void dv_end(int p) {
    constexpr int NY = (6 + LPE - 1) / LPE;
#pragma unroll
    for (int i = 0; i < NY; ++i) { const int k = p + i * LPE; parkf(k < 6 ? 57 + k : 63) = as_[i]; }
  }
  __device__ __forceinline__ void dv_gather() {   // after the group's sync: every lane needs the whole arm velocity
    const float4 a = sma.park(14), b = sma.park(15);
    za[0] = a.y; za[1] = a.z; za[2] = a.w; za[3] = b.x; za[4] = b.y; za[5] = b.z;
  }
};

// picks the physics_substep hook of a kernel instantiation
template <int EPW, bool ARM> struct ArmHook;
template <int EPW> struct ArmHook<EPW, false> {
  using type = NoArm;
  __device__ __forceinline__ static NoArm make(float4*, int) { return NoArm{}; }
};
template <int EPW> struct ArmHook<EPW, true> {
  using type = ArmChain<EPW>;
  __device__ __forceinline__ static ArmChain<EPW> make(float4* lds, int slot) {
    ArmChain<EPW> a;
    a.sma = LdsArm<EPW>{lds, slot};
    a.active_any = 0;
    return a;
  }
};

}  // namespace rex
