// rex_policy.h -- the ACTOR of the reference's PPO agents evaluated inside the step kernel (rex_step_policy /
// rex_step_segment_policy): what `algo.perform(prevob)` computes between two `batch_env.simulate(action)` calls of the
// reference's rollout loop (agents/tools/simulate.py:57-76, agents/ppo/algorithm.py:105-134), so that a closed-loop rollout
// -- a policy in the loop -- runs as ONE launch per segment like an open-loop one.
//
//   observ filter   agents/ppo/normalize.py:47-66: (o - mean) / std, clipped to +-5 -- the statistics are the caller's,
//                   frozen at rex_set_policy (obs_scale = 1 / (std + 1e-8))
//   network         agents/scripts/networks.py:66-110 ForwardGaussianPolicy: relu(W1 x + b1) -> relu(W2 . + b2) ->
//                   mean = tanh(W3 . + b3); logstd a free vector (configs.py:31-32: 200 and 100 units)
//   action          training: mean + exp(logstd) * N(0, 1) (`network.policy.sample`, algorithm.py:117); else the mean
//
// Mapping.  A wave carries EPW envs (lane groups, rex_kernels.h) and evaluates the network for all of them at once on the
// MATRIX cores: v_mfma_f32_4x4x1_16b_f32 is 16 independent 4 x 4 outer-product accumulations per instruction -- here 16 blocks
// of 4 UNITS x 4 ENVS, one input k per instruction: operand A = the weight W[k][unit] of the lane's unit (lane l of a pass owns
// unit l), operand B = the activation x[k][env] of env (l & 3) of the env group, D = 4 units x 1 env per lane (lane layout probed:
// tools/microbench/mfma_4x4x1_layout.hip).  The fp32 MFMA is bit for bit an fmaf chain (even and odd inputs run as two chains, added at the end), so the arithmetic is a plain fp32
// network's and does not depend on EPW, on the segment length or on where the weights come from.  A pass covers 128 units (two
// MFMAs per k and env group); the 200-unit and the 100-unit layer of the reference's actor are 2 + 1 passes.  Inputs are read
// four k at a time: weights packed [k / 4][unit][4] (rex_set_policy packs them) so that a lane's four weights are one b128 read,
// 1 KB contiguous per wave read; activations in LDS as [k / 4][env][4], a broadcast b128.  The weights are shared by every wave:
// where they fit next to the rows of the four waves of a workgroup (85 KB for 4-200-100-2) ONE copy is loaded into LDS per launch,
// else they are streamed from L2 every step, two chunks in flight.  The hidden activations use the contact-row region of LDS,
// idle between two env steps.
// What a perform() costs a wave (tools/microbench/policy_mb.hip, profiles/r06_mb_policy.txt): DESIGN.md section 5.
#pragma once
#include <type_traits>

namespace rex {

// float offsets of the packed actor (global buffer written by rex_pack_policy_kernel; the same layout in LDS); every block starts on a
// 16-byte boundary and the buffer ends with REX_POLICY_SLACK zeros: a lane whose unit lies beyond a layer's width reads on into the next
// block instead of branching (finite numbers, its results are dropped)
#define REX_POLICY_SLACK 640
__host__ __device__ __forceinline__ int pol_q4(int n) { return (n + 3) >> 2; }
struct PolOff { int w1, b1, w2, b2, w3, b3, logstd, mean, scale, total; };
__host__ __device__ __forceinline__ PolOff policy_offsets(int O, int A, int H1, int H2) {
  PolOff o;
  auto up4 = [](int n) { return (n + 3) & ~3; };
  o.w1 = 0;                                // [ceil(O / 4)][H1][4]: W1[4 q + r][j] at ((q H1 + j) 4 + r), zero beyond O
  o.b1 = o.w1 + pol_q4(O) * H1 * 4;        // [H1]
  o.w2 = o.b1 + up4(H1);                   // [ceil(H1 / 4)][H2][4]
  o.b2 = o.w2 + pol_q4(H1) * H2 * 4;       // [H2]
  o.w3 = o.b2 + up4(H2);                   // [H2][A]
  o.b3 = o.w3 + up4(H2 * A);
  o.logstd = o.b3 + up4(A);
  o.mean = o.logstd + up4(A);              // observ filter: mean [O], scale [O] (0 / 1 without a filter)
  o.scale = o.mean + up4(O);
  o.total = o.scale + up4(O) + REX_POLICY_SLACK;
  return o;
}

struct PolDev {
  const float* pk;                       // the packed actor (policy_offsets), library-owned
  const float* obs_in;                   // [n][obs_dim]: the observation the FIRST step of the launch acts on
  float* action_out; float* mean_out;    // [nsteps][n][action_dim] (mean_out nullable)
  int32_t h1, h2;
  float obs_clip;                        // > 0: the observ filter is on
  int32_t sample;                        // 1: Gaussian sample (training), 0: the mean (evaluation)
  uint32_t seed_lo, seed_hi;
  int32_t in_lds;                        // the launch carries dynamic LDS for the packed actor: one copy per workgroup
};
struct NoPol {};
template <bool POLICY> struct PolArg { using type = NoPol; };
template <> struct PolArg<true> { using type = PolDev; };

// floats of LDS scratch per env of the wave: x [4 ceil(O / 4)], meta [4], action [8], h1 [4 ceil(H1 / 4)], h2 [4 ceil(H2 / 4)]
__host__ __device__ __forceinline__ int policy_scratch_floats(int obs_dim, int h1, int h2) { return 4 * pol_q4(obs_dim) + 12 + 4 * pol_q4(h1) + 4 * pol_q4(h2); }
#define REX_POLICY_NOISE_BLOCK 64   /* Philox block numbers (gauss4) of the action sample: behind the sensor-noise call sites */
#ifdef REX_POL_PROF      /* tools/microbench/policy_mb.hip: cycle counters of the sections of a perform() */
__device__ long long g_pol_prof[8];
#define REX_POL_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) { const long long now_ = clock64(); g_pol_prof[k] += now_ - pol_t_; pol_t_ = now_; } } while (0)
#define REX_POL_STAMP0 long long pol_t_ = clock64()
#else
#define REX_POL_STAMP(k)
#define REX_POL_STAMP0
#endif

typedef float pol_f4 __attribute__((ext_vector_type(4)));

// every thread of the workgroup copies its share of the packed actor into LDS (once per launch; a __syncthreads() follows)
__device__ __forceinline__ void policy_weights_to_lds(const PolDev& p, int total_floats, float* wl, int tid, int nthreads) {
  const float4* s4 = reinterpret_cast<const float4*>(p.pk);
  float4* d4 = reinterpret_cast<float4*>(wl);
  for (int k = tid; k < total_floats / 4; k += nthreads) d4[k] = s4[k];
}

// One ReLU layer for the E envs of the wave on the matrix cores: out[j][e] = relu(B[j] + sum_k W[k][j] in[k][e]).
//   Wp  packed weights [Kq][N][4] (float4 index q N + j; LDS or global), Bv [N]; in / out: LDS activations [k / 4][E][4]
// A lone wave hides nothing by itself: the reads of a chunk of KC input quads (weights 2 KC b128, activations G KC b128) are issued a
// chunk ahead of the MFMAs that consume them (A / B register sets), and a (unit half, env group) keeps TWO accumulators -- even and odd
// inputs -- so that consecutive MFMAs on one accumulator are four instructions apart.  out = relu(even chain (bias first) + odd chain).
template <int E>
__device__ __forceinline__ void dense_relu_mfma(const float* Wp, const float* Bv, int Kq, int N, const float* in, float* out, int lane) {
  constexpr int G = E / 4;                                         // env groups of 4: one MFMA each per k and unit half
  constexpr int KC = G == 1 ? 4 : 2;                               // (<= 12 LDS reads in flight: the counter holds 15)
  const int blk = lane >> 2, jj = lane & 3;
  const float4* in4 = reinterpret_cast<const float4*>(in) + jj;
  float4* out4 = reinterpret_cast<float4*>(out);
  for (int ub = 0; ub < N; ub += 2 * REX_WAVE) {
    const float4* w4 = reinterpret_cast<const float4*>(Wp) + ub + lane;      // this lane's unit of the lower half; + 64: the upper
    const float4 bl = *reinterpret_cast<const float4*>(Bv + ub + 4 * blk), bh = *reinterpret_cast<const float4*>(Bv + ub + REX_WAVE + 4 * blk);
    pol_f4 d0[G][2], d1[G][2];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      d0[g][0] = pol_f4{bl.x, bl.y, bl.z, bl.w}; d1[g][0] = pol_f4{bh.x, bh.y, bh.z, bh.w};
      d0[g][1] = pol_f4{0.0f, 0.0f, 0.0f, 0.0f}; d1[g][1] = pol_f4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    struct Chunk { float4 w0[KC], w1[KC], x[KC][G]; };
    auto fetch = [&](int q0, Chunk& c) {                           // (quads beyond Kq: the index is clamped, the quad is not consumed)
#pragma unroll
      for (int k = 0; k < KC; ++k) {
        const int q = min(q0 + k, Kq - 1);
        c.w0[k] = w4[q * N]; c.w1[k] = w4[q * N + REX_WAVE];
#pragma unroll
        for (int g = 0; g < G; ++g) c.x[k][g] = in4[q * E + 4 * g];
      }
    };
    auto consume = [&](int q0, const Chunk& c) {
#pragma unroll
      for (int k = 0; k < KC; ++k) {
        if (q0 + k < Kq) {                                         // inputs 4 q .. 4 q + 3, in order: even ones into [0], odd ones into [1]
#pragma unroll
          for (int g = 0; g < G; ++g) {
            d0[g][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(c.w0[k].x, c.x[k][g].x, d0[g][0], 0, 0, 0); d1[g][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(c.w1[k].x, c.x[k][g].x, d1[g][0], 0, 0, 0);
            d0[g][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(c.w0[k].y, c.x[k][g].y, d0[g][1], 0, 0, 0); d1[g][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(c.w1[k].y, c.x[k][g].y, d1[g][1], 0, 0, 0);
          }
#pragma unroll
          for (int g = 0; g < G; ++g) {
            d0[g][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(c.w0[k].z, c.x[k][g].z, d0[g][0], 0, 0, 0); d1[g][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(c.w1[k].z, c.x[k][g].z, d1[g][0], 0, 0, 0);
            d0[g][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(c.w0[k].w, c.x[k][g].w, d0[g][1], 0, 0, 0); d1[g][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(c.w1[k].w, c.x[k][g].w, d1[g][1], 0, 0, 0);
          }
        }
      }
    };
    Chunk ca, cb;
    fetch(0, ca);
    for (int q0 = 0; q0 < Kq; q0 += 2 * KC) {
      fetch(q0 + KC, cb);
      consume(q0, ca);
      fetch(q0 + 2 * KC, ca);
      consume(q0 + KC, cb);
    }
    // D of lane 4 B + j, register i = unit ub (+ 64) + 4 B + i of env 4 g + j: four units of one env = one float4 of the next layer's input
    const int q_lo = (ub >> 2) + blk, q_hi = q_lo + REX_WAVE / 4;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const pol_f4 lo = d0[g][0] + d0[g][1], hi = d1[g][0] + d1[g][1];
      if (4 * q_lo < N) out4[q_lo * E + 4 * g + jj] = make_float4(fmaxf(lo[0], 0.0f), fmaxf(lo[1], 0.0f), fmaxf(lo[2], 0.0f), fmaxf(lo[3], 0.0f));
      if (4 * q_hi < N) out4[q_hi * E + 4 * g + jj] = make_float4(fmaxf(hi[0], 0.0f), fmaxf(hi[1], 0.0f), fmaxf(hi[2], 0.0f), fmaxf(hi[3], 0.0f));
    }
  }
}

template <int LP>
__device__ __forceinline__ float lanes_sum(float v) {     // over LP adjacent lanes (1, 2, 4, 8), every lane of the run gets the total
  if (LP >= 2) v += dpp_f<kDppXor1>(v);
  if (LP >= 4) v += dpp_f<kDppXor2>(v);
  if (LP >= 8) v += dpp_f<kDppHalfMirror>(v);
  return v;
}

// One perform(): the actions of this wave's envs for the step they are about to take, into act[0..action_dim) of every lane of an env's
// group (and into action_out / mean_out by one lane per action word).  `obs_prev` rows are the observations the envs returned last
// (reset, or the previous step of this segment: written by this very wave, in front of the workgroup fence that ends a step).
// sc: this wave's LDS scratch, 16-byte aligned, policy_scratch_floats() * EPW floats.  wl: the packed actor in LDS
// (policy_weights_to_lds, shared by the workgroup's waves), or null: streamed from p.pk.
template <int EPW, int LPE, bool ARM, bool INLDS>
__device__ __forceinline__ void policy_act(const DevCfg& c, const PolDev& p, const float* wl, float* sc, int lane, int slot, int pl, int leg0, int i, bool valid,
                                           int episode, int steps, const float* obs_prev, unsigned out_off, float* act) {
  constexpr int E = EPW;
  static_assert(EPW % 4 == 0 && EPW <= 16, "lane-group kernels only");
  const int O = c.obs_dim, A = c.action_dim, H1 = p.h1, H2 = p.h2;
  const PolOff o = policy_offsets(O, A, H1, H2);
  const float* pw = INLDS ? wl : p.pk;                             // (an LDS pointer or a global one: the two instantiations keep them apart)
  float* xs = sc;
  int* meta = reinterpret_cast<int*>(sc + 4 * pol_q4(O) * E);
  float* as = reinterpret_cast<float*>(meta) + 4 * E;
  float* h1s = as + 8 * E;
  float* h2s = h1s + 4 * pol_q4(H1) * E;
  const bool in_wave = lane < LPE * EPW;
  const bool leader = in_wave && pl == 0;                          // (padding slots shadow a real env: they act on its observation)
  const bool legown = in_wave && (LPE != 8 || (pl & 1) == 0);
  REX_POL_STAMP0;
  const bool filt = p.obs_clip > 0.0f;
  auto put = [&](int k, float ob) {                                // normalize.py:47-66, into the layer input [k / 4][env][4]
    if (filt) { ob = (ob - pw[o.mean + k]) * pw[o.scale + k]; ob = fminf(fmaxf(ob, -p.obs_clip), p.obs_clip); }
    xs[((k >> 2) * E + slot) * 4 + (k & 3)] = ob;
  };
  // ---- the observation rows of the wave's envs -> LDS, the lanes that hold (stored) a word bring it ----
  if (leader) {
#pragma unroll
    for (int k = 0; k < 4; ++k) put(k, obs_prev[(unsigned)(i * O + k)]);
    for (int k = O; k < 4 * pol_q4(O); ++k) xs[((k >> 2) * E + slot) * 4 + (k & 3)] = 0.0f;      // (the zero-weighted tail of the last input quad)
    meta[slot] = i; meta[E + slot] = episode; meta[2 * E + slot] = steps; meta[3 * E + slot] = valid ? 1 : 0;
  }
  if (O > 4) {                                                     // the motor angles of the gallop observation (gallop_env.py:349-356)
    if (legown) {
#pragma unroll
      for (int jl = 0; jl < 3; ++jl) { const int k = 4 + 3 * leg0 + jl; put(k, obs_prev[(unsigned)(i * O + k)]); }
    }
    if (ARM && leader) {
#pragma unroll
      for (int a = 0; a < 6; ++a) { const int k = 16 + a; put(k, obs_prev[(unsigned)(i * O + k)]); }
    }
  }
  mirror_sync();
  REX_POL_STAMP(0);
  // ---- the two ReLU layers ----
  dense_relu_mfma<E>(pw + o.w1, pw + o.b1, pol_q4(O), H1, xs, h1s, lane);
  mirror_sync();
  REX_POL_STAMP(1);
  dense_relu_mfma<E>(pw + o.w2, pw + o.b2, pol_q4(H1), H2, h1s, h2s, lane);     // the bulk: h1 x h2 x EPW fmas
  mirror_sync();
  REX_POL_STAMP(2);
  // ---- the mean layer and the sample: LP adjacent lanes per (action word, env), each a slice of the inputs, one DPP sum ----
  auto head = [&](auto lp_tag) {
    constexpr int LP = decltype(lp_tag)::value;
    for (int q0 = 0; q0 < A * E; q0 += REX_WAVE / LP) {
      const int q = q0 + lane / LP, sub = lane & (LP - 1);
      const bool live = q < A * E;
      const int a = live ? q / E : 0, e = q & (E - 1);
      // eight partial sums p_c = sum over the inputs k = c (mod 8), c = sub + LP m, then ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7)) + bias:
      // the same additions in the same order however many lanes share them (the mean does not depend on the envs per wave)
      constexpr int NACC = 8 / LP;
      float part[NACC];
#pragma unroll
      for (int m = 0; m < NACC; ++m) part[m] = 0.0f;
      const float* w3 = pw + o.w3 + a;
      for (int k0 = 0; k0 < H2; k0 += 8) {
#pragma unroll
        for (int m = 0; m < NACC; ++m) {
          const int k = k0 + sub + LP * m;
          if (k < H2) part[m] = fmaf(h2s[((k >> 2) * E + e) * 4 + (k & 3)], w3[k * A], part[m]);
        }
      }
      float acc;
      if constexpr (LP == 8) acc = lanes_sum<8>(part[0]);
      else if constexpr (LP == 4) acc = lanes_sum<4>(part[0]) + lanes_sum<4>(part[1]);
      else if constexpr (LP == 2) acc = (lanes_sum<2>(part[0]) + lanes_sum<2>(part[1])) + (lanes_sum<2>(part[2]) + lanes_sum<2>(part[3]));
      else acc = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
      acc += pw[o.b3 + a];
      REX_POL_STAMP(3);
      const float mean = tanhf(acc);
      float action = mean;
      const int ie = meta[e];
      if (p.sample) {                                              // network.policy.sample: a diagonal normal (algorithm.py:117,493-499)
        float z[4];
        gauss4(p.seed_lo, p.seed_hi, c.env_index_base + ie, meta[E + e], meta[2 * E + e], REX_POLICY_NOISE_BLOCK + (a >> 2), z);
        const float zs = (a & 2) ? ((a & 1) ? z[3] : z[2]) : ((a & 1) ? z[1] : z[0]);
        action = fmaf(expf(pw[o.logstd + a]), zs, mean);
      }
      REX_POL_STAMP(4);
      if (live && sub == 0) {
        as[q] = action;
        if (meta[3 * E + e]) {
          p.action_out[out_off + (unsigned)(ie * A + a)] = action;
          if (p.mean_out) p.mean_out[out_off + (unsigned)(ie * A + a)] = mean;
        }
      }
    }
  };
  const int pairs = A * E;
  if (pairs <= 8) head(std::integral_constant<int, 8>{});
  else if (pairs <= 16) head(std::integral_constant<int, 4>{});
  else if (pairs <= 32) head(std::integral_constant<int, 2>{});
  else head(std::integral_constant<int, 1>{});
  REX_POL_STAMP(5);
  mirror_sync();
  REX_POL_STAMP(6);
#pragma unroll
  for (int k = 0; k < 8; ++k) act[k] = k < A ? as[k * E + slot] : 0.0f;
  mirror_sync();   // (the step's first LDS writes -- the rows of its first substep -- come behind these reads)
  REX_POL_STAMP(7);
}

}  // namespace rex
