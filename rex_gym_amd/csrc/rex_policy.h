// rex_policy.h -- the ACTOR of the reference's PPO agents evaluated inside the step kernel (rex_step_policy /
// rex_step_segment_policy): what `algo.perform(prevob)` computes between two `batch_env.simulate(action)` calls of the
// reference's rollout loop (agents/tools/simulate.py:57-76, agents/ppo/algorithm.py:105-134), so that a closed-loop rollout
// -- a policy in the loop -- runs as ONE launch per segment like an open-loop one.
//
//   observ filter   agents/ppo/normalize.py:47-66: (o - mean) / std, clipped to +-5 -- the statistics are the caller's,
//                   frozen for the launch (PolDev::obs_mean / obs_scale = 1 / (std + 1e-8))
//   network         agents/scripts/networks.py:66-110 ForwardGaussianPolicy: relu(W1 x + b1) -> relu(W2 . + b2) ->
//                   mean = tanh(W3 . + b3); logstd a free vector (configs.py:31-32: 200 and 100 units)
//   action          training: mean + exp(logstd) * N(0, 1) (`network.policy.sample`, algorithm.py:117); else the mean
//
// Mapping.  A wave carries EPW envs (lane groups, rex_kernels.h) and evaluates the network for all of them at once with
// the NEURONS spread over its 64 lanes: lane l owns units l, l + 64, ... of a layer and keeps one accumulator per env;
// a layer's input -- [unit][env], envs fastest -- sits in LDS and is read as broadcast b128 words (every lane reads the
// same address), its weights come from L2 with the 64 lanes reading 64 consecutive floats of a weight row ([in][out]
// layout).  The 85 KB of weights of the 4-200-100-2 actor are shared by every wave of the launch and stay in L2; the
// hidden activations use the contact-row region of LDS, which is idle between two env steps.  At 4 envs per wave the
// upper 32 lanes -- which only repeat the lower 32 in the physics -- carry neurons of their own.
// The arithmetic (fma order: bias first, inputs ascending) does not depend on EPW, the segment length or the step's
// position in a segment: T launches of one step and one launch of T steps give the same bits.
#pragma once

namespace rex {

struct PolDev {
  const float* w1; const float* b1;      // [obs_dim][h1] (row k = the weights input k feeds), [h1]
  const float* w2; const float* b2;      // [h1][h2], [h2]
  const float* w3; const float* b3;      // [h2][action_dim], [action_dim]
  const float* logstd;                   // [action_dim]
  const float* obs_mean; const float* obs_scale;   // [obs_dim] each; obs_mean == nullptr: no observ filter
  const float* obs_in;                   // [n][obs_dim]: the observation the FIRST step of the launch acts on
  float* action_out; float* mean_out;    // [nsteps][n][action_dim] (mean_out nullable)
  int32_t h1, h2;
  float obs_clip;
  int32_t sample;                        // 1: Gaussian sample (training), 0: the mean (evaluation)
  uint32_t seed_lo, seed_hi;
};
struct NoPol {};
template <bool POLICY> struct PolArg { using type = NoPol; };
template <> struct PolArg<true> { using type = PolDev; };

// floats of LDS scratch per env of the wave: x [obs_dim], meta [4], action [8], h1, h2
__host__ __device__ __forceinline__ int policy_scratch_floats(int obs_dim, int h1, int h2) { return obs_dim + 12 + h1 + h2; }
#define REX_POLICY_NOISE_BLOCK 64   /* Philox block numbers (gauss4) of the action sample: behind the sensor-noise call sites */

// One perform(): the actions of this wave's envs for the step they are about to take, into act[0..action_dim) of every
// lane of an env's group (and into action_out / mean_out by one lane per action word).  `obs_prev` rows are the
// observations the envs returned last (reset, or the previous step of this segment: written by this very wave, in front
// of the workgroup fence that ends a step).  sc: LDS scratch, 16-byte aligned, policy_scratch_floats() * EPW floats.
template <int EPW, int LPE, bool ARM>
__device__ __forceinline__ void policy_act(const DevCfg& c, const PolDev& p, float* sc, int lane, int slot, int pl, int leg0, int i, bool valid,
                                           int episode, int steps, const float* obs_prev, unsigned out_off, float* act) {
  constexpr int E = EPW, E4 = EPW / 4;
  static_assert(EPW % 4 == 0 && EPW <= 16, "lane-group kernels only");
  const int O = c.obs_dim, A = c.action_dim, H1 = p.h1, H2 = p.h2;
  float* xs = sc;
  int* meta = reinterpret_cast<int*>(sc + O * E);
  float* as = sc + (O + 4) * E;
  float* h1s = sc + (O + 12) * E;
  float* h2s = h1s + H1 * E;
  const bool in_wave = lane < LPE * EPW;
  const bool leader = in_wave && pl == 0;                          // (padding slots shadow a real env: they act on its observation)
  const bool legown = in_wave && (LPE != 8 || (pl & 1) == 0);
  auto filtered = [&](int k, float o) {                            // normalize.py:47-66
    if (p.obs_mean) { o = (o - p.obs_mean[k]) * p.obs_scale[k]; o = fminf(fmaxf(o, -p.obs_clip), p.obs_clip); }
    return o;
  };
  // ---- the observation rows of the wave's envs -> LDS, the lanes that hold (stored) a word bring it ----
  if (leader) {
#pragma unroll
    for (int k = 0; k < 4; ++k) xs[k * E + slot] = filtered(k, obs_prev[(unsigned)(i * O + k)]);
    meta[slot] = i; meta[E + slot] = episode; meta[2 * E + slot] = steps; meta[3 * E + slot] = valid ? 1 : 0;
  }
  if (O > 4) {                                                     // the motor angles of the gallop observation (gallop_env.py:349-356)
    if (legown) {
#pragma unroll
      for (int jl = 0; jl < 3; ++jl) { const int k = 4 + 3 * leg0 + jl; xs[k * E + slot] = filtered(k, obs_prev[(unsigned)(i * O + k)]); }
    }
    if (ARM && leader) {
#pragma unroll
      for (int a = 0; a < 6; ++a) { const int k = 16 + a; xs[k * E + slot] = filtered(k, obs_prev[(unsigned)(i * O + k)]); }
    }
  }
  mirror_sync();
  // ---- layer 1 ----
  for (int j = lane; j < H1; j += REX_WAVE) {
    float acc[E];
    const float b = p.b1[j];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = b;
    for (int k = 0; k < O; ++k) {
      const float w = p.w1[k * H1 + j];
      const float4* x4 = reinterpret_cast<const float4*>(xs + k * E);
#pragma unroll
      for (int q = 0; q < E4; ++q) {
        const float4 x = x4[q];
        acc[4 * q] = fmaf(w, x.x, acc[4 * q]); acc[4 * q + 1] = fmaf(w, x.y, acc[4 * q + 1]);
        acc[4 * q + 2] = fmaf(w, x.z, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(w, x.w, acc[4 * q + 3]);
      }
    }
    float4* h4 = reinterpret_cast<float4*>(h1s + j * E);
#pragma unroll
    for (int q = 0; q < E4; ++q) h4[q] = make_float4(fmaxf(acc[4 * q], 0.0f), fmaxf(acc[4 * q + 1], 0.0f), fmaxf(acc[4 * q + 2], 0.0f), fmaxf(acc[4 * q + 3], 0.0f));
  }
  mirror_sync();
  // ---- layer 2: the bulk (h1 x h2 x EPW fmas); a lane runs two units side by side on one read of the input ----
  for (int jb = 0; jb < H2; jb += 2 * REX_WAVE) {
    const int j0 = jb + lane, j1 = jb + REX_WAVE + lane;
    const bool v0 = j0 < H2, v1 = j1 < H2;
    const int c0 = v0 ? j0 : 0, c1 = v1 ? j1 : 0;                  // (a lane without a unit computes unit 0 again and drops it)
    float a0[E], a1[E];
    const float b0 = p.b2[c0], b1 = p.b2[c1];
#pragma unroll
    for (int e = 0; e < E; ++e) { a0[e] = b0; a1[e] = b1; }
    const float* w0p = p.w2 + c0;
    const float* w1p = p.w2 + c1;
#pragma unroll 4
    for (int k = 0; k < H1; ++k) {
      const float w0 = w0p[k * H2], w1 = w1p[k * H2];
      const float4* x4 = reinterpret_cast<const float4*>(h1s + k * E);
#pragma unroll
      for (int q = 0; q < E4; ++q) {
        const float4 x = x4[q];
        a0[4 * q] = fmaf(w0, x.x, a0[4 * q]); a0[4 * q + 1] = fmaf(w0, x.y, a0[4 * q + 1]);
        a0[4 * q + 2] = fmaf(w0, x.z, a0[4 * q + 2]); a0[4 * q + 3] = fmaf(w0, x.w, a0[4 * q + 3]);
        a1[4 * q] = fmaf(w1, x.x, a1[4 * q]); a1[4 * q + 1] = fmaf(w1, x.y, a1[4 * q + 1]);
        a1[4 * q + 2] = fmaf(w1, x.z, a1[4 * q + 2]); a1[4 * q + 3] = fmaf(w1, x.w, a1[4 * q + 3]);
      }
    }
    if (v0) {
      float4* h4 = reinterpret_cast<float4*>(h2s + j0 * E);
#pragma unroll
      for (int q = 0; q < E4; ++q) h4[q] = make_float4(fmaxf(a0[4 * q], 0.0f), fmaxf(a0[4 * q + 1], 0.0f), fmaxf(a0[4 * q + 2], 0.0f), fmaxf(a0[4 * q + 3], 0.0f));
    }
    if (v1) {
      float4* h4 = reinterpret_cast<float4*>(h2s + j1 * E);
#pragma unroll
      for (int q = 0; q < E4; ++q) h4[q] = make_float4(fmaxf(a1[4 * q], 0.0f), fmaxf(a1[4 * q + 1], 0.0f), fmaxf(a1[4 * q + 2], 0.0f), fmaxf(a1[4 * q + 3], 0.0f));
    }
  }
  mirror_sync();
  // ---- the mean layer, the sample: one lane per (action word, env) ----
  for (int q = lane; q < A * E; q += REX_WAVE) {
    const int a = q / E, e = q & (E - 1);
    float acc = p.b3[a];
    for (int k = 0; k < H2; ++k) acc = fmaf(h2s[k * E + e], p.w3[k * A + a], acc);
    const float mean = tanhf(acc);
    float action = mean;
    const int ie = meta[e];
    if (p.sample) {                                                // network.policy.sample: a diagonal normal (algorithm.py:117,493-499)
      float z[4];
      gauss4(p.seed_lo, p.seed_hi, c.env_index_base + ie, meta[E + e], meta[2 * E + e], REX_POLICY_NOISE_BLOCK + (a >> 2), z);
      const float zs = (a & 2) ? ((a & 1) ? z[3] : z[2]) : ((a & 1) ? z[1] : z[0]);
      action = fmaf(expf(p.logstd[a]), zs, mean);
    }
    as[q] = action;
    if (meta[3 * E + e]) {
      p.action_out[out_off + (unsigned)(ie * A + a)] = action;
      if (p.mean_out) p.mean_out[out_off + (unsigned)(ie * A + a)] = mean;
    }
  }
  mirror_sync();
#pragma unroll
  for (int k = 0; k < 8; ++k) act[k] = k < A ? as[k * E + slot] : 0.0f;
  mirror_sync();   // (the step's first LDS writes -- the rows of its first substep -- come behind these reads)
}

}  // namespace rex
