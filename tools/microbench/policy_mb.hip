// micro-benchmark 6 (round 6): the actor inside the launch (csrc/rex_policy.h) -- what one perform() of a wave costs, one wave per SIMD
// (the step kernels' occupancy), with the weights
//   A  streamed from L2 every step (b128 loads of the packed [k / 4][unit][4] weights, 1 KB contiguous per wave load, two chunks in flight)
//   B  resident in LDS for the whole launch: four waves per workgroup share ONE copy (88 KB for 4-200-100-2), each wave keeps its own
//      activations; a lane's four weights are one ds_read_b128
// both on the matrix cores (v_mfma_f32_4x4x1_16b_f32: 16 blocks of 4 units x 4 envs per instruction).  Earlier forms of the same
// network, measured with this harness (docs/HISTORY.md, round 6): VALU v_pk_fma with dword weight loads from L2 41.7 k cycles per
// perform() at 4 envs per wave; VALU with the weights in LDS as [k][unit] (ds_read2st64_b32 + broadcast b128) 19-27 k, of which the
// 200 x 100 layer 19.8 k = 99 cycles per input: one wave per SIMD gets a fifth of the LDS rate for 4-byte reads.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -I rex_gym_amd/csrc -o /tmp/policy_mb tools/microbench/policy_mb.hip && /tmp/policy_mb
#define REX_POL_PROF 1
#include "rex_kernels.h"
#include <cstdio>
#include <vector>

using namespace rex;

template <int EPW>
__global__ __launch_bounds__(64) void k_stream(DevCfg c, PolDev p, const float* obs, int reps, long long* ticks, float* sink) {
  __shared__ float4 lds[120 * EPW];
  constexpr int LPE = EPW <= 8 ? 8 : 4;
  const int lane = threadIdx.x, slot = (lane / LPE) & (EPW - 1), pl = lane & (LPE - 1), leg0 = LPE == 8 ? pl >> 1 : pl;
  const int i = blockIdx.x * EPW + slot;
  float act[8], s = 0.0f;
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    policy_act<EPW, LPE, false, false>(c, p, nullptr, reinterpret_cast<float*>(lds), lane, slot, pl, leg0, i, true, 1, r, obs, 0u, act);
    s += act[0] + act[1];
  }
  const long long t1 = clock64();
  sink[blockIdx.x * 64 + lane] = s;
  if (lane == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

// B: LDS-resident weights, 4 waves per workgroup
template <int EPW>
__global__ __launch_bounds__(256) void k_lds(DevCfg c, PolDev p, const float* obs, int reps, long long* ticks, float* sink) {
  extern __shared__ float4 dyn[];
  float* wl = reinterpret_cast<float*>(dyn);                 // weights, then 4 x per-wave scratch
  const int O = c.obs_dim, A = c.action_dim, H1 = p.h1, H2 = p.h2;
  const int nw = policy_offsets(O, A, H1, H2).total;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  policy_weights_to_lds(p, nw, wl, threadIdx.x, 256);
  __syncthreads();
  float* sc = wl + ((nw + 3) & ~3) + wave * (120 * 4 * EPW);
  constexpr int LPE = EPW <= 8 ? 8 : 4;
  const int slot = (lane / LPE) & (EPW - 1), pl = lane & (LPE - 1), leg0 = LPE == 8 ? pl >> 1 : pl;
  const int i = (blockIdx.x * 4 + wave) * EPW + slot;
  float act[8], s = 0.0f;
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    policy_act<EPW, LPE, false, true>(c, p, wl, sc, lane, slot, pl, leg0, i, true, 1, r, obs, 0u, act);
    s += act[0] + act[1];
  }
  const long long t1 = clock64();
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

int main() {
  const int O = 4, A = 2, H1 = 200, H2 = 100, n = 16384;
  std::vector<float> h(64 * 1024);
  for (size_t k = 0; k < h.size(); ++k) h[k] = 0.02f * (float)((k * 2654435761u >> 20) & 31) - 0.3f;
  float* w; (void)hipMalloc(&w, h.size() * 4); (void)hipMemcpy(w, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  float *obs, *aout, *mout, *sink; long long* ticks;
  (void)hipMalloc(&obs, n * 32 * 4); (void)hipMemcpy(obs, h.data(), n * 4 * 4 < h.size() * 4 ? n * 4 * 4 : h.size() * 4, hipMemcpyHostToDevice);
  (void)hipMalloc(&aout, n * 8 * 4); (void)hipMalloc(&mout, n * 8 * 4); (void)hipMalloc(&sink, 1024 * 64 * 4); (void)hipMalloc(&ticks, 16);
  DevCfg c; memset(&c, 0, sizeof c); c.n = n; c.obs_dim = O; c.action_dim = A;
  PolDev p; memset(&p, 0, sizeof p);
  p.pk = w;      // (any finite numbers in the packed layout will do)
  p.obs_in = obs; p.action_out = aout; p.mean_out = mout; p.h1 = H1; p.h2 = H2; p.obs_clip = 5.0f; p.sample = 1; p.seed_lo = 1; p.seed_hi = 2;
  const int reps = 50;
  long long hticks;
  auto report = [&](const char* name, int epw) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&hticks, ticks, 8, hipMemcpyDeviceToHost);
    printf("%-60s %2d envs per wave: %8.0f cycles per perform()  (%s)\n", name, epw, (double)hticks / reps, hipGetErrorString(hipGetLastError()));
  };
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_stream<4>, dim3(1024), dim3(64), 0, 0, c, p, obs, reps, ticks, sink);
  report("A weights streamed from L2", 4);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_stream<8>, dim3(1024), dim3(64), 0, 0, c, p, obs, reps, ticks, sink);
  report("A weights streamed from L2", 8);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_stream<16>, dim3(1024), dim3(64), 0, 0, c, p, obs, reps, ticks, sink);
  report("A weights streamed from L2", 16);
  const int nw = policy_offsets(O, A, H1, H2).total;
  {
    const size_t bytes = (size_t)(nw + 4 * 120 * 4 * 4) * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    { long long z[8] = {0}; (void)hipDeviceSynchronize(); (void)hipMemcpyToSymbol(HIP_SYMBOL(rex::g_pol_prof), z, sizeof z); }
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_lds<4>, dim3(256), dim3(256), bytes, 0, c, p, obs, reps, ticks, sink);
    report("B weights resident in LDS, 4 waves per workgroup", 4);
    long long pp[8];
    (void)hipMemcpyFromSymbol(pp, HIP_SYMBOL(rex::g_pol_prof), sizeof pp);
    printf("   sections (cycles per perform(), 2 launches): obs->LDS %lld | layer 1 %lld | layer 2 %lld | mean layer %lld | tanh + sample %lld | stores %lld | fence %lld | act %lld\n",
           pp[0] / (2 * reps), pp[1] / (2 * reps), pp[2] / (2 * reps), pp[3] / (2 * reps), pp[4] / (2 * reps), pp[5] / (2 * reps), pp[6] / (2 * reps), pp[7] / (2 * reps));
    printf("   (stamped by wave 0 of workgroup 0 only)\n");
  }
  {
    const size_t bytes = (size_t)(nw + 4 * 120 * 4 * 8) * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_lds<8>, dim3(256), dim3(256), bytes, 0, c, p, obs, reps, ticks, sink);
    report("B weights resident in LDS, 4 waves per workgroup", 8);
  }
  return 0;
}
