// micro-benchmark 5 (round 5): the pipelined contact row of pgs_dv (rex_device.h) with FEWER cross-lane adds per row.
// Round 4 (row_ilp.hip, profiles/r04_microbench.md): a v_add_f32_dpp costs a lone wave 9-10 cycles whatever else is ready, the
// three of a row at 8 lanes per env are half of its 56-62 cycles.  This file measures the same row -- same arithmetic, row
// slices in registers, one wave per SIMD (1 024 one-wave workgroups) -- when an env's 9-vector (6 base + 3 own-leg components) is
// spread over
//   8 lanes (3 DPP adds; shipped at <= 8 envs per wave)         4 lanes (2 DPP adds; shipped at 16 envs per wave)
//   2 lanes (ONE DPP add: lane 0 = base 0-2 + leg 0-1, lane 1 = base 3-5 + leg 2: 5 / 4 fmas per lane)
//   2 "half groups" of an 8-lane group (the same split, every value replicated over 4 lanes: one row_half_mirror add)
//   1 lane  (NO cross-lane add: 9 fmas; the 24 x 9 row slice does not fit 256 VGPRs -- the compiler parks part of it in AGPRs)
// and, for the 2-lane forms, with the lane's partial inner product as one dependent fma chain or as two chains added.
// ACCEPT: a 24-row sweep at <= 45 cycles per row.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/row_lanes row_lanes.hip && /tmp/row_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ __forceinline__ float dppf(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// SPLIT: 8, 4, 2 (adjacent lanes), 20 (two half groups of 8 lanes), 1
template <int SPLIT> __device__ __forceinline__ float group_sum(float v) {
  if (SPLIT == 8) { v += dppf<0xB1>(v); v += dppf<0x4E>(v); v += dppf<0x141>(v); }
  if (SPLIT == 4) { v += dppf<0xB1>(v); v += dppf<0x4E>(v); }
  if (SPLIT == 2) { v += dppf<0xB1>(v); }
  if (SPLIT == 20) { v += dppf<0x141>(v); }
  return v;
}

template <int SPLIT, int ROWS, bool TREE>
__global__ __launch_bounds__(64) void k_rows(float* out, const float* __restrict__ in, int sweeps, long long* ticks) {
  constexpr int NY = SPLIT == 8 ? 1 : SPLIT == 4 ? 2 : SPLIT == 1 ? 6 : 3;   // base components per lane
  constexpr int NZ = SPLIT >= 4 && SPLIT != 20 ? 1 : SPLIT == 1 ? 3 : 2;     // own-leg components per lane (a zero pads the short lane)
  float Jy[ROWS][NY], Jz[ROWS][NZ], Kt[ROWS / 3], Ki[ROWS], cpl[ROWS], lam[ROWS];
  float ys[NY], zs[4][NZ];
  const int t = threadIdx.x;
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const float* p = in + (r * 16) * 64 + t;
#pragma unroll
    for (int i = 0; i < NY; ++i) Jy[r][i] = p[i * 64];
#pragma unroll
    for (int i = 0; i < NZ; ++i) Jz[r][i] = p[(6 + i) * 64];
    if (r < ROWS / 3) Kt[r] = p[9 * 64];
    Ki[r] = p[10 * 64]; cpl[r] = p[11 * 64]; lam[r] = 0.0f;
  }
#pragma unroll
  for (int i = 0; i < NY; ++i) ys[i] = 0.01f * (t & 7) + i;
#pragma unroll
  for (int l = 0; l < 4; ++l)
#pragma unroll
    for (int i = 0; i < NZ; ++i) zs[l][i] = 0.02f * l - 0.01f * i;
  float worst = 0.0f;
  const float thr = 1e-4f, mu = 0.5f;
  auto partial = [&](int r) __attribute__((always_inline)) {
    const int L = (r >> 1) & 3;
    float a = r < ROWS / 3 ? fmaf(Jz[r][0], zs[L][0], Kt[r]) : Jz[r][0] * zs[L][0];
    if (!TREE) {
#pragma unroll
      for (int i = 1; i < NZ; ++i) a = fmaf(Jz[r][i], zs[L][i], a);
#pragma unroll
      for (int i = 0; i < NY; ++i) a = fmaf(Jy[r][i], ys[i], a);
      return a;
    }
    float b = Jy[r][0] * ys[0];   // two chains: the leg part (+ target) and the base part
#pragma unroll
    for (int i = 1; i < NZ; ++i) a = fmaf(Jz[r][i], zs[L][i], a);
#pragma unroll
    for (int i = 1; i < NY; ++i) b = fmaf(Jy[r][i], ys[i], b);
    return a + b;
  };
  const long long t0 = clock64();
  for (int it = 0; it < sweeps; ++it) {
    float S = group_sum<SPLIT>(partial(0)), dlp = 0.0f;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int L = (r >> 1) & 3;
      const float sum = fmaf(cpl[r], dlp, S);
      float nl = fmaf(-Ki[r], sum, lam[r]);
      if (r < ROWS / 3) nl = fmaxf(nl, 0.0f);
      else { const float lm = mu * lam[(r - ROWS / 3) / 2]; nl = __builtin_amdgcn_fmed3f(nl, -lm, lm); }
      const float dl = nl - lam[r];
      if (r + 1 < ROWS) S = group_sum<SPLIT>(partial(r + 1));
      worst = fmaxf(worst, fmaf(-thr, Ki[r], fabsf(dl)));
      lam[r] = nl;
      dlp = dl;
#pragma unroll
      for (int i = 0; i < NY; ++i) ys[i] = fmaf(Jy[r][i], dl, ys[i]);
#pragma unroll
      for (int i = 0; i < NZ; ++i) zs[L][i] = fmaf(Jz[r][i], dl, zs[L][i]);
    }
  }
  const long long t1 = clock64();
  float s = worst + ys[0] + ys[NY - 1] + lam[0] + lam[ROWS - 1];
#pragma unroll
  for (int l = 0; l < 4; ++l) s += zs[l][0] + zs[l][NZ - 1];
  out[blockIdx.x * 64 + t] = s;
  if (t == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int SPLIT, int ROWS, bool TREE> void run_rows(const char* name) {
  float *out, *in; long long* ticks; long long h;
  const size_t nin = (size_t)ROWS * 16 * 64;
  (void)hipMalloc(&out, 1024 * 64 * 4); (void)hipMalloc(&ticks, 16); (void)hipMalloc(&in, nin * 4);
  float* hin = new float[nin];
  for (size_t k = 0; k < nin; ++k) hin[k] = 0.05f * (float)((k * 2654435761u >> 20) & 15) - 0.4f;
  (void)hipMemcpy(in, hin, nin * 4, hipMemcpyHostToDevice);
  const int sweeps = 200;
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k_rows<SPLIT, ROWS, TREE>), dim3(1024), dim3(64), 0, 0, out, in, sweeps, ticks); (void)hipDeviceSynchronize(); }
  (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
  hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_rows<SPLIT, ROWS, TREE>));
  printf("%-58s rows %2d %s: %.1f cycles per row   (%d registers, %zu B scratch)\n", name, ROWS, TREE ? "two chains" : "one chain ",
         (double)h / sweeps / ROWS, fa.numRegs, (size_t)fa.localSizeBytes);
  delete[] hin; (void)hipFree(out); (void)hipFree(ticks); (void)hipFree(in);
}
int main() {
  run_rows<8, 24, false>("8 lanes per env, 3 DPP adds (shipped, <= 8 envs per wave)");
  run_rows<4, 24, false>("4 lanes per env, 2 DPP adds (shipped, 16 envs per wave)");
  run_rows<2, 24, false>("2 lanes per env, 1 DPP add (quad_perm)");
  run_rows<2, 24, true>("2 lanes per env, 1 DPP add (quad_perm)");
  run_rows<20, 24, false>("2 half groups of an 8-lane group, 1 DPP add (half mirror)");
  run_rows<20, 24, true>("2 half groups of an 8-lane group, 1 DPP add (half mirror)");
  run_rows<1, 24, false>("1 lane per env, no cross-lane add");
  run_rows<1, 24, true>("1 lane per env, no cross-lane add");
  run_rows<8, 12, false>("8 lanes per env, 3 DPP adds");
  run_rows<2, 12, true>("2 lanes per env, 1 DPP add");
  run_rows<1, 12, true>("1 lane per env, no cross-lane add");
  return 0;
}
