// micro-benchmark 3 (round 4): does instruction-level parallelism pay on a lone wave?  One wave per SIMD (1 024 one-wave
// workgroups), as the step kernel runs.
//   (a) v_fma_f32 with THREE VGPR operands, 1 / 2 / 4 / 8 independent chains (round 3 measured SGPR operands only);
//   (b) v_add_f32_dpp on 1 / 2 / 4 / 8 independent chains;
//   (c) the pipelined contact row of pgs_dv (rex_device.h; 8 lanes per env: one base component per lane, 3-step DPP group
//       sum, coupling fma, impulse fma, clamp, difference, residual test, two update fmas) for ONE env per lane group and
//       for TWO independent envs interleaved in the same lanes (two register sets, the rows of the two envs alternating).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/row_ilp row_ilp.hip && /tmp/row_ilp
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
template <int CTRL> __device__ __forceinline__ float dppf(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float group_sum8(float v) { v += dppf<0xB1>(v); v += dppf<0x4E>(v); v += dppf<0x141>(v); return v; }
__device__ __forceinline__ float group_sum4(float v) { v += dppf<0xB1>(v); v += dppf<0x4E>(v); return v; }

template <int MODE, int CHAINS>   // MODE 0: fma, three VGPR operands; 1: v_add_f32_dpp
__global__ __launch_bounds__(64) void k_chain(float* out, long long* ticks) {
  float x[CHAINS], y[CHAINS], z[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) {
    x[c] = threadIdx.x * 0.001f + c; y[c] = 0.999f - 1e-6f * threadIdx.x; z[c] = 1e-3f * (c + 1);
    asm volatile("" : "+v"(y[c]), "+v"(z[c]));     // keep the operands in vector registers
  }
  const long long t0 = clock64();
  for (int i = 0; i < N / 16; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) x[c] = MODE == 0 ? fmaf(x[c], y[c], z[c]) : x[c] + dppf<0xB1>(x[c]);
    }
  }
  const long long t1 = clock64();
  float s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

// the contact sweep of pgs_dv, ROWS rows, ENVS independent envs per lane group, LPE lanes per env (8: one base component per
// lane; 4: two)
template <int ENVS, int ROWS, int LPE>
__global__ __launch_bounds__(64) void k_rows(float* out, const float* __restrict__ in, int sweeps, long long* ticks) {
  constexpr int NY = LPE == 8 ? 1 : 2;
  float Jy[ENVS][ROWS][NY], Jz[ENVS][ROWS], Kt[ENVS][ROWS], Ki[ENVS][ROWS], cpl[ENVS][ROWS], lam[ENVS][ROWS];
  float ys[ENVS][NY], zs[ENVS][4];
  const int t = threadIdx.x;
#pragma unroll
  for (int e = 0; e < ENVS; ++e) {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const float* p = in + ((e * ROWS + r) * 8) * 64 + t;
#pragma unroll
      for (int i = 0; i < NY; ++i) Jy[e][r][i] = p[i * 64];
      Jz[e][r] = p[2 * 64]; Kt[e][r] = p[3 * 64]; Ki[e][r] = p[4 * 64]; cpl[e][r] = p[5 * 64]; lam[e][r] = 0.0f;
    }
#pragma unroll
    for (int i = 0; i < NY; ++i) ys[e][i] = 0.01f * (t & 7) + e + i;
#pragma unroll
    for (int l = 0; l < 4; ++l) zs[e][l] = 0.02f * l - e;
  }
  float worst = 0.0f;
  const float thr = 1e-4f, mu = 0.5f;
  const long long t0 = clock64();
  for (int it = 0; it < sweeps; ++it) {
    float S[ENVS], dlp[ENVS];
#pragma unroll
    for (int e = 0; e < ENVS; ++e) {
      float part = fmaf(Jz[e][0], zs[e][0], Kt[e][0]);
#pragma unroll
      for (int i = 0; i < NY; ++i) part = fmaf(Jy[e][0][i], ys[e][i], part);
      S[e] = LPE == 8 ? group_sum8(part) : group_sum4(part);
      dlp[e] = 0.0f;
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
#pragma unroll
      for (int e = 0; e < ENVS; ++e) {      // the rows of the envs alternate: row r of env 0, row r of env 1, ...
        const int L = (r >> 1) & 3;
        const float sum = fmaf(cpl[e][r], dlp[e], S[e]);
        float nl = fmaf(-Ki[e][r], sum, lam[e][r]);
        if (r < ROWS / 3) nl = fmaxf(nl, 0.0f);
        else { const float lm = mu * lam[e][(r - ROWS / 3) / 2]; nl = __builtin_amdgcn_fmed3f(nl, -lm, lm); }
        const float dl = nl - lam[e][r];
        if (r + 1 < ROWS) {
          float part = fmaf(Jz[e][r + 1], zs[e][((r + 1) >> 1) & 3], Kt[e][r + 1]);
#pragma unroll
          for (int i = 0; i < NY; ++i) part = fmaf(Jy[e][r + 1][i], ys[e][i], part);
          S[e] = LPE == 8 ? group_sum8(part) : group_sum4(part);
        }
        worst = fmaxf(worst, fmaf(-thr, Ki[e][r], fabsf(dl)));
        lam[e][r] = nl;
        dlp[e] = dl;
#pragma unroll
        for (int i = 0; i < NY; ++i) ys[e][i] = fmaf(Jy[e][r][i], dl, ys[e][i]);
        zs[e][L] = fmaf(Jz[e][r], dl, zs[e][L]);
      }
    }
  }
  const long long t1 = clock64();
  float s = worst;
#pragma unroll
  for (int e = 0; e < ENVS; ++e) { s += ys[e][0] + zs[e][0] + zs[e][1] + zs[e][2] + zs[e][3] + lam[e][0] + lam[e][ROWS - 1]; }
  out[blockIdx.x * 64 + t] = s;
  if (t == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int MODE, int CHAINS> void run_chain(const char* name) {
  float* out; long long* ticks; long long h;
  (void)hipMalloc(&out, 1024 * 64 * 4); (void)hipMalloc(&ticks, 16);
  for (int rep = 0; rep < 2; ++rep) { k_chain<MODE, CHAINS><<<1024, 64>>>(out, ticks); (void)hipDeviceSynchronize(); }
  (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
  printf("%-34s chains %d: %.2f cycles per chain step, %.2f per instruction issued\n", name, CHAINS, (double)h / N, (double)h / N / CHAINS);
  (void)hipFree(out); (void)hipFree(ticks);
}
template <int ENVS, int ROWS, int LPE> void run_rows(const char* name) {
  float *out, *in; long long* ticks; long long h;
  const size_t nin = (size_t)ENVS * ROWS * 8 * 64;
  (void)hipMalloc(&out, 1024 * 64 * 4); (void)hipMalloc(&ticks, 16); (void)hipMalloc(&in, nin * 4);
  float* hin = new float[nin];
  for (size_t k = 0; k < nin; ++k) hin[k] = 0.05f * (float)((k * 2654435761u >> 20) & 15) - 0.4f;
  (void)hipMemcpy(in, hin, nin * 4, hipMemcpyHostToDevice);
  const int sweeps = 200;
  for (int rep = 0; rep < 2; ++rep) { k_rows<ENVS, ROWS, LPE><<<1024, 64>>>(out, in, sweeps, ticks); (void)hipDeviceSynchronize(); }
  (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
  printf("%-46s envs %d rows %2d lanes/env %d: %.1f cycles per row of one env, %.1f per row issued\n", name, ENVS, ROWS, LPE,
         (double)h / sweeps / ROWS, (double)h / sweeps / ROWS / ENVS);
  delete[] hin; (void)hipFree(out); (void)hipFree(ticks); (void)hipFree(in);
}
int main() {
  run_chain<0, 1>("fma, 3 VGPR operands"); run_chain<0, 2>("fma, 3 VGPR operands"); run_chain<0, 4>("fma, 3 VGPR operands"); run_chain<0, 8>("fma, 3 VGPR operands");
  run_chain<1, 1>("v_add_f32_dpp quad_perm"); run_chain<1, 2>("v_add_f32_dpp quad_perm"); run_chain<1, 4>("v_add_f32_dpp quad_perm"); run_chain<1, 8>("v_add_f32_dpp quad_perm");
  run_rows<1, 12, 8>("contact rows, one env per lane group");
  run_rows<2, 12, 8>("contact rows, two envs interleaved");
  run_rows<1, 24, 8>("contact rows, one env per lane group");
  run_rows<2, 24, 8>("contact rows, two envs interleaved");
  run_rows<1, 12, 4>("contact rows, one env per lane group");
  run_rows<2, 12, 4>("contact rows, two envs interleaved");
  run_rows<1, 24, 4>("contact rows, one env per lane group");
  run_rows<2, 24, 4>("contact rows, two envs interleaved");
  return 0;
}
