// micro-benchmark 2: cost of cross-lane sum steps with ONE wave per SIMD on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
typedef float v4f __attribute__((ext_vector_type(4)));
template <int CTRL> __device__ __forceinline__ float dppf(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int MODE> __device__ __forceinline__ float step(float x, float a) {
  if (MODE == 0) return x + dppf<0xB1>(x);                 // v_add_f32_dpp quad_perm [1,0,3,2]
  if (MODE == 1) return x + dppf<0x141>(x);                // row_half_mirror
  if (MODE == 2) return x + dppf<0x111>(x);                // row_shr:1
  if (MODE == 3) { float t; asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(t) : "v"(x)); return x + t; }
  if (MODE == 4) return x + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), 0x041F));   // swap adjacent lanes (bit mode xor 1)
  if (MODE == 5) return x + __shfl_xor(x, 1);
  if (MODE == 6) {   // 4x4x1 MFMA all-gather of the 4-lane block, then a tree of adds
    v4f c = {0.f, 0.f, 0.f, 0.f};
    v4f d = __builtin_amdgcn_mfma_f32_4x4x1f32(x, 1.0f, c, 0, 0, 0);
    return (d[0] + d[1]) + (d[2] + d[3]);
  }
  if (MODE == 7) return fmaf(x, a, 0.001f);
  if (MODE == 8) {   // the 3-step butterfly of 8 lanes as one unit
    x += dppf<0xB1>(x); x += dppf<0x4E>(x); x += dppf<0x141>(x); return x * a;
  }
  if (MODE == 9) {   // readlane-based: uniform broadcast (not a sum; cost reference)
    return x + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 1));
  }
  return x;
}
template <int MODE, int CHAINS>
__global__ __launch_bounds__(64) void k(float* out, float a, long long* ticks) {
  float x[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 0.001f + c;
  const long long t0 = clock64();
  for (int i = 0; i < N / 8; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) x[c] = step<MODE>(x[c], a) * (MODE == 7 ? 1.0f : 1.0f);
    }
  }
  const long long t1 = clock64();
  float s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
template <int MODE, int CHAINS>
void run(const char* name) {
  float* out; long long* ticks; long long h;
  (void)hipMalloc(&out, 1024 * 64 * 4); (void)hipMalloc(&ticks, 16);
  k<MODE, CHAINS><<<1024, 64>>>(out, 0.5f, ticks);
  (void)hipDeviceSynchronize();
  k<MODE, CHAINS><<<1024, 64>>>(out, 0.5f, ticks);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
  printf("%-46s chains %d: %.2f cycles per step per chain-step (%.2f per step issued)\n", name, CHAINS, (double)h / N, (double)h / N / CHAINS);
}
int main() {
  run<7, 1>("fma (reference)"); run<7, 4>("fma (reference)");
  run<0, 1>("v_add_f32_dpp quad_perm"); run<0, 4>("v_add_f32_dpp quad_perm");
  run<1, 1>("v_add_f32_dpp row_half_mirror"); run<1, 4>("v_add_f32_dpp row_half_mirror");
  run<2, 1>("v_add_f32_dpp row_shr:1"); run<2, 4>("v_add_f32_dpp row_shr:1");
  run<3, 1>("v_mov_b32_dpp + v_add"); run<3, 4>("v_mov_b32_dpp + v_add");
  run<4, 1>("ds_swizzle + v_add"); run<4, 4>("ds_swizzle + v_add");
  run<5, 1>("ds_bpermute (shfl_xor) + v_add"); run<5, 4>("ds_bpermute (shfl_xor) + v_add");
  run<6, 1>("mfma 4x4x1 gather + 3 adds (sum of 4 lanes)"); run<6, 4>("mfma 4x4x1 gather + 3 adds (sum of 4 lanes)");
  run<8, 1>("3 dpp adds + mul (sum of 8 lanes)"); run<8, 2>("3 dpp adds + mul (sum of 8 lanes)"); run<8, 4>("3 dpp adds + mul (sum of 8 lanes)");
  run<9, 1>("v_readlane + v_add"); run<9, 4>("v_readlane + v_add");
  return 0;
}
