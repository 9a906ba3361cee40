// micro-benchmark: VALU issue / dependent latency with ONE wave per SIMD (1024 one-wave blocks) on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 8192
template <int CHAINS, bool DPP>
__global__ __launch_bounds__(64) void k(float* out, float a, float b, long long* ticks) {
  float x[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 0.001f + c;
  const long long t0 = clock64();
  const long long w0 = wall_clock64();
  for (int i = 0; i < N / 16; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        if (DPP) x[c] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x[c]), 0xB1, 0xF, 0xF, true));
        else x[c] = fmaf(x[c], a, b);
      }
    }
  }
  const long long t1 = clock64();
  const long long w1 = wall_clock64();
  float s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = w1 - w0; }
}
template <int CHAINS, bool DPP>
void run(const char* name, int blocks) {
  float* out; long long* ticks; long long h[2];
  hipMalloc(&out, blocks * 64 * 4); hipMalloc(&ticks, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<CHAINS, DPP><<<blocks, 64>>>(out, 0.999f, 0.001f, ticks);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<CHAINS, DPP><<<blocks, 64>>>(out, 0.999f, 0.001f, ticks);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
  const double instr = (double)N * CHAINS;
  printf("%-28s blocks %5d: launch %.4f ms; wave: %lld clock64 ticks, %lld x10ns -> %.2f ns per instruction, %.2f clock64 ticks per instruction, clock64 tick = %.3f ns\n",
         name, blocks, ms, h[0], h[1], h[1] * 10.0 / instr, h[0] / instr, h[1] * 10.0 / h[0]);
}
int main() {
  run<1, false>("fma chain x1", 1024);
  run<2, false>("fma chains x2", 1024);
  run<4, false>("fma chains x4", 1024);
  run<8, false>("fma chains x8", 1024);
  run<1, true>("dpp-add chain x1", 1024);
  run<2, true>("dpp-add chains x2", 1024);
  run<4, true>("dpp-add chains x4", 1024);
  run<4, false>("fma chains x4, 2 waves/SIMD", 2048);
  run<1, false>("fma chain x1, 2 waves/SIMD", 2048);
  run<1, false>("fma chain x1, 4 waves/SIMD", 4096);
  return 0;
}
