// probe: operand / result lane layout of v_mfma_f32_4x4x1_16b_f32 (gfx950) -- which (lane, register) of D receives a[la] * b[lb]
// hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_layout tools/microbench/mfma_4x4x1_layout.hip && /tmp/mfma_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void probe(float* out) {
  const int la = blockIdx.x, lb = blockIdx.y, lane = threadIdx.x;
  const float a = lane == la ? 1.0f : 0.0f, b = lane == lb ? 1.0f : 0.0f;
  f4 c = {0.0f, 0.0f, 0.0f, 0.0f};
  const f4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  float* o = out + (((size_t)la * 64 + lb) * 64 + lane) * 4;
  o[0] = d[0]; o[1] = d[1]; o[2] = d[2]; o[3] = d[3];
}
int main() {
  float* d; const size_t n = 64ull * 64 * 64 * 4;
  (void)hipMalloc(&d, n * 4);
  hipLaunchKernelGGL(probe, dim3(64, 64), dim3(64), 0, 0, d);
  float* h = new float[n];
  (void)hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
  int shown = 0, total = 0, ok = 0;
  for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r)
    if (h[(((size_t)la * 64 + lb) * 64 + l) * 4 + r] != 0.0f) {
      ++total;
      // hypothesis: block = la / 4 = lb / 4; i = la % 4; j = lb % 4; D[i][j] in lane 4 * block + j, register i
      if (la / 4 == lb / 4 && l == 4 * (la / 4) + lb % 4 && r == la % 4) ++ok;
      if (shown < 24) { printf("a in lane %2d, b in lane %2d -> D lane %2d reg %d\n", la, lb, l, r); ++shown; }
    }
  printf("nonzero results: %d (expected 16 blocks x 16 = 256), matching 'D[i][j] of block B: lane 4B + j, register i; A[i]: lane 4B + i; B[j]: lane 4B + j': %d\n", total, ok);
  return 0;
}
