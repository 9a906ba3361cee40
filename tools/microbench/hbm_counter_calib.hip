// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS library's access pattern (VERDICT round 3, item 7;
// MI355X_MICROARCH.md: "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own
// access pattern").  Kernels that move a known number of bytes:
//   calib_words<EPW>  the step kernel's state traffic: a word-major [W][N] fp32 block, 64-thread workgroups of EPW envs, every
//                     lane of an env's lane group loads the env's W words with 4-byte loads (the group's lanes read the same
//                     addresses), lane 0 of the group stores W words to a second block; workgroup -> env mapping as in
//                     rex_step_kernel (the workgroups of one 64-byte sector share an XCD).  Bytes: W N 4 read, W N 4 written.
//   calib_stream      a wide streaming copy, 16 bytes per lane (the pattern the guide's factor 1/2 was measured on).
// Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace` (separate passes), then
// tools/hbm_calib.py turns the two databases into counter-per-byte factors.
//   hipcc --offload-arch=gfx950 -O3 -o hbm_counter_calib hbm_counter_calib.hip && ./hbm_counter_calib 4096 262144
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define W 54   /* state words of mark 'base' (include/rexsim.h) */

template <int EPW>
__global__ __launch_bounds__(64) void calib_words(const float* __restrict__ in, float* __restrict__ out, int n) {
  constexpr int LPE = EPW <= 8 ? 8 : 4;
  const int lane = threadIdx.x, slot = (lane / LPE) & (EPW - 1), pl = lane & (LPE - 1);
  int blk = (int)blockIdx.x;
  if (EPW < 16) {
    constexpr int G = 16 / EPW;
    const int full = ((int)gridDim.x / (8 * G)) * (8 * G);
    if (blk < full) { const int xcd = blk & 7, q = blk >> 3; blk = ((q / G) * 8 + xcd) * G + (q % G); }
  }
  const int gi = blk * EPW + slot;
  const int i = gi < n ? gi : n - 1;
  float v[W];
#pragma unroll
  for (int w = 0; w < W; ++w) v[w] = in[(unsigned)(w * n + i)];
  float s = 0.0f;
#pragma unroll
  for (int w = 0; w < W; ++w) s += v[w];
  if (lane < LPE * EPW && gi < n && pl == 0) {
#pragma unroll
    for (int w = 0; w < W; ++w) out[(unsigned)(w * n + i)] = v[w] + s;
  }
}
__global__ __launch_bounds__(256) void calib_stream(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) { float4 v = in[i]; v.x += 1.0f; out[i] = v; }
}
template <int EPW> void run_words(const float* in, float* out, int n, int reps) {
  for (int r = 0; r < reps; ++r) calib_words<EPW><<<(n + EPW - 1) / EPW, 64>>>(in, out, n);
  (void)hipDeviceSynchronize();
}
int main(int argc, char** argv) {
  const int reps = 12;
  for (int a = 1; a < argc; ++a) {
    const int n = atoi(argv[a]);
    const size_t floats = (size_t)W * n;
    float *in, *out;
    (void)hipMalloc(&in, floats * 4); (void)hipMalloc(&out, floats * 4);
    (void)hipMemset(in, 0, floats * 4); (void)hipMemset(out, 0, floats * 4);
    (void)hipDeviceSynchronize();
    if (n <= 4096) run_words<4>(in, out, n, reps); else if (n <= 8192) run_words<8>(in, out, n, reps); else run_words<16>(in, out, n, reps);
    for (int r = 0; r < reps; ++r) calib_stream<<<(unsigned)((floats / 4 + 255) / 256), 256>>>((const float4*)in, (float4*)out, floats / 4);
    (void)hipDeviceSynchronize();
    printf("{\"n\": %d, \"words\": %d, \"bytes_read\": %zu, \"bytes_written\": %zu, \"launches_each\": %d}\n", n, W, floats * 4, floats * 4, reps);
    (void)hipFree(in); (void)hipFree(out);
  }
  return 0;
}
