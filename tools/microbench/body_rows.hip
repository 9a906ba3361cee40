// micro-benchmark 4 (round 4, preparation for RexPosesEnv): what a constraint row costs when its slice comes from LDS.
// One wave per SIMD, 4 envs per wave, 8 lanes per env (the `<4, base, BODY>` kernel's shape): a lane owns one base component
// (word oy of a row) and one leg component (word oz); rows are float4[row][3 chunks][4 slots] in LDS as in rex_device.h.
//   A  the shipped form of body_row / the joint-limit rows: every row statically unrolled behind a wave-uniform test, its three LDS
//      reads issued right before use, impulses in registers;
//   B  a software-pipelined loop over a wave-uniform LIST of the occupied rows (in sweep order): the slice, the impulse and the
//      friction limit of row i+1 are read from LDS while row i is solved; impulses and limits live in LDS; the leg whose velocity
//      the row touches is a scalar (wave-uniform) selector.
//   C  B with the list held in scalar registers and the loop unrolled (no scalar memory load inside the loop).
// Reports cycles per row for 12 and 36 occupied rows.   hipcc --offload-arch=gfx950 -O3 -o /tmp/body_rows body_rows.hip && /tmp/body_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#define EPW 4
#define NROWS 36
#define ROWB (3 * EPW * 16)   /* bytes from a row to the next */
template <int CTRL> __device__ __forceinline__ float dppf(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float group_sum8(float v) { v += dppf<0xB1>(v); v += dppf<0x4E>(v); v += dppf<0x141>(v); return v; }
__device__ __forceinline__ float ldsf(const char* base, int off) { return *reinterpret_cast<const float*>(base + off); }

__device__ __forceinline__ void fill(float4* rows, float* lam, float* lim, int lane) {
  for (int k = lane; k < NROWS * 3 * EPW; k += 64) rows[k] = make_float4(0.01f * (k % 7) - 0.02f, 0.02f * (k % 5), 0.5f + 0.01f * (k % 3), 0.0f);
  for (int k = lane; k < NROWS * EPW; k += 64) { lam[k] = 0.0f; lim[k] = 0.3f; }
  __syncthreads();
}

template <int R>   // A: static rows, loads next to their use, impulses in registers
__global__ __launch_bounds__(64) void k_static(float* out, unsigned mask_lo, unsigned mask_hi, int sweeps, long long* ticks) {
  __shared__ float4 rows[NROWS * 3 * EPW];
  __shared__ float lamL[NROWS * EPW], limL[NROWS * EPW];
  const int lane = threadIdx.x, slot = (lane >> 3) & 3, p = lane & 7;
  fill(rows, lamL, limL, lane);
  const char* base = reinterpret_cast<const char*>(rows);
  const int oy = (((p < 6 ? p : 11) >> 2) * EPW + slot) * 16 + ((p < 6 ? p : 11) & 3) * 4;
  const int oz = (((p < 3 ? 6 + p : 11) >> 2) * EPW + slot) * 16 + ((p < 3 ? 6 + p : 11) & 3) * 4;
  float ys = 0.01f * p, zs[4] = {0.1f, 0.2f, 0.3f, 0.4f}, lam[R], worst = 0.0f;
#pragma unroll
  for (int r = 0; r < R; ++r) lam[r] = 0.0f;
  const unsigned long long mask = ((unsigned long long)mask_hi << 32) | mask_lo;
  const long long t0 = clock64();
  for (int it = 0; it < sweeps; ++it) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (!((mask >> r) & 1ull)) continue;                       // wave-uniform
      const int leg = (r >> 1) & 3;
      const float jy = ldsf(base, r * ROWB + oy), jz = ldsf(base, r * ROWB + oz);
      const float4 c2 = rows[(r * 3 + 2) * EPW + slot];
      const float vel = group_sum8(fmaf(jy, ys, jz * zs[leg]));
      float nl = fmaf(-c2.z, vel, lam[r] + c2.y);
      if (r < R / 3) nl = fmaxf(nl, 0.0f); else { const float lm = 0.5f * lam[(r - R / 3) / 2]; nl = __builtin_amdgcn_fmed3f(nl, -lm, lm); }
      const float dl = nl - lam[r];
      lam[r] = nl;
      worst = fmaxf(worst, fmaf(-1e-4f, c2.z, fabsf(dl)));
      ys = fmaf(jy, dl, ys); zs[leg] = fmaf(jz, dl, zs[leg]);
    }
  }
  const long long t1 = clock64();
  out[blockIdx.x * 64 + lane] = ys + zs[0] + zs[1] + zs[2] + zs[3] + worst + lam[0] + lam[R - 1];
  if (lane == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

// B: pipelined loop over a uniform row list; impulses / friction limits in LDS
__global__ __launch_bounds__(64) void k_list(float* out, const int* __restrict__ list, int n, int nnormal, int sweeps, long long* ticks) {
  __shared__ float4 rows[NROWS * 3 * EPW];
  __shared__ float lamL[NROWS * EPW], limL[NROWS * EPW];
  const int lane = threadIdx.x, slot = (lane >> 3) & 3, p = lane & 7;
  fill(rows, lamL, limL, lane);
  const char* base = reinterpret_cast<const char*>(rows);
  const int oy = (((p < 6 ? p : 11) >> 2) * EPW + slot) * 16 + ((p < 6 ? p : 11) & 3) * 4;
  const int oz = (((p < 3 ? 6 + p : 11) >> 2) * EPW + slot) * 16 + ((p < 3 ? 6 + p : 11) & 3) * 4;
  float ys = 0.01f * p, z0 = 0.1f, z1 = 0.2f, z2 = 0.3f, z3 = 0.4f, worst = 0.0f;
  const long long t0 = clock64();
  for (int it = 0; it < sweeps; ++it) {
    int r = list[0];                                                      // scalar loads: the list is wave-uniform
    float jy = ldsf(base, r * ROWB + oy), jz = ldsf(base, r * ROWB + oz);
    float4 c2 = rows[(r * 3 + 2) * EPW + slot];
    float lam = lamL[r * EPW + slot], lim = limL[r * EPW + slot];
    for (int i = 0; i < n; ++i) {
      const int rn = list[i + 1 < n ? i + 1 : i];
      // the next row's slice, impulse and limit: issued now, needed one row later
      const float jyn = ldsf(base, rn * ROWB + oy), jzn = ldsf(base, rn * ROWB + oz);
      const float4 c2n = rows[(rn * 3 + 2) * EPW + slot];
      const float lamn = lamL[rn * EPW + slot], limn = limL[rn * EPW + slot];
      const int leg = (r >> 1) & 3;                                       // scalar
      const float zl = leg == 0 ? z0 : (leg == 1 ? z1 : (leg == 2 ? z2 : z3));
      const float vel = group_sum8(fmaf(jy, ys, jz * zl));
      float nl = fmaf(-c2.z, vel, lam + c2.y);
      const bool normal = i < nnormal;                                    // scalar
      nl = normal ? fmaxf(nl, 0.0f) : __builtin_amdgcn_fmed3f(nl, -lim, lim);
      const float dl = nl - lam;
      lamL[r * EPW + slot] = nl;
      if (normal) limL[(nnormal + 2 * i) * EPW + slot] = 0.5f * nl;       // (the friction rows of this point read it later in the sweep)
      worst = fmaxf(worst, fmaf(-1e-4f, c2.z, fabsf(dl)));
      ys = fmaf(jy, dl, ys);
      const float zn = fmaf(jz, dl, zl);
      z0 = leg == 0 ? zn : z0; z1 = leg == 1 ? zn : z1; z2 = leg == 2 ? zn : z2; z3 = leg == 3 ? zn : z3;
      r = rn; jy = jyn; jz = jzn; c2 = c2n; lam = lamn; lim = limn;
    }
  }
  const long long t1 = clock64();
  out[blockIdx.x * 64 + lane] = ys + z0 + z1 + z2 + z3 + worst;
  if (lane == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

// C: as B, but the list is read into scalar registers ONCE per substep and the loop over it is unrolled (no scalar memory load in
//    the loop: s_load and LDS share lgkmcnt and return out of order, so a wait for the list entry is a wait for every prefetch)
__global__ __launch_bounds__(64) void k_list_sgpr(float* out, const int* __restrict__ list, int n, int nnormal, int sweeps, long long* ticks) {
  __shared__ float4 rows[NROWS * 3 * EPW];
  __shared__ float lamL[NROWS * EPW], limL[NROWS * EPW];
  const int lane = threadIdx.x, slot = (lane >> 3) & 3, p = lane & 7;
  fill(rows, lamL, limL, lane);
  const char* base = reinterpret_cast<const char*>(rows);
  const int oy = (((p < 6 ? p : 11) >> 2) * EPW + slot) * 16 + ((p < 6 ? p : 11) & 3) * 4;
  const int oz = (((p < 3 ? 6 + p : 11) >> 2) * EPW + slot) * 16 + ((p < 3 ? 6 + p : 11) & 3) * 4;
  float ys = 0.01f * p, z0 = 0.1f, z1 = 0.2f, z2 = 0.3f, z3 = 0.4f, worst = 0.0f;
  int L[NROWS + 1];
#pragma unroll
  for (int i = 0; i < NROWS; ++i) L[i] = __builtin_amdgcn_readfirstlane(list[i < n ? i : n - 1]);
  L[NROWS] = L[NROWS - 1];
  const long long t0 = clock64();
  for (int it = 0; it < sweeps; ++it) {
    float jy = ldsf(base, L[0] * ROWB + oy), jz = ldsf(base, L[0] * ROWB + oz);
    float4 c2 = rows[(L[0] * 3 + 2) * EPW + slot];
    float lam = lamL[L[0] * EPW + slot], lim = limL[L[0] * EPW + slot];
#pragma unroll
    for (int i = 0; i < NROWS; ++i) {
      if (i >= n) break;                                                  // wave-uniform
      const int r = L[i], rn = L[i + 1];
      const float jyn = ldsf(base, rn * ROWB + oy), jzn = ldsf(base, rn * ROWB + oz);
      const float4 c2n = rows[(rn * 3 + 2) * EPW + slot];
      const float lamn = lamL[rn * EPW + slot], limn = limL[rn * EPW + slot];
      const int leg = (r >> 1) & 3;
      const float zl = leg == 0 ? z0 : (leg == 1 ? z1 : (leg == 2 ? z2 : z3));
      const float vel = group_sum8(fmaf(jy, ys, jz * zl));
      float nl = fmaf(-c2.z, vel, lam + c2.y);
      const bool normal = i < nnormal;
      nl = normal ? fmaxf(nl, 0.0f) : __builtin_amdgcn_fmed3f(nl, -lim, lim);
      const float dl = nl - lam;
      lamL[r * EPW + slot] = nl;
      if (normal) limL[(nnormal + 2 * i) * EPW + slot] = 0.5f * nl;
      worst = fmaxf(worst, fmaf(-1e-4f, c2.z, fabsf(dl)));
      ys = fmaf(jy, dl, ys);
      const float zn = fmaf(jz, dl, zl);
      z0 = leg == 0 ? zn : z0; z1 = leg == 1 ? zn : z1; z2 = leg == 2 ? zn : z2; z3 = leg == 3 ? zn : z3;
      jy = jyn; jz = jzn; c2 = c2n; lam = lamn; lim = limn;
    }
  }
  const long long t1 = clock64();
  out[blockIdx.x * 64 + lane] = ys + z0 + z1 + z2 + z3 + worst;
  if (lane == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

// E (not yet measured: round 4 ran out of GPU minutes -- a candidate for the next round): A with the slice of row r+1 read while
//    row r is solved, unconditionally (the rows stay static, the impulses in registers; a prefetch of an unoccupied row is wasted)
template <int R>
__global__ __launch_bounds__(64) void k_static_ahead(float* out, unsigned mask_lo, unsigned mask_hi, int sweeps, long long* ticks) {
  __shared__ float4 rows[NROWS * 3 * EPW];
  __shared__ float lamL[NROWS * EPW], limL[NROWS * EPW];
  const int lane = threadIdx.x, slot = (lane >> 3) & 3, p = lane & 7;
  fill(rows, lamL, limL, lane);
  const char* base = reinterpret_cast<const char*>(rows);
  const int oy = (((p < 6 ? p : 11) >> 2) * EPW + slot) * 16 + ((p < 6 ? p : 11) & 3) * 4;
  const int oz = (((p < 3 ? 6 + p : 11) >> 2) * EPW + slot) * 16 + ((p < 3 ? 6 + p : 11) & 3) * 4;
  float ys = 0.01f * p, zs[4] = {0.1f, 0.2f, 0.3f, 0.4f}, lam[R], worst = 0.0f;
#pragma unroll
  for (int r = 0; r < R; ++r) lam[r] = 0.0f;
  const unsigned long long mask = ((unsigned long long)mask_hi << 32) | mask_lo;
  const long long t0 = clock64();
  for (int it = 0; it < sweeps; ++it) {
    float jy = ldsf(base, oy), jz = ldsf(base, oz);
    float4 c2 = rows[2 * EPW + slot];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int rn = r + 1 < R ? r + 1 : r;
      const float jyn = ldsf(base, rn * ROWB + oy), jzn = ldsf(base, rn * ROWB + oz);
      const float4 c2n = rows[(rn * 3 + 2) * EPW + slot];
      if ((mask >> r) & 1ull) {                                  // wave-uniform
        const int leg = (r >> 1) & 3;
        const float vel = group_sum8(fmaf(jy, ys, jz * zs[leg]));
        float nl = fmaf(-c2.z, vel, lam[r] + c2.y);
        if (r < R / 3) nl = fmaxf(nl, 0.0f); else { const float lm = 0.5f * lam[(r - R / 3) / 2]; nl = __builtin_amdgcn_fmed3f(nl, -lm, lm); }
        const float dl = nl - lam[r];
        lam[r] = nl;
        worst = fmaxf(worst, fmaf(-1e-4f, c2.z, fabsf(dl)));
        ys = fmaf(jy, dl, ys); zs[leg] = fmaf(jz, dl, zs[leg]);
      }
      jy = jyn; jz = jzn; c2 = c2n;
    }
  }
  const long long t1 = clock64();
  out[blockIdx.x * 64 + lane] = ys + zs[0] + zs[1] + zs[2] + zs[3] + worst + lam[0] + lam[R - 1];
  if (lane == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

// D: C with the list where the step kernel would have it -- in scalar registers for real (a kernel argument by value; C's local copy
//    was re-read from memory by the compiler): no memory load on the way to the next row's LDS addresses
struct RowList { int r[NROWS + 1]; };
__global__ __launch_bounds__(64) void k_list_args(float* out, RowList L, int n, int nnormal, int sweeps, long long* ticks) {
  __shared__ float4 rows[NROWS * 3 * EPW];
  __shared__ float lamL[NROWS * EPW], limL[NROWS * EPW];
  const int lane = threadIdx.x, slot = (lane >> 3) & 3, p = lane & 7;
  fill(rows, lamL, limL, lane);
  const char* base = reinterpret_cast<const char*>(rows);
  const int oy = (((p < 6 ? p : 11) >> 2) * EPW + slot) * 16 + ((p < 6 ? p : 11) & 3) * 4;
  const int oz = (((p < 3 ? 6 + p : 11) >> 2) * EPW + slot) * 16 + ((p < 3 ? 6 + p : 11) & 3) * 4;
  float ys = 0.01f * p, z0 = 0.1f, z1 = 0.2f, z2 = 0.3f, z3 = 0.4f, worst = 0.0f;
  const long long t0 = clock64();
  for (int it = 0; it < sweeps; ++it) {
    float jy = ldsf(base, L.r[0] * ROWB + oy), jz = ldsf(base, L.r[0] * ROWB + oz);
    float4 c2 = rows[(L.r[0] * 3 + 2) * EPW + slot];
    float lam = lamL[L.r[0] * EPW + slot], lim = limL[L.r[0] * EPW + slot];
#pragma unroll
    for (int i = 0; i < NROWS; ++i) {
      if (i >= n) break;                                                  // wave-uniform
      const int r = L.r[i], rn = L.r[i + 1];
      const float jyn = ldsf(base, rn * ROWB + oy), jzn = ldsf(base, rn * ROWB + oz);
      const float4 c2n = rows[(rn * 3 + 2) * EPW + slot];
      const float lamn = lamL[rn * EPW + slot], limn = limL[rn * EPW + slot];
      const int leg = (r >> 1) & 3;
      const float zl = leg == 0 ? z0 : (leg == 1 ? z1 : (leg == 2 ? z2 : z3));
      const float vel = group_sum8(fmaf(jy, ys, jz * zl));
      float nl = fmaf(-c2.z, vel, lam + c2.y);
      const bool normal = i < nnormal;
      nl = normal ? fmaxf(nl, 0.0f) : __builtin_amdgcn_fmed3f(nl, -lim, lim);
      const float dl = nl - lam;
      lamL[r * EPW + slot] = nl;
      if (normal) limL[(nnormal + 2 * i) * EPW + slot] = 0.5f * nl;
      worst = fmaxf(worst, fmaf(-1e-4f, c2.z, fabsf(dl)));
      ys = fmaf(jy, dl, ys);
      const float zn = fmaf(jz, dl, zl);
      z0 = leg == 0 ? zn : z0; z1 = leg == 1 ? zn : z1; z2 = leg == 2 ? zn : z2; z3 = leg == 3 ? zn : z3;
      jy = jyn; jz = jzn; c2 = c2n; lam = lamn; lim = limn;
    }
  }
  const long long t1 = clock64();
  out[blockIdx.x * 64 + lane] = ys + z0 + z1 + z2 + z3 + worst;
  if (lane == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

int main() {
  float* out; long long* ticks; int* list; long long h;
  (void)hipMalloc(&out, 1024 * 64 * 4); (void)hipMalloc(&ticks, 16); (void)hipMalloc(&list, 64 * 4);
  const int sweeps = 300;
  for (int n : {12, 36}) {
    int hl[64];
    for (int i = 0; i < n; ++i) hl[i] = i;
    (void)hipMemcpy(list, hl, sizeof(hl), hipMemcpyHostToDevice);
    const unsigned long long mask = n == 36 ? 0xFFFFFFFFFull : 0xFFFull;
    for (int rep = 0; rep < 2; ++rep) { k_static<36><<<1024, 64>>>(out, (unsigned)mask, (unsigned)(mask >> 32), sweeps, ticks); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
    printf("A static rows, loads at use, impulses in registers : %2d occupied rows: %.1f cycles per row\n", n, (double)h / sweeps / n);
    for (int rep = 0; rep < 2; ++rep) { k_list<<<1024, 64>>>(out, list, n, n / 3, sweeps, ticks); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
    printf("B pipelined loop over the occupied-row list, LDS impulses : %2d occupied rows: %.1f cycles per row\n", n, (double)h / sweeps / n);
    for (int rep = 0; rep < 2; ++rep) { k_list_sgpr<<<1024, 64>>>(out, list, n, n / 3, sweeps, ticks); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
    printf("C the same with the list in scalar registers, loop unrolled : %2d occupied rows: %.1f cycles per row\n", n, (double)h / sweeps / n);
    for (int rep = 0; rep < 2; ++rep) { k_static_ahead<36><<<1024, 64>>>(out, (unsigned)mask, (unsigned)(mask >> 32), sweeps, ticks); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
    printf("E static rows, next row read one row ahead, impulses in registers : %2d occupied rows: %.1f cycles per row\n", n, (double)h / sweeps / n);
    RowList rl; for (int i = 0; i <= NROWS; ++i) rl.r[i] = i < n ? i : n - 1;
    for (int rep = 0; rep < 2; ++rep) { k_list_args<<<1024, 64>>>(out, rl, n, n / 3, sweeps, ticks); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
    printf("D the list as a kernel argument (scalar registers), loop unrolled : %2d occupied rows: %.1f cycles per row\n", n, (double)h / sweeps / n);
  }
  return 0;
}
