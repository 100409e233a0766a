// Dependent-issue latencies of the instructions on the Cholesky pivot chain (developer tool, not part of the library):
//   hipcc -O3 --offload-arch=gfx950 scripts/lat_probe.hip -o scripts/lat_probe && scripts/lat_probe
// One wave, alone on its SIMD; every figure is cycles (s_memtime) per instruction of a dependent chain of N.
#include <hip/hip_runtime.h>
#include <cstdio>

#define N 256
// the counter read is tied to x on both sides, so that the measured chain can move neither above nor below it
#define TICK(t) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(x) : : "memory")
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

__global__ void k_lat(double* out, long long* cyc, double seed) {
  __shared__ double sm[256];
  const int lane = threadIdx.x;
  double x = seed + lane * 1e-3, y = 1.0 + 1e-9 * lane;
  long long t0, t1;
  int slot = 0;
  // 0: dependent v_fma_f64
  TICK(t0);
#pragma unroll
  for (int i = 0; i < N; i++) x = fma(x, y, 1e-9);
  TICK(t1);
  if (lane == 0) cyc[slot] = t1 - t0;
  slot++;
  // 1: independent v_fma_f64 (8 chains)
  double a[8];
#pragma unroll
  for (int k = 0; k < 8; k++) a[k] = x + k;
  TICK(t0);
#pragma unroll
  for (int i = 0; i < N / 8; i++)
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = fma(a[k], y, 1e-9);
  x = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  TICK(t1);
  if (lane == 0) cyc[slot] = t1 - t0;
  slot++;
#pragma unroll
  for (int k = 0; k < 8; k++) x += a[k];
  // 2: dependent v_mul_f64
  TICK(t0);
#pragma unroll
  for (int i = 0; i < N; i++) x = x * y;
  TICK(t1);
  if (lane == 0) cyc[slot] = t1 - t0;
  slot++;
  // 3: dependent v_rcp_f64
  TICK(t0);
#pragma unroll
  for (int i = 0; i < N; i++) x = __builtin_amdgcn_rcp(x);
  TICK(t1);
  if (lane == 0) cyc[slot] = t1 - t0;
  slot++;
  // 4: independent v_rcp_f64 (8 chains)
#pragma unroll
  for (int k = 0; k < 8; k++) a[k] = x + k;
  TICK(t0);
#pragma unroll
  for (int i = 0; i < N / 8; i++)
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = __builtin_amdgcn_rcp(a[k]);
  x = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  TICK(t1);
  if (lane == 0) cyc[slot] = t1 - t0;
  slot++;
#pragma unroll
  for (int k = 0; k < 8; k++) x += a[k];
  // 5: dependent v_rcp_f32 with conversions (cvt f64->f32, rcp, cvt f32->f64)
  TICK(t0);
#pragma unroll
  for (int i = 0; i < N; i++) x = (double)__builtin_amdgcn_rcpf((float)x);
  TICK(t1);
  if (lane == 0) cyc[slot] = t1 - t0;
  slot++;
  // 6: readlane (both halves) -> fma with the uniform value -> readlane ...
  TICK(t0);
#pragma unroll
  for (int i = 0; i < N; i++) {
    const double u = readlane_f64(x, (i * 7) & 63);
    x = fma(u, y, x);
  }
  TICK(t1);
  if (lane == 0) cyc[slot] = t1 - t0;
  slot++;
  // 7: LDS round trip: ds_write_b64 own slot, uniform-address ds_read_b64, dependent
  TICK(t0);
#pragma unroll
  for (int i = 0; i < N; i++) {
    sm[lane] = x;
    const double u = sm[(i * 5) & 63];
    x = u + 1e-9 * lane;
  }
  TICK(t1);
  if (lane == 0) cyc[slot] = t1 - t0;
  slot++;
  // 8: v_rsq_f64 dependent
  TICK(t0);
#pragma unroll
  for (int i = 0; i < N; i++) x = __builtin_amdgcn_rsq(x) + 1.0;
  TICK(t1);
  if (lane == 0) cyc[slot] = t1 - t0;
  slot++;
  // 9: DPP row broadcast style move: v_mov_dpp (quad_perm) dependent with an add
  TICK(t0);
#pragma unroll
  for (int i = 0; i < N; i++) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0x4E, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0x4E, 0xF, 0xF, true);
    x = __hiloint2double(hi, lo) + 1e-9;
  }
  TICK(t1);
  if (lane == 0) cyc[slot] = t1 - t0;
  slot++;
  // 10: dependent f64 MFMA 16x16x4 (accumulator chain)
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 acc = {x, x, x, x};
  TICK(t0);
#pragma unroll
  for (int i = 0; i < N / 4; i++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, acc, 0, 0, 0);
  x = acc[0];
  TICK(t1);
  if (lane == 0) cyc[slot] = t1 - t0;
  slot++;
  x += acc[0] + acc[1] + acc[2] + acc[3];
  // 11: independent f64 MFMA 16x16x4 (4 accumulators)
  d4 ac[4];
#pragma unroll
  for (int k = 0; k < 4; k++) ac[k] = d4{x, x + k, x, x};
  TICK(t0);
#pragma unroll
  for (int i = 0; i < N / 16; i++)
#pragma unroll
    for (int k = 0; k < 4; k++) ac[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, ac[k], 0, 0, 0);
  x = (ac[0][0] + ac[1][0]) + (ac[2][0] + ac[3][0]);
  TICK(t1);
  if (lane == 0) cyc[slot] = t1 - t0;
  slot++;
#pragma unroll
  for (int k = 0; k < 4; k++) x += ac[k][0] + ac[k][3];
  // 12: independent v_fma_f64 with three distinct VGPR-pair sources (the shape of the potrf dot-product terms)
  {
    double b[8], c[8];
    TICK(t0);
#pragma unroll
    for (int k = 0; k < 8; k++) { a[k] = x + k; b[k] = y + 0.5 * k; c[k] = x - k; }
#pragma unroll
    for (int i = 0; i < N / 8; i++)
#pragma unroll
      for (int k = 0; k < 8; k++) c[k] = fma(a[k], b[(k + i) & 7], c[k]);
    x = ((c[0] + c[1]) + (c[2] + c[3])) + ((c[4] + c[5]) + (c[6] + c[7]));
    TICK(t1);
    if (lane == 0) cyc[slot] = t1 - t0;
    slot++;
    // 13: the same with the multiplier in an SGPR pair (uniform value)
    const double us = readlane_f64(y, 3);
    TICK(t0);
#pragma unroll
    for (int k = 0; k < 8; k++) { c[k] = x - k; a[k] = x + k; }
#pragma unroll
    for (int i = 0; i < N / 8; i++)
#pragma unroll
      for (int k = 0; k < 8; k++) c[k] = fma(a[k], us, c[k]);
    x = ((c[0] + c[1]) + (c[2] + c[3])) + ((c[4] + c[5]) + (c[6] + c[7]));
    TICK(t1);
    if (lane == 0) cyc[slot] = t1 - t0;
    slot++;
  }
  // 13b: independent fmac with one shared VGPR-pair multiplier: c[k] = fma(ns, w[k], c[k])
  {
    double w8[8], c8[8];
    TICK(t0);
#pragma unroll
    for (int k = 0; k < 8; k++) { w8[k] = y + 0.25 * k; c8[k] = x - k; }
    const double ns = x * 1e-3;
#pragma unroll
    for (int i = 0; i < N / 8; i++)
#pragma unroll
      for (int k = 0; k < 8; k++) c8[k] = fma(ns, w8[(k + i) & 7], c8[k]);
    x = ((c8[0] + c8[1]) + (c8[2] + c8[3])) + ((c8[4] + c8[5]) + (c8[6] + c8[7]));
    TICK(t1);
    if (lane == 0) cyc[slot] = t1 - t0;
    slot++;
  }
  // 14: independent readlane pairs (8 different registers), results summed on the scalar side
  {
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = x + k;
    TICK(t0);
    double acc8 = 0.0;
#pragma unroll
    for (int i = 0; i < N / 8; i++) {
      double u[8];
#pragma unroll
      for (int k = 0; k < 8; k++) u[k] = readlane_f64(a[k], (i + k) & 63);
      acc8 += ((u[0] + u[1]) + (u[2] + u[3])) + ((u[4] + u[5]) + (u[6] + u[7]));
    }
    x = acc8;
    TICK(t1);
    if (lane == 0) cyc[slot] = t1 - t0;
    slot++;
  }
  // 15: uniform-address ds_read_b128, 8 in flight
  {
    sm[lane] = x; sm[64 + lane] = x + 1; sm[128 + lane] = x + 2;
    __builtin_amdgcn_s_waitcnt(0);
    typedef double d2 __attribute__((ext_vector_type(2)));
    TICK(t0);
    double acc8 = 0.0;
#pragma unroll
    for (int i = 0; i < N / 8; i++) {
      d2 u[8];
#pragma unroll
      for (int k = 0; k < 8; k++) u[k] = *reinterpret_cast<const d2*>(&sm[2 * ((i * 8 + k) & 63)]);
#pragma unroll
      for (int k = 0; k < 8; k++) acc8 += u[k].x + u[k].y;
    }
    x = acc8;
    TICK(t1);
    if (lane == 0) cyc[slot] = t1 - t0;
    slot++;
  }
  // 16: ds_bpermute round trip (both halves) + add, dependent
  TICK(t0);
#pragma unroll
  for (int i = 0; i < N; i++) {
    const int lo = __builtin_amdgcn_ds_bpermute(((lane + 17) & 63) * 4, __double2loint(x));
    const int hi = __builtin_amdgcn_ds_bpermute(((lane + 17) & 63) * 4, __double2hiint(x));
    x = __hiloint2double(hi, lo) + 1e-9;
  }
  TICK(t1);
  if (lane == 0) cyc[slot] = t1 - t0;
  slot++;
  // 17: v_permlane32_swap pair + add, dependent
  TICK(t0);
#pragma unroll
  for (int i = 0; i < N; i++) {
    const auto l = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(x), false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(x), false, false);
    x = __hiloint2double(h[0], l[0]) + 1e-9;
  }
  TICK(t1);
  if (lane == 0) cyc[slot] = t1 - t0;
  slot++;
  // accuracy of v_rcp_f64 (max relative error over the lanes, in units of 2^-53), and of rcp + one quadratic step
  {
    const double d = 1.0 + lane * 0.013 + seed * 1e-3;
    const double r = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, r, 1.0);
    out[64 + lane] = e;
    const double r2 = fma(r, e, r);
    out[128 + lane] = fma(-d, r2, 1.0);
  }
  out[lane] = x;
}

int main() {
  double* out;
  long long* cyc;
  hipMalloc(&out, 256 * sizeof(double));
  hipMalloc(&cyc, 32 * sizeof(long long));
  hipMemset(cyc, 0, 32 * sizeof(long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, out, cyc, 1.25);
    hipEventRecord(e1);
    hipDeviceSynchronize();
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long h[32];
  double ho[256];
  hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  hipMemcpy(ho, out, sizeof ho, hipMemcpyDeviceToHost);
  const char* names[] = {"dep v_fma_f64", "indep v_fma_f64 (8 chains)", "dep v_mul_f64", "dep v_rcp_f64", "indep v_rcp_f64 (8)", "dep cvt+rcp_f32+cvt",
                         "readlane x2 -> fma", "ds_write + bcast ds_read", "dep v_rsq_f64 + add", "dpp mov x2 + add", "dep mfma f64 16x16x4", "indep mfma f64 (4 acc)",
                         "indep fma, 3 VGPR pairs", "indep fma, SGPR multiplier", "indep fmac, shared multiplier", "indep readlane pair (+adds)", "bcast ds_read_b128 (+2 adds)", "ds_bpermute x2 + add", "permlane32_swap x2 + add"};
  const int cnt[] = {N, N, N, N, N, N, N, N, N, N, N / 4, N / 4, N, N, N, N, N, N, N};
  long long tot = 0;
  for (int i = 0; i < 19; i++) { printf("%-28s %8.1f cycles each\n", names[i], (double)h[i] / cnt[i]); tot += h[i]; }
  printf("kernel %.3f ms for %lld counted cycles -> counter runs at >= %.0f MHz\n", ms, tot, tot / (ms * 1e3));
  double emax = 0, e2max = 0;
  for (int l = 0; l < 64; l++) { emax = fmax(emax, fabs(ho[64 + l])); e2max = fmax(e2max, fabs(ho[128 + l])); }
  printf("v_rcp_f64 max |1 - d r| = %.3e (2^%.1f); after one quadratic step %.3e\n", emax, log2(emax), e2max);
  return 0;
}
