// Stand-alone probe of the 16 x 16 tile factorisation of chol.hip (developer tool, not part of the library):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude scripts/potrf_probe.hip -o scripts/potrf_probe && scripts/potrf_probe
// One wave factors a 16 x 16 SPD tile that lies in LDS exactly as panel_col0 keeps it, the cycle count of the call and the
// error against a host factorisation are printed.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#define MSFM_POTRF_PROBE 1
#include "../metricsfm_amd/csrc/chol_potrf16.h"
#ifndef POTRF
#define POTRF potrf16_m
#endif

__global__ __launch_bounds__(256) void k_probe(const double* A /*[64][64]*/, double* Lout /*[64][64]*/, double* Dout /*[64][16]*/, long long* cyc, int* fail) {
  __shared__ double sm[80 + 64 * DV + 64 * LDT];
  double* dvec = sm;
  double* dinv = sm + 80;
  double* Ls = sm + 80 + 64 * DV;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int e = tid; e < 64 * 64; e += 256) Ls[(e >> 6) * LDT + (e & 63)] = A[e];
  __syncthreads();
  if (wave == 0) {
    long long t0, t1;
    int dummy = lane;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(dummy) : : "memory");
    POTRF<0>(Ls, dinv, dvec, lane, fail);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(dummy) : : "memory");
    if (lane == 0) cyc[0] = t1 - t0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(dummy) : : "memory");
    POTRF<2>(Ls, dinv, dvec, lane, fail);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(dummy) : : "memory");
    if (lane == 0) cyc[1] = t1 - t0;
  }
  __syncthreads();
  for (int e = tid; e < 64 * 64; e += 256) Lout[e] = Ls[(e >> 6) * LDT + (e & 63)];
  for (int e = tid; e < 64 * 16; e += 256) Dout[e] = dinv[(e >> 4) * DV + (e & 15)];
}

int main() {
  std::vector<double> A(64 * 64, 0.0), L(64 * 64), D(64 * 16);
  std::mt19937_64 g(5);
  std::uniform_real_distribution<double> U(-1, 1);
  for (int t = 0; t < 4; t++)   // four independent SPD diagonal tiles, garbage elsewhere
    for (int r = 0; r < 16; r++) {
      for (int c = 0; c < r; c++) A[(16 * t + r) * 64 + 16 * t + c] = A[(16 * t + c) * 64 + 16 * t + r] = U(g);
      A[(16 * t + r) * 64 + 16 * t + r] = 17.0 + U(g);
    }
  double *dA, *dL, *dD;
  long long* dc;
  int* df;
  hipMalloc(&dA, sizeof(double) * 4096); hipMalloc(&dL, sizeof(double) * 4096); hipMalloc(&dD, sizeof(double) * 1024);
  hipMalloc(&dc, 64); hipMalloc(&df, 16); hipMemset(df, 0, 16);
  hipMemcpy(dA, A.data(), sizeof(double) * 4096, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(256), 0, 0, dA, dL, dD, dc, df);
    hipDeviceSynchronize();
  }
  long long hc[2]; int hf;
  hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost); hipMemcpy(&hf, df, 4, hipMemcpyDeviceToHost);
  hipMemcpy(L.data(), dL, sizeof(double) * 4096, hipMemcpyDeviceToHost); hipMemcpy(D.data(), dD, sizeof(double) * 1024, hipMemcpyDeviceToHost);
  printf("tile 0: %lld cycles, tile 2: %lld cycles (16 pivots each), fail %d\n", hc[0], hc[1], hf);
  for (int t : {0, 2}) {
    // host Cholesky of the tile
    double R[16][16] = {};
    for (int c = 0; c < 16; c++) {
      double s = A[(16 * t + c) * 64 + 16 * t + c];
      for (int k = 0; k < c; k++) s -= R[c][k] * R[c][k];
      R[c][c] = std::sqrt(s);
      for (int r = c + 1; r < 16; r++) {
        double v = A[(16 * t + r) * 64 + 16 * t + c];
        for (int k = 0; k < c; k++) v -= R[r][k] * R[c][k];
        R[r][c] = v / R[c][c];
      }
    }
    double eL = 0, eI = 0;
    for (int r = 0; r < 16; r++)
      for (int c = 0; c <= r; c++) eL = std::fmax(eL, std::fabs(L[(16 * t + r) * 64 + 16 * t + c] - R[r][c]));
    // dinv[(16 t + r) * 16 + l] must be Linv[r][l]: check  sum_l Dinv[r][l] R[l][c] = delta(r, c)
    for (int r = 0; r < 16; r++)
      for (int c = 0; c < 16; c++) {
        double s = 0;
        for (int l = 0; l < 16; l++) s += D[(16 * t + r) * 16 + l] * R[l][c];
        eI = std::fmax(eI, std::fabs(s - (r == c ? 1.0 : 0.0)));
      }
    printf("tile %d: max |L - L_host| = %.3e, max |Linv L - I| = %.3e\n", t, eL, eI);
  }
  return 0;
}
