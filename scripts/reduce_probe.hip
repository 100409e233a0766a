// Developer check of wave_reduce_scatter against wave_sum (bitwise), for several counts:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/reduce_probe.hip -o scripts/reduce_probe && scripts/reduce_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "../metricsfm_amd/csrc/ba_device.h"

template <int N>
__global__ void k_check(double* ref, double* got, int* gidx) {
  const int lane = threadIdx.x;
  double v[N], w[N];
  for (int k = 0; k < N; k++) { v[k] = sin(1.0 + lane * 0.37 + k * 1.7) * (1 + k); w[k] = v[k]; }
  for (int k = 0; k < N; k++) { const double s = wave_sum(w[k]); if (lane == 0) ref[k] = s; }
  double val; int idx;
  wave_reduce_scatter<N>(v, lane, val, idx);
  got[lane] = val; gidx[lane] = idx;
}

template <int N>
static int run() {
  double *ref, *got; int* gidx;
  hipMalloc(&ref, 64 * 8); hipMalloc(&got, 64 * 8); hipMalloc(&gidx, 64 * 4);
  hipLaunchKernelGGL(k_check<N>, dim3(1), dim3(64), 0, 0, ref, got, gidx);
  double hr[64], hg[64]; int hi[64];
  hipMemcpy(hr, ref, 64 * 8, hipMemcpyDeviceToHost); hipMemcpy(hg, got, 64 * 8, hipMemcpyDeviceToHost); hipMemcpy(hi, gidx, 64 * 4, hipMemcpyDeviceToHost);
  int seen[64] = {0}, bad = 0;
  for (int l = 0; l < 64; l++) {
    if (hi[l] < 0) continue;
    if (hi[l] >= N) { bad++; continue; }
    seen[hi[l]]++;
    if (memcmp(&hr[hi[l]], &hg[l], 8) != 0) { if (bad < 4) printf("  N=%d lane %d idx %d: %.17g vs %.17g\n", N, l, hi[l], hg[l], hr[hi[l]]); bad++; }
  }
  for (int k = 0; k < N; k++) if (seen[k] != 1) { if (bad < 8) printf("  N=%d value %d held by %d lanes\n", N, k, seen[k]); bad++; }
  printf("N=%d: %s\n", N, bad ? "MISMATCH" : "identical");
  return bad;
}

int main() {
  int bad = 0;
  bad += run<1>(); bad += run<2>(); bad += run<3>(); bad += run<6>(); bad += run<18>(); bad += run<27>(); bad += run<36>(); bad += run<60>(); bad += run<64>();
  return bad ? 1 : 0;
}
