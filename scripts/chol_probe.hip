// Cycle-counter probe of the fused Cholesky panel kernel (developer tool, not part of the library):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/chol_probe.hip metricsfm_amd/csrc/ctx.o -o /tmp/chol_probe
// Probes are taken by thread 0 of workgroup `PROBE_WG` of the launch with j0 == g_probe_j0.
#include <hip/hip_runtime.h>
__device__ long long g_probe[32];
__device__ int g_probe_j0 = -1;
__device__ int g_probe_wg = 0;
__device__ volatile int g_probe_on = 0;
#define MSFM_PROBE_ARM(j0v) do { if (threadIdx.x == 0 && (int)blockIdx.x == g_probe_wg) g_probe_on = ((j0v) == g_probe_j0); } while (0)
#define MSFM_PROBE(i) do { if (threadIdx.x == 0 && (int)blockIdx.x == g_probe_wg && g_probe_on) g_probe[i] = clock64(); } while (0)
#include "../metricsfm_amd/csrc/chol.hip"
#include <cstdio>
#include <vector>
#include <random>
#include <cmath>
#include <algorithm>

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 3003;
  const int pj0 = argc > 2 ? atoi(argv[2]) : 640;
  const int pwg = argc > 3 ? atoi(argv[3]) : 0;
  const int K = argc > 4 ? atoi(argv[4]) : 1;
  const int npad = (n + 1 + 63) / 64 * 64;
  msfm_ctx* ctx = nullptr;
  if (msfm_ctx_create(0, &ctx) != 0) { printf("no ctx\n"); return 1; }
  std::vector<double> h((size_t)npad * npad, 0.0);
  std::mt19937_64 g(7);
  std::uniform_real_distribution<double> U(-1, 1);
  (void)K;   // (dense order only: the multilevel plans are exercised through msfm_ba_*)
  for (int r = 0; r < n; r++) {
    for (int c = 0; c < r; c++) {
      h[(size_t)r * npad + c] = U(g);
    }
    h[(size_t)r * npad + r] = n + 1.0;
  }
  for (int c = 0; c < n; c++) h[(size_t)n * npad + c] = U(g);
  double *M, *work, *w, *z; int* fail;
  hipMalloc(&M, sizeof(double) * h.size());
  hipMalloc(&work, sizeof(double) * (size_t)npad * 144);
  hipMalloc(&w, sizeof(double) * npad); hipMalloc(&z, sizeof(double) * npad);
  hipMalloc(&fail, 16); hipMemset(fail, 0, 16);
  hipMemcpyToSymbol(HIP_SYMBOL(g_probe_j0), &pj0, sizeof(int));
  hipMemcpyToSymbol(HIP_SYMBOL(g_probe_wg), &pwg, sizeof(int));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; rep++) {
    hipMemcpy(M, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    hipEventRecord(e0, ctx->stream);
    int rc = msfm_chol_factor_solve(ctx, M, npad, n, work, w, z, fail, nullptr);
    hipEventRecord(e1, ctx->stream);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long pr[32];
    hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_probe), sizeof pr);
    int hf; hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
    printf("rep %d rc %d fail %d total %.3f ms; probes (cycles since p0):", rep, rc, hf, ms);
    for (int i = 0; i < 16; i++) printf(" [%d]%lld", i, pr[i] ? pr[i] - pr[0] : -1);
    printf("\n");
    if (rep == 0) {
      std::vector<double> zh(npad);
      hipMemcpy(zh.data(), z, sizeof(double) * npad, hipMemcpyDeviceToHost);
      double rmax = 0, bmax = 0;
      for (int r = 0; r < n; r++) {
        double acc = 0;
        for (int c = 0; c < n; c++) acc += (c <= r ? h[(size_t)r * npad + c] : h[(size_t)c * npad + r]) * zh[c];
        rmax = std::max(rmax, std::fabs(acc - h[(size_t)n * npad + r]));
        bmax = std::max(bmax, std::fabs(h[(size_t)n * npad + r]));
      }
      printf("residual max |S z - b| = %.3e (|b|max %.3e)\n", rmax, bmax);
    }
  }
  return 0;
}
