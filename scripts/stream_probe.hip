// Do four independent panel chains overlap when they are enqueued on four HIP streams?  (developer experiment)
#include <hip/hip_runtime.h>
#include "../metricsfm_amd/csrc/chol.hip"
#include <cstdio>
#include <vector>
#include <random>
#include <chrono>
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 768, K = argc > 2 ? atoi(argv[2]) : 4;
  const int npad = (n + 1 + 63) / 64 * 64;
  msfm_ctx* ctx[8];
  std::vector<double> h((size_t)npad * npad, 0.0);
  std::mt19937_64 g(7);
  std::uniform_real_distribution<double> U(-1, 1);
  for (int r = 0; r < n; r++) { for (int c = 0; c < r; c++) h[(size_t)r * npad + c] = U(g); h[(size_t)r * npad + r] = n + 1.0; }
  for (int c = 0; c < n; c++) h[(size_t)n * npad + c] = U(g);
  double *M[8], *work[8], *w[8], *z[8]; int* fail[8];
  for (int k = 0; k < K; k++) {
    if (msfm_ctx_create(0, &ctx[k]) != 0) return 1;   // each ctx owns its own stream
    hipMalloc(&M[k], sizeof(double) * h.size()); hipMalloc(&work[k], sizeof(double) * (size_t)npad * 144);
    hipMalloc(&w[k], sizeof(double) * npad); hipMalloc(&z[k], sizeof(double) * npad); hipMalloc(&fail[k], 16); hipMemset(fail[k], 0, 16);
  }
  for (int mode = 0; mode < 2; mode++)
    for (int rep = 0; rep < 3; rep++) {
      for (int k = 0; k < K; k++) hipMemcpy(M[k], h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
      hipDeviceSynchronize();
      auto t0 = std::chrono::steady_clock::now();
      for (int k = 0; k < K; k++) msfm_chol_factor_solve(ctx[mode ? k : 0], M[k], npad, n, work[k], w[k], z[k], fail[k], nullptr);
      hipDeviceSynchronize();
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("%s: %d factor+solve of n=%d in %.1f us\n", mode ? "K streams" : "1 stream ", K, n, us);
    }
  return 0;
}
