// accuracy of v_rcp_f64 / v_rsq_f64 and cost of readlane+fma patterns (developer microbenchmark)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* r, double* q, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { r[i] = __builtin_amdgcn_rcp(x[i]); q[i] = __builtin_amdgcn_rsq(x[i]); }
}
__device__ __forceinline__ double rl(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// chain test: dependent rcp+NR+readlane+fma, 1024 steps
__global__ void k_chain(double* out, long long* cyc, double seed) {
  double p = seed + threadIdx.x * 1e-3, acc = 1.0;
  long long t0 = clock64();
#pragma unroll 16
  for (int it = 0; it < 1024; it++) {
    double d = rl(p, it & 63);
    double x = __builtin_amdgcn_rcp(d);
    double e = fma(-d, x, 1.0);
    x = fma(x, e, x);
    p = fma(-p, x * 0.25, p + 1.0);
  }
  long long t1 = clock64();
  // throughput test: 16 independent (readlane pair + fma) per iteration
  double a[16];
#pragma unroll
  for (int j = 0; j < 16; j++) a[j] = seed + j;
  long long t2 = clock64();
  for (int it = 0; it < 256; it++) {
    double s[16];
#pragma unroll
    for (int j = 0; j < 16; j++) s[j] = rl(p, (it + j) & 63);
#pragma unroll
    for (int j = 0; j < 16; j++) a[j] = fma(-acc, s[j], a[j]);
  }
  long long t3 = clock64();
  // plain fma throughput, 16 independent chains
  for (int it = 0; it < 256; it++) {
#pragma unroll
    for (int j = 0; j < 16; j++) a[j] = fma(a[j], 1.0000001, 1e-9);
  }
  long long t4 = clock64();
  double sum = p;
#pragma unroll
  for (int j = 0; j < 16; j++) sum += a[j];
  out[threadIdx.x] = sum;
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t3 - t2; cyc[2] = t4 - t3; cyc[3] = wall_clock64(); }
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), r(n), q(n);
  for (int i = 0; i < n; i++) x[i] = std::exp((i % 4001 - 2000) * 0.01) * (1.0 + (i * 2654435761u % 1000003) / 1000003.0);
  double *dx, *dr, *dq; hipMalloc(&dx, n * 8); hipMalloc(&dr, n * 8); hipMalloc(&dq, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dr, dq, n);
  hipMemcpy(r.data(), dr, n * 8, hipMemcpyDeviceToHost); hipMemcpy(q.data(), dq, n * 8, hipMemcpyDeviceToHost);
  double er = 0, eq = 0;
  for (int i = 0; i < n; i++) { er = std::max(er, std::fabs(r[i] * x[i] - 1.0)); eq = std::max(eq, std::fabs(q[i] * q[i] * x[i] - 1.0) * 0.5); }
  printf("max rel err rcp %.3e (2^%.1f)  rsq %.3e (2^%.1f)\n", er, std::log2(er), eq, std::log2(eq));
  long long* dc; hipMalloc(&dc, 64); long long hc[4];
  for (int rep = 0; rep < 2; rep++) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k_chain<<<1, 64>>>(dr, dc, 3.0); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hc, dc, 32, hipMemcpyDeviceToHost);
    printf("chain: %.1f cyc/step; rl+fma: %.1f cyc per (2 readlane + fma); fma: %.2f cyc each; kernel %.1f us, total cycles %lld -> %.2f GHz\n", hc[0] / 1024.0, hc[1] / 4096.0, hc[2] / 4096.0,
           ms * 1e3, hc[0] + hc[1] + hc[2], (hc[0] + hc[1] + hc[2]) / (ms * 1e6));
  }
  return 0;
}
