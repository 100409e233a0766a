// How many integer vector operations hide behind v_mfma_i32_32x32x32_i8 in ONE wave's stream (developer probe):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/mfma_valu_probe.hip -o scripts/mfma_valu_probe && scripts/mfma_valu_probe
// One workgroup per CU, WAVES waves per SIMD; each iteration issues two MFMAs (two accumulators) and NV operations of the
// matcher's selection chain (v_lshl_or_b32, v_min_u32, v_med3_u32 on registers the MFMAs do not touch).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

template <int NV>
__global__ __launch_bounds__(1024) void k(int iters, int* out, long long* cyc) {
  i32x16 a0, a1;
  for (int k2 = 0; k2 < 16; k2++) { a0[k2] = threadIdx.x + k2; a1[k2] = threadIdx.x * 3 + k2; }
  i32x4 x = {(int)threadIdx.x, 1, 2, 3}, y = {5, 6, (int)threadIdx.x, 8};
  unsigned k0 = 0xffffffffu, k1 = 0xffffffffu, v[8];
  for (int j = 0; j < 8; j++) v[j] = threadIdx.x * 17 + j * 1000003;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, a0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NV / 6; j++) {
      const unsigned key = (v[j & 7] << 9) | (unsigned)it;
      const unsigned n0 = min(k0, key);
      unsigned r;
      asm volatile("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(k0), "v"(k1), "v"(key));
      k1 = r; k0 = n0;
    }
    __builtin_amdgcn_sched_barrier(0);
    a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, a1, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NV / 6; j++) {
      const unsigned key = (v[(j + 3) & 7] << 9) | (unsigned)it;
      const unsigned n0 = min(k0, key);
      unsigned r;
      asm volatile("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(k0), "v"(k1), "v"(key));
      k1 = r; k0 = n0;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  long long t1 = __builtin_readcyclecounter();
  int s = k0 ^ k1;
  for (int k2 = 0; k2 < 16; k2++) s ^= a0[k2] ^ a1[k2];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NV>
void run(int waves_per_simd) {
  int* out; long long* cyc;
  hipMalloc(&out, sizeof(int) * 256 * 1024); hipMalloc(&cyc, 8);
  const int iters = 4000;
  hipLaunchKernelGGL(k<NV>, dim3(256), dim3(256 * waves_per_simd), 0, 0, iters, out, cyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<NV>, dim3(256), dim3(256 * waves_per_simd), 0, 0, iters, out, cyc);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  // wall clock per MFMA over one SIMD, in ns, and the implied integer rate of the chip (1024 SIMDs, 65536 operations per MFMA)
  const double ns_per_mfma = ms * 1e6 / ((double)iters * 2 * waves_per_simd);
  printf("waves/SIMD %d, %2d vector ops per 2 MFMAs: %.1f ticks per iteration (per wave); %.2f ns per MFMA per SIMD = %.2f POP/s over the chip; tick = %.3f ns\n", waves_per_simd, NV,
         (double)h / iters, ns_per_mfma, 65536.0 * 1024 / ns_per_mfma * 1e-6, ms * 1e6 / (double)h);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int w : {1, 2, 4}) { run<0>(w); run<6>(w); run<12>(w); run<24>(w); run<36>(w); run<48>(w); }
  return 0;
}
