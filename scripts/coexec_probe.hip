// Can one SIMD execute matrix and vector instructions at the same time?  (developer tool)
//   hipcc -O3 --offload-arch=gfx950 scripts/coexec_probe.hip -o scripts/coexec_probe && scripts/coexec_probe
// One workgroup on one CU.  mode 0: 64 int8 MFMAs (32x32x32) per wave, 1: 512 independent v_min_u32 / v_med3 style integer ops,
// 2: both interleaved in ONE wave (8 vector ops behind every MFMA, independent of it), 3: 512-thread workgroup - waves 0..3 run
// the MFMAs, waves 4..7 (the SIMD partners) the vector ops.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
#define NM 64
__global__ __launch_bounds__(512) void k(int mode, int* out, long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  i32x4 a = {lane, lane + 1, lane + 2, lane + 3}, b = {lane * 3, 1, 2, 3};
  i32x16 acc0, acc1;
  for (int i = 0; i < 16; i++) { acc0[i] = i; acc1[i] = -i; }
  unsigned v[8];
  for (int i = 0; i < 8; i++) v[i] = lane * 7 + i;
  unsigned x = lane;
  const bool do_m = mode == 0 || mode == 2 || (mode == 3 && wave < 4);
  const bool do_v = mode == 1 || mode == 2 || (mode == 3 && wave >= 4);
  __syncthreads();
  long long t0, t1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(x) : : "memory");
  if (mode == 2) {
#pragma unroll
    for (int i = 0; i < NM / 2; i++) {
      acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc0, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = min(v[j] * 3u + x, v[(j + 1) & 7] ^ 0x55u);
      acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc1, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = min(v[j] * 3u + x, v[(j + 1) & 7] ^ 0x55u);
    }
  } else {
    if (do_m) {
#pragma unroll
      for (int i = 0; i < NM / 2; i++) {
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc1, 0, 0, 0);
      }
    }
    if (do_v) {
#pragma unroll
      for (int i = 0; i < NM; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = min(v[j] * 3u + x, v[(j + 1) & 7] ^ 0x55u);
    }
  }
  for (int i = 0; i < 16; i++) x += acc0[i] + acc1[i];
  for (int i = 0; i < 8; i++) x += v[i];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(x) : : "memory");
  if (lane == 0) cyc[wave] = t1 - t0;
  out[threadIdx.x] = x;
}
int main() {
  int* out; long long* cyc;
  hipMalloc(&out, 2048); hipMalloc(&cyc, 64);
  const char* names[] = {"64 MFMA i8 32x32x32 alone", "512 x (mul+add, xor, min) alone", "interleaved in one wave", "MFMA waves 0-3 | vector waves 4-7"};
  for (int mode = 0; mode < 4; mode++) {
    long long h[8];
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(k, dim3(1), dim3(mode == 3 ? 512 : 256), 0, 0, mode, out, cyc);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    printf("%-40s wave0 %lld cycles", names[mode], h[0]);
    if (mode == 3) printf(", wave4 %lld cycles", h[4]);
    printf("\n");
  }
  return 0;
}
