// Which MFMA shape sustains more FLOP/s under this chip's clock management (developer probe, round 5)?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/mfma_shape_probe.hip -o scripts/mfma_shape_probe && scripts/mfma_shape_probe
// MI355X_MICROARCH.md (DVFS give-back, item 7) reports ~1.15 x the FLOP/s for v_mfma_f32_16x16x32_bf16 against 32x32x16 in bare loops on
// random data at equal cycles per FLOP.  The matchers use 32x32x16 (f16) and 32x32x32 (i8); this probe times, per shape, a loop of
// independent accumulators fed from registers (one workgroup of WAVES waves per SIMD on every CU, random operands), long enough
// (> 100 ms per case, after a warm-up launch) for the clock to settle.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }

// SHAPE 0: f16 32x32x16 (4 accumulators x 16 regs), 1: f16 16x16x32 (16 accumulators x 4 regs), 2: i8 32x32x32, 3: i8 16x16x64
template <int SHAPE>
__global__ __launch_bounds__(512) void k(int iters, float* out) {
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 17u;
  f16x8 a[4], b[4];
  i32x4 ia[4], ib[4];
  for (int j = 0; j < 4; j++) {
    for (int e = 0; e < 8; e++) { a[j][e] = (_Float16)((float)(rnd(s) >> 20) * (1.0f / 4096.0f) - 0.5f); b[j][e] = (_Float16)((float)(rnd(s) >> 20) * (1.0f / 4096.0f) - 0.5f); }
    for (int e = 0; e < 4; e++) { ia[j][e] = (int)rnd(s); ib[j][e] = (int)rnd(s); }
  }
  float acc_sum = 0.f;
  if (SHAPE == 0) {
    f32x16 c[4];
    for (int j = 0; j < 4; j++) for (int e = 0; e < 16; e++) c[j][e] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int j = 0; j < 4; j++) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[j & 3], c[j], 0, 0, 0);
    }
    for (int j = 0; j < 4; j++) for (int e = 0; e < 16; e++) acc_sum += c[j][e];
  } else if (SHAPE == 1) {
    f32x4 c[16];
    for (int j = 0; j < 16; j++) for (int e = 0; e < 4; e++) c[j][e] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int j = 0; j < 16; j++) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j & 3], b[j & 3], c[j], 0, 0, 0);
    }
    for (int j = 0; j < 16; j++) for (int e = 0; e < 4; e++) acc_sum += c[j][e];
  } else if (SHAPE == 2) {
    i32x16 c[4];
    for (int j = 0; j < 4; j++) for (int e = 0; e < 16; e++) c[j][e] = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int j = 0; j < 4; j++) c[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ia[j], ib[j & 3], c[j], 0, 0, 0);
    }
    for (int j = 0; j < 4; j++) for (int e = 0; e < 16; e++) acc_sum += (float)c[j][e];
  } else {
    i32x4 c[16];
    for (int j = 0; j < 16; j++) for (int e = 0; e < 4; e++) c[j][e] = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int j = 0; j < 16; j++) c[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ia[j & 3], ib[j & 3], c[j], 0, 0, 0);
    }
    for (int j = 0; j < 16; j++) for (int e = 0; e < 4; e++) acc_sum += (float)c[j][e];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc_sum;
}

template <int SHAPE>
void run(const char* name, int waves_per_simd, double ops_per_iter_per_wave) {
  float* out;
  hipMalloc(&out, sizeof(float) * 256 * 512);
  const int iters = 600000;
  const int threads = 256 * waves_per_simd;
  hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(threads), 0, 0, iters, out);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(threads), 0, 0, iters, out);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves = 256.0 * 4 * waves_per_simd;
  const double tops = ops_per_iter_per_wave * iters * waves / (ms * 1e-3) / 1e12;
  printf("%-16s %d wave(s)/SIMD: %8.2f ms  %8.1f T(FL)OP/s\n", name, waves_per_simd, ms, tops);
  hipFree(out);
}

int main() {
  for (int w = 1; w <= 2; w++) {
    run<0>("f16 32x32x16", w, 4.0 * 32 * 32 * 16 * 2);
    run<1>("f16 16x16x32", w, 16.0 * 16 * 16 * 32 * 2);
    run<2>("i8  32x32x32", w, 4.0 * 32 * 32 * 32 * 2);
    run<3>("i8  16x16x64", w, 16.0 * 16 * 16 * 64 * 2);
  }
  return 0;
}
