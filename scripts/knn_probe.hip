// Issue-rate probe for the inner loop of k_knn2_i8: how long do 8 x v_mfma_i32_32x32x32_i8 and 96 single-issue
// integer VALU updates (v_lshl_or_b32, v_min_u32, v_med3_u32) take alone, back to back, and interleaved, with 1, 2
// or 4 waves per SIMD?   usage: knn_probe <waves_per_simd>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32;
__device__ __forceinline__ u32 umed3(u32 a, u32 b, u32 c) {
  u32 r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_probe(int iters, const int* __restrict__ in, int* __restrict__ out, long long* __restrict__ cyc) {
  const int lane = threadIdx.x & 63;
  i32x4 av[4], bqa[4], bqb[4];
  for (int k = 0; k < 4; k++) {
    av[k] = *reinterpret_cast<const i32x4*>(in + 4 * (lane + 64 * k));
    bqa[k] = *reinterpret_cast<const i32x4*>(in + 1024 + 4 * (lane + 64 * k));
    bqb[k] = *reinterpret_cast<const i32x4*>(in + 2048 + 4 * (lane + 64 * k));
  }
  i32x16 a0, b0, a1, b1, cc;
  for (int i = 0; i < 16; i++) { a0[i] = in[i] + lane; b0[i] = in[16 + i] + lane; a1[i] = a0[i] ^ 5; b1[i] = b0[i] ^ 9; cc[i] = i; }
  u32 ak0 = 0xffffffffu, ak1 = 0xffffffffu, bk0 = 0xffffffffu, bk1 = 0xffffffffu;
  auto select4 = [&](const i32x16& xa, const i32x16& xb, int g) {
#pragma unroll
    for (int reg = 4 * g; reg < 4 * g + 4; reg++) {
      const u32 keya = ((u32)xa[reg] << 9) | (u32)cc[reg];
      const u32 keyb = ((u32)xb[reg] << 9) | (u32)cc[reg];
      const u32 na0 = min(ak0, keya);
      ak1 = umed3(ak0, ak1, keya);
      ak0 = na0;
      const u32 nb0 = min(bk0, keyb);
      bk1 = umed3(bk0, bk1, keyb);
      bk0 = nb0;
    }
  };
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {  // MFMA only
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[ks], bqa[ks], a1, 0, 0, 0);
        b1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[ks], bqb[ks], b1, 0, 0, 0);
      }
    } else if (MODE == 1) {  // VALU only (on a set that changes a little every iteration)
#pragma unroll
      for (int ks = 0; ks < 4; ks++) { __builtin_amdgcn_sched_barrier(0); select4(a0, b0, ks); }
      a0[0] += 1; b0[0] += 1;
    } else if (MODE == 2) {  // MFMA then VALU on the fresh result (what the shipped kernel does)
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[ks], bqa[ks], a1, 0, 0, 0);
        b1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[ks], bqb[ks], b1, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 4; ks++) select4(a1, b1, ks);
      __builtin_amdgcn_sched_barrier(0);
    } else if (MODE == 3) {  // interleaved: MFMAs into set 1 between the updates from set 0, then the roles swap
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        __builtin_amdgcn_sched_barrier(0);
        a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[ks], bqa[ks], a1, 0, 0, 0);
        b1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[ks], bqb[ks], b1, 0, 0, 0);
        select4(a0, b0, ks);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        __builtin_amdgcn_sched_barrier(0);
        a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[ks], bqa[ks], a0, 0, 0, 0);
        b0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[ks], bqb[ks], b0, 0, 0, 0);
        select4(a1, b1, ks);
      }
      __builtin_amdgcn_sched_barrier(0);
    } else if (MODE == 4) {  // one MFMA, then a quarter of the updates, eight times (finer interleave)
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        __builtin_amdgcn_sched_barrier(0);
        a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[ks], bqa[ks], a1, 0, 0, 0);
#pragma unroll
        for (int reg = 4 * ks; reg < 4 * ks + 2; reg++) {
          const u32 keya = ((u32)a0[reg] << 9) | (u32)cc[reg]; const u32 keyb = ((u32)b0[reg] << 9) | (u32)cc[reg];
          const u32 na0 = min(ak0, keya); ak1 = umed3(ak0, ak1, keya); ak0 = na0;
          const u32 nb0 = min(bk0, keyb); bk1 = umed3(bk0, bk1, keyb); bk0 = nb0;
        }
        __builtin_amdgcn_sched_barrier(0);
        b1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[ks], bqb[ks], b1, 0, 0, 0);
#pragma unroll
        for (int reg = 4 * ks + 2; reg < 4 * ks + 4; reg++) {
          const u32 keya = ((u32)a0[reg] << 9) | (u32)cc[reg]; const u32 keyb = ((u32)b0[reg] << 9) | (u32)cc[reg];
          const u32 na0 = min(ak0, keya); ak1 = umed3(ak0, ak1, keya); ak0 = na0;
          const u32 nb0 = min(bk0, keyb); bk1 = umed3(bk0, bk1, keyb); bk0 = nb0;
        }
      }
      a0[0] += 1; b0[0] += 1;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = clock64();
  int acc = (int)(ak0 ^ ak1 ^ bk0 ^ bk1);
  for (int i = 0; i < 16; i++) acc ^= a0[i] ^ b0[i] ^ a1[i] ^ b1[i];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 1;   // waves per SIMD = workgroups of 256 threads per CU
  const int iters = 2000, ncu = 256;
  int *in, *out; long long* cyc;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, ncu * 8 * 256 * 4); hipMalloc(&cyc, ncu * 8 * 8);
  std::vector<int> h(4096); for (int i = 0; i < 4096; i++) h[i] = (i * 2654435761u) >> 8;
  hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  const char* names[5] = {"8 MFMA", "96 VALU", "8 MFMA then 96 VALU (dependent)", "2 x [8 MFMA interleaved with 96 VALU] (per pair)", "8 x [1 MFMA + 12 VALU]"};
  for (int mode = 0; mode < 5; mode++) {
    const int grid = ncu * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      switch (mode) {
        case 0: hipLaunchKernelGGL(k_probe<0>, dim3(grid), dim3(256), 0, 0, iters, in, out, cyc); break;
        case 1: hipLaunchKernelGGL(k_probe<1>, dim3(grid), dim3(256), 0, 0, iters, in, out, cyc); break;
        case 2: hipLaunchKernelGGL(k_probe<2>, dim3(grid), dim3(256), 0, 0, iters, in, out, cyc); break;
        case 3: hipLaunchKernelGGL(k_probe<3>, dim3(grid), dim3(256), 0, 0, iters, in, out, cyc); break;
        case 4: hipLaunchKernelGGL(k_probe<4>, dim3(grid), dim3(256), 0, 0, iters, in, out, cyc); break;
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> c(grid); hipMemcpy(c.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : c) avg += v; avg /= grid;
    // clock64 ticks at 100 MHz on this part; report wall ns per iteration per wave and per SIMD
    printf("%-52s waves/SIMD %d: %.1f ns per iteration per wave (clock64 %.1f ticks), SIMD time per iteration %.1f ns\n", names[mode], wps,
           ms * 1e6 / iters, avg / iters, ms * 1e6 / iters / wps);
  }
  return 0;
}
