// Exhaustive 2-nearest-neighbour descriptor matching on MI355X.
// Replaces flann_build_index + flann_find_nearest_neighbors_index
// (SfM/src/graph/fine_matching_graph.cc:72-81,99; SfM/src/slam_gps.cc:438-447,463) and fuses
// the ratio tests of fine_matching_graph.cc:116-133.
//
// Fast path (descriptors integer-valued in [0,255], e.g. SIFT bytes stored as float):
//   d(a,b) = |a|^2 + |b|^2 - 2 a.b is evaluated as a bf16 MFMA GEMM with fp32 accumulators;
//   every operand (<= 255, and -2*b <= 510) is exact in bf16 and every partial sum is an
//   integer below 2^24, so the distances are exact and the indices identical to a binary64
//   brute force.  One workgroup = 4 waves = 128 query descriptors of one image pair; each
//   wave keeps its 32 queries as MFMA B fragments in registers for the whole sweep, train
//   descriptors stream through an XOR-swizzled LDS tile shared by the 4 waves; the running
//   top-2 per query lives in packed (distance << 8 | row) keys, 4 VALU ops per candidate.
// General path (any float32 descriptors): exact binary64 brute force, sequential in k —
//   bit-identical to the oracle's definition; slow, used only when the data is not integral.
#include <algorithm>
#include <cstdlib>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32;

#define DIM 128
#define TT 64          // train rows per LDS tile
#define WIN 256        // rows per packed-key window
#define QPB 256        // queries per workgroup (4 waves x 64)
#define NORM_BIAS 8388608.0f  // 2^23: keeps a2 - 2ab positive for the integer key

struct msfm_descset {
  msfm_ctx* ctx;
  int n_images, dim;
  std::vector<int> count;
  std::vector<DevBuf<float>*> f32;       // [count][dim]
  std::vector<DevBuf<unsigned short>*> bf16;  // [count][dim], train copy (plain) — query copy is scaled by -2 on load
  std::vector<DevBuf<float>*> norm;      // [count]  |a|^2
  // int8 forms for the i8 MFMA kernel: train rows a-128, query rows 127-b, and the per-row terms of
  //   |a-b|^2 = 2 sum (a-128)(127-b) + sum (a-127)^2 + sum (128-b)^2 - 128
  std::vector<DevBuf<signed char>*> ti8, qi8;   // [count][128]
  std::vector<DevBuf<int>*> tcin, tpar, qbeta;  // (alpha>>1)+2^21 ; alpha&1 ; beta
  // split-bf16 forms for non-integral descriptors: v ~ hi + lo (two bf16 terms, 16 significant bits)
  std::vector<DevBuf<unsigned short>*> shi, slo;  // [count][128]
  std::vector<DevBuf<float>*> sn2;                // [count]  |v|^2 (binary64 sum rounded once)
  std::vector<float> sn2max;                      // per image max |v|^2 (host copy)
  DevBuf<unsigned> n2max_dev;
  DevBuf<int> nonint;                    // OR of "not integer in [0,255]" over all uploads
  int h_nonint = 0;
  // bumped by every upload: a match result remembers the generation its device pointer tables were built at and
  // refuses to run or to be read once an image has been replaced under it
  unsigned long generation = 0;
};

// ---- prep: f32 -> bf16, squared norms, integrality flag -------------------------------
__global__ __launch_bounds__(256) void k_desc_prep(const float* __restrict__ d, int count, unsigned short* __restrict__ out,
                                                    float* __restrict__ norm, int* __restrict__ nonint) {
  // one wave per row, lane handles 2 of the 128 values
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= count) return;
  const float2 v = reinterpret_cast<const float2*>(d + (size_t)row * DIM)[lane];
  const bool bad = !(v.x >= 0.f && v.x <= 255.f && v.x == truncf(v.x) && v.y >= 0.f && v.y <= 255.f && v.y == truncf(v.y));
  if (bad) atomicOr(nonint, 1);
  ushort2 o;
  o.x = (unsigned short)(__float_as_uint(v.x) >> 16);  // exact for integers <= 255 (low mantissa bits are zero)
  o.y = (unsigned short)(__float_as_uint(v.y) >> 16);
  reinterpret_cast<ushort2*>(out + (size_t)row * DIM)[lane] = o;
  float s = v.x * v.x + v.y * v.y;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) norm[row] = s;
}

// int8 operands + the per-row integer terms (see msfm_descset).  One wave per row.
__global__ __launch_bounds__(256) void k_desc_prep_i8(const float* __restrict__ d, int count, signed char* __restrict__ ti8,
                                                       signed char* __restrict__ qi8, int* __restrict__ tcin, int* __restrict__ tpar,
                                                       int* __restrict__ qbeta) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= count) return;
  const float2 v = reinterpret_cast<const float2*>(d + (size_t)row * DIM)[lane];
  const int a0 = (int)v.x, a1 = (int)v.y;  // only meaningful for integer-valued data in [0,255] (flag from k_desc_prep)
  char2 t, q;
  t.x = (signed char)(a0 - 128); t.y = (signed char)(a1 - 128);
  q.x = (signed char)(127 - a0); q.y = (signed char)(127 - a1);
  reinterpret_cast<char2*>(ti8 + (size_t)row * DIM)[lane] = t;
  reinterpret_cast<char2*>(qi8 + (size_t)row * DIM)[lane] = q;
  int alpha = (a0 - 127) * (a0 - 127) + (a1 - 127) * (a1 - 127);
  int beta = (128 - a0) * (128 - a0) + (128 - a1) * (128 - a1);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { alpha += __shfl_xor(alpha, off, 64); beta += __shfl_xor(beta, off, 64); }
  if (lane == 0) { tcin[row] = (alpha >> 1) + (1 << 21); tpar[row] = alpha & 1; qbeta[row] = beta - 128; }
}

// ---- fast path --------------------------------------------------------------------------
struct PairTask {
  const unsigned short* train;  // bf16 [n_train][128]
  const unsigned short* query;
  const float* tnorm;
  const float* qnorm;
  int n_train, n_query;
  int out_off;                  // offset of this pair's queries in the flat outputs
};

__device__ __forceinline__ u32 umed3(u32 a, u32 b, u32 c) {
  u32 r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// The two ratio tests of fine_matching_graph.cc:116-130, evaluated independently of each other as the reference does
// (a match can be "good" without being in the "all" set when ratio_good > ratio_all).  -1: in neither set.
__device__ __forceinline__ int32_t ratio_code(float d0, float d1, int id0, float ratio_good, float ratio_all, int* n_all, int* n_good) {
  const float ratio = d0 / d1;  // fine_matching_graph.cc:118
  const bool good = ratio < ratio_good, all = ratio < ratio_all;
  if (good) atomicAdd(n_good, 1);
  if (all) atomicAdd(n_all, 1);
  if (!good && !all) return -1;
  return id0 | (good ? MSFM_MATCH_GOOD : 0) | (all ? 0 : MSFM_MATCH_NOT_ALL);
}

// merge candidate (d, i) into the sorted pair (d0,i0) <= (d1,i1); ties keep the earlier (lower index first)
__device__ __forceinline__ void top2_insert(u32& d0, int& i0, u32& d1, int& i1, u32 d, int i) {
  const bool lt0 = d < d0 || (d == d0 && i < i0);
  const bool lt1 = d < d1 || (d == d1 && i < i1);
  if (lt0) { d1 = d0; i1 = i0; d0 = d; i0 = i; }
  else if (lt1) { d1 = d; i1 = i; }
}

// merge a window's two packed keys into the lane's running (distance, row) top-2
__device__ __forceinline__ void flush_window(u32& k0, u32& k1, u32& D0, int& I0, u32& D1, int& I1, int base) {
  if (k0 != 0xffffffffu) top2_insert(D0, I0, D1, I1, k0 >> 8, base + (int)(k0 & 255u));
  if (k1 != 0xffffffffu) top2_insert(D0, I0, D1, I1, k1 >> 8, base + (int)(k1 & 255u));
  k0 = k1 = 0xffffffffu;
}

__device__ __forceinline__ void load_query_frags(const unsigned short* qp, int h, bf16x8* bq) {
#pragma unroll
  for (int ks = 0; ks < 8; ks++) {
    const uint4 raw = *reinterpret_cast<const uint4*>(qp + ks * 16 + h * 8);
    u32 w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      // each 16-bit half: bf16 v -> -2v (exponent + 1, sign set); v == 0 stays +0
      u32 lo16 = w[j] & 0xffffu, hi16 = w[j] >> 16;
      lo16 = lo16 ? ((lo16 + 0x0080u) | 0x8000u) : 0u;
      hi16 = hi16 ? ((hi16 + 0x0080u) | 0x8000u) : 0u;
      w[j] = lo16 | (hi16 << 16);
    }
    bq[ks] = __builtin_bit_cast(bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
  }
}

// 1 workgroup = 4 waves = 256 queries of one pair; each wave keeps 2 x 32 queries as B fragments
// (every A fragment read from LDS feeds two MFMAs).  Train tiles of 64 rows are double-buffered:
// the global loads of tile t+1 are issued before the MFMAs of tile t and written to the other LDS
// buffer afterwards, one barrier per tile.
__global__ __launch_bounds__(256, 2) void k_knn2_bf16(const PairTask* __restrict__ tasks, const int* __restrict__ tile_first, int n_pairs,
                                                       float ratio_good, float ratio_all, int32_t* __restrict__ code,
                                                       int* __restrict__ ids, float* __restrict__ sqd, int* __restrict__ n_all,
                                                       int* __restrict__ n_good) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_a[2][TT * 256];
  __shared__ __attribute__((aligned(16))) float lds_n[2][TT];
  int lo = 0, hi = n_pairs - 1;
  const int bid = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tile_first[mid] <= bid) lo = mid; else hi = mid - 1;
  }
  const int pair = lo;
  const PairTask T = tasks[pair];
  const int q0 = (bid - tile_first[pair]) * QPB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  const int qa = q0 + wave * 64 + r, qb = qa + 32;
  const bool va = qa < T.n_query, vb = qb < T.n_query;
  bf16x8 bqa[8], bqb[8];
  load_query_frags(T.query + (size_t)(va ? qa : 0) * DIM, h, bqa);
  load_query_frags(T.query + (size_t)(vb ? qb : 0) * DIM, h, bqb);
  u32 aD0 = 0xffffffffu, aD1 = 0xffffffffu, bD0 = 0xffffffffu, bD1 = 0xffffffffu;
  int aI0 = 0x7fffffff, aI1 = 0x7fffffff, bI0 = 0x7fffffff, bI1 = 0x7fffffff;
  u32 ak0 = 0xffffffffu, ak1 = 0xffffffffu, bk0 = 0xffffffffu, bk1 = 0xffffffffu;
  const int n_tiles = (T.n_train + TT - 1) / TT;
  // staging assignment: thread -> 4 chunks of 16 B (row = c >> 4, chunk = c & 15), XOR-swizzled per row
  uint4 stage[4];
  float stage_n = 0.f;
  auto fetch = [&](int tile) {
    const int t0 = tile * TT;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int c = tid + 256 * i, row = c >> 4, ch = c & 15;
      stage[i] = make_uint4(0, 0, 0, 0);
      if (t0 + row < T.n_train) stage[i] = *reinterpret_cast<const uint4*>(T.train + (size_t)(t0 + row) * DIM + ch * 8);
    }
    if (tid < TT) stage_n = (t0 + tid < T.n_train) ? T.tnorm[t0 + tid] + NORM_BIAS : 16777215.0f;  // padding rows lose every comparison
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int c = tid + 256 * i, row = c >> 4, ch = c & 15;
      *reinterpret_cast<uint4*>(&lds_a[buf][row * 256 + ((ch ^ (row & 15)) << 4)]) = stage[i];
    }
    if (tid < TT) lds_n[buf][tid] = stage_n;
  };
  fetch(0);
  commit(0);
  __syncthreads();
  int cur = 0;
  for (int tile = 0; tile < n_tiles; tile++) {
    if (tile + 1 < n_tiles) fetch(tile + 1);
    const unsigned char* la = lds_a[cur];
    const float* ln = lds_n[cur];
#pragma unroll
    for (int st = 0; st < 2; st++) {
      f32x16 acca, accb;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const f32x4 nv = *reinterpret_cast<const f32x4*>(&ln[st * 32 + 8 * g + 4 * h]);
        acca[4 * g + 0] = nv.x; acca[4 * g + 1] = nv.y; acca[4 * g + 2] = nv.z; acca[4 * g + 3] = nv.w;
      }
      accb = acca;
      const int row = st * 32 + r;
#pragma unroll
      for (int ks = 0; ks < 8; ks++) {
        const int ch = 2 * ks + h;
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(la + row * 256 + ((ch ^ (row & 15)) << 4));
        acca = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bqa[ks], acca, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bqb[ks], accb, 0, 0, 0);
      }
      const int wbase = ((tile & 3) * 2 + st) * 32;
#pragma unroll
      for (int reg = 0; reg < 16; reg++) {
        const u32 idx = (u32)(wbase + (reg & 3) + 8 * (reg >> 2));
        const u32 keya = ((u32)acca[reg] << 8) | idx;  // v_cvt_u32_f32 of an exact integer < 2^24
        const u32 keyb = ((u32)accb[reg] << 8) | idx;
        const u32 na0 = min(ak0, keya);
        ak1 = umed3(ak0, ak1, keya);
        ak0 = na0;
        const u32 nb0 = min(bk0, keyb);
        bk1 = umed3(bk0, bk1, keyb);
        bk0 = nb0;
      }
    }
    if ((tile & 3) == 3 || tile == n_tiles - 1) {
      const int base = (tile & ~3) * TT + 4 * h;
      flush_window(ak0, ak1, aD0, aI0, aD1, aI1, base);
      flush_window(bk0, bk1, bD0, bI0, bD1, bI1, base);
    }
    if (tile + 1 < n_tiles) commit(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int s = 0; s < 2; s++) {
    u32 D0 = s ? bD0 : aD0, D1 = s ? bD1 : aD1;
    int I0 = s ? bI0 : aI0, I1 = s ? bI1 : aI1;
    const int q = s ? qb : qa;
    const bool qvalid = s ? vb : va;
    // merge the two lane halves (rows 4h+..) of each query
    const u32 pd0 = __shfl_xor(D0, 32, 64), pd1 = __shfl_xor(D1, 32, 64);
    const int pi0 = __shfl_xor(I0, 32, 64), pi1 = __shfl_xor(I1, 32, 64);
    top2_insert(D0, I0, D1, I1, pd0, pi0);
    top2_insert(D0, I0, D1, I1, pd1, pi1);
    if (h == 0 && qvalid) {
      const int b2 = (int)T.qnorm[q];
      const float d0 = (float)((int)D0 - 8388608 + b2), d1 = (float)((int)D1 - 8388608 + b2);
      const size_t o = (size_t)T.out_off + q;
      if (ids) { ids[2 * o] = I0; ids[2 * o + 1] = I1; sqd[2 * o] = d0; sqd[2 * o + 1] = d1; }
      if (code) code[o] = ratio_code(d0, d1, I0, ratio_good, ratio_all, &n_all[pair], &n_good[pair]);
    }
  }
}

// ---- int8 fast path ---------------------------------------------------------------------------
// Same structure as k_knn2_bf16 on v_mfma_i32_32x32x32_i8 (twice the bf16 MFMA rate, K = 32 per
// instruction, exact int32 accumulation).  With A = a-128 and B' = 127-b (both fit int8),
//   |a-b|^2 = 2 A.B' + alpha(a) + beta(b),  alpha = sum (a-127)^2,  beta = sum (128-b)^2 - 128,
// the accumulator is started at (alpha >> 1) + 2^21 so m = A.B' + (alpha>>1) + 2^21 >= 0 and
// 2 m + (alpha & 1) orders the candidates of one query exactly like the distance.  The per-row word
// c = (alpha & 1) << 8 | row-in-window comes from LDS, so a candidate costs three VALU ops:
// v_lshl_or_b32 (key = m << 9 | c), v_min_u32, v_med3_u32.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

struct PairTask8 {
  const signed char* train;  // [n_train][128]  a - 128
  const signed char* query;  // [n_query][128]  127 - b
  const int* tcin;
  const int* tpar;
  const int* qbeta;
  int n_train, n_query, out_off;
};

__device__ __forceinline__ void flush_window8(u32& k0, u32& k1, u32& D0, int& I0, u32& D1, int& I1, int base) {
  if (k0 != 0xffffffffu) top2_insert(D0, I0, D1, I1, ((k0 >> 9) << 1) | ((k0 >> 8) & 1u), base + (int)(k0 & 255u));
  if (k1 != 0xffffffffu) top2_insert(D0, I0, D1, I1, ((k1 >> 9) << 1) | ((k1 >> 8) & 1u), base + (int)(k1 & 255u));
  k0 = k1 = 0xffffffffu;
}

__global__ __launch_bounds__(256, 2) void k_knn2_i8(const PairTask8* __restrict__ tasks, const int* __restrict__ tile_first, int n_pairs,
                                                     float ratio_good, float ratio_all, int32_t* __restrict__ code,
                                                     int* __restrict__ ids, float* __restrict__ sqd, int* __restrict__ n_all,
                                                     int* __restrict__ n_good) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_a[2][TT * 128];
  __shared__ __attribute__((aligned(16))) int lds_cin[2][TT];
  __shared__ __attribute__((aligned(16))) int lds_c[2][TT];
  int lo = 0, hi = n_pairs - 1;
  const int bid = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tile_first[mid] <= bid) lo = mid; else hi = mid - 1;
  }
  const int pair = lo;
  const PairTask8 T = tasks[pair];
  const int q0 = (bid - tile_first[pair]) * QPB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  const int qa = q0 + wave * 64 + r, qb = qa + 32;
  const bool va = qa < T.n_query, vb = qb < T.n_query;
  i32x4 bqa[4], bqb[4];
  {
    const signed char* pa = T.query + (size_t)(va ? qa : 0) * DIM;
    const signed char* pb = T.query + (size_t)(vb ? qb : 0) * DIM;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      bqa[ks] = *reinterpret_cast<const i32x4*>(pa + ks * 32 + h * 16);
      bqb[ks] = *reinterpret_cast<const i32x4*>(pb + ks * 32 + h * 16);
    }
  }
  u32 aD0 = 0xffffffffu, aD1 = 0xffffffffu, bD0 = 0xffffffffu, bD1 = 0xffffffffu;
  int aI0 = 0x7fffffff, aI1 = 0x7fffffff, bI0 = 0x7fffffff, bI1 = 0x7fffffff;
  u32 ak0 = 0xffffffffu, ak1 = 0xffffffffu, bk0 = 0xffffffffu, bk1 = 0xffffffffu;
  const int n_tiles = (T.n_train + TT - 1) / TT;
  // staging: 64 rows x 8 chunks of 16 B = 512 chunks, 2 per thread; chunk' = chunk ^ ((row >> 1) & 7)
  uint4 stage[2];
  int stage_cin = 0, stage_c = 0;
  auto fetch = [&](int tile) {
    const int t0 = tile * TT;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
      stage[i] = make_uint4(0, 0, 0, 0);
      if (t0 + row < T.n_train) stage[i] = *reinterpret_cast<const uint4*>(T.train + (size_t)(t0 + row) * DIM + ch * 16);
    }
    if (tid < TT) {
      const bool in = t0 + tid < T.n_train;
      stage_cin = in ? T.tcin[t0 + tid] : 0x7fffff;  // padding rows (all-zero operands) lose every comparison
      stage_c = ((in ? T.tpar[t0 + tid] : 0) << 8) | ((tile & 3) * TT + tid);
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
      *reinterpret_cast<uint4*>(&lds_a[buf][row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)]) = stage[i];
    }
    if (tid < TT) { lds_cin[buf][tid] = stage_cin; lds_c[buf][tid] = stage_c; }
  };
  fetch(0);
  commit(0);
  __syncthreads();
  int cur = 0;
  for (int tile = 0; tile < n_tiles; tile++) {
    if (tile + 1 < n_tiles) fetch(tile + 1);
    const unsigned char* la = lds_a[cur];
#pragma unroll
    for (int st = 0; st < 2; st++) {
      i32x16 acca, accb, cc;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const i32x4 nv = *reinterpret_cast<const i32x4*>(&lds_cin[cur][st * 32 + 8 * g + 4 * h]);
        const i32x4 cv = *reinterpret_cast<const i32x4*>(&lds_c[cur][st * 32 + 8 * g + 4 * h]);
        acca[4 * g + 0] = nv.x; acca[4 * g + 1] = nv.y; acca[4 * g + 2] = nv.z; acca[4 * g + 3] = nv.w;
        cc[4 * g + 0] = cv.x; cc[4 * g + 1] = cv.y; cc[4 * g + 2] = cv.z; cc[4 * g + 3] = cv.w;
      }
      accb = acca;
      const int row = st * 32 + r;
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        const int ch = 2 * ks + h;
        const i32x4 a = *reinterpret_cast<const i32x4*>(la + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
        acca = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bqa[ks], acca, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bqb[ks], accb, 0, 0, 0);
      }
#pragma unroll
      for (int reg = 0; reg < 16; reg++) {
        const u32 keya = ((u32)acca[reg] << 9) | (u32)cc[reg];
        const u32 keyb = ((u32)accb[reg] << 9) | (u32)cc[reg];
        const u32 na0 = min(ak0, keya);
        ak1 = umed3(ak0, ak1, keya);
        ak0 = na0;
        const u32 nb0 = min(bk0, keyb);
        bk1 = umed3(bk0, bk1, keyb);
        bk0 = nb0;
      }
    }
    if ((tile & 3) == 3 || tile == n_tiles - 1) {
      const int base = (tile & ~3) * TT;
      flush_window8(ak0, ak1, aD0, aI0, aD1, aI1, base);
      flush_window8(bk0, bk1, bD0, bI0, bD1, bI1, base);
    }
    if (tile + 1 < n_tiles) commit(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int s = 0; s < 2; s++) {
    u32 D0 = s ? bD0 : aD0, D1 = s ? bD1 : aD1;
    int I0 = s ? bI0 : aI0, I1 = s ? bI1 : aI1;
    const int q = s ? qb : qa;
    const bool qvalid = s ? vb : va;
    const u32 pd0 = __shfl_xor(D0, 32, 64), pd1 = __shfl_xor(D1, 32, 64);
    const int pi0 = __shfl_xor(I0, 32, 64), pi1 = __shfl_xor(I1, 32, 64);
    top2_insert(D0, I0, D1, I1, pd0, pi0);
    top2_insert(D0, I0, D1, I1, pd1, pi1);
    if (h == 0 && qvalid) {
      const int beta = T.qbeta[q];
      const float d0 = (float)((int)D0 - (1 << 22) + beta), d1 = (float)((int)D1 - (1 << 22) + beta);
      const size_t o = (size_t)T.out_off + q;
      if (ids) { ids[2 * o] = I0; ids[2 * o + 1] = I1; sqd[2 * o] = d0; sqd[2 * o + 1] = d1; }
      if (code) code[o] = ratio_code(d0, d1, I0, ratio_good, ratio_all, &n_all[pair], &n_good[pair]);
    }
  }
}

// ---- general path: exact binary64 brute force, k sequential ------------------------------
struct PairTaskF {
  const float* train;
  const float* query;
  int n_train, n_query, out_off;
};

__global__ __launch_bounds__(64) void k_knn2_exact(const PairTaskF* __restrict__ tasks, const int* __restrict__ tile_first, int n_pairs,
                                                    float ratio_good, float ratio_all, int32_t* __restrict__ code,
                                                    int* __restrict__ ids, float* __restrict__ sqd, int* __restrict__ n_all,
                                                    int* __restrict__ n_good) {
  __shared__ float ta[64 * (DIM + 1)];
  int lo = 0, hi = n_pairs - 1;
  const int bid = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tile_first[mid] <= bid) lo = mid; else hi = mid - 1;
  }
  const int pair = lo;
  const PairTaskF T = tasks[pair];
  const int q = (bid - tile_first[pair]) * 64 + threadIdx.x;
  const bool qvalid = q < T.n_query;
  float qv[DIM];
  const float* qp = T.query + (size_t)(qvalid ? q : 0) * DIM;
#pragma unroll
  for (int k = 0; k < DIM; k++) qv[k] = qp[k];
  double d0 = __builtin_inf(), d1 = __builtin_inf();
  int i0 = -1, i1 = -1;
  for (int t0 = 0; t0 < T.n_train; t0 += 64) {
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * DIM; e += 64) {
      const int row = e / DIM, k = e % DIM;
      ta[row * (DIM + 1) + k] = (t0 + row < T.n_train) ? T.train[(size_t)(t0 + row) * DIM + k] : 0.f;
    }
    __syncthreads();
    const int nrow = min(64, T.n_train - t0);
    for (int row = 0; row < nrow; row++) {
      const float* a = &ta[row * (DIM + 1)];
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < DIM; k++) {
        const double d = (double)a[k] - (double)qv[k];
        s = fma(d, d, s);
      }
      const int t = t0 + row;
      if (s < d0) { d1 = d0; i1 = i0; d0 = s; i0 = t; }
      else if (s < d1) { d1 = s; i1 = t; }
    }
  }
  if (qvalid) {
    const float f0 = (float)d0, f1 = (float)d1;
    const size_t o = (size_t)T.out_off + q;
    if (ids) { ids[2 * o] = i0; ids[2 * o + 1] = i1; sqd[2 * o] = f0; sqd[2 * o + 1] = f1; }
    if (code) code[o] = ratio_code(f0, f1, i0, ratio_good, ratio_all, &n_all[pair], &n_good[pair]);
  }
}

// ---- split-bf16 path for non-integral descriptors ---------------------------------------------
// v is split into two bf16 terms (hi + lo, 16 significant bits); a.b ~ ah.bh + ah.bl + al.bh on the
// bf16 MFMA with fp32 accumulation gives every distance to within eps (bound below).  Each query
// keeps its 4 best approximate candidates; k_rerank evaluates those 4 exactly (binary64, k
// sequential, the oracle's definition) and accepts the answer only if no candidate outside the
// shortlist can beat it:  d4_approx - eps > d2_exact.  Queries that fail the test (exact duplicates,
// near ties) are redone by the exact brute force k_exact_flagged, so the result is exact for every
// finite input; how many took the slow road is reported in the result object.
__device__ __forceinline__ unsigned short f32_to_bf16_rne(float v) {
  const u32 u = __float_as_uint(v);
  return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);  // finite inputs only
}

__global__ __launch_bounds__(256) void k_desc_prep_split(const float* __restrict__ d, int count, unsigned short* __restrict__ hi,
                                                          unsigned short* __restrict__ lo, float* __restrict__ n2,
                                                          unsigned* __restrict__ n2max) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= count) return;
  const float2 v = reinterpret_cast<const float2*>(d + (size_t)row * DIM)[lane];
  ushort2 h, l;
  h.x = f32_to_bf16_rne(v.x); h.y = f32_to_bf16_rne(v.y);
  l.x = f32_to_bf16_rne(v.x - __uint_as_float((u32)h.x << 16));
  l.y = f32_to_bf16_rne(v.y - __uint_as_float((u32)h.y << 16));
  reinterpret_cast<ushort2*>(hi + (size_t)row * DIM)[lane] = h;
  reinterpret_cast<ushort2*>(lo + (size_t)row * DIM)[lane] = l;
  double s = (double)v.x * v.x + (double)v.y * v.y;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) {
    const float f = (float)s;
    n2[row] = f;
    atomicMax(n2max, __float_as_uint(f));  // non-negative floats order like their bit patterns
  }
}

struct PairTaskS {
  const unsigned short *thi, *tlo, *qhi, *qlo;
  const float *tn2, *qn2;
  const float *tf32, *qf32;
  float tn2max;
  int n_train, n_query, out_off;
};

#define QPS 128  // queries per workgroup in the split kernel (4 waves x 32)

__device__ __forceinline__ void load_query_frags_split(const unsigned short* qp, int h, bf16x8* bq) {
#pragma unroll
  for (int ks = 0; ks < 8; ks++) {
    const uint4 raw = *reinterpret_cast<const uint4*>(qp + ks * 16 + h * 8);
    u32 w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      // bf16 x -> -2x: exponent + 1, sign flipped; zero / subnormal -> 0
      u32 a = w[j] & 0xffffu, b = w[j] >> 16;
      a = (a & 0x7F80u) ? (((a + 0x0080u) ^ 0x8000u) & 0xffffu) : 0u;
      b = (b & 0x7F80u) ? (((b + 0x0080u) ^ 0x8000u) & 0xffffu) : 0u;
      w[j] = a | (b << 16);
    }
    bq[ks] = __builtin_bit_cast(bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
  }
}

// sorted insert of key x into k0 <= k1 <= k2 <= k3
__device__ __forceinline__ void top4_insert(u32& k0, u32& k1, u32& k2, u32& k3, u32 x) {
  u32 t = max(k0, x); k0 = min(k0, x);
  u32 t2 = max(k1, t); k1 = min(k1, t);
  u32 t3 = max(k2, t2); k2 = min(k2, t2);
  k3 = min(k3, t3);
}
// (value bits, index) lists, 4 entries, ordered by (value, index)
__device__ __forceinline__ void list4_insert(u32 (&v)[4], int (&id)[4], u32 x, int xi) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const bool lt = x < v[j] || (x == v[j] && xi < id[j]);
    const u32 tv = lt ? v[j] : x;
    const int ti = lt ? id[j] : xi;
    v[j] = lt ? x : v[j];
    id[j] = lt ? xi : id[j];
    x = tv; xi = ti;
  }
}

__global__ __launch_bounds__(256, 2) void k_knn2_split(const PairTaskS* __restrict__ tasks, const int* __restrict__ tile_first, int n_pairs,
                                                        int* __restrict__ cand /*[q][4]*/, float* __restrict__ d4a /*[q]*/) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_hi[2 * TT * 256];
  __shared__ __attribute__((aligned(16))) unsigned char lds_lo[2 * TT * 256];
  __shared__ __attribute__((aligned(16))) float lds_n[2 * TT];
  int lo_ = 0, hi_ = n_pairs - 1;
  const int bid = blockIdx.x;
  while (lo_ < hi_) {
    const int mid = (lo_ + hi_ + 1) >> 1;
    if (tile_first[mid] <= bid) lo_ = mid; else hi_ = mid - 1;
  }
  const int pair = lo_;
  const PairTaskS T = tasks[pair];
  const int q0 = (bid - tile_first[pair]) * QPS;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  const int q = q0 + wave * 32 + r;
  const bool qvalid = q < T.n_query;
  bf16x8 bh[8], bl[8];
  load_query_frags_split(T.qhi + (size_t)(qvalid ? q : 0) * DIM, h, bh);
  load_query_frags_split(T.qlo + (size_t)(qvalid ? q : 0) * DIM, h, bl);
  const float b2 = T.qn2[qvalid ? q : 0];
  // every approximate distance is within eps of the exact one; shifting by 2 eps keeps them positive
  const float eps = 2.44140625e-4f * sqrtf(T.tn2max * b2) + 4.76837158e-7f * (T.tn2max + b2) + 1e-30f;
  const float shift = b2 + 2.0f * eps;
  u32 k0 = 0xffffffffu, k1 = 0xffffffffu, k2 = 0xffffffffu, k3 = 0xffffffffu;
  u32 gv[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
  int gi[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
  const int n_tiles = (T.n_train + TT - 1) / TT;
  uint4 sh[4], sl[4];
  float stage_n = 0.f;
  auto fetch = [&](int tile) {
    const int t0 = tile * TT;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int c = tid + 256 * i, row = c >> 4, ch = c & 15;
      sh[i] = make_uint4(0, 0, 0, 0); sl[i] = make_uint4(0, 0, 0, 0);
      if (t0 + row < T.n_train) {
        sh[i] = *reinterpret_cast<const uint4*>(T.thi + (size_t)(t0 + row) * DIM + ch * 8);
        sl[i] = *reinterpret_cast<const uint4*>(T.tlo + (size_t)(t0 + row) * DIM + ch * 8);
      }
    }
    if (tid < TT) stage_n = (t0 + tid < T.n_train) ? T.tn2[t0 + tid] : 3.0e38f;  // padding rows lose every comparison
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int c = tid + 256 * i, row = c >> 4, ch = c & 15;
      const int off = buf * TT * 256 + row * 256 + ((ch ^ (row & 15)) << 4);
      *reinterpret_cast<uint4*>(lds_hi + off) = sh[i];
      *reinterpret_cast<uint4*>(lds_lo + off) = sl[i];
    }
    if (tid < TT) lds_n[buf * TT + tid] = stage_n;
  };
  fetch(0);
  commit(0);
  __syncthreads();
  int cur = 0;
  for (int tile = 0; tile < n_tiles; tile++) {
    if (tile + 1 < n_tiles) fetch(tile + 1);
#pragma unroll
    for (int st = 0; st < 2; st++) {
      f32x16 acc;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const f32x4 nv = *reinterpret_cast<const f32x4*>(&lds_n[cur * TT + st * 32 + 8 * g + 4 * h]);
        acc[4 * g + 0] = nv.x; acc[4 * g + 1] = nv.y; acc[4 * g + 2] = nv.z; acc[4 * g + 3] = nv.w;
      }
      const int row = st * 32 + r;
#pragma unroll
      for (int ks = 0; ks < 8; ks++) {
        const int ch = 2 * ks + h;
        const int off = cur * TT * 256 + row * 256 + ((ch ^ (row & 15)) << 4);
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(lds_hi + off);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(lds_lo + off);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[ks], acc, 0, 0, 0);
      }
      const int wbase = ((tile & 3) * 2 + st) * 32;
#pragma unroll
      for (int reg = 0; reg < 16; reg++) {
        const float v = fmaxf(acc[reg] + shift, 0.0f);  // > 0 whenever eps is a valid bound; the clamp only guards NaN-free ordering
        const u32 key = (__float_as_uint(v) & 0xffffff00u) | (u32)(wbase + (reg & 3) + 8 * (reg >> 2));
        top4_insert(k0, k1, k2, k3, key);
      }
    }
    if ((tile & 3) == 3 || tile == n_tiles - 1) {
      const int base = (tile & ~3) * TT + 4 * h;
      const u32 ks4[4] = {k0, k1, k2, k3};
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (ks4[j] != 0xffffffffu) list4_insert(gv, gi, ks4[j] & 0xffffff00u, base + (int)(ks4[j] & 255u));
      k0 = k1 = k2 = k3 = 0xffffffffu;
    }
    if (tile + 1 < n_tiles) commit(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  // merge the two lane halves of each query
  {
    u32 pv[4]; int pi[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { pv[j] = __shfl_xor(gv[j], 32, 64); pi[j] = __shfl_xor(gi[j], 32, 64); }
#pragma unroll
    for (int j = 0; j < 4; j++) list4_insert(gv, gi, pv[j], pi[j]);
  }
  if (h == 0 && qvalid) {
    const size_t o = (size_t)T.out_off + q;
#pragma unroll
    for (int j = 0; j < 4; j++) cand[4 * o + j] = (gv[j] == 0xffffffffu || gi[j] >= T.n_train) ? -1 : gi[j];  // padding rows are not candidates
    // approximate distance of the 4th candidate (lower end of its truncation bucket), un-shifted
    d4a[o] = (gv[3] == 0xffffffffu) ? 3.0e38f : __uint_as_float(gv[3]) - 2.0f * eps;  // key value = approx distance + 2 eps
  }
}

// exact distance, the oracle's definition: binary64, k sequential, one fma per term
__device__ __forceinline__ double exact_sqdist(const float* __restrict__ a, const float* __restrict__ b) {
  double s = 0.0;
#pragma unroll 8
  for (int k = 0; k < DIM; k++) {
    const double d = (double)a[k] - (double)b[k];
    s = fma(d, d, s);
  }
  return s;
}

__global__ __launch_bounds__(256) void k_rerank(const PairTaskS* __restrict__ tasks, const int* __restrict__ qpair /*[total_q]*/,
                                                 long total_q, const int* __restrict__ cand, const float* __restrict__ d4a,
                                                 int* __restrict__ ids, float* __restrict__ sqd, int* __restrict__ flagged,
                                                 int* __restrict__ n_flagged) {
  const long o = (long)blockIdx.x * 256 + threadIdx.x;
  if (o >= total_q) return;
  const int pair = qpair[o];
  const PairTaskS T = tasks[pair];
  const int q = (int)(o - T.out_off);
  const float* qv = T.qf32 + (size_t)q * DIM;
  double d0 = __builtin_inf(), d1 = __builtin_inf();
  int i0 = -1, i1 = -1;
  for (int j = 0; j < 4; j++) {
    const int t = cand[4 * o + j];
    if (t < 0) continue;
    const double s = exact_sqdist(T.tf32 + (size_t)t * DIM, qv);
    if (s < d0 || (s == d0 && t < i0)) { d1 = d0; i1 = i0; d0 = s; i0 = t; }
    else if (s < d1 || (s == d1 && t < i1)) { d1 = s; i1 = t; }
  }
  ids[2 * o] = i0; ids[2 * o + 1] = i1;
  sqd[2 * o] = (float)d0; sqd[2 * o + 1] = (float)d1;
  if (T.n_train > 4) {
    // can a row outside the shortlist beat the second best?  Its exact distance is at least
    // d4_approx - eps (eps: split-bf16 products dropped, fp32 accumulation, key truncation).
    const float b2 = T.qn2[q];
    const double eps = 2.44140625e-4 * sqrt((double)T.tn2max * b2) + 4.76837158e-7 * ((double)T.tn2max + b2) + 1e-30;
    const double d4 = (double)d4a[o];
    // (d4a is the lower end of the 4th key's truncation bucket, so the dropped mantissa bits are already on the safe side;
    //  the 2^-22 |d4| term covers the fp32 rounding of the shift itself)
    const double bound = d4 - eps - 2.4e-7 * fabs(d4);
    if (!(bound > d1)) flagged[atomicAdd(n_flagged, 1)] = (int)o;
  }
}

// exact brute force for the queries the shortlist could not certify: one wave per flagged query
__global__ __launch_bounds__(64) void k_exact_flagged(const PairTaskS* __restrict__ tasks, const int* __restrict__ qpair,
                                                       const int* __restrict__ flagged, const int* __restrict__ n_flagged,
                                                       int* __restrict__ ids, float* __restrict__ sqd) {
  const int nf = *n_flagged, lane = threadIdx.x;
  for (int f = blockIdx.x; f < nf; f += gridDim.x) {
    const long o = flagged[f];
    const PairTaskS T = tasks[qpair[o]];
    const int q = (int)(o - T.out_off);
    const float* qv = T.qf32 + (size_t)q * DIM;
    double d0 = __builtin_inf(), d1 = __builtin_inf();
    int i0 = 0x7fffffff, i1 = 0x7fffffff;
    for (int t = lane; t < T.n_train; t += 64) {
      const double s = exact_sqdist(T.tf32 + (size_t)t * DIM, qv);
      if (s < d0) { d1 = d0; i1 = i0; d0 = s; i0 = t; }
      else if (s < d1) { d1 = s; i1 = t; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double e0 = __shfl_xor(d0, off, 64), e1 = __shfl_xor(d1, off, 64);
      const int j0 = __shfl_xor(i0, off, 64), j1 = __shfl_xor(i1, off, 64);
      // merge two sorted pairs, ties -> lower index
      if (e0 < d0 || (e0 == d0 && j0 < i0)) {
        const bool second_is_mine = d0 < e1 || (d0 == e1 && i0 < j1);
        d1 = second_is_mine ? d0 : e1; i1 = second_is_mine ? i0 : j1;
        d0 = e0; i0 = j0;
      } else if (e0 < d1 || (e0 == d1 && j0 < i1)) {
        d1 = e0; i1 = j0;
      }
    }
    if (lane == 0) { ids[2 * o] = i0; ids[2 * o + 1] = i1; sqd[2 * o] = (float)d0; sqd[2 * o + 1] = (float)d1; }
  }
}

// ratio tests on final (ids, sqd): fine_matching_graph.cc:116-133
__global__ __launch_bounds__(256) void k_codes(const int* __restrict__ qpair, long total_q, const int* __restrict__ ids,
                                                const float* __restrict__ sqd, float ratio_good, float ratio_all,
                                                int32_t* __restrict__ code, int* __restrict__ n_all, int* __restrict__ n_good) {
  const long o = (long)blockIdx.x * 256 + threadIdx.x;
  if (o >= total_q) return;
  code[o] = ratio_code(sqd[2 * o], sqd[2 * o + 1], ids[2 * o], ratio_good, ratio_all, &n_all[qpair[o]], &n_good[qpair[o]]);
}

// ---- host ---------------------------------------------------------------------------------
MSFM_API int msfm_descset_create(msfm_ctx* ctx, int n_images, int dim, msfm_descset** out) {
  if (!ctx || !out || n_images <= 0) return MSFM_E_INVAL;
  if (dim != DIM) return msfm_set_error(ctx, MSFM_E_INVAL, "descriptor dim %d not supported (128 = SIFT)", dim);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  msfm_descset* s = new msfm_descset();
  s->ctx = ctx; s->n_images = n_images; s->dim = dim;
  ctx->children++;
  s->count.assign(n_images, 0);
  s->f32.assign(n_images, nullptr); s->bf16.assign(n_images, nullptr); s->norm.assign(n_images, nullptr);
  s->shi.assign(n_images, nullptr); s->slo.assign(n_images, nullptr); s->sn2.assign(n_images, nullptr); s->sn2max.assign(n_images, 0.f);
  s->ti8.assign(n_images, nullptr); s->qi8.assign(n_images, nullptr); s->tcin.assign(n_images, nullptr); s->tpar.assign(n_images, nullptr); s->qbeta.assign(n_images, nullptr);
  if (s->n2max_dev.alloc(1) != hipSuccess || s->nonint.alloc(1) != hipSuccess || hipMemsetAsync(s->nonint.p, 0, sizeof(int), ctx->stream) != hipSuccess) {
    delete s;
    return msfm_set_error(ctx, MSFM_E_NOMEM, "descset alloc");
  }
  *out = s;
  return MSFM_OK;
}

MSFM_API void msfm_descset_destroy(msfm_descset* s) {
  if (!s) return;
  msfm_ctx* ctx = s->ctx;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto p : s->f32) delete p;
  for (auto p : s->bf16) delete p;
  for (auto p : s->norm) delete p;
  for (auto p : s->shi) delete p;
  for (auto p : s->slo) delete p;
  for (auto p : s->sn2) delete p;
  for (auto p : s->ti8) delete p;
  for (auto p : s->qi8) delete p;
  for (auto p : s->tcin) delete p;
  for (auto p : s->tpar) delete p;
  for (auto p : s->qbeta) delete p;
  delete s;
  msfm_ctx_child_released(ctx);
}

MSFM_API int msfm_descset_count(const msfm_descset* s, int image) {
  if (!s || image < 0 || image >= s->n_images) return MSFM_E_INVAL;
  return s->count[image];
}

MSFM_API int msfm_descset_upload(msfm_descset* s, int image, const float* desc, int count) {
  if (!s || image < 0 || image >= s->n_images || count < 0 || (count > 0 && !desc)) return MSFM_E_INVAL;
  msfm_ctx* ctx = s->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  // kernels of an earlier msfm_match_pairs may still be reading this image's buffers: blocks go back to the pool only
  // after their stream has drained (common.h)
  HIP_TRY(ctx, hipStreamSynchronize(st));
  s->generation++;
  delete s->f32[image]; delete s->bf16[image]; delete s->norm[image];
  delete s->shi[image]; delete s->slo[image]; delete s->sn2[image];
  s->shi[image] = new DevBuf<unsigned short>(); s->slo[image] = new DevBuf<unsigned short>(); s->sn2[image] = new DevBuf<float>();
  s->sn2max[image] = 0.f;
  delete s->ti8[image]; delete s->qi8[image]; delete s->tcin[image]; delete s->tpar[image]; delete s->qbeta[image];
  s->f32[image] = new DevBuf<float>(); s->bf16[image] = new DevBuf<unsigned short>(); s->norm[image] = new DevBuf<float>();
  s->ti8[image] = new DevBuf<signed char>(); s->qi8[image] = new DevBuf<signed char>();
  s->tcin[image] = new DevBuf<int>(); s->tpar[image] = new DevBuf<int>(); s->qbeta[image] = new DevBuf<int>();
  s->count[image] = count;
  if (count == 0) return MSFM_OK;
  HIP_TRY(ctx, s->f32[image]->alloc((size_t)count * DIM));
  HIP_TRY(ctx, s->bf16[image]->alloc((size_t)count * DIM));
  HIP_TRY(ctx, s->norm[image]->alloc(count));
  HIP_TRY(ctx, s->ti8[image]->alloc((size_t)count * DIM)); HIP_TRY(ctx, s->qi8[image]->alloc((size_t)count * DIM));
  HIP_TRY(ctx, s->tcin[image]->alloc(count)); HIP_TRY(ctx, s->tpar[image]->alloc(count)); HIP_TRY(ctx, s->qbeta[image]->alloc(count));
  HIP_TRY(ctx, s->f32[image]->upload(desc, (size_t)count * DIM, st));
  hipLaunchKernelGGL(k_desc_prep, dim3(cdiv(count, 4)), dim3(256), 0, st, s->f32[image]->p, count, s->bf16[image]->p,
                     s->norm[image]->p, s->nonint.p);
  hipLaunchKernelGGL(k_desc_prep_i8, dim3(cdiv(count, 4)), dim3(256), 0, st, s->f32[image]->p, count, s->ti8[image]->p,
                     s->qi8[image]->p, s->tcin[image]->p, s->tpar[image]->p, s->qbeta[image]->p);
  HIP_TRY(ctx, s->shi[image]->alloc((size_t)count * DIM)); HIP_TRY(ctx, s->slo[image]->alloc((size_t)count * DIM));
  HIP_TRY(ctx, s->sn2[image]->alloc(count));
  HIP_TRY(ctx, hipMemsetAsync(s->n2max_dev.p, 0, sizeof(unsigned), st));
  hipLaunchKernelGGL(k_desc_prep_split, dim3(cdiv(count, 4)), dim3(256), 0, st, s->f32[image]->p, count, s->shi[image]->p,
                     s->slo[image]->p, s->sn2[image]->p, s->n2max_dev.p);
  HIP_TRY(ctx, hipGetLastError());
  unsigned n2bits = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&s->h_nonint, s->nonint.p, sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(&n2bits, s->n2max_dev.p, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  memcpy(&s->sn2max[image], &n2bits, sizeof(float));
  return MSFM_OK;
}

struct msfm_match_result {
  msfm_descset* set;
  msfm_ctx* ctx = nullptr;   // kept separately: the descriptor set may be destroyed before its results
  int n_pairs = 0;
  long total_q = 0;
  bool keep_knn = false;
  float ratio_good = 0.6f, ratio_all = 0.85f;
  std::vector<int> pairs, out_off, nq;
  DevBuf<int32_t> code;
  DevBuf<int> ids, n_all, n_good, tile_first;
  DevBuf<float> sqd;
  DevBuf<PairTask> tasks;
  DevBuf<PairTask8> tasks8;
  DevBuf<PairTaskF> tasksf;
  DevBuf<PairTaskS> taskss;
  DevBuf<int> qpair, cand, flagged, n_flagged;
  DevBuf<float> d4a;
  int n_tiles_split = 0;
  bool use_exact = false;  // MSFM_KNN_EXACT=1: brute-force FP64 kernel for non-integral data instead of split-bf16 + re-rank
  bool use_bf16 = false;  // MSFM_KNN_BF16=1 selects the bf16 MFMA kernel instead of the int8 one
  int n_tiles = 0;
  bool exact_path = false;
  unsigned long generation = 0;   // of the descriptor set when the task tables were built
};

static int check_generation(const msfm_match_result* R) {
  if (R->generation != R->set->generation)
    return msfm_set_error(R->set->ctx, MSFM_E_INVAL, "an image of the descriptor set was uploaded again after this match result was created; "
                                                     "create a new result with msfm_match_pairs");
  return MSFM_OK;
}

static int launch_match(msfm_match_result* R) {
  msfm_ctx* ctx = R->set->ctx;
  hipStream_t st = ctx->stream;
  HIP_TRY(ctx, hipMemsetAsync(R->n_all.p, 0, sizeof(int) * R->n_pairs, st));
  HIP_TRY(ctx, hipMemsetAsync(R->n_good.p, 0, sizeof(int) * R->n_pairs, st));
  if (R->n_tiles == 0) return MSFM_OK;
  if (!R->exact_path && !R->use_bf16) {
    KTimer t(ctx, "knn2_i8_mfma");
    hipLaunchKernelGGL(k_knn2_i8, dim3(R->n_tiles), dim3(256), 0, st, R->tasks8.p, R->tile_first.p, R->n_pairs, R->ratio_good,
                       R->ratio_all, R->code.p, R->keep_knn ? R->ids.p : (int*)nullptr, R->keep_knn ? R->sqd.p : (float*)nullptr,
                       R->n_all.p, R->n_good.p);
  } else if (!R->exact_path) {
    KTimer t(ctx, "knn2_bf16_mfma");
    hipLaunchKernelGGL(k_knn2_bf16, dim3(R->n_tiles), dim3(256), 0, st, R->tasks.p, R->tile_first.p, R->n_pairs, R->ratio_good,
                       R->ratio_all, R->code.p, R->keep_knn ? R->ids.p : (int*)nullptr, R->keep_knn ? R->sqd.p : (float*)nullptr,
                       R->n_all.p, R->n_good.p);
  } else if (!R->use_exact) {
    {
      KTimer t(ctx, "knn2_split_bf16_mfma");
      hipLaunchKernelGGL(k_knn2_split, dim3(R->n_tiles_split), dim3(256), 0, st, R->taskss.p, R->tile_first.p, R->n_pairs,
                         R->cand.p, R->d4a.p);
    }
    {
      KTimer t(ctx, "knn2_rerank_f64");
      HIP_TRY(ctx, hipMemsetAsync(R->n_flagged.p, 0, sizeof(int), st));
      const int nb = cdiv(R->total_q, 256);
      hipLaunchKernelGGL(k_rerank, dim3(nb), dim3(256), 0, st, R->taskss.p, R->qpair.p, R->total_q, R->cand.p, R->d4a.p, R->ids.p,
                         R->sqd.p, R->flagged.p, R->n_flagged.p);
      hipLaunchKernelGGL(k_exact_flagged, dim3((int)std::min<long>(R->total_q, 4096)), dim3(64), 0, st, R->taskss.p, R->qpair.p,
                         R->flagged.p, R->n_flagged.p, R->ids.p, R->sqd.p);
      hipLaunchKernelGGL(k_codes, dim3(nb), dim3(256), 0, st, R->qpair.p, R->total_q, R->ids.p, R->sqd.p, R->ratio_good, R->ratio_all,
                         R->code.p, R->n_all.p, R->n_good.p);
    }
  } else {
    KTimer t(ctx, "knn2_exact_f64");
    hipLaunchKernelGGL(k_knn2_exact, dim3(R->n_tiles), dim3(64), 0, st, R->tasksf.p, R->tile_first.p, R->n_pairs, R->ratio_good,
                       R->ratio_all, R->code.p, R->keep_knn ? R->ids.p : (int*)nullptr, R->keep_knn ? R->sqd.p : (float*)nullptr,
                       R->n_all.p, R->n_good.p);
  }
  HIP_TRY(ctx, hipGetLastError());
  return MSFM_OK;
}

MSFM_API int msfm_match_pairs(msfm_descset* s, const int* pairs, int n_pairs, float ratio_good, float ratio_all, int keep_knn,
                              msfm_match_result** out) {
  if (!s || !out || n_pairs < 0 || (n_pairs > 0 && !pairs)) return MSFM_E_INVAL;
  msfm_ctx* ctx = s->ctx;
  *out = nullptr;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  for (int p = 0; p < n_pairs; p++) {
    const int a = pairs[2 * p], b = pairs[2 * p + 1];
    if (a < 0 || a >= s->n_images || b < 0 || b >= s->n_images) return msfm_set_error(ctx, MSFM_E_INVAL, "pair %d: image index out of range", p);
    if (s->count[a] < 2) return msfm_set_error(ctx, MSFM_E_INVAL, "pair %d: train image %d has %d < 2 descriptors", p, a, s->count[a]);
  }
  msfm_match_result* R = new msfm_match_result();
  struct Guard { msfm_match_result* p; msfm_ctx* c; ~Guard() { if (p) { delete p; msfm_ctx_child_released(c); } } } guard{R, ctx};
  ctx->children++;
  R->generation = s->generation;
  R->ctx = ctx;
  R->set = s; R->n_pairs = n_pairs; R->keep_knn = keep_knn != 0; R->ratio_good = ratio_good; R->ratio_all = ratio_all;
  R->pairs.assign(pairs, pairs + 2 * (size_t)n_pairs);
  R->exact_path = s->h_nonint != 0;
  { const char* e = getenv("MSFM_KNN_BF16"); R->use_bf16 = e && e[0] == '1'; }
  { const char* e = getenv("MSFM_KNN_EXACT"); R->use_exact = e && e[0] == '1'; }
  const bool split = R->exact_path && !R->use_exact;
  const int qpb = split ? QPS : (R->exact_path ? 64 : QPB);
  std::vector<PairTaskS> taskss(n_pairs);
  std::vector<int> qpair;
  std::vector<int> tile_first(n_pairs + 1, 0);
  std::vector<PairTask> tasks(n_pairs);
  std::vector<PairTaskF> tasksf(n_pairs);
  std::vector<PairTask8> tasks8(n_pairs);
  long off = 0;
  long tiles = 0;
  for (int p = 0; p < n_pairs; p++) {
    const int a = pairs[2 * p], b = pairs[2 * p + 1];
    const int nq = s->count[b];
    R->out_off.push_back((int)off);
    R->nq.push_back(nq);
    tile_first[p] = (int)tiles;
    tasks[p] = PairTask{s->bf16[a]->p, nq ? s->bf16[b]->p : nullptr, s->norm[a]->p, nq ? s->norm[b]->p : nullptr, s->count[a], nq, (int)off};
    tasks8[p] = PairTask8{s->ti8[a]->p, nq ? s->qi8[b]->p : nullptr, s->tcin[a]->p, s->tpar[a]->p, nq ? s->qbeta[b]->p : nullptr, s->count[a], nq, (int)off};
    tasksf[p] = PairTaskF{s->f32[a]->p, nq ? s->f32[b]->p : nullptr, s->count[a], nq, (int)off};
    taskss[p] = PairTaskS{s->shi[a]->p, s->slo[a]->p, nq ? s->shi[b]->p : nullptr, nq ? s->slo[b]->p : nullptr, s->sn2[a]->p,
                          nq ? s->sn2[b]->p : nullptr, s->f32[a]->p, nq ? s->f32[b]->p : nullptr, s->sn2max[a], s->count[a], nq, (int)off};
    if (split) qpair.insert(qpair.end(), (size_t)nq, p);
    off += nq;
    tiles += cdiv(nq, qpb);
    if (off > 0x7fffffffL || tiles > 0x7fffffffL) return msfm_set_error(ctx, MSFM_E_INVAL, "too many queries in one call; split the pair list");
  }
  tile_first[n_pairs] = (int)tiles;
  R->total_q = off;
  R->n_tiles = (int)tiles;
  R->n_tiles_split = split ? (int)tiles : 0;
  hipStream_t st = ctx->stream;
  HIP_TRY(ctx, R->code.alloc(std::max<long>(1, off)));
  if (split) {
    R->keep_knn = true;  // the re-rank stage produces (ids, sqd) anyway
    HIP_TRY(ctx, R->qpair.from(qpair.empty() ? std::vector<int>(1, 0) : qpair, st));
    HIP_TRY(ctx, R->cand.alloc(std::max<long>(1, 4 * off))); HIP_TRY(ctx, R->d4a.alloc(std::max<long>(1, off)));
    HIP_TRY(ctx, R->flagged.alloc(std::max<long>(1, off))); HIP_TRY(ctx, R->n_flagged.alloc(1));
    if (n_pairs) HIP_TRY(ctx, R->taskss.from(taskss, st));
  }
  if (R->keep_knn) { HIP_TRY(ctx, R->ids.alloc(std::max<long>(1, 2 * off))); HIP_TRY(ctx, R->sqd.alloc(std::max<long>(1, 2 * off))); }
  HIP_TRY(ctx, R->n_all.alloc(std::max(1, n_pairs))); HIP_TRY(ctx, R->n_good.alloc(std::max(1, n_pairs)));
  HIP_TRY(ctx, R->tile_first.from(tile_first, st));
  if (n_pairs) { HIP_TRY(ctx, R->tasks.from(tasks, st)); HIP_TRY(ctx, R->tasksf.from(tasksf, st)); HIP_TRY(ctx, R->tasks8.from(tasks8, st)); }
  HIP_TRY(ctx, hipStreamSynchronize(st));
  MSFM_TRY(launch_match(R));
  guard.p = nullptr;
  *out = R;
  return MSFM_OK;
}

MSFM_API int msfm_match_pairs_rerun(msfm_descset* s, msfm_match_result* R) {
  if (!s || !R || R->set != s) return MSFM_E_INVAL;
  MSFM_TRY(check_generation(R));
  HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
  return launch_match(R);
}

MSFM_API int msfm_match_result_counts(msfm_match_result* R, int* n_all, int* n_good) {
  if (!R) return MSFM_E_INVAL;
  msfm_ctx* ctx = R->set->ctx;
  if (R->n_pairs == 0) return MSFM_OK;
  if (n_all) HIP_TRY(ctx, hipMemcpyAsync(n_all, R->n_all.p, sizeof(int) * R->n_pairs, hipMemcpyDeviceToHost, ctx->stream));
  if (n_good) HIP_TRY(ctx, hipMemcpyAsync(n_good, R->n_good.p, sizeof(int) * R->n_pairs, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MSFM_OK;
}

MSFM_API int msfm_match_result_fetch(msfm_match_result* R, int pair, int32_t* code, int* ids, float* sqdists) {
  if (!R || pair < 0 || pair >= R->n_pairs) return MSFM_E_INVAL;
  msfm_ctx* ctx = R->set->ctx;
  MSFM_TRY(check_generation(R));
  if ((ids || sqdists) && !R->keep_knn) return msfm_set_error(ctx, MSFM_E_INVAL, "result was created without keep_knn");
  const size_t o = R->out_off[pair], n = R->nq[pair];
  hipStream_t st = ctx->stream;
  if (n == 0) return MSFM_OK;
  if (code) HIP_TRY(ctx, hipMemcpyAsync(code, R->code.p + o, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
  if (ids) HIP_TRY(ctx, hipMemcpyAsync(ids, R->ids.p + 2 * o, sizeof(int) * 2 * n, hipMemcpyDeviceToHost, st));
  if (sqdists) HIP_TRY(ctx, hipMemcpyAsync(sqdists, R->sqd.p + 2 * o, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  return MSFM_OK;
}

MSFM_API int msfm_match_result_stats(msfm_match_result* R, int* n_queries, int* n_slow_path) {
  if (!R) return MSFM_E_INVAL;
  msfm_ctx* ctx = R->set->ctx;
  int nf = 0;
  if (R->n_flagged.p) {
    HIP_TRY(ctx, hipMemcpyAsync(&nf, R->n_flagged.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (n_queries) *n_queries = (int)R->total_q;
  if (n_slow_path) *n_slow_path = nf;
  return MSFM_OK;
}

MSFM_API void msfm_match_result_destroy(msfm_match_result* R) {
  if (!R) return;
  msfm_ctx* ctx = R->ctx;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  delete R;
  msfm_ctx_child_released(ctx);
}

// The FLANN-shaped entry point: one train set, one query set, host buffers in and out.
MSFM_API int msfm_knn2_f32(msfm_ctx* ctx, const float* train, int n_train, const float* query, int n_query, int dim, int* ids,
                           float* sqdists) {
  if (!ctx || !train || n_query < 0 || (n_query > 0 && (!query || !ids || !sqdists))) return MSFM_E_INVAL;
  if (n_train < 2) return msfm_set_error(ctx, MSFM_E_INVAL, "msfm_knn2_f32 needs n_train >= 2 (got %d)", n_train);
  msfm_descset* s = nullptr;
  MSFM_TRY(msfm_descset_create(ctx, 2, dim, &s));
  int rc = msfm_descset_upload(s, 0, train, n_train);
  if (rc == MSFM_OK) rc = msfm_descset_upload(s, 1, query, n_query);
  msfm_match_result* R = nullptr;
  const int pr[2] = {0, 1};
  if (rc == MSFM_OK) rc = msfm_match_pairs(s, pr, 1, 0.6f, 0.85f, 1, &R);
  if (rc == MSFM_OK) rc = msfm_match_result_fetch(R, 0, nullptr, ids, sqdists);
  msfm_match_result_destroy(R);
  msfm_descset_destroy(s);
  return rc;
}
