// Exhaustive 2-nearest-neighbour descriptor matching on MI355X.
// Replaces flann_build_index + flann_find_nearest_neighbors_index
// (SfM/src/graph/fine_matching_graph.cc:72-81,99; SfM/src/slam_gps.cc:438-447,463) and fuses
// the ratio tests of fine_matching_graph.cc:116-133.
//
// Fast path (descriptors integer-valued in [0,255], e.g. SIFT bytes stored as float): k_knn2_i8.
//   |a-b|^2 = 2 sum (a-128)(127-b) + alpha(a) + beta(b) is evaluated as an int8 MFMA GEMM
//   (v_mfma_i32_32x32x32_i8, int32 accumulators): operands a-128 and 127-b fit int8 exactly and every sum is an
//   exact integer, so the distances are exact and the indices identical to a binary64 brute force.  One workgroup
//   = 4 waves = 256 query descriptors of one image pair; each wave keeps its 64 queries as MFMA B fragments in
//   registers for the whole sweep, train descriptors stream through a double-buffered XOR-swizzled LDS tile shared
//   by the 4 waves; the running top-2 per query lives in packed keys (m << 9 | parity << 8 | row), 3 VALU operations
//   per candidate (lshl_or, min, med3).  MSFM_KNN_BF16=1 selects the older bf16 kernel (128 queries per workgroup,
//   fp32 accumulators, 4 VALU operations per candidate) for comparison.
// General path (any float32 descriptors): one f16 MFMA product per term with a rigorous error bound, the four
//   smallest approximate distances per half-wave lane, exact binary64 re-evaluation of the plausible candidates and a
//   certificate that nothing outside the lists can win; the rare uncertified query is redone by exact brute force.
//   Bit-identical to the oracle's definition for every finite input (k_knn2_f16 below).
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>

#include "common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32;

#define DIM 128
#define TT 64          // train rows per LDS tile
#define WIN 256        // rows per packed-key window
#define QPB 256        // queries per workgroup (4 waves x 64)
#define NORM_BIAS 8388608.0f  // 2^23: keeps a2 - 2ab positive for the integer key

struct msfm_match_result;
static void orphan_result(msfm_match_result* R);
struct msfm_descset {
  msfm_ctx* ctx;
  int n_images, dim;
  std::vector<msfm_match_result*> results;   // live results of this set: orphaned (set = nullptr) when the set is destroyed first
  std::vector<DevBuf<float>*> kp;            // [count][2] keypoint positions (msfm_descset_upload_keypoints), nullptr: none
  std::vector<int> count;
  std::vector<DevBuf<float>*> f32;       // [count][dim]
  std::vector<DevBuf<unsigned short>*> bf16;  // [count][dim], train copy (plain) — query copy is scaled by -2 on load
  std::vector<DevBuf<float>*> norm;      // [count]  |a|^2
  // int8 forms for the i8 MFMA kernel: train rows a-128, query rows 127-b, and the per-row terms of
  //   |a-b|^2 = 2 sum (a-128)(127-b) + sum (a-127)^2 + sum (128-b)^2 - 128
  std::vector<DevBuf<signed char>*> ti8, qi8;   // [count][128]
  std::vector<DevBuf<int>*> tcin, tpar, qbeta;  // (alpha>>1)+2^21 ; alpha&1 ; beta
  // f16 forms for non-integral descriptors (made at match time, once the common power-of-two scale is known):
  // f16(s v), -2 f16(s v), fl32(s^2 |v|^2) and its per-image maximum
  std::vector<DevBuf<unsigned short>*> th16, qh16;  // [count][128]
  std::vector<DevBuf<float>*> n2s, rerr;            // [count]  fl32(s^2 |v|^2), |s v - f16(s v)|_2 (rounded up)
  std::vector<float> n2s_max, rerr_max;             // per image (host copies)
  std::vector<int> f16_exp;                         // log2 of the scale an image's f16 forms were made with (INT_MIN: none)
  float vabs_max = 0.f;                             // largest |value| uploaded so far
  DevBuf<unsigned> vmax_dev, n2smax_dev;            // [1], [2 n_images]: n2s maxima, then rerr maxima
  DevBuf<signed char> zero_row;          // 128 zero bytes (k_knn2_i8 reads them for train rows past the end)
  DevBuf<int> nonint;                    // OR of "not integer in [0,255]" over all uploads
  int h_nonint = 0;
  // bumped by every upload: a match result remembers the generation its device pointer tables were built at and
  // refuses to run or to be read once an image has been replaced under it
  unsigned long generation = 0;
  // bumped by every keypoint upload: the gate tables of a SLAM match result hold raw pointers into kp[]
  unsigned long kp_generation = 0;
};

// ---- prep: f32 -> bf16, squared norms, integrality flag -------------------------------
__global__ __launch_bounds__(256) void k_desc_prep(const float* __restrict__ d, int count, unsigned short* __restrict__ out,
                                                    float* __restrict__ norm, int* __restrict__ nonint, unsigned* __restrict__ vmax) {
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
  float m = fmaxf(fabsf(v.x), fabsf(v.y));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off, 64); m = fmaxf(m, __shfl_xor(m, off, 64)); }
  if (lane == 0) { norm[row] = s; atomicMax(vmax, __float_as_uint(m)); }
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
// ratio_good < 0 selects the SLAM form (slam_gps.cc:469-477): `if (ratio > th) continue;` with th = ratio_all - a ratio EQUAL
// to the threshold and a NaN ratio (0 / 0: two exact duplicates of the query) are kept, unlike `ratio < th` above.  The code is
// then the bare train index, n_good counts the survivors of this test and n_all is left to the prior F / H gates (k_slam_gate).
__device__ __forceinline__ int32_t ratio_code(float d0, float d1, int id0, float ratio_good, float ratio_all, int* n_all, int* n_good) {
  const float ratio = d0 / d1;  // fine_matching_graph.cc:118, slam_gps.cc:470
  if (ratio_good < 0.f) {
    if (ratio > ratio_all) return -1;
    atomicAdd(n_good, 1);
    return id0;
  }
  const bool good = ratio < ratio_good, all = ratio < ratio_all;
  if (good) atomicAdd(n_good, 1);
  if (all) atomicAdd(n_all, 1);
  if (!good && !all) return -1;
  return id0 | (good ? MSFM_MATCH_GOOD : 0) | (all ? 0 : MSFM_MATCH_NOT_ALL);
}

// merge candidate (d, i) into the sorted pair (d0,i0) <= (d1,i1); ties keep the earlier (lower index first)
// (selects, not branches: with `if (lt0) ... else if (lt1) ...` the compiler kept the four values of every query in scratch
//  memory and wrote them through computed addresses - ten dependent scratch round trips per window flush)
__device__ __forceinline__ void top2_insert(u32& d0, int& i0, u32& d1, int& i1, u32 d, int i) {
  const bool lt0 = d < d0 || (d == d0 && i < i0);
  const bool lt1 = d < d1 || (d == d1 && i < i1);
  const u32 nd1 = lt0 ? d0 : (lt1 ? d : d1);
  const int ni1 = lt0 ? i0 : (lt1 ? i : i1);
  d0 = lt0 ? d : d0;
  i0 = lt0 ? i : i0;
  d1 = nd1;
  i1 = ni1;
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

#ifdef MSFM_KNN_HITSTATS
__device__ unsigned long long g_knn_hits[8][8];   // [train tile / 8][slots of a group of four with a candidate under the threshold]
extern "C" __attribute__((visibility("default"))) int msfm_dbg_knn_hits(unsigned long long* out, int reset) {
  if (reset) { static unsigned long long z[64]; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_knn_hits), z, sizeof z); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_knn_hits), 64 * sizeof(unsigned long long));
}
#endif
struct PairTask8 {
  const signed char* train;  // [n_train][128]  a - 128
  const signed char* query;  // [n_query][128]  127 - b
  const int* tcin;
  const int* tpar;
  const int* qbeta;
  int n_train, n_query, out_off;
  const signed char* zero_row;   // 128 zero bytes: what the tile fetch reads for rows past the end
};

__device__ __forceinline__ void flush_window8(u32& k0, u32& k1, u32& D0, int& I0, u32& D1, int& I1, int base) {
  // (an empty slot becomes the candidate (0xffffffff, INT_MAX), which loses every comparison: no branch)
  const bool e0 = k0 == 0xffffffffu, e1 = k1 == 0xffffffffu;
  top2_insert(D0, I0, D1, I1, e0 ? 0xffffffffu : (((k0 >> 9) << 1) | ((k0 >> 8) & 1u)), e0 ? 0x7fffffff : base + (int)(k0 & 255u));
  top2_insert(D0, I0, D1, I1, e1 ? 0xffffffffu : (((k1 >> 9) << 1) | ((k1 >> 8) & 1u)), e1 ? 0x7fffffff : base + (int)(k1 & 255u));
  k0 = k1 = 0xffffffffu;
}

__global__ __launch_bounds__(256, 4) void k_knn2_i8(const PairTask8* __restrict__ tasks, const int* __restrict__ tile_first, int n_pairs,
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
  u32 aTl = 0x7fffffffu, bTl = 0x7fffffffu;
  const int n_tiles = (T.n_train + TT - 1) / TT;
  // staging: 64 rows x 8 chunks of 16 B = 512 chunks, 2 per thread; chunk' = chunk ^ ((row >> 1) & 7)
  // Round 5: the train tile goes from memory straight into LDS (global_load_lds_dwordx4, as in k_knn2_f16: the XOR swizzle on the
  // source side; a row past the end reads a row of zeros - its operands must be zero, or its sum could leave the 23 bits a key
  // has for it).  (The same fetch through registers without a branch, padding decided at the commit, had measured slower here:
  // 1 791 -> 1 736 Mmatches/s, ten registers spilled around the loop at 128.)
  int stage_cin = 0, stage_c = 0;
  auto fetch = [&](int tile, int buf) {
    const int t0 = tile * TT;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int piece = 2 * wave + i, row = 8 * piece + (lane >> 3), slot = lane & 7;
      const signed char* src = (t0 + row < T.n_train ? T.train + (size_t)(t0 + row) * DIM : T.zero_row) + ((slot ^ ((row >> 1) & 7)) << 4);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(&lds_a[buf][piece * 1024]), 16, 0, 0);
    }
    if (tid < TT) {
      const bool in = t0 + tid < T.n_train;
      stage_cin = in ? T.tcin[t0 + tid] : 0x7fffff;  // padding rows (all-zero operands) lose every comparison
      stage_c = ((in ? T.tpar[t0 + tid] : 0) << 8) | ((tile & 3) * TT + tid);
    }
  };
  auto commit = [&](int buf) {
    if (tid < TT) { lds_cin[buf][tid] = stage_cin; lds_c[buf][tid] = stage_c; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the wave's own LDS-DMA pieces have landed (s_barrier does not wait for them)
  };
  fetch(0, 0);
  commit(0);
  __syncthreads();
  int cur = 0;
  // Measured round 3 (config-2-sized images, 2 256 pairs; scripts/mfma_valu_probe.hip for the bare instruction streams): a wave
  // that issues its eight MFMAs and then its 96 selection operations spends 15.7 ns per MFMA in the first block and about 1.9 ns
  // per vector operation in the second, and the blocks of the three or four waves of a SIMD do not overlap (41-44 ns per MFMA
  // per SIMD here, MFMA pipe 0.36 busy).  The bare stream with twelve operations BETWEEN consecutive MFMAs runs at 26 ns per
  // MFMA, but three pipelined forms of this loop did not get there: selecting the previous half's accumulators between this
  // half's MFMAs with a second accumulator set (200-228 registers, two waves per SIMD; with the next half's operands and the
  // tile after next staged ahead as well) 6.8-8.2 ms against 6.2, and letting the two accumulator sets of a lane take turns
  // (b selected under a's MFMAs; 168-178 registers) 6.7-6.9 ms.  What did pay: top2_insert without branches (above).
  for (int tile = 0; tile < n_tiles; tile++) {
    if (tile + 1 < n_tiles) fetch(tile + 1, cur ^ 1);
    const unsigned char* la = lds_a[cur];
#pragma unroll
    for (int st = 0; st < 2; st++) {
      i32x16 acca, accb;
#if defined(MSFM_KNN_NOFILTER) || !defined(MSFM_KNN_CC_LDS)
      i32x16 cc;
#endif
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const i32x4 nv = *reinterpret_cast<const i32x4*>(&lds_cin[cur][st * 32 + 8 * g + 4 * h]);
        acca[4 * g + 0] = nv.x; acca[4 * g + 1] = nv.y; acca[4 * g + 2] = nv.z; acca[4 * g + 3] = nv.w;
#if defined(MSFM_KNN_NOFILTER) || !defined(MSFM_KNN_CC_LDS)
        const i32x4 cv = *reinterpret_cast<const i32x4*>(&lds_c[cur][st * 32 + 8 * g + 4 * h]);
        cc[4 * g + 0] = cv.x; cc[4 * g + 1] = cv.y; cc[4 * g + 2] = cv.z; cc[4 * g + 3] = cv.w;
#endif
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
      // Round 4: one compare per candidate.  A candidate whose m exceeds the m of the lane's current second best - of the
      // window (ak1) and of the list over all rows so far (aD1) - cannot enter the top two, whatever its low bits: its key is
      // not formed at all.  The thresholds are taken once per 32-row step (a stale threshold is only larger: conservative);
      // the three selection operations run for a register slot only when some lane of the wave has a candidate under its
      // threshold in one of the two query sets - late in the sweep that is one slot in five (a row enters a lane's top two
      // with probability 2 / (rows seen + 1)).  MSFM_KNN_NOFILTER (compile time) keeps the unconditional form.
#ifndef MSFM_KNN_NOFILTER
#ifndef MSFM_KNN_GROUP
#define MSFM_KNN_GROUP 4
#endif
#ifndef MSFM_KNN_NOSHARE
      // (aTl / bTl: the second smallest m over the lists of BOTH lanes that hold this query - see the flush below)
      u32 thra = min(ak1 >> 9, aTl), thrb = min(bk1 >> 9, bTl);
#else
      u32 thra = min(ak1 >> 9, aD1 >> 1), thrb = min(bk1 >> 9, bD1 >> 1);
#endif
      // all the compares of a group of register slots first (their results are wave masks in scalar registers - no compare ->
      // branch latency per slot), then the selection for the slots whose mask is not empty.  The key formation is volatile
      // assembly so that it stays inside the conditional block (the compiler would otherwise hoist it in front of the tests).
#ifdef MSFM_KNN_SPLIT_SETS
      // (variant: one test per query set and slot - half the candidates per test, twice the tests)
#pragma unroll
      for (int g = 0; g < 16 / MSFM_KNN_GROUP; g++) {
        unsigned long long ha[MSFM_KNN_GROUP], hb[MSFM_KNN_GROUP];
#pragma unroll
        for (int j = 0; j < MSFM_KNN_GROUP; j++) {
          ha[j] = __builtin_amdgcn_ballot_w64((u32)acca[MSFM_KNN_GROUP * g + j] <= thra);
          hb[j] = __builtin_amdgcn_ballot_w64((u32)accb[MSFM_KNN_GROUP * g + j] <= thrb);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < MSFM_KNN_GROUP; j++) {
          const int reg = MSFM_KNN_GROUP * g + j;
          if (ha[j] != 0ull) {
            u32 keya;
            asm volatile("v_lshl_or_b32 %0, %1, 9, %2" : "=v"(keya) : "v"(acca[reg]), "v"(cc[reg]));
            const u32 na0 = min(ak0, keya);
            ak1 = umed3(ak0, ak1, keya);
            ak0 = na0;
          }
          if (hb[j] != 0ull) {
            u32 keyb;
            asm volatile("v_lshl_or_b32 %0, %1, 9, %2" : "=v"(keyb) : "v"(accb[reg]), "v"(cc[reg]));
            const u32 nb0 = min(bk0, keyb);
            bk1 = umed3(bk0, bk1, keyb);
            bk0 = nb0;
          }
        }
      }
#else
#pragma unroll
      for (int g = 0; g < 16 / MSFM_KNN_GROUP; g++) {
        unsigned long long hm[MSFM_KNN_GROUP];
#ifdef MSFM_KNN_THR_GROUP
        if (g > 0) { thra = min(ak1 >> 9, aD1 >> 1); thrb = min(bk1 >> 9, bD1 >> 1); }
#endif
#pragma unroll
        for (int j = 0; j < MSFM_KNN_GROUP; j++)
          hm[j] = __builtin_amdgcn_ballot_w64((u32)acca[MSFM_KNN_GROUP * g + j] <= thra) | __builtin_amdgcn_ballot_w64((u32)accb[MSFM_KNN_GROUP * g + j] <= thrb);
        __builtin_amdgcn_sched_barrier(0);
#ifdef MSFM_KNN_HITSTATS
        if (lane == 0) {
          int nh = 0;
          for (int j = 0; j < MSFM_KNN_GROUP; j++) nh += hm[j] != 0ull;
          atomicAdd(&g_knn_hits[min(tile >> 3, 7)][nh], 1ull);
        }
#endif
#pragma unroll
        for (int j = 0; j < MSFM_KNN_GROUP; j++) {
          if (hm[j] == 0ull) continue;
          const int reg = MSFM_KNN_GROUP * g + j;
#ifndef MSFM_KNN_CC_LDS
          const u32 cw = (u32)cc[reg];
#else
          // (MSFM_KNN_CC_LDS: the row word fetched only here - sixteen registers less in the loop; measured slower: 1 707 against 1 750)
          const u32 cw = (u32)lds_c[cur][st * 32 + 8 * (reg >> 2) + 4 * h + (reg & 3)];
#endif
          u32 keya, keyb;
          asm volatile("v_lshl_or_b32 %0, %1, 9, %2" : "=v"(keya) : "v"(acca[reg]), "v"(cw));
          asm volatile("v_lshl_or_b32 %0, %1, 9, %2" : "=v"(keyb) : "v"(accb[reg]), "v"(cw));
          const u32 na0 = min(ak0, keya);
          ak1 = umed3(ak0, ak1, keya);
          ak0 = na0;
          const u32 nb0 = min(bk0, keyb);
          bk1 = umed3(bk0, bk1, keyb);
          bk0 = nb0;
        }
      }
#endif
#else
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
#endif
    }
    if ((tile & 3) == 3 || tile == n_tiles - 1) {
      const int base = (tile & ~3) * TT;
      flush_window8(ak0, ak1, aD0, aI0, aD1, aI1, base);
      flush_window8(bk0, bk1, bD0, bI0, bD1, bI1, base);
#ifndef MSFM_KNN_NOSHARE
      // A query's train rows are split between lanes r and r + 32, each with its own list.  A candidate worse than the second
      // best of the two lists TOGETHER cannot be among the query's two nearest, whichever lane it falls to: that bound (formed
      // here, where the lists change - once per 256 rows) halves the candidates that pass the compare filter late in the sweep
      // (a row enters a lane's own top two with probability 4 / rows seen, the query's with 2 / rows seen).
      {
        const auto a0 = __builtin_amdgcn_permlane32_swap(aD0 >> 1, aD0 >> 1, false, false);
        const auto a1 = __builtin_amdgcn_permlane32_swap(aD1 >> 1, aD1 >> 1, false, false);
        aTl = min(max((u32)a0[0], (u32)a0[1]), min((u32)a1[0], (u32)a1[1]));
        const auto b0 = __builtin_amdgcn_permlane32_swap(bD0 >> 1, bD0 >> 1, false, false);
        const auto b1 = __builtin_amdgcn_permlane32_swap(bD1 >> 1, bD1 >> 1, false, false);
        bTl = min(max((u32)b0[0], (u32)b0[1]), min((u32)b1[0], (u32)b1[1]));
      }
#endif
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

// ---- certified f16 path for non-integral descriptors --------------------------------------------
// The reference's extractors hand over non-integral floats (feature_extractor_vl_sift.cpp:202: 512.0F * x, never cast).
// One f16 MFMA product per term gives every squared distance to within a bound E that follows from the operand
// rounding (unit roundoff 2^-11), the fp32 accumulation and the rounding of the norms; all operands are scaled by a
// power of two s (exact) so that the largest |value| of the descriptor set sits in [2^13, 2^14):
//   v(row) = fl32(s^2 |a|^2) + SHIFT - 2 f16(s a) . f16(s b)          (MFMA, fp32 accumulators)
//   s^2 |a - b|^2  is in  [v - G - E, v + G' + E],  G = SHIFT - fl32(s^2 |b|^2)
// SHIFT (one constant per pair) keeps v positive, so the float bit patterns order like unsigned integers and a
// candidate is the key (bits & ~255) | row-in-window.  Every half-wave lane keeps the FOUR smallest keys of its half of
// the train rows (med3 chain: 5 VALU operations per candidate).  In the epilogue the two halves of a query exchange
// their lists, the candidates that can still be among the two nearest are evaluated exactly (binary64, k sequential -
// the oracle's definition) and the result is accepted only if no row outside the lists can beat it:
//   min over halves of (4th smallest v) - G - E  >  s^2 * d_second.
// Queries that fail the test (exact duplicates, near ties) are appended to a list and redone by exact brute force
// (k_exact_flagged), so the result is exact for every finite input; how many took that road is reported in the
// result object (msfm_match_result_stats).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#ifndef MSFM_KNN_F16_PROBE
#define MSFM_KNN_F16_PROBE 0
#endif

__global__ __launch_bounds__(256) void k_desc_prep_f16(const float* __restrict__ d, int count, float scale, unsigned short* __restrict__ th,
                                                        unsigned short* __restrict__ qh, float* __restrict__ n2s, float* __restrict__ rerr,
                                                        unsigned* __restrict__ n2smax, unsigned* __restrict__ rerrmax) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= count) return;
  const float2 v = reinterpret_cast<const float2*>(d + (size_t)row * DIM)[lane];
  const float sx = v.x * scale, sy = v.y * scale;     // exact: scale is a power of two (no overflow by its choice)
  const _Float16 hx = (_Float16)sx, hy = (_Float16)sy;  // round to nearest even
  const _Float16 qx = (_Float16)(-2.0f * (float)hx), qy = (_Float16)(-2.0f * (float)hy);   // exact doubling (|s v| < 2^14)
  ushort2 t, q;
  t.x = __builtin_bit_cast(unsigned short, hx); t.y = __builtin_bit_cast(unsigned short, hy);
  q.x = __builtin_bit_cast(unsigned short, qx); q.y = __builtin_bit_cast(unsigned short, qy);
  reinterpret_cast<ushort2*>(th + (size_t)row * DIM)[lane] = t;
  reinterpret_cast<ushort2*>(qh + (size_t)row * DIM)[lane] = q;
  double s = (double)sx * sx + (double)sy * sy;
  // the rounding error this row really carries into the MFMA, |s v - f16(s v)| per element - or |s v| where the matrix
  // pipe may flush a subnormal f16 operand to zero, whichever is larger
  auto elem_err = [](float x, _Float16 hx) {
    const double e = fabs((double)x - (double)(float)hx);
    return fabsf((float)hx) < 6.103515625e-5f ? fmax(e, fabs((double)x)) : e;
  };
  const double ex = elem_err(sx, hx), ey = elem_err(sy, hy);
  double r2 = ex * ex + ey * ey;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off, 64); r2 += __shfl_xor(r2, off, 64); }
  if (lane == 0) {
    const float f = (float)s;
    n2s[row] = f;
    atomicMax(n2smax, __float_as_uint(f));  // non-negative floats order like their bit patterns
    const float rr = (float)(sqrt(r2) * 1.000001) + 1e-30f;   // rounded up
    rerr[row] = rr;
    atomicMax(rerrmax, __float_as_uint(rr));
  }
}

struct PairTaskH {
  const unsigned short* th;   // f16(s a)      [n_train][128]
  const unsigned short* qh;   // -2 f16(s b)   [n_query][128]
  const float* tn2s;          // fl32(s^2 |a|^2)
  const float* qn2s;
  const float* qre;           // |s b - f16(s b)|_2 of every query row (rounded up)
  const float* tf32;          // the descriptors as uploaded, for the exact evaluations
  const float* qf32;
  float a2max_s;              // max of tn2s over the train image
  float remax_a;              // max rounding-error norm over the train image
  float shift;                // SHIFT
  float s2;                   // s^2
  int n_train, n_query, out_off;
  int group, group_off;       // pairs with the same train image form a group; its flagged queries share one list
};

// Error bound E (scaled units) of one approximate distance v against the exact s^2 |a - b|^2 (derivation: DESIGN.md §4):
//   operands: |sum (f16(s a) f16(s b) - s^2 a b)| <= r_a |f16(s b)| + |s a| r_b <= r_a (|s b| + r_b) + |s a| r_b by
//   Cauchy-Schwarz, with r the 2-norm of a row's ACTUAL rounding-error vector (computed at preparation time; for an
//   unlisted train row only the image maximum is known); twice that in the distance;
//   accumulation: fp32 sums of <= 137 terms, counted twice: 1.7e-5 (2.02 |s a||s b| + |C|), C the accumulator start;
//   the roundings of the two norms, of C and of the key truncation's upper end: 1.2e-7 (...); 2 % slack on top.
__host__ __device__ __forceinline__ double f16_error_bound(double a2max, double ramax, double b2, double rb, double shift) {
  const double na = sqrt(a2max), nb = sqrt(b2);
  const double dot = ramax * (nb + rb) + na * rb;
  return 1.02 * (2.0 * dot + 1.7e-5 * (2.02 * na * nb + a2max + shift) + 1.2e-7 * (a2max + b2 + shift)) + 1e-5;
}

// exact distance, the oracle's definition: binary64, k sequential, one fma per term
__device__ __forceinline__ double exact_sqdist(const float* __restrict__ a, const float* __restrict__ b) {
  double s = 0.0;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
#pragma unroll 4
  for (int k = 0; k < DIM / 4; k++) {
    const float4 x = a4[k], y = b4[k];
    double d = (double)x.x - (double)y.x; s = fma(d, d, s);
    d = (double)x.y - (double)y.y; s = fma(d, d, s);
    d = (double)x.z - (double)y.z; s = fma(d, d, s);
    d = (double)x.w - (double)y.w; s = fma(d, d, s);
  }
  return s;
}

// sorted insert of (value bits, index) into a 4-entry list ordered by (value, index)
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

// merge (e0, j0) <= (e1, j1) into (d0, i0) <= (d1, i1); ties -> lower index
__device__ __forceinline__ void merge_top2(double& d0, int& i0, double& d1, int& i1, double e0, int j0, double e1, int j1) {
  if (e0 < d0 || (e0 == d0 && j0 < i0)) {
    const bool second_is_mine = d0 < e1 || (d0 == e1 && i0 < j1);
    d1 = second_is_mine ? d0 : e1; i1 = second_is_mine ? i0 : j1;
    d0 = e0; i0 = j0;
  } else if (e0 < d1 || (e0 == d1 && j0 < i1)) {
    d1 = e0; i1 = j0;
  }
}

// The four smallest keys of a finished 256-row window (sorted, rows base + (key & 255)) merged into the lane's sorted list
// over all rows so far: one bitonic step (min of g[i], w[3-i]) leaves the four smallest of the eight, four
// compare-exchanges sort them.  Entries carry their window base beside the key; equal keys of different windows may land
// in either order (their values agree to the key's resolution, which is all the certificate uses).
__device__ __forceinline__ void merge_window4(u32 (&g)[4], int (&gb)[4], u32 w0, u32 w1, u32 w2, u32 w3, int wb) {
  const u32 w[4] = {w3, w2, w1, w0};
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const bool t = w[i] < g[i];
    g[i] = t ? w[i] : g[i];
    gb[i] = t ? wb : gb[i];
  }
#define MSFM_CE(a, b) { const bool t = g[b] < g[a]; const u32 ka = g[a], kb = g[b]; const int ba = gb[a], bb = gb[b]; \
                        g[a] = t ? kb : ka; g[b] = t ? ka : kb; gb[a] = t ? bb : ba; gb[b] = t ? ba : bb; }
  MSFM_CE(0, 2) MSFM_CE(1, 3) MSFM_CE(0, 1) MSFM_CE(2, 3)
#undef MSFM_CE
}

// Epilogue of one query (two lanes: r, r + 32, each with the list of its half of the train rows).
//
// Round 5: what the reference's loop consumes of a query is the index of its nearest row and the outcome of `ratio < th`
// (fine_matching_graph.cc:116-130; slam_gps.cc:469-477) - the distances themselves only when the caller keeps the 2-NN arrays.
// Every listed row's exact scaled distance lies in [val - G - E, val (1 + 2^-15) - G + E] and every other row's above
// `unl - G - E` (unl: the smaller fourth key of the two lists and the bound of the rows the compare filter dropped).  When
// the interval of the smallest value lies strictly below everybody else's lower bound, the nearest row is known; the second
// nearest DISTANCE then lies between the smallest lower bound and the smallest upper bound of the others, the float ratio is
// monotone in both distances (float conversion and float division are monotone), and if `ratio < th` comes out the same at
// both ends of its interval it is the exact answer: the code is written without a single exact evaluation (94 % of the
// queries on 512-norm SIFT-like data, scripts/knn_f16_stats.py; 0.13 exact evaluations per query instead of 2.1).  Otherwise -
// and whenever the 2-NN arrays are kept - the candidates are evaluated by the oracle's definition as before.
// (the exact evaluations of a wave are pooled: every lane files its candidates of both query sets in a work list in LDS, the
// wave walks the list sixty-four entries at a time - one lane per (query, row), all lanes busy - and every lane then reads its
// own results back.  Per-slot evaluation `if (eval_j) exact_sqdist(...)` made the whole wave wait through up to eight rounds of
// 128 dependent binary64 operations whenever one lane had a candidate in that slot.)
#ifdef MSFM_KNN_F16_STATS   // developer counters (scripts/knn_f16_stats.py): valid queries, queries decided from intervals, exact evaluations, list rounds
__device__ unsigned long long g_f16_stats[8];
extern "C" __attribute__((visibility("default"))) int msfm_dbg_f16_stats(unsigned long long* out, int reset) {
  if (reset) { static unsigned long long z[8]; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_f16_stats), z, sizeof z); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_f16_stats), 8 * sizeof(unsigned long long));
}
#endif
// Is the outcome of the ratio tests the same for every pair of distances d0 in [lo0, hi0], d1 in [lo1, hi1] (hi0 < lo1)?  The
// float distances the exact path would form lie in [(float)lo, (float)hi] (rounding is monotone), its float ratio in
// [r_lo, r_hi] (float division is monotone in both operands); the 1e-30 guard keeps zeros, subnormal ratios and 0 / 0 on the
// exact path.  fa / fb: a point of the interval (any gives the certain outcome).
__device__ __forceinline__ bool ratio_certain(double lo0, double hi0, double lo1, double hi1, float ratio_good, float ratio_all, float& fa, float& fb) {
  const float f_lo0 = (float)lo0, f_hi0 = (float)hi0, f_lo1 = (float)lo1, f_hi1 = (float)hi1;
  if (!(lo0 > 1e-30 && hi0 < lo1 && f_hi1 < 3.0e38f)) return false;
  const float r_lo = f_lo0 / f_hi1, r_hi = f_hi0 / f_lo1;
  fa = f_hi0; fb = f_lo1;
  if (ratio_good < 0.f) return (r_lo > ratio_all) == (r_hi > ratio_all);
  return ((r_lo < ratio_good) == (r_hi < ratio_good)) && ((r_lo < ratio_all) == (r_hi < ratio_all));
}
struct EpiQuery {
  u32 gv[4];
  int gi[4];
  bool eval[4];
  int pos[4];
  double unl_bound;   // lower bound (scaled exact distance) of every row outside the two lists
  bool certain;
  float fa, fb;
  int id1;
};

__device__ __forceinline__ void epi_prepare(const PairTaskH& T, int q, bool qvalid, const u32 (&gk)[4], const int (&gbase)[4], bool keep_knn,
                                            float ratio_good, float ratio_all, EpiQuery& Q) {
  u32 ov[4];
  int oi[4];
  u32 (&gv)[4] = Q.gv;
  int (&gi)[4] = Q.gi;
#pragma unroll
  for (int j = 0; j < 4; j++) {   // key -> (value bits, train row); an empty slot stays out of range
    gv[j] = gk[j] & 0xffffff00u;
    gi[j] = gk[j] == 0xffffffffu ? 0x7fffffff : gbase[j] + (int)(gk[j] & 255u);
    ov[j] = __shfl_xor(gv[j], 32, 64);
    oi[j] = __shfl_xor(gi[j], 32, 64);
  }
  const double inf = __builtin_inf();
  const double b2 = (double)T.qn2s[qvalid ? q : 0], shift = (double)T.shift;
  const double E = f16_error_bound((double)T.a2max_s, (double)T.remax_a, b2, (double)T.qre[qvalid ? q : 0], shift);
  const double G = shift - b2;
  // rows outside the two lists: v >= the 4th key of their half ...
  const double b_mine = gi[3] < T.n_train ? (double)__uint_as_float(gv[3]) : inf;
  const double b_other = oi[3] < T.n_train ? (double)__uint_as_float(ov[3]) : inf;
  double unl = fmin(b_mine, b_other);
#ifndef MSFM_KNN_F16_NOFILTER
  // ... and the rows the compare filter of the main loop dropped were above a threshold that was never smaller than
  // T = (second smallest key of the two final lists, low byte filled) (1 + 2^-15) + 2.01 E  (the loop adds 2.05 E, rounded up; a
  // window's or an earlier flush's second smallest is never below the final one, and the two smallest rows of a query are
  // never dropped nor pushed out of a four-key list), so T bounds them from below as the fourth keys bound the evicted rows.
  {
    u32 k1 = 0xffffffffu, k2 = 0xffffffffu;   // two smallest keys over both lists
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const u32 kk = (j < 4 ? gv[j] : ov[j - 4]) | 255u;   // (gv / ov: the keys with their low byte cleared; an empty slot is all ones)
      const u32 n1 = min(k1, kk);
      k2 = min(max(k1, kk), k2);
      k1 = n1;
    }
    if (k2 < 0x7f800000u) unl = fmin(unl, (double)__uint_as_float(k2) * (1.0 + 3.0517578125e-5) + 2.01 * E);
  }
#endif
  Q.unl_bound = unl - G - E;
  // scaled-distance interval of a listed row: [val - G - E, val (1 + 2^-15) - G + E]  (val = key with its low byte cleared)
  // the two smallest values over both lists (both lanes of a query see the same eight entries: every decision below that
  // both take is taken alike)
  u32 v1 = 0xffffffffu, v2 = 0xffffffffu;
  int id1 = 0x7fffffff;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const u32 kv = j < 4 ? gv[j] : ov[j - 4];
    const int ki = j < 4 ? gi[j] : oi[j - 4];
    if (ki < T.n_train) {
      if (kv < v1 || (kv == v1 && ki < id1)) { v2 = v1; v1 = kv; id1 = ki; }
      else if (kv < v2) v2 = kv;
    }
  }
  bool certain = false;
  float fa = 0.f, fb = 1.f;
#ifndef MSFM_KNN_F16_NOFASTCODE
  if (!keep_knn && v2 < 0x7f800000u) {
    const double inv_s2 = 1.0 / (double)T.s2;   // a power of two: exact
    const double x1 = (double)__uint_as_float(v1), x2 = (double)__uint_as_float(v2);
    // (the 1e-12: the oracle's 128 binary64 fused steps stay within 1.5e-14 of the real-number distance that E bounds)
    const double lo0 = (x1 - G - E) * inv_s2 * (1.0 - 1e-12), hi0 = (x1 * (1.0 + 3.0517578125e-5) - G + E) * inv_s2 * (1.0 + 1e-12);
    const double lo1 = (fmin(x2, unl) - G - E) * inv_s2 * (1.0 - 1e-12), hi1 = (x2 * (1.0 + 3.0517578125e-5) - G + E) * inv_s2 * (1.0 + 1e-12);
    certain = ratio_certain(lo0, hi0, lo1, hi1, ratio_good, ratio_all, fa, fb);
  }
#endif
  Q.certain = certain; Q.fa = fa; Q.fb = fb; Q.id1 = id1;
  double h1 = inf, h2 = inf;  // the two smallest upper bounds over both lists
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const u32 kv = j < 4 ? gv[j] : ov[j - 4];
    const int ki = j < 4 ? gi[j] : oi[j - 4];
    if (ki < T.n_train) {
      const double hi = (double)__uint_as_float(kv) * (1.0 + 3.0517578125e-5) - G + E;
      if (hi < h1) { h2 = h1; h1 = hi; } else if (hi < h2) h2 = hi;
    }
  }
  // the candidates that can still be among the two nearest: lower bound <= second smallest upper bound
#pragma unroll
  for (int j = 0; j < 4; j++)
    Q.eval[j] = qvalid && !certain && gi[j] < T.n_train && ((double)__uint_as_float(gv[j]) - G - E) <= h2;
}

// files the candidates of one query set in the wave's work list (positions by ballot + prefix count: deterministic)
__device__ __forceinline__ void epi_file(EpiQuery& Q, int q, int2* __restrict__ ent, int& n) {
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(Q.eval[j]);
    const int pos = n + (int)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
    if (Q.eval[j]) ent[pos] = make_int2(Q.gi[j], q);
    Q.pos[j] = pos;
    n += __builtin_popcountll(m);
  }
}

// Second look at the queries the f16 intervals could not decide (codes-only matching): their candidates' distances evaluated in
// binary32 by the whole wave (r32[], below) are within 2e-6 of the real value - (x - y) rounded once, squared and accumulated by
// four fused steps per lane and five tree steps: eleven roundings of 2^-24, three times covered - against E / d^2 ~ 1.6e-3 of the
// f16 product.  With b0 < b1 the two smallest of them: the nearest row is known if the intervals do not touch, the second nearest
// DISTANCE lies in b1's interval whichever row it belongs to (every other evaluated candidate is no smaller, the listed rows that
// were not evaluated and the unlisted rows are bounded away as in the exact path), and the ratio's interval is 8e-6 wide instead
// of 6e-3: all but ~0.3 % of these queries are decided here (0.126 -> 0.0003 exact evaluations per query on 512-norm SIFT-like data,
// the wave's round of 128 dependent binary64 operations runs in 0.8 % of the waves instead of 97 %).  On that data the kernel's time
// did not change (18.75 ms either way on 4 032 pairs: the round was hidden behind the CU's other workgroup); the pass is kept for
// data whose ratios crowd a threshold, where the f16 intervals decide little (-DMSFM_KNN_F16_NOREFINE: without it).
__device__ __forceinline__ void epi_refine(const PairTaskH& T, EpiQuery& Q, const float* __restrict__ r32, float ratio_good, float ratio_all) {
  float b0 = __builtin_inff(), b1 = __builtin_inff();
  int j0 = 0x7fffffff, j1 = 0x7fffffff;
  auto ins = [&](float v, int id) {
    const bool lt0 = v < b0 || (v == b0 && id < j0), lt1 = v < b1 || (v == b1 && id < j1);
    const float nb1 = lt0 ? b0 : (lt1 ? v : b1);
    const int nj1 = lt0 ? j0 : (lt1 ? id : j1);
    b0 = lt0 ? v : b0; j0 = lt0 ? id : j0; b1 = nb1; j1 = nj1;
  };
  bool any = false;
#pragma unroll
  for (int j = 0; j < 4; j++)
    if (Q.eval[j]) { ins(r32[Q.pos[j]], Q.gi[j]); any = true; }
  {
    const float p0 = __shfl_xor(b0, 32, 64), p1 = __shfl_xor(b1, 32, 64);
    const int q0 = __shfl_xor(j0, 32, 64), q1 = __shfl_xor(j1, 32, 64);
    ins(p0, q0); ins(p1, q1);
    const int any_other = __shfl_xor((int)any, 32, 64);   // (unconditionally: `any || shuffle` would skip the exchange in the lanes that have candidates)
    any = any || any_other != 0;
  }
  if (Q.certain || !any || !(b1 < 3.0e38f)) return;
  const double eps = 2e-6;
  const double lo0 = (double)b0 * (1.0 - eps), hi0 = (double)b0 * (1.0 + eps), lo1 = (double)b1 * (1.0 - eps), hi1 = (double)b1 * (1.0 + eps);
  if (!(Q.unl_bound > hi1 * (double)T.s2 * (1.0 + 1e-12))) return;   // a row outside the lists might be the second nearest: exact path / slow path
  float fa = 0.f, fb = 1.f;
  if (!ratio_certain(lo0, hi0, lo1, hi1, ratio_good, ratio_all, fa, fb)) return;
  Q.certain = true; Q.fa = fa; Q.fb = fb; Q.id1 = j0;
#pragma unroll
  for (int j = 0; j < 4; j++) Q.eval[j] = false;
}

__device__ __forceinline__ void epi_finish(const PairTaskH& T, int pair, int q, bool qvalid, int h, const EpiQuery& Q, const double* __restrict__ res,
                                           float ratio_good, float ratio_all, int32_t* __restrict__ code, int* __restrict__ ids,
                                           float* __restrict__ sqd, int* __restrict__ n_all, int* __restrict__ n_good,
                                           int* __restrict__ flagged, int* __restrict__ nf_group, int* __restrict__ n_flagged) {
  const double inf = __builtin_inf();
  double d0 = inf, d1 = inf;
  int i0 = 0x7fffffff, i1 = 0x7fffffff;
#pragma unroll
  for (int j = 0; j < 4; j++)
    if (Q.eval[j]) merge_top2(d0, i0, d1, i1, res[Q.pos[j]], Q.gi[j], inf, 0x7fffffff);
  {
    const double e0 = __shfl_xor(d0, 32, 64), e1 = __shfl_xor(d1, 32, 64);
    const int j0 = __shfl_xor(i0, 32, 64), j1 = __shfl_xor(i1, 32, 64);
    merge_top2(d0, i0, d1, i1, e0, j0, e1, j1);
  }
  if (h != 0 || !qvalid) return;
  const size_t o = (size_t)T.out_off + q;
  if (Q.certain) {
    code[o] = ratio_code(Q.fa, Q.fb, Q.id1, ratio_good, ratio_all, &n_all[pair], &n_good[pair]);
    return;
  }
  if (!(Q.unl_bound > d1 * (double)T.s2 * (1.0 + 1e-12))) {   // cannot certify: exact brute force later, filed under the train image's group
    flagged[T.group_off + atomicAdd(&nf_group[T.group], 1)] = (int)o;
    atomicAdd(n_flagged, 1);
    return;
  }
  const float f0 = (float)d0, f1 = (float)d1;
  if (ids) { ids[2 * o] = i0; ids[2 * o + 1] = i1; sqd[2 * o] = f0; sqd[2 * o + 1] = f1; }
  code[o] = ratio_code(f0, f1, i0, ratio_good, ratio_all, &n_all[pair], &n_good[pair]);
}

// 1 workgroup = 4 waves = 256 queries of one pair (two query sets of 32 per wave, every A fragment feeds two MFMAs);
// train tiles of 64 rows double-buffered through XOR-swizzled LDS, as the integer kernels.
#ifndef MSFM_KNN_F16_WGS
#define MSFM_KNN_F16_WGS 3   // (round 5: three workgroups per CU since the tile fetch needs no staging registers: 875 -> 960 Mmatches/s on 4 032 pairs)
#endif
__global__ __launch_bounds__(256, MSFM_KNN_F16_WGS) void k_knn2_f16(const PairTaskH* __restrict__ tasks, const int* __restrict__ tile_first, int n_pairs,
                                                      float ratio_good, float ratio_all, int32_t* __restrict__ code, int* __restrict__ ids,
                                                      float* __restrict__ sqd, int* __restrict__ n_all, int* __restrict__ n_good,
                                                      int* __restrict__ flagged, int* __restrict__ nf_group, int* __restrict__ n_flagged,
                                                      int debug_mode) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_a[2][TT * 256];
  __shared__ __attribute__((aligned(16))) float lds_n[2][TT];
  int lo = 0, hi = n_pairs - 1;
  const int bid = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tile_first[mid] <= bid) lo = mid; else hi = mid - 1;
  }
  const int pair = lo;
  const PairTaskH T = tasks[pair];
  const int q0 = (bid - tile_first[pair]) * QPB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, h = lane >> 5;
  const int qa = q0 + wave * 64 + r, qb = qa + 32;
  const bool va = qa < T.n_query, vb = qb < T.n_query;
  f16x8 bqa[8], bqb[8];
  {
    const unsigned short* pa = T.qh + (size_t)(va ? qa : 0) * DIM;
    const unsigned short* pb = T.qh + (size_t)(vb ? qb : 0) * DIM;
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
      bqa[ks] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(pa + ks * 16 + h * 8));
      bqb[ks] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(pb + ks * 16 + h * 8));
    }
  }
  // window keys (4 smallest of the current 256-row window) and the lane's lists over all rows so far
  u32 a0 = 0xffffffffu, a1 = 0xffffffffu, a2 = 0xffffffffu, a3 = 0xffffffffu;
  u32 b0 = 0xffffffffu, b1 = 0xffffffffu, b2 = 0xffffffffu, b3 = 0xffffffffu;
  u32 gva[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, gvb[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
  int gia[4] = {0, 0, 0, 0}, gib[4] = {0, 0, 0, 0};   // window bases of the list entries
  const int n_tiles = (T.n_train + TT - 1) / TT;
  // Round 5: the train tile goes from memory straight into LDS (global_load_lds_dwordx4: 1 KB per wave instruction, sixteen bytes
  // per lane, no staging registers and no ds_write): the XOR swizzle of the tile is applied on the SOURCE side - LDS position
  // (row, slot s) holds the row's 16-byte chunk s ^ (row & 15), so the lane that fills slot s fetches that chunk.  A row past the
  // end reads the last row (its norm makes it lose every comparison).  Sixteen registers less: three workgroups per CU fit.
  float stage_n = 0.f;
  typedef const float __attribute__((address_space(1)))* g_f32p;
  const g_f32p tn_g = (g_f32p)(uintptr_t)T.tn2s;
  auto fetch = [&](int tile, int buf) {
    const int t0 = tile * TT;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int piece = 4 * wave + i, row = 4 * piece + (lane >> 4), slot = lane & 15;
      const unsigned short* src = T.th + (size_t)min(t0 + row, T.n_train - 1) * DIM + ((slot ^ (row & 15)) << 3);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(&lds_a[buf][piece * 1024]), 16, 0, 0);
    }
    if (tid < TT) stage_n = tn_g[(unsigned)min(t0 + tid, T.n_train - 1)];
  };
  auto commit = [&](int buf, int tile) {
    if (tid < TT) lds_n[buf][tid] = (tile * TT + tid < T.n_train) ? stage_n + T.shift : 3.0e38f;  // padding rows lose every comparison
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the wave's own LDS-DMA pieces have landed (s_barrier does not wait for them)
  };
  u32 keymask;
  asm volatile("v_mov_b32 %0, 0xffffff00" : "=v"(keymask));
#ifndef MSFM_KNN_F16_NOFILTER
  // Round 5: one compare per candidate, against the query's SECOND smallest value so far plus twice the error bound (below).
  // aU / bU: threshold bits from the lists of both lanes of a query (formed at the window flush), mA / mB: 2.05 E of the lane's
  // two queries, rounded up.
  u32 aU = 0xffffffffu, bU = 0xffffffffu;
  float mA, mB;
  {
    const double sh = (double)T.shift;
    const double Ea = f16_error_bound((double)T.a2max_s, (double)T.remax_a, (double)T.qn2s[va ? qa : 0], (double)T.qre[va ? qa : 0], sh);
    const double Eb = f16_error_bound((double)T.a2max_s, (double)T.remax_a, (double)T.qn2s[vb ? qb : 0], (double)T.qre[vb ? qb : 0], sh);
    mA = __uint_as_float(__float_as_uint((float)(2.05 * Ea)) + 1u);
    mB = __uint_as_float(__float_as_uint((float)(2.05 * Eb)) + 1u);
  }
#endif
  fetch(0, 0);
  commit(0, 0);
  __syncthreads();
  int cur = 0;
  for (int tile = 0; tile < n_tiles; tile++) {
    if (tile + 1 < n_tiles) fetch(tile + 1, cur ^ 1);
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
        const f16x8 a = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(la + row * 256 + ((ch ^ (row & 15)) << 4)));
        acca = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bqa[ks], acca, 0, 0, 0);
        accb = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bqb[ks], accb, 0, 0, 0);
      }
      const int wbase = ((tile & 3) * 2 + st) * 32;
#ifndef MSFM_KNN_F16_NOFILTER
      // Round 5: one compare per candidate.  With v2 the second smallest accumulator value a query has met - in this lane's
      // current window (a1) or in the lists of both its lanes at the last flush (aU) - a row whose value exceeds
      // v2 (1 + 2^-15) + 2 E has an exact distance above that of two other rows (every value is within E of the exact scaled
      // distance, the key truncation costs 2^-15): it is not among the two nearest and no certificate needs it, so its key is
      // not formed.  (Round 4 compared with the FOURTH smallest, the bound of the lists themselves: twice to four times the
      // hit rate, measured slower than the unconditional chain.)  An empty window key is a NaN pattern: above every finite
      // value as an unsigned integer, so everything passes.  All compares of a group of four register slots first (wave masks
      // in scalar registers), then the ten selection operations for the slots with a hit in either query set.
#if MSFM_KNN_F16_PROBE == 1     // (timing probes, wrong results: no candidate ever passes)
      const u32 ta = 0u, tb = 0u;
#else
      const u32 ta = min(__float_as_uint(__builtin_fmaf(__uint_as_float(a1 | 255u), 1.0000306f, mA)), aU);
      const u32 tb = min(__float_as_uint(__builtin_fmaf(__uint_as_float(b1 | 255u), 1.0000306f, mB)), bU);
#endif
#if MSFM_KNN_F16_PROBE == 2     // (no selection at all: the accumulators are consumed by two operations per step)
      a0 = min(a0, __float_as_uint(acca[0]) ^ __float_as_uint(acca[15])); b0 = min(b0, __float_as_uint(accb[0]) ^ __float_as_uint(accb[15]));
#else
#pragma unroll
      for (int g = 0; g < 4; g++) {
        unsigned long long hm[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
          hm[j] = __builtin_amdgcn_ballot_w64(__float_as_uint(acca[4 * g + j]) <= ta) | __builtin_amdgcn_ballot_w64(__float_as_uint(accb[4 * g + j]) <= tb);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (hm[j] == 0ull) continue;
          const int reg = 4 * g + j;
          const u32 idx = (u32)(wbase + (reg & 3) + 8 * (reg >> 2));
          u32 ka, kb;
          asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(ka) : "v"(acca[reg]), "v"(keymask), "v"(idx));
          asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(kb) : "v"(accb[reg]), "v"(keymask), "v"(idx));
          a3 = umed3(a2, a3, ka); a2 = umed3(a1, a2, ka); a1 = umed3(a0, a1, ka); a0 = min(a0, ka);
          b3 = umed3(b2, b3, kb); b2 = umed3(b1, b2, kb); b1 = umed3(b0, b1, kb); b0 = min(b0, kb);
        }
      }
#endif
#else
#pragma unroll
      for (int reg = 0; reg < 16; reg++) {
        const u32 idx = (u32)(wbase + (reg & 3) + 8 * (reg >> 2));
        // one v_and_or_b32 per key: the mask sits in a VGPR (a VOP3 takes one scalar / literal operand only).  The
        // accumulators must be read by an instruction the compiler sees: it places the MFMA -> VALU wait states
        const u32 ka = (__float_as_uint(acca[reg]) & keymask) | idx;
        const u32 kb = (__float_as_uint(accb[reg]) & keymask) | idx;
        a3 = umed3(a2, a3, ka); a2 = umed3(a1, a2, ka); a1 = umed3(a0, a1, ka); a0 = min(a0, ka);
        b3 = umed3(b2, b3, kb); b2 = umed3(b1, b2, kb); b1 = umed3(b0, b1, kb); b0 = min(b0, kb);
      }
#endif
    }
    if ((tile & 3) == 3 || tile == n_tiles - 1) {
      const int base = (tile & ~3) * TT + 4 * h;
      merge_window4(gva, gia, a0, a1, a2, a3, base);
      merge_window4(gvb, gib, b0, b1, b2, b3, base);
      a0 = a1 = a2 = a3 = 0xffffffffu;
      b0 = b1 = b2 = b3 = 0xffffffffu;
#ifndef MSFM_KNN_F16_NOFILTER
      // the second smallest key over the sorted lists of BOTH lanes of a query: min(max(x0, y0), min(x1, y1))
      {
        const auto s0 = __builtin_amdgcn_permlane32_swap(gva[0], gva[0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(gva[1], gva[1], false, false);
        const u32 v2 = min(max((u32)s0[0], (u32)s0[1]), min((u32)s1[0], (u32)s1[1])) | 255u;
        aU = __float_as_uint(__builtin_fmaf(__uint_as_float(v2), 1.0000306f, mA));
        const auto t0 = __builtin_amdgcn_permlane32_swap(gvb[0], gvb[0], false, false);
        const auto t1 = __builtin_amdgcn_permlane32_swap(gvb[1], gvb[1], false, false);
        const u32 w2 = min(max((u32)t0[0], (u32)t0[1]), min((u32)t1[0], (u32)t1[1])) | 255u;
        bU = __float_as_uint(__builtin_fmaf(__uint_as_float(w2), 1.0000306f, mB));
      }
#endif
    }
    if (tile + 1 < n_tiles) commit(cur ^ 1, tile + 1);
    __syncthreads();
    cur ^= 1;
  }
  if (debug_mode == 1) {   // timing experiment: main loop only
    if (va && h == 0) code[(size_t)T.out_off + qa] = (int)gva[0] + gia[1];
    if (vb && h == 0) code[(size_t)T.out_off + qb] = (int)gvb[0] + gib[1];
    return;
  }
  // Epilogue.  Every wave is past its last read of the train tiles (the loop ends with a barrier): their 32 KB become four
  // wave-private work lists - 512 (train row, query row) entries and 512 results each (a lane files at most eight candidates).
  EpiQuery A, B;
  epi_prepare(T, qa, va, gva, gia, ids != nullptr, ratio_good, ratio_all, A);
  epi_prepare(T, qb, vb, gvb, gib, ids != nullptr, ratio_good, ratio_all, B);
  if (debug_mode == 2) {   // timing experiment: main loop + interval decisions only
    if (va && h == 0) code[(size_t)T.out_off + qa] = A.id1 + (int)A.certain;
    if (vb && h == 0) code[(size_t)T.out_off + qb] = B.id1 + (int)B.certain;
    return;
  }
  int2* ent = reinterpret_cast<int2*>(&lds_a[0][0] + wave * 8192);
  double* res = reinterpret_cast<double*>(&lds_a[0][0] + wave * 8192 + 4096);
  int n_ent = 0;
  epi_file(A, qa, ent, n_ent);
  epi_file(B, qb, ent, n_ent);
#ifdef MSFM_KNN_F16_STATS
  {
    const int nv = __builtin_popcountll(__builtin_amdgcn_ballot_w64(va && h == 0)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(vb && h == 0));
    const int nc = __builtin_popcountll(__builtin_amdgcn_ballot_w64(va && h == 0 && A.certain)) + __builtin_popcountll(__builtin_amdgcn_ballot_w64(vb && h == 0 && B.certain));
    if (lane == 0) { atomicAdd(&g_f16_stats[0], (unsigned long long)nv); atomicAdd(&g_f16_stats[1], (unsigned long long)nc);
                     atomicAdd(&g_f16_stats[2], (unsigned long long)n_ent); atomicAdd(&g_f16_stats[3], (unsigned long long)((n_ent + 63) / 64)); }
  }
#endif
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#ifndef MSFM_KNN_F16_NOREFINE
  if (!ids && n_ent > 0) {
    // binary32 distances of every filed (train row, query row) by the whole wave: lanes 0-31 fetch the train row, lanes 32-63 the
    // query row, sixteen coalesced bytes each; eight entries' loads in flight at a time (the float32 rows are cold: the sweep read
    // the f16 forms)
    float* r32 = reinterpret_cast<float*>(res);
    for (int e0 = 0; e0 < n_ent; e0 += 8) {
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int2 x = ent[min(e0 + j, n_ent - 1)];
        const float* src = (h ? T.qf32 + (size_t)x.y * DIM : T.tf32 + (size_t)x.x * DIM) + 4 * r;
        v[j] = *reinterpret_cast<const f32x4*>(src);
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        if (e0 + j < n_ent) {   // (uniform over the wave)
          const float d0 = v[j].x - __shfl_xor(v[j].x, 32, 64), d1 = v[j].y - __shfl_xor(v[j].y, 32, 64);
          const float d2 = v[j].z - __shfl_xor(v[j].z, 32, 64), d3 = v[j].w - __shfl_xor(v[j].w, 32, 64);
          float sum = d0 * d0;
          sum = __builtin_fmaf(d1, d1, sum); sum = __builtin_fmaf(d2, d2, sum); sum = __builtin_fmaf(d3, d3, sum);
#pragma unroll
          for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
          if (lane == 0) r32[e0 + j] = sum;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    epi_refine(T, A, r32, ratio_good, ratio_all);
    epi_refine(T, B, r32, ratio_good, ratio_all);
    __builtin_amdgcn_wave_barrier();   // (every lane has read its binary32 values: the exact round reuses the space)
    n_ent = 0;
    epi_file(A, qa, ent, n_ent);
    epi_file(B, qb, ent, n_ent);
#ifdef MSFM_KNN_F16_STATS
    if (lane == 0) { atomicAdd(&g_f16_stats[4], (unsigned long long)n_ent); atomicAdd(&g_f16_stats[5], (unsigned long long)((n_ent + 63) / 64)); }
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
#endif
  for (int e = lane; e < (debug_mode == 3 ? 0 : n_ent); e += 64) {   // (3: timing experiment without the exact evaluations)
    const int2 x = ent[e];
    res[e] = exact_sqdist(T.tf32 + (size_t)x.x * DIM, T.qf32 + (size_t)x.y * DIM);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  epi_finish(T, pair, qa, va, h, A, res, ratio_good, ratio_all, code, ids, sqd, n_all, n_good, flagged, nf_group, n_flagged);
  epi_finish(T, pair, qb, vb, h, B, res, ratio_good, ratio_all, code, ids, sqd, n_all, n_good, flagged, nf_group, n_flagged);
}

// Exact brute force for the queries the lists could not certify.  The f16 kernel files every such query under the group
// of its train image (consecutive pairs with the same idx1 - the reference's loop order, fine_matching_graph.cc:58,87 -
// share one list), so one workgroup per group takes them SIXTEEN at a time against each staged block of train rows: the
// 2 MB train image is read once per batch, not once per query.
// Phase 1 sweeps all train rows in binary32 (64 rows at a time, transposed through LDS: coalesced global reads, every
// lane walks its own row against four queries of its wave, whose values arrive through the scalar cache): d32 is within
// a relative 1.6e-5 of the exact value (130 roundings of 2^-24 on a sum of squares); each lane keeps its four smallest
// per query.  Phase 2 (one wave = four queries): with g1 the second smallest d32 of the whole image, only rows with
// d32 <= g1 (1 + 3.3e-5) can be among the two nearest; those few are evaluated by the oracle's definition (binary64, k
// sequential).  A query for which some lane's four slots all qualify (many duplicates) goes through the plain binary64
// sweep instead.  Writes the final (ids, sqdists), code and counts of the flagged queries.
#define XROW 132   // LDS row stride in floats: 16-byte aligned rows, conflict-free b128 reads across 16 lanes
#define XQ 16      // flagged queries per batch (4 per wave)
__global__ __launch_bounds__(256) void k_exact_flagged(const PairTaskH* __restrict__ tasks, const int* __restrict__ group_first /*[n_groups+1]*/,
                                                        const int* __restrict__ pair_off /*[n_pairs+1]*/, const int* __restrict__ flagged,
                                                        const int* __restrict__ nf_group, float ratio_good, float ratio_all,
                                                        int32_t* __restrict__ code, int* __restrict__ ids, float* __restrict__ sqd,
                                                        int* __restrict__ n_all, int* __restrict__ n_good) {
  __shared__ __attribute__((aligned(16))) float rows[64 * XROW];
  // gridDim.y workgroups share a group: workgroup y takes batches y, y + gridDim.y, ...
  const int group = blockIdx.x;
  const int nf = nf_group[group];
  if ((int)blockIdx.y * XQ >= nf) return;
  const int p_first = group_first[group], p_last = group_first[group + 1] - 1;
  const PairTaskH T = tasks[p_first];   // train side (tf32, n_train) and group_off are the same for every pair of the group
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const double inf = __builtin_inf();
  for (int f0 = (int)blockIdx.y * XQ; f0 < nf; f0 += (int)gridDim.y * XQ) {
    const int nq = min(XQ, nf - f0);
    // the four queries of this wave: wave-uniform row pointers, so their values arrive through the scalar cache and feed
    // the vector instructions as SGPR operands (staging them in LDS made the kernel LDS-bandwidth bound)
    const float* qp[4];
    int qo[4], qpair[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int slot = min(4 * wave + j, nq - 1);
      const int oq = __builtin_amdgcn_readfirstlane(flagged[T.group_off + f0 + slot]);
      int lo = p_first, hi = p_last;   // pair of flat query offset oq
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (pair_off[mid] <= oq) lo = mid; else hi = mid - 1;
      }
      lo = __builtin_amdgcn_readfirstlane(lo);
      qo[j] = oq;
      qpair[j] = lo;
      qp[j] = tasks[lo].qf32 + (size_t)(oq - pair_off[lo]) * DIM;
    }
    u32 lv[4][4];
    int li[4][4];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int k = 0; k < 4; k++) { lv[j][k] = 0xffffffffu; li[j][k] = 0x7fffffff; }
    // (round 5: the next block of train rows is on its way from memory while this one is worked on - a sweep is one workgroup
    //  walking a whole image, 64 blocks whose load latency used to stand in front of each of them: it is what a per-train-image
    //  call of a few hundred pairs waits for at the end)
    f32x4 stg[8];
    auto fetch_rows = [&](int t0) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int e = tid + 256 * i, row = e >> 5, c4 = e & 31;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        stg[i] = zero;
        if (t0 + row < T.n_train) stg[i] = *reinterpret_cast<const f32x4*>(T.tf32 + (size_t)(t0 + row) * DIM + 4 * c4);
      }
    };
    fetch_rows(0);
    for (int t0 = 0; t0 < T.n_train; t0 += 64) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int e = tid + 256 * i, row = e >> 5, c4 = e & 31;
        *reinterpret_cast<f32x4*>(&rows[row * XROW + 4 * c4]) = stg[i];
      }
      if (t0 + 64 < T.n_train) fetch_rows(t0 + 64);
      __syncthreads();
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 4
      for (int k = 0; k < DIM / 4; k++) {
        const float4 x = *reinterpret_cast<const float4*>(&rows[lane * XROW + 4 * k]);
        // constant address space + wave-uniform address = s_load_dwordx4 (the descriptors are never written by this kernel)
        typedef const f32x4 __attribute__((address_space(4)))* cf4p;
        const f32x4 y0 = ((cf4p)(uintptr_t)qp[0])[k];
        const f32x4 y1 = ((cf4p)(uintptr_t)qp[1])[k];
        const f32x4 y2 = ((cf4p)(uintptr_t)qp[2])[k];
        const f32x4 y3 = ((cf4p)(uintptr_t)qp[3])[k];
        float d;
        d = x.x - y0.x; s0 = fmaf(d, d, s0); d = x.y - y0.y; s0 = fmaf(d, d, s0); d = x.z - y0.z; s0 = fmaf(d, d, s0); d = x.w - y0.w; s0 = fmaf(d, d, s0);
        d = x.x - y1.x; s1 = fmaf(d, d, s1); d = x.y - y1.y; s1 = fmaf(d, d, s1); d = x.z - y1.z; s1 = fmaf(d, d, s1); d = x.w - y1.w; s1 = fmaf(d, d, s1);
        d = x.x - y2.x; s2 = fmaf(d, d, s2); d = x.y - y2.y; s2 = fmaf(d, d, s2); d = x.z - y2.z; s2 = fmaf(d, d, s2); d = x.w - y2.w; s2 = fmaf(d, d, s2);
        d = x.x - y3.x; s3 = fmaf(d, d, s3); d = x.y - y3.y; s3 = fmaf(d, d, s3); d = x.z - y3.z; s3 = fmaf(d, d, s3); d = x.w - y3.w; s3 = fmaf(d, d, s3);
      }
      if (t0 + lane < T.n_train) {
        const float sv[4] = {s0, s1, s2, s3};
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const u32 sb = (sv[j] == sv[j]) ? __float_as_uint(sv[j]) : 0u;   // a NaN (inf - inf) qualifies for the exact evaluation
          if (sb < lv[j][3] || (sb == lv[j][3] && t0 + lane < li[j][3])) list4_insert(lv[j], li[j], sb, t0 + lane);
        }
      }
    }
    // phase 2: wave w finishes queries 4w .. 4w+3 of the batch
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int slot = 4 * wave + j;
      if (slot >= nq) break;   // wave-uniform
      const int o = qo[j], pair = qpair[j];
      const float* qv = qp[j];
      float g0 = __uint_as_float(lv[j][0] == 0xffffffffu ? 0x7f800000u : lv[j][0]);
      float g1 = __uint_as_float(lv[j][1] == 0xffffffffu ? 0x7f800000u : lv[j][1]);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float e0 = __shfl_xor(g0, off, 64), e1 = __shfl_xor(g1, off, 64);
        const float n0 = fminf(g0, e0);
        g1 = fminf(fmaxf(g0, e0), fminf(g1, e1));
        g0 = n0;
      }
      const float tau = g1 * 1.000033f + 1e-37f;   // inf stays inf; the absolute term covers sums in the subnormal range
      const bool overflow = li[j][3] < T.n_train && __uint_as_float(lv[j][3]) <= tau;
      double d0 = inf, d1 = inf;
      int i0 = 0x7fffffff, i1 = 0x7fffffff;
      if (__any(overflow)) {
        for (int t = lane; t < T.n_train; t += 64) {   // plain binary64 sweep (many rows tie within the binary32 resolution)
          const double sx = exact_sqdist(T.tf32 + (size_t)t * DIM, qv);
          merge_top2(d0, i0, d1, i1, sx, t, inf, 0x7fffffff);
        }
      } else {
#pragma unroll
        for (int c = 0; c < 4; c++) {
          if (li[j][c] < T.n_train && __uint_as_float(lv[j][c]) <= tau) {
            const double sx = exact_sqdist(T.tf32 + (size_t)li[j][c] * DIM, qv);
            merge_top2(d0, i0, d1, i1, sx, li[j][c], inf, 0x7fffffff);
          }
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double e0 = __shfl_xor(d0, off, 64), e1 = __shfl_xor(d1, off, 64);
        const int j0 = __shfl_xor(i0, off, 64), j1 = __shfl_xor(i1, off, 64);
        merge_top2(d0, i0, d1, i1, e0, j0, e1, j1);
      }
      if (lane == 0) {
        const float f0v = (float)d0, f1v = (float)d1;
        if (ids) { ids[2 * (size_t)o] = i0; ids[2 * (size_t)o + 1] = i1; sqd[2 * (size_t)o] = f0v; sqd[2 * (size_t)o + 1] = f1v; }
        code[o] = ratio_code(f0v, f1v, i0, ratio_good, ratio_all, &n_all[pair], &n_good[pair]);
      }
    }
  }
}

// ---- host ---------------------------------------------------------------------------------
MSFM_API int msfm_descset_create(msfm_ctx* ctx, int n_images, int dim, msfm_descset** out) {
  if (!ctx || !out || n_images <= 0) return MSFM_E_INVAL;
  if (dim != DIM) return msfm_set_error(ctx, MSFM_E_INVAL, "descriptor dim %d not supported (128 = SIFT)", dim);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  msfm_descset* s = new msfm_descset();
  s->ctx = ctx; s->n_images = n_images; s->dim = dim;
  s->count.assign(n_images, 0);
  s->kp.assign(n_images, nullptr);
  s->f32.assign(n_images, nullptr); s->bf16.assign(n_images, nullptr); s->norm.assign(n_images, nullptr);
  s->th16.assign(n_images, nullptr); s->qh16.assign(n_images, nullptr); s->n2s.assign(n_images, nullptr); s->n2s_max.assign(n_images, 0.f);
  s->rerr.assign(n_images, nullptr); s->rerr_max.assign(n_images, 0.f);
  s->f16_exp.assign(n_images, INT_MIN);
  s->ti8.assign(n_images, nullptr); s->qi8.assign(n_images, nullptr); s->tcin.assign(n_images, nullptr); s->tpar.assign(n_images, nullptr); s->qbeta.assign(n_images, nullptr);
  if (s->vmax_dev.alloc(1) != hipSuccess || s->n2smax_dev.alloc(2 * (size_t)n_images) != hipSuccess || s->nonint.alloc(1) != hipSuccess ||
      s->zero_row.alloc(256) != hipSuccess || hipMemsetAsync(s->zero_row.p, 0, 256, ctx->stream) != hipSuccess ||
      hipMemsetAsync(s->nonint.p, 0, sizeof(int), ctx->stream) != hipSuccess || hipMemsetAsync(s->vmax_dev.p, 0, sizeof(unsigned), ctx->stream) != hipSuccess) {
    delete s;
    return msfm_set_error(ctx, MSFM_E_NOMEM, "descset alloc");
  }
  ctx->children++;   // only a set that exists counts: a failed create must not keep the context alive for good
  *out = s;
  return MSFM_OK;
}

MSFM_API void msfm_descset_destroy(msfm_descset* s) {
  if (!s) return;
  msfm_ctx* ctx = s->ctx;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (msfm_match_result* r : s->results) orphan_result(r);   // their codes / counts stay readable, a rerun is refused
  for (auto p : s->kp) delete p;
  for (auto p : s->f32) delete p;
  for (auto p : s->bf16) delete p;
  for (auto p : s->norm) delete p;
  for (auto p : s->th16) delete p;
  for (auto p : s->qh16) delete p;
  for (auto p : s->n2s) delete p;
  for (auto p : s->rerr) delete p;
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
  delete s->kp[image]; s->kp[image] = nullptr;   // positions belong to the features they were uploaded with
  s->kp_generation++;
  delete s->f32[image]; delete s->bf16[image]; delete s->norm[image];
  delete s->th16[image]; delete s->qh16[image]; delete s->n2s[image]; delete s->rerr[image];
  s->th16[image] = new DevBuf<unsigned short>(); s->qh16[image] = new DevBuf<unsigned short>(); s->n2s[image] = new DevBuf<float>();
  s->rerr[image] = new DevBuf<float>();
  s->n2s_max[image] = 0.f;
  s->f16_exp[image] = INT_MIN;
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
                     s->norm[image]->p, s->nonint.p, s->vmax_dev.p);
  hipLaunchKernelGGL(k_desc_prep_i8, dim3(cdiv(count, 4)), dim3(256), 0, st, s->f32[image]->p, count, s->ti8[image]->p,
                     s->qi8[image]->p, s->tcin[image]->p, s->tpar[image]->p, s->qbeta[image]->p);
  HIP_TRY(ctx, hipGetLastError());
  unsigned vbits = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&s->h_nonint, s->nonint.p, sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(&vbits, s->vmax_dev.p, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  memcpy(&s->vabs_max, &vbits, sizeof(float));   // running maximum over every upload so far (never reset)
  return MSFM_OK;
}

// f16 forms of every image at the common power-of-two scale (largest |value| of the set in [2^13, 2^14)); images whose
// forms were made at another scale (or never) are redone from their resident float32 copy.
static int ensure_f16_forms(msfm_descset* s) {
  msfm_ctx* ctx = s->ctx;
  hipStream_t st = ctx->stream;
  int e = 0;
  if (s->vabs_max > 0.f && std::isfinite(s->vabs_max)) {
    int ex = 0;
    (void)frexpf(s->vabs_max, &ex);   // vabs_max = m 2^ex, m in [0.5, 1)  ->  vabs_max * 2^(14 - ex) in [2^13, 2^14)
    e = 14 - ex;
  }
  e = std::max(-100, std::min(100, e));
  const float scale = ldexpf(1.0f, e);
  bool any = false;
  for (int i = 0; i < s->n_images; i++) {
    if (s->count[i] == 0 || s->f16_exp[i] == e) continue;
    const int count = s->count[i];
    if (!any) HIP_TRY(ctx, hipMemsetAsync(s->n2smax_dev.p, 0, sizeof(unsigned) * 2 * s->n_images, st));
    any = true;
    if (s->th16[i]->n != (size_t)count * DIM) {
      HIP_TRY(ctx, s->th16[i]->alloc((size_t)count * DIM)); HIP_TRY(ctx, s->qh16[i]->alloc((size_t)count * DIM));
      HIP_TRY(ctx, s->n2s[i]->alloc(count)); HIP_TRY(ctx, s->rerr[i]->alloc(count));
    }
    hipLaunchKernelGGL(k_desc_prep_f16, dim3(cdiv(count, 4)), dim3(256), 0, st, s->f32[i]->p, count, scale, s->th16[i]->p, s->qh16[i]->p,
                       s->n2s[i]->p, s->rerr[i]->p, s->n2smax_dev.p + i, s->n2smax_dev.p + s->n_images + i);
    s->f16_exp[i] = e;
    s->n2s_max[i] = -1.f;   // to be read back
  }
  if (any) {
    HIP_TRY(ctx, hipGetLastError());
    std::vector<unsigned> bits(2 * (size_t)s->n_images);
    HIP_TRY(ctx, hipMemcpyAsync(bits.data(), s->n2smax_dev.p, sizeof(unsigned) * bits.size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    for (int i = 0; i < s->n_images; i++)
      if (s->n2s_max[i] < 0.f) { memcpy(&s->n2s_max[i], &bits[i], sizeof(float)); memcpy(&s->rerr_max[i], &bits[s->n_images + i], sizeof(float)); }
  }
  return MSFM_OK;
}

// The two prior-geometry gates of SLAMGPS::FeatureMatching (slam_gps.cc:478-499) on the survivors of the ratio test.  One
// thread per query feature; binary64 arithmetic in the reference's order (cv::Mat products = sums from k = 0 without
// contraction, so every product and sum is rounded on its own).  A rejected match gets code -1, a kept one counts in n_all.
struct SlamGateTask {
  const float* kp1;   // train image (id1) positions [.][2]
  const float* kp2;   // query image (id2) positions
  int nq, off;
  double F[9], H[9];  // row-major, Fs[i][j] / Hs[i][j]
};
__global__ __launch_bounds__(256) void k_slam_gate(const SlamGateTask* __restrict__ tasks, double th_epipolar, double th_homography,
                                                    int32_t* __restrict__ code, int* __restrict__ n_all) {
  const SlamGateTask& t = tasks[blockIdx.y];
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= t.nq) return;
  const int32_t c = code[t.off + m];
  if (c < 0) return;
  const int i0 = c & MSFM_MATCH_ID_MASK;
  const double x1 = (double)t.kp1[2 * i0], y1 = (double)t.kp1[2 * i0 + 1];
  const double x2 = (double)t.kp2[2 * m], y2 = (double)t.kp2[2 * m + 1];
  auto row = [&](const double* A, int r) {   // (A pt1)_r with pt1 = (x1, y1, 1): cv::gemm's inner loop, k = 0, 1, 2
    return __dadd_rn(__dadd_rn(__dmul_rn(A[3 * r], x1), __dmul_rn(A[3 * r + 1], y1)), __dmul_rn(A[3 * r + 2], 1.0));
  };
  // check2 (slam_gps.cc:478-488): distance of pt2 to the epipolar line F pt1
  const double l0 = row(t.F, 0), l1 = row(t.F, 1), l2 = row(t.F, 2);
  const double dot = __dadd_rn(__dadd_rn(__dmul_rn(l0, x2), __dmul_rn(l1, y2)), __dmul_rn(l2, 1.0));
  const double epi = __ddiv_rn(fabs(dot), __dsqrt_rn(__dadd_rn(__dmul_rn(l0, l0), __dmul_rn(l1, l1))));
  bool keep = !(epi > th_epipolar);
  if (keep) {
    // check3 (slam_gps.cc:490-499): transfer distance under the prior homography
    const double h0 = row(t.H, 0), h1 = row(t.H, 1), h2 = row(t.H, 2);
    const double sc = __ddiv_rn(1.0, h2);
    const double dx = __dadd_rn(x2, -__dmul_rn(h0, sc)), dy = __dadd_rn(y2, -__dmul_rn(h1, sc));
    const double hd = __dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
    keep = !(hd > th_homography);
  }
  if (keep) atomicAdd(&n_all[blockIdx.y], 1);
  else code[t.off + m] = -1;
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
  DevBuf<PairTaskH> tasksh;
  DevBuf<int> flagged, n_flagged, pair_off, group_first, nf_group;
  int n_groups = 0;
  bool use_exact = false;  // MSFM_KNN_EXACT=1: brute-force FP64 kernel for non-integral data instead of the certified f16 path
  bool use_bf16 = false;  // MSFM_KNN_BF16=1 selects the bf16 MFMA kernel instead of the int8 one
  int n_tiles = 0;
  bool exact_path = false;
  unsigned long generation = 0;   // of the descriptor set when the task tables were built
  unsigned long kp_generation = 0;   // of its keypoints (read by the SLAM gates only)
  // SLAM form (msfm_match_pairs_slam): prior F / H gates behind the ratio test
  bool slam = false;
  DevBuf<SlamGateTask> gate_tasks;
  int gate_max_nq = 0;
  double th_epipolar = 0, th_homography = 0;
};

// The set is going away before one of its results: the result keeps what it owns (codes, counts, 2-NN arrays) and its
// context; everything that would read the set's descriptors (rerun, the slow-path kernels) is refused from now on.
static void orphan_result(msfm_match_result* R) { R->set = nullptr; }

static int check_generation(const msfm_match_result* R) {
  if (!R->set) return msfm_set_error(R->ctx, MSFM_E_INVAL, "the descriptor set of this match result has been destroyed");
  if (R->generation != R->set->generation)
    return msfm_set_error(R->ctx, MSFM_E_INVAL, "an image of the descriptor set was uploaded again after this match result was created; "
                                                     "create a new result with msfm_match_pairs");
  if (R->slam && R->kp_generation != R->set->kp_generation)
    return msfm_set_error(R->ctx, MSFM_E_INVAL, "the keypoints of an image were uploaded again after this SLAM match result was created "
                                                     "(its gate tables point at the old positions); create a new result with msfm_match_pairs_slam");
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
      KTimer t(ctx, "knn2_f16_mfma");
      HIP_TRY(ctx, hipMemsetAsync(R->n_flagged.p, 0, sizeof(int), st));
      HIP_TRY(ctx, hipMemsetAsync(R->nf_group.p, 0, sizeof(int) * std::max(1, R->n_groups), st));   // per-group flagged counters
      hipLaunchKernelGGL(k_knn2_f16, dim3(R->n_tiles), dim3(256), 0, st, R->tasksh.p, R->tile_first.p, R->n_pairs, R->ratio_good, R->ratio_all,
                         R->code.p, R->keep_knn ? R->ids.p : (int*)nullptr, R->keep_knn ? R->sqd.p : (float*)nullptr, R->n_all.p, R->n_good.p,
                         R->flagged.p, R->nf_group.p, R->n_flagged.p, getenv("MSFM_KNN_DEBUG") ? atoi(getenv("MSFM_KNN_DEBUG")) : 0);
    }
    {
      KTimer t(ctx, "knn2_exact_flagged_f64");
      if (R->n_groups)
        hipLaunchKernelGGL(k_exact_flagged, dim3(R->n_groups, std::max(1, std::min(64, cdiv(2048, R->n_groups)))), dim3(256), 0, st, R->tasksh.p, R->group_first.p, R->pair_off.p, R->flagged.p, R->nf_group.p, R->ratio_good,
                           R->ratio_all, R->code.p, R->keep_knn ? R->ids.p : (int*)nullptr, R->keep_knn ? R->sqd.p : (float*)nullptr, R->n_all.p,
                           R->n_good.p);
    }
  } else {
    KTimer t(ctx, "knn2_exact_f64");
    hipLaunchKernelGGL(k_knn2_exact, dim3(R->n_tiles), dim3(64), 0, st, R->tasksf.p, R->tile_first.p, R->n_pairs, R->ratio_good,
                       R->ratio_all, R->code.p, R->keep_knn ? R->ids.p : (int*)nullptr, R->keep_knn ? R->sqd.p : (float*)nullptr,
                       R->n_all.p, R->n_good.p);
  }
  if (R->slam && R->gate_max_nq > 0) {
    KTimer t(ctx, "slam_prior_gates");
    hipLaunchKernelGGL(k_slam_gate, dim3(cdiv(R->gate_max_nq, 256), R->n_pairs), dim3(256), 0, st, R->gate_tasks.p, R->th_epipolar,
                       R->th_homography, R->code.p, R->n_all.p);
  }
  HIP_TRY(ctx, hipGetLastError());
  return MSFM_OK;
}

// slam != nullptr: the SLAM form - ratio test `> th` (ratio_good < 0 marks it for the kernels), then the prior F / H gates.
static int match_pairs_impl(msfm_descset* s, const int* pairs, int n_pairs, float ratio_good, float ratio_all, int keep_knn,
                            const msfm_slam_match_options* slam, const double* Fs, const double* Hs, msfm_match_result** out) {
  if (!s || !out || n_pairs < 0 || (n_pairs > 0 && !pairs)) return MSFM_E_INVAL;
  msfm_ctx* ctx = s->ctx;
  *out = nullptr;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  for (int p = 0; p < n_pairs; p++) {
    const int a = pairs[2 * p], b = pairs[2 * p + 1];
    if (a < 0 || a >= s->n_images || b < 0 || b >= s->n_images) return msfm_set_error(ctx, MSFM_E_INVAL, "pair %d: image index out of range", p);
    if (s->count[a] < 2) return msfm_set_error(ctx, MSFM_E_INVAL, "pair %d: train image %d has %d < 2 descriptors", p, a, s->count[a]);
    if (slam && (!s->kp[a] || (s->count[b] > 0 && !s->kp[b])))
      return msfm_set_error(ctx, MSFM_E_INVAL, "pair %d: the prior F / H gates need the keypoints of both images (msfm_descset_upload_keypoints)", p);
  }
  msfm_match_result* R = new msfm_match_result();
  struct Guard { msfm_match_result* p; msfm_ctx* c; ~Guard() { if (p) { delete p; msfm_ctx_child_released(c); } } } guard{R, ctx};
  ctx->children++;
  R->generation = s->generation;
  R->kp_generation = s->kp_generation;
  R->ctx = ctx;
  R->set = s; R->n_pairs = n_pairs; R->keep_knn = keep_knn != 0; R->ratio_good = ratio_good; R->ratio_all = ratio_all;
  R->pairs.assign(pairs, pairs + 2 * (size_t)n_pairs);
  R->exact_path = s->h_nonint != 0;
  { const char* e = getenv("MSFM_KNN_BF16"); R->use_bf16 = e && e[0] == '1'; }
  { const char* e = getenv("MSFM_KNN_EXACT"); R->use_exact = e && e[0] == '1'; }
  const bool certified = R->exact_path && !R->use_exact;
  if (certified) MSFM_TRY(ensure_f16_forms(s));
  const int qpb = (R->exact_path && !certified) ? 64 : QPB;
  std::vector<PairTaskH> tasksh(certified ? n_pairs : 0);
  std::vector<int> pair_off(n_pairs + 1, 0), group_first;
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
    tasks8[p] = PairTask8{s->ti8[a]->p, nq ? s->qi8[b]->p : nullptr, s->tcin[a]->p, s->tpar[a]->p, nq ? s->qbeta[b]->p : nullptr, s->count[a], nq, (int)off, s->zero_row.p};
    tasksf[p] = PairTaskF{s->f32[a]->p, nq ? s->f32[b]->p : nullptr, s->count[a], nq, (int)off};
    pair_off[p] = (int)off;
    if (certified) {
      const float ex = (float)s->f16_exp[a];
      const double a2max = s->n2s_max[a], b2max = nq ? s->n2s_max[b] : 0.0;
      // SHIFT >= max |b|^2 + 2 E keeps every accumulator positive; E itself depends (weakly) on SHIFT: two sweeps settle it
      double shift = b2max;
      for (int it = 0; it < 3; it++) shift = b2max + 2.0 * f16_error_bound(a2max, s->rerr_max[a], b2max, nq ? s->rerr_max[b] : 0.0, shift * 1.001);
      float shift_f = (float)(shift * 1.001);
      shift_f = nextafterf(shift_f, INFINITY);
      if (p == 0 || pairs[2 * (p - 1)] != a) group_first.push_back(p);   // a new train image starts a new group
      tasksh[p] = PairTaskH{s->th16[a]->p, nq ? s->qh16[b]->p : nullptr, s->n2s[a]->p, nq ? s->n2s[b]->p : nullptr, nq ? s->rerr[b]->p : nullptr,
                            s->f32[a]->p, nq ? s->f32[b]->p : nullptr, (float)a2max, s->rerr_max[a], shift_f, ldexpf(1.0f, 2 * (int)ex), s->count[a], nq,
                            (int)off, (int)group_first.size() - 1, pair_off[group_first.back()]};
    }
    off += nq;
    tiles += cdiv(nq, qpb);
    if (off > 0x7fffffffL || tiles > 0x7fffffffL) return msfm_set_error(ctx, MSFM_E_INVAL, "too many queries in one call; split the pair list");
  }
  tile_first[n_pairs] = (int)tiles;
  R->total_q = off;
  R->n_tiles = (int)tiles;
  pair_off[n_pairs] = (int)off;
  hipStream_t st = ctx->stream;
  HIP_TRY(ctx, R->code.alloc(std::max<long>(1, off)));
  if (certified) {
    HIP_TRY(ctx, R->flagged.alloc(std::max<long>(1, off))); HIP_TRY(ctx, R->n_flagged.alloc(1));
    HIP_TRY(ctx, hipMemsetAsync(R->n_flagged.p, 0, sizeof(int), st));
    group_first.push_back(n_pairs);
    R->n_groups = (int)group_first.size() - 1;
    HIP_TRY(ctx, R->pair_off.from(pair_off, st));
    HIP_TRY(ctx, R->group_first.from(group_first, st));
    HIP_TRY(ctx, R->nf_group.alloc(std::max(1, R->n_groups)));   // per-group counters of flagged queries
    if (n_pairs) HIP_TRY(ctx, R->tasksh.from(tasksh, st));
  }
  if (R->keep_knn) { HIP_TRY(ctx, R->ids.alloc(std::max<long>(1, 2 * off))); HIP_TRY(ctx, R->sqd.alloc(std::max<long>(1, 2 * off))); }
  HIP_TRY(ctx, R->n_all.alloc(std::max(1, n_pairs))); HIP_TRY(ctx, R->n_good.alloc(std::max(1, n_pairs)));
  HIP_TRY(ctx, R->tile_first.from(tile_first, st));
  if (n_pairs) { HIP_TRY(ctx, R->tasks.from(tasks, st)); HIP_TRY(ctx, R->tasksf.from(tasksf, st)); HIP_TRY(ctx, R->tasks8.from(tasks8, st)); }
  if (slam) {
    std::vector<SlamGateTask> gt(n_pairs);
    for (int p = 0; p < n_pairs; p++) {
      const int a = pairs[2 * p], b = pairs[2 * p + 1];
      gt[p].kp1 = s->kp[a]->p; gt[p].kp2 = R->nq[p] ? s->kp[b]->p : nullptr; gt[p].nq = R->nq[p]; gt[p].off = R->out_off[p];
      for (int k = 0; k < 9; k++) { gt[p].F[k] = Fs[9 * (size_t)p + k]; gt[p].H[k] = Hs[9 * (size_t)p + k]; }
      R->gate_max_nq = std::max(R->gate_max_nq, R->nq[p]);
    }
    R->slam = true;
    // the reference compares binary64 distances with float thresholds: th_epipolar, and `40 * th_distance` formed in float
    R->th_epipolar = (double)slam->th_epipolar;
    R->th_homography = (double)(40.0f * slam->th_distance);
    if (n_pairs) HIP_TRY(ctx, R->gate_tasks.from(gt, st));
  }
  HIP_TRY(ctx, hipStreamSynchronize(st));
  MSFM_TRY(launch_match(R));
  guard.p = nullptr;
  s->results.push_back(R);
  *out = R;
  return MSFM_OK;
}

MSFM_API int msfm_match_pairs(msfm_descset* s, const int* pairs, int n_pairs, float ratio_good, float ratio_all, int keep_knn,
                              msfm_match_result** out) {
  if (s && !(ratio_good >= 0.f)) return msfm_set_error(s->ctx, MSFM_E_INVAL, "ratio_good must be >= 0 (got %g)", (double)ratio_good);
  return match_pairs_impl(s, pairs, n_pairs, ratio_good, ratio_all, keep_knn, nullptr, nullptr, nullptr, out);
}

MSFM_API void msfm_slam_match_default_options(msfm_slam_match_options* o) {
  if (!o) return;
  o->th_first_second_ratio = 0.80f;   // slam_gps.cc:320
  o->th_epipolar = 2.0f;              // 2.0 / resize_ratio, :316
  o->th_distance = 5.0f;              // 5.0 / resize_ratio, :317
}

MSFM_API int msfm_match_pairs_slam(msfm_descset* s, const int* pairs, int n_pairs, const double* F, const double* H,
                                   const msfm_slam_match_options* opt, int keep_knn, msfm_match_result** out) {
  if (!s || !opt || (n_pairs > 0 && (!F || !H))) return MSFM_E_INVAL;
  return match_pairs_impl(s, pairs, n_pairs, -1.0f, opt->th_first_second_ratio, keep_knn, opt, F, H, out);
}

MSFM_API int msfm_descset_upload_keypoints(msfm_descset* s, int image, const float* xy, int count) {
  if (!s || image < 0 || image >= s->n_images || count < 0 || (count > 0 && !xy)) return MSFM_E_INVAL;
  msfm_ctx* ctx = s->ctx;
  if (count != s->count[image])
    return msfm_set_error(ctx, MSFM_E_INVAL, "image %d has %d descriptors, %d keypoints given (upload the descriptors first)", image, s->count[image], count);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // a gate kernel may still be reading the old positions
  s->kp_generation++;   // SLAM results built on the old positions refuse to run from now on (check_generation)
  delete s->kp[image];
  s->kp[image] = new DevBuf<float>();
  if (count == 0) return MSFM_OK;
  HIP_TRY(ctx, s->kp[image]->alloc(2 * (size_t)count));
  HIP_TRY(ctx, s->kp[image]->upload(xy, 2 * (size_t)count, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // the caller's buffer is not retained
  return MSFM_OK;
}

int match_result_view(msfm_match_result* R, MatchView* v) {
  if (!R || !v) return MSFM_E_INVAL;
  MSFM_TRY(check_generation(R));
  const msfm_descset* s = R->set;
  v->ctx = R->ctx; v->n_images = s->n_images; v->n_pairs = R->n_pairs; v->total_q = R->total_q;
  v->pairs = R->pairs.data(); v->out_off = R->out_off.data(); v->nq = R->nq.data();
  v->code = R->code.p; v->n_all = R->n_all.p; v->n_good = R->n_good.p;
  v->count = s->count;
  v->kp.assign(s->n_images, nullptr);
  for (int i = 0; i < s->n_images; i++) if (s->kp[i]) v->kp[i] = s->kp[i]->p;
  v->slam = R->slam;
  return MSFM_OK;
}

MSFM_API int msfm_match_pairs_rerun(msfm_descset* s, msfm_match_result* R) {
  if (!s || !R || R->set != s) return MSFM_E_INVAL;   // (an orphaned result has set == nullptr)
  MSFM_TRY(check_generation(R));
  HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
  return launch_match(R);
}

MSFM_API int msfm_match_result_counts(msfm_match_result* R, int* n_all, int* n_good) {
  if (!R) return MSFM_E_INVAL;
  msfm_ctx* ctx = R->ctx;   // (the counts live in the result itself: readable even after the set is gone)
  if (R->n_pairs == 0) return MSFM_OK;
  if (n_all) HIP_TRY(ctx, hipMemcpyAsync(n_all, R->n_all.p, sizeof(int) * R->n_pairs, hipMemcpyDeviceToHost, ctx->stream));
  if (n_good) HIP_TRY(ctx, hipMemcpyAsync(n_good, R->n_good.p, sizeof(int) * R->n_pairs, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MSFM_OK;
}

MSFM_API int msfm_match_result_fetch(msfm_match_result* R, int pair, int32_t* code, int* ids, float* sqdists) {
  if (!R || pair < 0 || pair >= R->n_pairs) return MSFM_E_INVAL;
  msfm_ctx* ctx = R->ctx;
  if (R->set) MSFM_TRY(check_generation(R));   // an orphaned result (its set destroyed) still owns its codes and 2-NN arrays
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
  msfm_ctx* ctx = R->ctx;
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
  if (R->set) {
    auto& v = R->set->results;
    v.erase(std::remove(v.begin(), v.end(), R), v.end());
  }
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
