// Batched fundamental-matrix RANSAC for gfx950 — the geometric-verification stage that follows the
// ratio tests in the reference's matching loop:
//   GeoVerification::GeoVerificationFundamental   SfM/src/utils/geo_verification.cc:30-58
//     -> cv::findFundamentalMat(pt1, pt2, status, cv::FM_RANSAC, 3.0)   (OpenCV 2.4, not in the tree)
//   called per image pair from FineMatchingGraph::BuildMatchGraph, fine_matching_graph.cc:138-153.
// OpenCV's FM_RANSAC is restated from its published algorithm (CvFMEstimator): 7-point minimal
// solver (null space of the 7x9 epipolar system, cubic det(l F1 + (1-l) F2) = 0, up to three
// models per sample), symmetric squared point-to-epipolar-line error max(d1^2, d2^2) <= 3^2,
// confidence 0.99, at most 2000 samples with the adaptive stop of cvRANSACUpdateNumIters, no final
// refit.  OpenCV's random stream is not reproducible outside OpenCV, so parity with the reference
// is statistical; parity with oracle/ (same counter-based sampler) is exact:
// this file uses only + - * / sqrt on doubles, in a fixed order, with contraction off.
//
// Models: one GPU thread = one sample (k_fransac_models: the 7-point solver, work matrix in LDS).  Scoring: one LANE = one
// MATCH and the model is uniform over the wave (k_fransac_count: its nine numbers arrive through the scalar cache as SGPR
// operands, a count is the population count of the wave's inlier mask) - the first form of this file scored with one thread
// per sample walking all matches, its three models in scratch memory and lanes idle wherever a sample had fewer models:
// 5.2 ms for the first 128 samples of the 9 120 pairs of 96 images, of which the solver was 0.24.  All samples of a range are
// scored in parallel; the adaptive stop is then replayed over the per-sample inlier counts in sample order, which selects
// exactly the model the sequential loop would have kept.
#include "common.h"

#include <cfloat>
#include <mutex>
#include <unordered_map>
#include <cmath>

#pragma clang fp contract(off)

#define GEO_WAVE 64

__host__ __device__ static inline uint64_t geo_sm64(uint64_t& s) {
  s += 0x9E3779B97F4A7C15ull;
  uint64_t z = s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

struct GeoModels {
  int n;
  double F[3][9];
};

__device__ static inline double geo_det3(const double* m) {
  return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}
__device__ static inline void geo_cof3(const double* m, double* c) {
  c[0] = m[4] * m[8] - m[5] * m[7];
  c[1] = -(m[3] * m[8] - m[5] * m[6]);
  c[2] = m[3] * m[7] - m[4] * m[6];
  c[3] = -(m[1] * m[8] - m[2] * m[7]);
  c[4] = m[0] * m[8] - m[2] * m[6];
  c[5] = -(m[0] * m[7] - m[1] * m[6]);
  c[6] = m[1] * m[5] - m[2] * m[4];
  c[7] = -(m[0] * m[5] - m[2] * m[3]);
  c[8] = m[0] * m[4] - m[1] * m[3];
}

// Real roots of c3 x^3 + c2 x^2 + c1 x + c0 by bisection inside the Cauchy bound, two Newton
// steps and deflation: arithmetic and sqrt only, so that host and device agree to the last bit.
__device__ static inline int geo_cubic(double c3, double c2, double c1, double c0, double* roots) {
  const double a = c2 / c3, b = c1 / c3, c = c0 / c3;
  if (!(fabs(a) <= DBL_MAX && fabs(b) <= DBL_MAX && fabs(c) <= DBL_MAX)) {
    // not a cubic (c3 == 0 or overflow): at most the quadratic / linear roots
    if (c2 != 0.0) {
      const double p = c1 / c2, q = c0 / c2, disc = p * p - 4.0 * q;
      if (!(disc >= 0.0)) return 0;
      const double sq = sqrt(disc), t = -0.5 * (p + (p >= 0.0 ? sq : -sq));
      int n = 0;
      roots[n++] = t;
      if (t != 0.0) roots[n++] = q / t;
      return n;
    }
    if (c1 != 0.0) { roots[0] = -c0 / c1; return 1; }
    return 0;
  }
  double R = fabs(a);
  if (fabs(b) > R) R = fabs(b);
  if (fabs(c) > R) R = fabs(c);
  R = 1.0 + R;
  double lo = -R, hi = R;
  for (int it = 0; it < 100; it++) {
    const double mid = 0.5 * (lo + hi);
    const double f = ((mid + a) * mid + b) * mid + c;
    if (f <= 0.0) lo = mid; else hi = mid;
  }
  double r = 0.5 * (lo + hi);
  for (int it = 0; it < 2; it++) {
    const double f = ((r + a) * r + b) * r + c;
    const double fp = (3.0 * r + 2.0 * a) * r + b;
    if (fp != 0.0) {
      const double rn = r - f / fp;
      if (fabs(rn) <= DBL_MAX) r = rn;
    }
  }
  int n = 0;
  roots[n++] = r;
  const double p = a + r, q = b + r * p, disc = p * p - 4.0 * q;
  if (disc >= 0.0) {
    const double sq = sqrt(disc), t = -0.5 * (p + (p >= 0.0 ? sq : -sq));
    roots[n++] = t;
    if (t != 0.0) roots[n++] = q / t;
  }
  return n;
}

// The 7-point solver for sample `h` of pair `pair`.  A: this thread's 7x9 work matrix, element
// (r, c) at A[(r * 9 + c) * stride]  (LDS, one column of a [63][stride] array per thread).
__device__ static inline void geo_solve7(uint64_t seed, int pair, int h, int N, const float2* __restrict__ p1,
                                          const float2* __restrict__ p2, double* A, int stride, GeoModels& out) {
  out.n = 0;
  uint64_t s = seed ^ ((uint64_t)pair * 0xD1342543DE82EF95ull) ^ ((uint64_t)h * 0xA24BAED4963EE407ull);
  int idx[7];
  for (int k = 0; k < 7; k++) {
    for (;;) {
      const int v = (int)(geo_sm64(s) % (uint64_t)N);
      bool dup = false;
      for (int j = 0; j < k; j++) dup = dup || (idx[j] == v);
      if (!dup) { idx[k] = v; break; }
    }
  }
#define AT(r, c) A[((r) * 9 + (c)) * stride]
  for (int k = 0; k < 7; k++) {
    const double x1 = p1[idx[k]].x, y1 = p1[idx[k]].y, x2 = p2[idx[k]].x, y2 = p2[idx[k]].y;
    AT(k, 0) = x2 * x1; AT(k, 1) = x2 * y1; AT(k, 2) = x2;
    AT(k, 3) = y2 * x1; AT(k, 4) = y2 * y1; AT(k, 5) = y2;
    AT(k, 6) = x1; AT(k, 7) = y1; AT(k, 8) = 1.0;
  }
  int perm[9];
  for (int c = 0; c < 9; c++) perm[c] = c;
  // Gauss-Jordan with full pivoting (first maximum in row-major order)
  for (int i = 0; i < 7; i++) {
    int pr = i, pc = i;
    double best = -1.0;
    for (int r = i; r < 7; r++)
      for (int c = i; c < 9; c++) {
        const double v = fabs(AT(r, c));
        if (v > best) { best = v; pr = r; pc = c; }
      }
    if (!(best > 0.0)) return;  // degenerate sample (or NaN input): no model
    if (pr != i)
      for (int c = 0; c < 9; c++) { const double t = AT(i, c); AT(i, c) = AT(pr, c); AT(pr, c) = t; }
    if (pc != i) {
      for (int r = 0; r < 7; r++) { const double t = AT(r, i); AT(r, i) = AT(r, pc); AT(r, pc) = t; }
      // perm lives in registers: swap by value
      int pi = 0, pp = 0;
      for (int c = 0; c < 9; c++) { if (c == i) pi = perm[c]; if (c == pc) pp = perm[c]; }
      for (int c = 0; c < 9; c++) { if (c == i) perm[c] = pp; else if (c == pc) perm[c] = pi; }
    }
    const double piv = AT(i, i);
    for (int c = i; c < 9; c++) AT(i, c) = AT(i, c) / piv;
    for (int r = 0; r < 7; r++) {
      if (r == i) continue;
      const double f = AT(r, i);
      for (int c = i; c < 9; c++) AT(r, c) = AT(r, c) - f * AT(i, c);
    }
  }
  // null space: x = (-B[:,k], e_k) in the permuted column order
  double f1[9], f2[9];
  for (int j = 0; j < 9; j++) {
    const double xa = j < 7 ? -AT(j, 7) : (j == 7 ? 1.0 : 0.0);
    const double xb = j < 7 ? -AT(j, 8) : (j == 8 ? 1.0 : 0.0);
    for (int c = 0; c < 9; c++)
      if (c == perm[j]) { f1[c] = xa; f2[c] = xb; }
  }
#undef AT
  double G[9], cf[9];
  for (int k = 0; k < 9; k++) G[k] = f1[k] - f2[k];
  const double c0 = geo_det3(f2), c3 = geo_det3(G);
  geo_cof3(f2, cf);
  double c1 = 0.0;
  for (int k = 0; k < 9; k++) c1 = c1 + cf[k] * G[k];
  geo_cof3(G, cf);
  double c2 = 0.0;
  for (int k = 0; k < 9; k++) c2 = c2 + cf[k] * f2[k];
  double roots[3];
  const int nr = geo_cubic(c3, c2, c1, c0, roots);
  for (int q = 0; q < nr; q++) {
    const double lam = roots[q];
    double F[9];
    bool fin = true;
    for (int k = 0; k < 9; k++) { F[k] = f2[k] + lam * G[k]; fin = fin && (fabs(F[k]) <= DBL_MAX); }
    if (!fin) continue;
    const double mu = F[8];
    if (fabs(mu) > DBL_EPSILON) {
      const double inv = 1.0 / mu;
      for (int k = 0; k < 9; k++) F[k] = F[k] * inv;
    }
    for (int k = 0; k < 9; k++) out.F[out.n][k] = F[k];
    out.n++;
  }
}

// max(d1^2 / |l1|^2, d2^2 / |l2|^2) with l2 = F x1 (line in image 2), l1 = F^T x2
__device__ static inline bool geo_inlier(const double* F, double x1, double y1, double x2, double y2, double th2) {
  double a = F[0] * x1 + F[1] * y1 + F[2];
  double b = F[3] * x1 + F[4] * y1 + F[5];
  double c = F[6] * x1 + F[7] * y1 + F[8];
  const double s2 = 1.0 / (a * a + b * b);
  const double d2 = x2 * a + y2 * b + c;
  a = F[0] * x2 + F[3] * y2 + F[6];
  b = F[1] * x2 + F[4] * y2 + F[7];
  c = F[2] * x2 + F[5] * y2 + F[8];
  const double s1 = 1.0 / (a * a + b * b);
  const double d1 = x1 * a + y1 * b + c;
  const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
  const double err = e1 > e2 ? e1 : e2;
  return err <= th2;  // false for NaN
}

// The models of samples [h0, h1) of the pairs in `slot_pair`: grid (ceil((h1 - h0) / 64), n_slots), thread = sample.
// models[slot][h - h0]: up to three fundamental matrices and their number.
struct GeoModelRec { double F[3][9]; int n, pad; };
__global__ __launch_bounds__(GEO_WAVE) void k_fransac_models(int h0, int h1, const int* __restrict__ slot_pair, const int* __restrict__ n_slots_dev,
                                                              const int* __restrict__ off, const float2* __restrict__ pt1, const float2* __restrict__ pt2,
                                                              uint64_t seed, GeoModelRec* __restrict__ models) {
  __shared__ double A[63 * GEO_WAVE];
  if (n_slots_dev && (int)blockIdx.y >= *n_slots_dev) return;
  const int slot = blockIdx.y, pair = slot_pair[slot], h = h0 + blockIdx.x * GEO_WAVE + threadIdx.x;
  const int o = off[pair], N = off[pair + 1] - o;
  if (h >= h1) return;
  GeoModels m;
  m.n = 0;
  geo_solve7(seed, pair, h, N, pt1 + o, pt2 + o, A + threadIdx.x, GEO_WAVE, m);
  GeoModelRec* r = models + (size_t)slot * (h1 - h0) + (h - h0);
  for (int q = 0; q < 3; q++)
    for (int k = 0; k < 9; k++) r->F[q][k] = q < m.n ? m.F[q][k] : 0.0;
  r->n = m.n;
  r->pad = 0;
}

// Inlier counts of those models: grid (ceil((h1 - h0) / 128), n_slots), 256 threads; the pair's matches wait in LDS (1024
// at a time), wave w takes samples [32 w, 32 w + 32) of the block's 128 one after the other, lane = match.
// counts[slot][h - h0][3] (-1: no such model).
#define GEO_CNT_SAMPLES 128
// counts[slot][cs samples][3]: this launch's samples start at sample `co` of a slot's row (a range taken in several pieces shares one array).
__global__ __launch_bounds__(256) void k_fransac_count(int h0, int h1, const int* __restrict__ slot_pair, const int* __restrict__ n_slots_dev,
                                                        const int* __restrict__ off, const float2* __restrict__ pt1, const float2* __restrict__ pt2,
                                                        double th2, const GeoModelRec* __restrict__ models, int* __restrict__ counts, int cs, int co) {
  __shared__ float4 pts[1024];
  __shared__ int cnt_s[GEO_CNT_SAMPLES * 3];
  if (n_slots_dev && (int)blockIdx.y >= *n_slots_dev) return;
  const int slot = blockIdx.y, pair = slot_pair[slot];
  const int o = off[pair], N = off[pair + 1] - o;
  const int HS = h1 - h0;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int s_first = blockIdx.x * GEO_CNT_SAMPLES + 32 * wave;   // first sample (relative to h0) of this wave
  for (int e = threadIdx.x; e < GEO_CNT_SAMPLES * 3; e += 256) cnt_s[e] = 0;
  for (int base = 0; base < N; base += 1024) {
    const int nb = min(1024, N - base);
    __syncthreads();
    for (int e = threadIdx.x; e < nb; e += 256) {
      const float2 a = pt1[o + base + e], b = pt2[o + base + e];
      pts[e] = make_float4(a.x, a.y, b.x, b.y);
    }
    __syncthreads();
    for (int j = 0; j < 32; j++) {
      const int hs = s_first + j;
      if (hs >= HS) break;
      const GeoModelRec* r = models + (size_t)slot * HS + hs;
      const int nm = r->n;
      for (int q = 0; q < nm; q++) {
        double F[9];
#pragma unroll
        for (int k = 0; k < 9; k++) F[k] = r->F[q][k];
        int c = 0;
        for (int e = lane; e < nb; e += 64) {
          const float4 p = pts[e];
          c += geo_inlier(F, p.x, p.y, p.z, p.w, th2) ? 1 : 0;
        }
        for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
        if (lane == 0) cnt_s[(32 * wave + j) * 3 + q] += c;   // (only this wave touches these entries)
      }
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < GEO_CNT_SAMPLES * 3; e += 256) {
    const int hs = blockIdx.x * GEO_CNT_SAMPLES + e / 3, q = e % 3;
    if (hs < HS) counts[((size_t)slot * cs + co + hs) * 3 + q] = q < models[(size_t)slot * HS + hs].n ? cnt_s[e] : -1;
  }
}

// One wave per pair: replay OpenCV's sequential loop over the counts (niters shrinks whenever a
// better model appears: cvRANSACUpdateNumIters, tabulated on the host as R[g] per distinct N), recompute
// the winning model, write F, the inlier mask and the verdict of GeoVerificationFundamental.
// Two passes: pass 1 (grid = the pairs with enough matches, `slot_pair`) replays the first H1 samples only; pairs whose budget
// is still larger than H1 are appended to `need_list` (and finished by pass 2 over all H samples: grid = that list, the
// counts of samples [H1, H) in counts2[k]), the others are final.
__global__ __launch_bounds__(GEO_WAVE) void k_fransac_select(int H, int H1, int pass, const int* __restrict__ slot_pair, int* __restrict__ need_list,
                                                              int* __restrict__ need_count, const int* __restrict__ off, const float2* __restrict__ pt1,
                                                              const float2* __restrict__ pt2, uint64_t seed, double th2,
                                                              int min_inliers, const int* __restrict__ counts1, const int* __restrict__ counts2,
                                                              const int* __restrict__ niters_tab, const int* __restrict__ tab_off, double* __restrict__ Fout,
                                                              uint8_t* __restrict__ inlier, int* __restrict__ n_inliers,
                                                              uint8_t* __restrict__ ok) {
  __shared__ double A[63];
  __shared__ double Fw[9];
  __shared__ int win[3];
  __shared__ int cl[3 * 1024];
  const int lane = threadIdx.x;
  if (pass == 2 && (int)blockIdx.x >= *need_count) return;
  const int slot = pass == 2 ? need_list[blockIdx.x] : (int)blockIdx.x;
  const int pair = slot_pair[slot];
  const int o = off[pair], N = off[pair + 1] - o;
  const int Hscan = pass == 1 ? H1 : H;
  if (lane == 0) { win[0] = -1; win[1] = 0; win[2] = H; }
  {
    const int* R = niters_tab + tab_off[pair];  // N + 1 entries (the table of this N)
    int best = 6;                          // a model must beat modelPoints - 1
    for (int h0 = 0; h0 < Hscan; h0 += 1024) {
      __syncthreads();
      if (h0 >= win[2]) break;  // uniform: win[2] is shared
      const int nh = min(1024, Hscan - h0);
      for (int e = lane; e < 3 * nh; e += GEO_WAVE) {
        const int h = h0 + e / 3, q = e % 3;
        cl[e] = h < H1 ? counts1[((size_t)slot * H1 + h) * 3 + q] : counts2[((size_t)blockIdx.x * (H - H1) + (h - H1)) * 3 + q];
      }
      __syncthreads();
      if (lane == 0) {
        int niters = win[2];
        for (int h = h0; h < h0 + nh && h < niters; h++)
          for (int q = 0; q < 3; q++) {
            const int g = cl[(h - h0) * 3 + q];
            if (g > best) {
              best = g; win[0] = h; win[1] = q;
              const int r = R[g];
              if (r < niters) niters = r;
            }
          }
        win[2] = niters;
      }
    }
  }
  __syncthreads();
  if (pass == 1) {
    const bool more = win[2] > H1 && H1 < H;  // the sequential loop would have gone on past H1
    if (more) {
      if (lane == 0) need_list[atomicAdd(need_count, 1)] = slot;
      return;
    }
  }
  const int wh = win[0];
  if (wh < 0) {
    for (int e = lane; e < N; e += GEO_WAVE) inlier[o + e] = 0;
    if (lane == 0) {
      for (int k = 0; k < 9; k++) Fout[(size_t)pair * 9 + k] = 0.0;
      n_inliers[pair] = 0;
      ok[pair] = 0;
    }
    return;
  }
  if (lane == 0) {
    GeoModels m;
    geo_solve7(seed, pair, wh, N, pt1 + o, pt2 + o, A, 1, m);
    for (int k = 0; k < 9; k++) { Fw[k] = m.F[win[1]][k]; Fout[(size_t)pair * 9 + k] = Fw[k]; }
  }
  __syncthreads();
  double F[9];
  for (int k = 0; k < 9; k++) F[k] = Fw[k];
  int c = 0;
  for (int e = lane; e < N; e += GEO_WAVE) {
    const float2 a = pt1[o + e], b = pt2[o + e];
    const bool in = geo_inlier(F, a.x, a.y, b.x, b.y, th2);
    inlier[o + e] = in ? 1 : 0;
    c += in ? 1 : 0;
  }
  for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
  if (lane == 0) {
    n_inliers[pair] = c;
    ok[pair] = c >= min_inliers ? 1 : 0;  // match_inliers.size() < 30 -> false
  }
}

// l = F [x1, y1, 1]; l /= hypot(l0, l1); inlier iff |l . [x2, y2, 1]| < th   (geo_verification.cc:60-79),
// for every pair whose F was accepted.
__global__ __launch_bounds__(256) void k_epipolar_batch(int total, const int* __restrict__ pair_of, const float2* __restrict__ pt1,
                                                         const float2* __restrict__ pt2, const double* __restrict__ F,
                                                         const uint8_t* __restrict__ ok, double th, uint8_t* __restrict__ inlier) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int p = pair_of[e];
  if (ok && !ok[p]) { inlier[e] = 0; return; }
  const double* f = F + (size_t)p * 9;
  const double x1 = pt1[e].x, y1 = pt1[e].y, x2 = pt2[e].x, y2 = pt2[e].y;
  double l0 = f[0] * x1 + f[1] * y1 + f[2];
  double l1 = f[3] * x1 + f[4] * y1 + f[5];
  double l2 = f[6] * x1 + f[7] * y1 + f[8];
  const double n = sqrt(l0 * l0 + l1 * l1);
  l0 = l0 / n; l1 = l1 / n; l2 = l2 / n;
  const double dis = l0 * x2 + l1 * y2 + l2;
  inlier[e] = fabs(dis) < th ? 1 : 0;
}

__global__ void k_gather_int(int n, const int* __restrict__ idx, const int* __restrict__ src, int* __restrict__ dst) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}

// cvRANSACUpdateNumIters(p, ep, model_points, max_iters) for ep = (N - g) / N, clamped to max_iters
static int geo_update_num_iters(double p, double ep, int model_points, int max_iters) {
  p = std::max(p, 0.0); p = std::min(p, 1.0);
  ep = std::max(ep, 0.0); ep = std::min(ep, 1.0);
  double num = std::max(1.0 - p, DBL_MIN);
  double denom = 1.0 - std::pow(1.0 - ep, model_points);
  if (denom < DBL_MIN) return 0;
  num = std::log(num);
  denom = std::log(denom);
  if (denom >= 0 || -num >= max_iters * (-denom)) return max_iters;
  return (int)std::lrint(num / denom);
}

MSFM_API void msfm_fransac_default_options(msfm_fransac_options* o) {
  if (!o) return;
  o->threshold = 3.0;
  o->confidence = 0.99;
  o->max_iterations = 2000;
  o->min_points = 30;
  o->min_inliers = 30;
  o->seed = 0x4D53464D46ull;
}

// The verification of a batch of pairs on points that are already resident: d1 / d2 hold the pairs' matches back to back,
// h_offsets (host) and d_off (device) delimit them; results stay on the device.  msfm_fundamental_ransac_batch is this
// between an upload and a download, msfm_chain_verify (chain.hip) calls it on points gathered from the match codes.
int geo_fransac_dev(msfm_ctx* ctx, int n_pairs, const int* offsets, const int* d_off, const float* d1, const float* d2,
                    const msfm_fransac_options* opt, double* dF, uint8_t* d_in, int* d_nin, uint8_t* d_ok) {
  hipStream_t s = ctx->stream;
  const int H = opt->max_iterations;
  // R[g] = cvRANSACUpdateNumIters for g inliers of N: a pow and two logs per entry.  The table of a pair depends on its N
  // alone, so one table per DISTINCT N is formed (and kept from call to call while confidence and sample limit stay the same:
  // the matching loop verifies thousands of pairs with one set of options) - at 9 120 pairs of ~125 good matches each the
  // per-pair tables were 1.1 M evaluations, 7 of the 15 ms of msfm_chain_verify on sixteen host threads.
  std::vector<int> tab, tab_off(std::max(1, n_pairs), 0);
  {
    struct TabCache { std::mutex mu; double conf = -1.0; int H = -1; std::unordered_map<int, std::vector<int>> by_n; size_t entries = 0; };
    static TabCache cache;
    std::lock_guard<std::mutex> lock(cache.mu);
    if (cache.conf != opt->confidence || cache.H != H || cache.entries > (size_t)32 << 20) {
      cache.by_n.clear(); cache.entries = 0; cache.conf = opt->confidence; cache.H = H;
    }
    std::vector<int> missing;
    for (int p = 0; p < n_pairs; p++) {
      const int N = offsets[p + 1] - offsets[p];
      if (N >= opt->min_points && N >= 8 && cache.by_n.find(N) == cache.by_n.end()) { cache.by_n[N]; missing.push_back(N); }
    }
    std::vector<std::vector<int>*> slot(missing.size());
    for (size_t k = 0; k < missing.size(); k++) { slot[k] = &cache.by_n[missing[k]]; slot[k]->resize((size_t)missing[k] + 1); cache.entries += (size_t)missing[k] + 1; }
    par_ranges(missing.size(), host_threads(), [&](int, size_t k0, size_t k1) {
      for (size_t k = k0; k < k1; k++) {
        const int N = missing[k];
        int* R = slot[k]->data();
        for (int g = 0; g <= N; g++) R[g] = geo_update_num_iters(opt->confidence, (double)(N - g) / N, 7, H);
      }
    }, 4);
    std::unordered_map<int, int> at;   // N -> offset of its table in this call's upload
    for (int p = 0; p < n_pairs; p++) {
      const int N = offsets[p + 1] - offsets[p];
      if (!(N >= opt->min_points && N >= 8)) continue;   // (the kernels do not look at the table of such a pair)
      auto it = at.find(N);
      if (it == at.end()) {
        it = at.emplace(N, (int)tab.size()).first;
        const std::vector<int>& R = cache.by_n[N];
        tab.insert(tab.end(), R.begin(), R.end());
      }
      tab_off[p] = it->second;
    }
    if (tab.empty()) tab.push_back(H);
  }
  DevBuf<int> d_tab_off;
  HIP_TRY(ctx, d_tab_off.from(tab_off, s));
  DevBuf<int> d_tab;
  HIP_TRY(ctx, d_tab.from(tab, s));
  const double th2 = opt->threshold * opt->threshold;
  // pairs without enough matches (GeoVerificationFundamental: pt1.size() < 30 -> false) get their verdict here; the others
  // form the slot list the kernels run over
  std::vector<int> slot_pair;
  for (int p = 0; p < n_pairs; p++) {
    const int N = offsets[p + 1] - offsets[p];
    if (N >= opt->min_points && N >= 8) slot_pair.push_back(p);
  }
  const int n_slots = (int)slot_pair.size();
  HIP_TRY(ctx, hipMemsetAsync(dF, 0, sizeof(double) * 9 * (size_t)n_pairs, s));
  HIP_TRY(ctx, hipMemsetAsync(d_in, 0, (size_t)std::max(1, offsets[n_pairs]), s));
  HIP_TRY(ctx, hipMemsetAsync(d_nin, 0, sizeof(int) * (size_t)n_pairs, s));
  HIP_TRY(ctx, hipMemsetAsync(d_ok, 0, (size_t)n_pairs, s));
  if (n_slots == 0) { HIP_TRY(ctx, hipStreamSynchronize(s)); return MSFM_OK; }
  // Most pairs stop after a few dozen samples (cvRANSACUpdateNumIters): score the first H1 samples of every pair,
  // replay them, and run the remaining H - H1 samples only for the pairs whose budget is still open.
  const int H1 = std::min(H, 128);
  DevBuf<int> d_slot_pair, d_need, d_counts1, d_counts2;
  DevBuf<GeoModelRec> d_models1, d_models2;
  HIP_TRY(ctx, d_slot_pair.from(slot_pair, s));
  HIP_TRY(ctx, d_need.alloc((size_t)n_slots + 1));   // [0]: how many, [1..]: the slots
  HIP_TRY(ctx, hipMemsetAsync(d_need.p, 0, sizeof(int), s));
  HIP_TRY(ctx, d_counts1.alloc((size_t)n_slots * H1 * 3));
  HIP_TRY(ctx, d_models1.alloc((size_t)n_slots * H1));
  const float2* p1 = reinterpret_cast<const float2*>(d1);
  const float2* p2 = reinterpret_cast<const float2*>(d2);
  // the models of a sample range live in memory only between the two kernels: a long range is taken in pieces of `piece`
  // samples that reuse one model buffer (224 bytes per sample and pair: 1 872 samples at once would be 420 KB per pair)
  auto score = [&](int h0, int h1, int piece, int ns, const int* slots, GeoModelRec* models, int* counts, const char* name) {
    KTimer t(ctx, name);
    for (int a = h0; a < h1; a += piece) {
      const int b = std::min(h1, a + piece);
      for (int p0 = 0; p0 < ns; p0 += 32768) {  // grid.y limit
        const int np = std::min(32768, ns - p0);
        hipLaunchKernelGGL(k_fransac_models, dim3(cdiv(b - a, GEO_WAVE), np), dim3(GEO_WAVE), 0, s, a, b, slots + p0, (const int*)nullptr, d_off,
                           p1, p2, opt->seed, models + (size_t)p0 * piece);
        hipLaunchKernelGGL(k_fransac_count, dim3(cdiv(b - a, GEO_CNT_SAMPLES), np), dim3(256), 0, s, a, b, slots + p0, (const int*)nullptr, d_off,
                           p1, p2, th2, models + (size_t)p0 * piece, counts + (size_t)p0 * (h1 - h0) * 3, h1 - h0, a - h0);
      }
    }
  };
  auto select = [&](int pass, int grid) {
    KTimer t(ctx, "geo_fransac_select");
    hipLaunchKernelGGL(k_fransac_select, dim3(grid), dim3(GEO_WAVE), 0, s, H, H1, pass, d_slot_pair.p, d_need.p + 1, d_need.p, d_off, p1, p2, opt->seed, th2,
                       opt->min_inliers, d_counts1.p, d_counts2.p, d_tab.p, d_tab_off.p, dF, d_in, d_nin, d_ok);
  };
  score(0, H1, H1, n_slots, d_slot_pair.p, d_models1.p, d_counts1.p, "geo_fransac_score");
  select(1, n_slots);
  if (H1 < H) {
    int n_need = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&n_need, d_need.p, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    if (n_need > 0) {
      const int piece = std::min(H - H1, 256);
      HIP_TRY(ctx, d_counts2.alloc((size_t)n_need * (H - H1) * 3));
      HIP_TRY(ctx, d_models2.alloc((size_t)n_need * piece));
      // the list holds SLOTS; the kernels of the second range want pairs
      DevBuf<int> d_need_pair;
      HIP_TRY(ctx, d_need_pair.alloc(n_need));
      hipLaunchKernelGGL(k_gather_int, dim3(cdiv(n_need, 256)), dim3(256), 0, s, n_need, d_need.p + 1, d_slot_pair.p, d_need_pair.p);
      score(H1, H, piece, n_need, d_need_pair.p, d_models2.p, d_counts2.p, "geo_fransac_score_rest");
      select(2, n_need);
      HIP_TRY(ctx, hipGetLastError());
      HIP_TRY(ctx, hipStreamSynchronize(s));
      return MSFM_OK;
    }
  }
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipStreamSynchronize(s));   // the tables and counters above are released on return
  return MSFM_OK;
}

MSFM_API int msfm_fundamental_ransac_batch(msfm_ctx* ctx, int n_pairs, const int* offsets, const float* pt1, const float* pt2,
                                           const msfm_fransac_options* opt, double* F, uint8_t* inlier, int* n_inliers, uint8_t* ok) {
  if (!ctx || n_pairs < 0 || !offsets || !opt || !F || !n_inliers || !ok) return MSFM_E_INVAL;
  if (opt->max_iterations < 1 || opt->max_iterations > 65536 || !(opt->threshold > 0.0)) return msfm_set_error(ctx, MSFM_E_INVAL, "fransac: bad options");
  if (n_pairs == 0) return MSFM_OK;
  if (offsets[0] != 0) return msfm_set_error(ctx, MSFM_E_INVAL, "fransac: offsets[0] must be 0");
  for (int p = 0; p < n_pairs; p++)
    if (offsets[p + 1] < offsets[p]) return msfm_set_error(ctx, MSFM_E_INVAL, "fransac: offsets must be non-decreasing");
  const int total = offsets[n_pairs];
  if (total > 0 && (!pt1 || !pt2 || !inlier)) return MSFM_E_INVAL;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  DevBuf<int> d_off, d_nin;
  DevBuf<float> d1, d2;
  DevBuf<double> dF;
  DevBuf<uint8_t> d_in, d_ok;
  HIP_TRY(ctx, d_off.alloc((size_t)n_pairs + 1));
  HIP_TRY(ctx, d_off.upload(offsets, (size_t)n_pairs + 1, s));
  HIP_TRY(ctx, d1.alloc(2 * (size_t)std::max(1, total)));
  HIP_TRY(ctx, d2.alloc(2 * (size_t)std::max(1, total)));
  HIP_TRY(ctx, d1.upload(pt1, 2 * (size_t)total, s));
  HIP_TRY(ctx, d2.upload(pt2, 2 * (size_t)total, s));
  HIP_TRY(ctx, dF.alloc((size_t)n_pairs * 9));
  HIP_TRY(ctx, d_in.alloc((size_t)std::max(1, total)));
  HIP_TRY(ctx, d_nin.alloc(n_pairs));
  HIP_TRY(ctx, d_ok.alloc(n_pairs));
  MSFM_TRY(geo_fransac_dev(ctx, n_pairs, offsets, d_off.p, d1.p, d2.p, opt, dF.p, d_in.p, d_nin.p, d_ok.p));
  HIP_TRY(ctx, hipMemcpyAsync(F, dF.p, sizeof(double) * 9 * (size_t)n_pairs, hipMemcpyDeviceToHost, s));
  if (total) HIP_TRY(ctx, hipMemcpyAsync(inlier, d_in.p, (size_t)total, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(n_inliers, d_nin.p, sizeof(int) * (size_t)n_pairs, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(ok, d_ok.p, (size_t)n_pairs, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return MSFM_OK;
}

// The closed-form filter of a batch on resident points (pair_of[e] = pair of match e): msfm_epipolar_filter_batch without its
// transfers; msfm_chain_verify calls it on the "all" sets gathered from the match codes.
int geo_epipolar_batch_dev(msfm_ctx* ctx, int total, const int* d_pair_of, const float* d1, const float* d2, const double* dF,
                           const uint8_t* d_ok, double th, uint8_t* d_in) {
  if (total == 0) return MSFM_OK;
  KTimer t(ctx, "geo_epipolar_filter");
  hipLaunchKernelGGL(k_epipolar_batch, dim3(cdiv(total, 256)), dim3(256), 0, ctx->stream, total, d_pair_of, reinterpret_cast<const float2*>(d1),
                     reinterpret_cast<const float2*>(d2), dF, d_ok, th, d_in);
  HIP_TRY(ctx, hipGetLastError());
  return MSFM_OK;
}

MSFM_API int msfm_epipolar_filter_batch(msfm_ctx* ctx, int n_pairs, const int* offsets, const float* pt1, const float* pt2,
                                        const double* F, const uint8_t* ok, double th, uint8_t* inlier) {
  if (!ctx || n_pairs < 0 || !offsets || !F) return MSFM_E_INVAL;
  if (n_pairs == 0) return MSFM_OK;
  const int total = offsets[n_pairs];
  if (total == 0) return MSFM_OK;
  if (!pt1 || !pt2 || !inlier) return MSFM_E_INVAL;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  std::vector<int> pair_of(total);
  for (int p = 0; p < n_pairs; p++) {
    if (offsets[p + 1] < offsets[p]) return msfm_set_error(ctx, MSFM_E_INVAL, "epipolar: offsets must be non-decreasing");
    for (int e = offsets[p]; e < offsets[p + 1]; e++) pair_of[e] = p;
  }
  DevBuf<int> d_po;
  DevBuf<float> d1, d2;
  DevBuf<double> dF;
  DevBuf<uint8_t> d_in, d_ok;
  HIP_TRY(ctx, d_po.from(pair_of, s));
  HIP_TRY(ctx, d1.alloc(2 * (size_t)total)); HIP_TRY(ctx, d2.alloc(2 * (size_t)total));
  HIP_TRY(ctx, d1.upload(pt1, 2 * (size_t)total, s)); HIP_TRY(ctx, d2.upload(pt2, 2 * (size_t)total, s));
  HIP_TRY(ctx, dF.alloc(9 * (size_t)n_pairs)); HIP_TRY(ctx, dF.upload(F, 9 * (size_t)n_pairs, s));
  HIP_TRY(ctx, d_in.alloc(total));
  if (ok) { HIP_TRY(ctx, d_ok.alloc(n_pairs)); HIP_TRY(ctx, d_ok.upload(ok, n_pairs, s)); }
  {
    KTimer t(ctx, "geo_epipolar_filter");
    hipLaunchKernelGGL(k_epipolar_batch, dim3(cdiv(total, 256)), dim3(256), 0, s, total, d_po.p, reinterpret_cast<const float2*>(d1.p),
                       reinterpret_cast<const float2*>(d2.p), dF.p, ok ? d_ok.p : nullptr, th, d_in.p);
  }
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(inlier, d_in.p, (size_t)total, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return MSFM_OK;
}
