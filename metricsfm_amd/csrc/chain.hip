// The chain between matching and bundle adjustment with every intermediate result resident in HBM.
//
// The reference walks it through the host for every image pair and every point:
//   match lists -> GeoVerificationFundamental on the "good" set, closed-form filter on the "all" set
//                  (SfM/src/graph/fine_matching_graph.cc:138-153, :182-186)
//   -> data association of SLAMGPS::Triangulation (SfM/src/slam_gps.cc:565-635)
//   -> Point3D::Trianglate2 per point, points with fewer than three views or a failed triangulation marked bad (:638-648)
//   -> the residual blocks of BundleAdjuster::RunOptimizetion (SfM/src/optimizer.cc:59-129).
// Round 2 had a batched entry point for each step, each taking and returning host arrays: the 2-NN kernel of a pair takes
// 40 us inside a 0.63 ms call, the triangulation of config 3's tracks 0.06 ms inside 1.3 ms.  msfm_chain strings the same
// kernels together on device buffers: the match codes stay where msfm_match_pairs left them, the keypoint positions were
// uploaded once beside the descriptors (msfm_descset_upload_keypoints), and what crosses PCIe between the descriptor
// upload and the download of the adjusted parameters is a few integers per image pair (match counts, needed on the host
// to size the next step) plus the cameras.  Every step gives bit for bit what its host-array counterpart gives on the same
// input (tests/test_gpu_chain.py).
#include <algorithm>
#include <cstring>
#include <memory>

#include <rocprim/rocprim.hpp>

#include "common.h"
#include <chrono>
#include "../../include/msfm.h"

namespace chn {

// exclusive position of a set flag among the flags of the workgroup (256 threads), and the workgroup's total
__device__ __forceinline__ int block_rank(bool f, int* wave_tot /*[4] LDS*/, int& total) {
  const unsigned long long b = __ballot(f);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int within = __popcll(b & ((1ull << lane) - 1ull));
  __syncthreads();   // (wave_tot is reused by consecutive calls)
  if (lane == 0) wave_tot[wave] = __popcll(b);
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; w++) base += wave_tot[w];
  total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
  return base + within;
}

struct PairSrc {
  const float* kp1;   // train image positions
  const float* kp2;   // query image positions
  int nq, off;        // codes of the pair at code[off .. off + nq)
  int off_good, off_all;
};

// One workgroup per pair: the loop of fine_matching_graph.cc:116-133 on the codes - matches_good / matches_all and their
// point lists in query order - written at the pair's offsets.
__global__ __launch_bounds__(256) void k_gather_sets(const PairSrc* __restrict__ src, const int32_t* __restrict__ code,
                                                      int* __restrict__ m_good, float* __restrict__ g1, float* __restrict__ g2,
                                                      int* __restrict__ m_all, float* __restrict__ a1, float* __restrict__ a2,
                                                      int* __restrict__ pair_of_all) {
  __shared__ int wt[4];
  const PairSrc S = src[blockIdx.x];
  int ng = 0, na = 0;
  for (int m0 = 0; m0 < S.nq; m0 += 256) {
    const int m = m0 + threadIdx.x;
    const int32_t c = m < S.nq ? code[S.off + m] : -1;
    const bool any = c >= 0;
    const bool good = any && (c & MSFM_MATCH_GOOD) != 0, all = any && (c & MSFM_MATCH_NOT_ALL) == 0;
    const int id = c & MSFM_MATCH_ID_MASK;
    float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f;
    if (any) { x1 = S.kp1[2 * id]; y1 = S.kp1[2 * id + 1]; x2 = S.kp2[2 * m]; y2 = S.kp2[2 * m + 1]; }
    int tot;
    const int rg = block_rank(good, wt, tot);
    if (good) {
      const size_t e = (size_t)S.off_good + ng + rg;
      m_good[2 * e] = id; m_good[2 * e + 1] = m;
      g1[2 * e] = x1; g1[2 * e + 1] = y1; g2[2 * e] = x2; g2[2 * e + 1] = y2;
    }
    ng += tot;
    const int ra = block_rank(all, wt, tot);
    if (all) {
      const size_t e = (size_t)S.off_all + na + ra;
      m_all[2 * e] = id; m_all[2 * e + 1] = m;
      a1[2 * e] = x1; a1[2 * e + 1] = y1; a2[2 * e] = x2; a2[2 * e + 1] = y2;
      pair_of_all[e] = blockIdx.x;
    }
    na += tot;
  }
}

// inliers of the "all" set per pair (zero for a pair whose RANSAC failed: its mask is all zero)
__global__ __launch_bounds__(256) void k_count_inliers(const int* __restrict__ off_all, const uint8_t* __restrict__ in_all, int* __restrict__ n_fin) {
  __shared__ int wt[4];
  const int b = off_all[blockIdx.x], e = off_all[blockIdx.x + 1];
  int n = 0;
  for (int i0 = b; i0 < e; i0 += 256) {
    const int i = i0 + threadIdx.x;
    int tot;
    (void)block_rank(i < e && in_all[i] != 0, wt, tot);
    n += tot;
  }
  if (threadIdx.x == 0) n_fin[blockIdx.x] = n;
}

// matches_inliers of every pair (fine_matching_graph.cc:150-153), in the order of matches_all
__global__ __launch_bounds__(256) void k_compact_matches(const int* __restrict__ off_all, const int* __restrict__ off_fin, const uint8_t* __restrict__ in_all,
                                                          const int* __restrict__ m_all, int* __restrict__ m_fin) {
  __shared__ int wt[4];
  const int b = off_all[blockIdx.x], e = off_all[blockIdx.x + 1];
  int n = 0;
  for (int i0 = b; i0 < e; i0 += 256) {
    const int i = i0 + threadIdx.x;
    const bool f = i < e && in_all[i] != 0;
    int tot;
    const int r = block_rank(f, wt, tot);
    if (f) {
      const size_t o = (size_t)off_fin[blockIdx.x] + n + r;
      m_fin[2 * o] = m_all[2 * (size_t)i]; m_fin[2 * o + 1] = m_all[2 * (size_t)i + 1];
    }
    n += tot;
  }
}

// positions of a track's observations as doubles (structure.h:65-66 keeps them as Eigen::Vector2d), from the resident keypoints
__global__ __launch_bounds__(256) void k_track_xy(int n_obs, const int* __restrict__ img, const int* __restrict__ feat, const float* const* __restrict__ kp,
                                                   double* __restrict__ xy) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_obs) return;
  const float* k = kp[img[e]];
  xy[2 * (size_t)e] = (double)k[2 * (size_t)feat[e]];
  xy[2 * (size_t)e + 1] = (double)k[2 * (size_t)feat[e] + 1];
}

// a track becomes a point of the bundle adjustment when its triangulation was accepted and it has min_views observations
__global__ __launch_bounds__(256) void k_keep(int n_tracks, const int* __restrict__ off, const uint8_t* __restrict__ ok, int min_views, int* __restrict__ keep,
                                               int* __restrict__ keep_obs) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t > n_tracks) return;
  if (t == n_tracks) { keep[t] = 0; keep_obs[t] = 0; return; }
  const int k = off[t + 1] - off[t];
  const int f = ok[t] && k >= min_views;
  keep[t] = f;
  keep_obs[t] = f ? k : 0;
}

// the flat problem arrays of optimizer.cc:59-129 for the kept tracks: points ascending, a point's observations in ascending
// image order (std::map), weight 1.0 for two views and `weight_ge3` for more (:69-78)
__global__ __launch_bounds__(256) void k_ba_arrays(int n_tracks, const int* __restrict__ off, const int* __restrict__ img, const double* __restrict__ xy,
                                                    const double* __restrict__ X, const int* __restrict__ keep, const int* __restrict__ new_pt,
                                                    const int* __restrict__ new_obs, double weight_ge3, int* __restrict__ obs_cam,
                                                    int* __restrict__ obs_pt, double* __restrict__ obs_xy, double* __restrict__ point,
                                                    double* __restrict__ pt_weight, int* __restrict__ track_of_point) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n_tracks || !keep[t]) return;
  const int p = new_pt[t], o0 = new_obs[t], b = off[t], k = off[t + 1] - b;
  point[3 * (size_t)p] = X[3 * (size_t)t]; point[3 * (size_t)p + 1] = X[3 * (size_t)t + 1]; point[3 * (size_t)p + 2] = X[3 * (size_t)t + 2];
  pt_weight[p] = k == 2 ? 1.0 : weight_ge3;
  track_of_point[p] = t;
  for (int j = 0; j < k; j++) {
    obs_cam[o0 + j] = img[b + j];
    obs_pt[o0 + j] = p;
    obs_xy[2 * (size_t)(o0 + j)] = xy[2 * (size_t)(b + j)];
    obs_xy[2 * (size_t)(o0 + j) + 1] = xy[2 * (size_t)(b + j) + 1];
  }
}

}  // namespace chn

struct msfm_chain {
  msfm_ctx* ctx = nullptr;
  int n_images = 0, n_pairs = 0;
  std::vector<int> pairs, count, feat_off;
  // The chain keeps its OWN copy of the keypoints (one flat buffer, kp[i] points into it): the descriptor set may re-upload
  // or drop them, or be destroyed, while the chain lives (verify and triangulate read them long after create).
  DevBuf<float> kp_own;
  std::vector<const float*> kp;
  DevBuf<const float*> d_kp;
  // verification
  bool verified = false;
  std::vector<int> n_good, n_all;        // sizes of matches_good / matches_all per pair (read back once at create time)
  std::vector<int> n_fin, off_fin;       // matches_inliers per pair and their prefix sums
  std::vector<uint8_t> ok;
  std::vector<double> F;
  DevBuf<int> d_pair, d_moff, d_match, d_nf, d_fo;
  // tracks
  bool have_tracks = false;
  msfm_track_dev tracks;
  // triangulation
  bool triangulated = false;
  DevBuf<double> xy, X, mse;
  DevBuf<uint8_t> tok;
  // bundle adjustment
  int n_ba_points = 0, n_ba_obs = 0;
  DevBuf<int> track_of_point;
};

#define CH_TRY(e) HIP_TRY(ctx, (e))

MSFM_API int msfm_chain_create(msfm_match_result* res, msfm_chain** out) {
  if (!res || !out) return MSFM_E_INVAL;
  *out = nullptr;
  MatchView v;
  MSFM_TRY(match_result_view(res, &v));
  msfm_ctx* ctx = v.ctx;
  if (v.slam) return msfm_set_error(ctx, MSFM_E_INVAL, "msfm_chain_create: the codes of msfm_match_pairs_slam carry no good / all sets");
  for (int p = 0; p < v.n_pairs; p++) {
    const int a = v.pairs[2 * p], b = v.pairs[2 * p + 1];
    if (!v.kp[a] || (v.count[b] > 0 && !v.kp[b]))
      return msfm_set_error(ctx, MSFM_E_INVAL, "msfm_chain_create: pair %d needs the keypoints of both images (msfm_descset_upload_keypoints)", p);
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  std::unique_ptr<msfm_chain> C(new msfm_chain());
  C->ctx = ctx; C->n_images = v.n_images; C->n_pairs = v.n_pairs;
  C->pairs.assign(v.pairs, v.pairs + 2 * (size_t)v.n_pairs);
  C->count = v.count;
  C->feat_off.assign(v.n_images + 1, 0);
  for (int i = 0; i < v.n_images; i++) C->feat_off[i + 1] = C->feat_off[i] + v.count[i];
  hipStream_t s = ctx->stream;
  C->kp.assign(v.n_images, nullptr);
  CH_TRY(C->kp_own.alloc(2 * (size_t)std::max(1, C->feat_off[v.n_images])));
  for (int i = 0; i < v.n_images; i++) {
    if (!v.kp[i] || v.count[i] == 0) continue;
    float* dst = C->kp_own.p + 2 * (size_t)C->feat_off[i];
    CH_TRY(hipMemcpyAsync(dst, v.kp[i], sizeof(float) * 2 * (size_t)v.count[i], hipMemcpyDeviceToDevice, s));
    C->kp[i] = dst;
  }
  // ---- matches_good / matches_all of every pair, gathered from the codes ----
  std::vector<int> ng(std::max(1, v.n_pairs)), na(std::max(1, v.n_pairs));
  if (v.n_pairs) {
    CH_TRY(hipMemcpyAsync(ng.data(), v.n_good, sizeof(int) * v.n_pairs, hipMemcpyDeviceToHost, s));
    CH_TRY(hipMemcpyAsync(na.data(), v.n_all, sizeof(int) * v.n_pairs, hipMemcpyDeviceToHost, s));
    CH_TRY(hipStreamSynchronize(s));
  }
  C->n_good = ng; C->n_all = na;   // sizes of matches_good / matches_all per pair: msfm_chain_verify gathers them
  ctx->children++;
  *out = C.release();
  return MSFM_OK;
}

MSFM_API int msfm_chain_verify(msfm_chain* C, msfm_match_result* res, const msfm_fransac_options* opt, double th_filter) {
  if (!C || !res || !opt) return MSFM_E_INVAL;
  msfm_ctx* ctx = C->ctx;
  if (C->verified) return msfm_set_error(ctx, MSFM_E_INVAL, "msfm_chain_verify: already done");
  if (opt->max_iterations < 1 || opt->max_iterations > 65536 || !(opt->threshold > 0.0)) return msfm_set_error(ctx, MSFM_E_INVAL, "fransac: bad options");
  MatchView v;
  MSFM_TRY(match_result_view(res, &v));
  if (v.ctx != ctx || v.n_pairs != C->n_pairs) return msfm_set_error(ctx, MSFM_E_INVAL, "msfm_chain_verify: not the match result the chain was created from");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int np = C->n_pairs;
  static const bool laps = getenv("MSFM_CHAIN_LAPS") != nullptr;   // host-side wall clock of the stages, to stderr
  const auto lap_t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (laps) fprintf(stderr, "msfm: chain verify %-24s %7.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - lap_t0).count());
  };
  C->ok.assign(std::max(1, np), 0);
  C->F.assign(9 * (size_t)std::max(1, np), 0.0);
  std::vector<int> off_g(np + 1, 0), off_a(np + 1, 0);
  for (int p = 0; p < np; p++) { off_g[p + 1] = off_g[p] + C->n_good[p]; off_a[p + 1] = off_a[p] + C->n_all[p]; }
  const int tg = off_g[np], ta = off_a[np];
  std::vector<chn::PairSrc> src(std::max(1, np));
  for (int p = 0; p < np; p++) {
    const int a = C->pairs[2 * p], b = C->pairs[2 * p + 1];
    src[p] = chn::PairSrc{C->kp[a], v.nq[p] ? C->kp[b] : nullptr, v.nq[p], v.out_off[p], off_g[p], off_a[p]};
  }
  DevBuf<chn::PairSrc> d_src;
  DevBuf<int> d_offg, d_offa, m_good, m_all, pair_of_all, d_nin, d_nfin, d_offfin;
  DevBuf<float> g1, g2, a1, a2;
  DevBuf<double> dF;
  DevBuf<uint8_t> in_g, in_a, d_ok;
  CH_TRY(d_src.from(src, s)); CH_TRY(d_offg.from(off_g, s)); CH_TRY(d_offa.from(off_a, s));
  CH_TRY(m_good.alloc(2 * (size_t)std::max(1, tg))); CH_TRY(g1.alloc(2 * (size_t)std::max(1, tg))); CH_TRY(g2.alloc(2 * (size_t)std::max(1, tg)));
  CH_TRY(m_all.alloc(2 * (size_t)std::max(1, ta))); CH_TRY(a1.alloc(2 * (size_t)std::max(1, ta))); CH_TRY(a2.alloc(2 * (size_t)std::max(1, ta)));
  CH_TRY(pair_of_all.alloc(std::max(1, ta)));
  CH_TRY(dF.alloc(9 * (size_t)std::max(1, np))); CH_TRY(in_g.alloc(std::max(1, tg))); CH_TRY(in_a.alloc(std::max(1, ta)));
  CH_TRY(d_nin.alloc(std::max(1, np))); CH_TRY(d_ok.alloc(std::max(1, np))); CH_TRY(d_nfin.alloc(std::max(1, np)));
  C->n_fin.assign(std::max(1, np), 0);
  C->off_fin.assign(np + 1, 0);
  if (np) {
    {
      KTimer t(ctx, "chain_gather_sets");
      hipLaunchKernelGGL(chn::k_gather_sets, dim3(np), dim3(256), 0, s, d_src.p, v.code, m_good.p, g1.p, g2.p, m_all.p, a1.p, a2.p, pair_of_all.p);
    }
    CH_TRY(hipGetLastError());
    lap("buffers + gather enqueued");
    // GeoVerificationFundamental on the good sets (fine_matching_graph.cc:141), then the closed-form filter on the all sets of
    // the pairs that passed (:145-147)
    MSFM_TRY(geo_fransac_dev(ctx, np, off_g.data(), d_offg.p, g1.p, g2.p, opt, dF.p, in_g.p, d_nin.p, d_ok.p));
    lap("fundamental matrices");
    CH_TRY(hipMemsetAsync(in_a.p, 0, std::max(1, ta), s));
    MSFM_TRY(geo_epipolar_batch_dev(ctx, ta, pair_of_all.p, a1.p, a2.p, dF.p, d_ok.p, th_filter, in_a.p));
    {
      KTimer t(ctx, "chain_compact_matches");
      hipLaunchKernelGGL(chn::k_count_inliers, dim3(np), dim3(256), 0, s, d_offa.p, in_a.p, d_nfin.p);
    }
    CH_TRY(hipMemcpyAsync(C->n_fin.data(), d_nfin.p, sizeof(int) * np, hipMemcpyDeviceToHost, s));
    CH_TRY(hipMemcpyAsync(C->ok.data(), d_ok.p, np, hipMemcpyDeviceToHost, s));
    CH_TRY(hipMemcpyAsync(C->F.data(), dF.p, sizeof(double) * 9 * (size_t)np, hipMemcpyDeviceToHost, s));
    CH_TRY(hipStreamSynchronize(s));
    for (int p = 0; p < np; p++) C->off_fin[p + 1] = C->off_fin[p] + C->n_fin[p];
    lap("filter + counts back");
  }
  const int M = C->off_fin[np];
  CH_TRY(C->d_moff.from(C->off_fin, s));
  CH_TRY(C->d_match.alloc(2 * (size_t)std::max(1, M)));
  CH_TRY(C->d_pair.from(C->pairs.empty() ? std::vector<int>(2, 0) : C->pairs, s));
  CH_TRY(C->d_nf.from(C->count, s));
  CH_TRY(C->d_fo.from(C->feat_off, s));
  if (np) {
    KTimer t(ctx, "chain_compact_matches");
    hipLaunchKernelGGL(chn::k_compact_matches, dim3(np), dim3(256), 0, s, d_offa.p, C->d_moff.p, in_a.p, m_all.p, C->d_match.p);
  }
  CH_TRY(hipGetLastError());
  CH_TRY(hipStreamSynchronize(s));   // the scratch above goes back to the pool
  lap("compacted");
  C->verified = true;
  return MSFM_OK;
}

MSFM_API int msfm_chain_matches(msfm_chain* C, int* n_matches, uint8_t* ok, double* F) {
  if (!C) return MSFM_E_INVAL;
  if (!C->verified) return msfm_set_error(C->ctx, MSFM_E_INVAL, "msfm_chain_matches: msfm_chain_verify first");
  for (int p = 0; p < C->n_pairs; p++) {
    if (n_matches) n_matches[p] = C->n_fin[p];
    if (ok) ok[p] = C->ok[p];
    if (F) for (int k = 0; k < 9; k++) F[9 * (size_t)p + k] = C->F[9 * (size_t)p + k];
  }
  return MSFM_OK;
}

MSFM_API int msfm_chain_fetch_matches(msfm_chain* C, int pair, int* matches) {
  if (!C || pair < 0 || pair >= C->n_pairs) return MSFM_E_INVAL;
  msfm_ctx* ctx = C->ctx;
  if (!C->verified) return msfm_set_error(ctx, MSFM_E_INVAL, "msfm_chain_fetch_matches: msfm_chain_verify first");
  const int n = C->n_fin[pair];
  if (n == 0) return MSFM_OK;
  if (!matches) return MSFM_E_INVAL;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  CH_TRY(hipMemcpyAsync(matches, C->d_match.p + 2 * (size_t)C->off_fin[pair], sizeof(int) * 2 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  CH_TRY(hipStreamSynchronize(ctx->stream));
  return MSFM_OK;
}

MSFM_API int msfm_chain_build_tracks(msfm_chain* C, int* n_tracks, int* n_observations) {
  if (!C) return MSFM_E_INVAL;
  msfm_ctx* ctx = C->ctx;
  if (!C->verified) return msfm_set_error(ctx, MSFM_E_INVAL, "msfm_chain_build_tracks: msfm_chain_verify first");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!C->have_tracks) {
    if ((long)C->feat_off[C->n_images] > 0x7fffffffL) return msfm_set_error(ctx, MSFM_E_INVAL, "more than 2^31 features");
    MSFM_TRY(tracks_build_dev(ctx, C->n_images, C->feat_off, C->d_nf.p, C->d_fo.p, C->n_pairs, C->d_pair.p, C->d_moff.p, C->d_match.p,
                              C->off_fin[C->n_pairs], &C->tracks));
    C->have_tracks = true;
  }
  if (n_tracks) *n_tracks = C->tracks.n_tracks;
  if (n_observations) *n_observations = C->tracks.n_obs;
  return MSFM_OK;
}

MSFM_API int msfm_chain_fetch_tracks(msfm_chain* C, int* track_off, int* obs_image, int* obs_feature) {
  if (!C) return MSFM_E_INVAL;
  msfm_ctx* ctx = C->ctx;
  if (!C->have_tracks) return msfm_set_error(ctx, MSFM_E_INVAL, "msfm_chain_fetch_tracks: msfm_chain_build_tracks first");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const msfm_track_dev& T = C->tracks;
  if (track_off) CH_TRY(hipMemcpyAsync(track_off, T.off.p, sizeof(int) * ((size_t)T.n_tracks + 1), hipMemcpyDeviceToHost, s));
  if (obs_image && T.n_obs) CH_TRY(hipMemcpyAsync(obs_image, T.img.p, sizeof(int) * (size_t)T.n_obs, hipMemcpyDeviceToHost, s));
  if (obs_feature && T.n_obs) CH_TRY(hipMemcpyAsync(obs_feature, T.feat.p, sizeof(int) * (size_t)T.n_obs, hipMemcpyDeviceToHost, s));
  CH_TRY(hipStreamSynchronize(s));
  return MSFM_OK;
}

MSFM_API int msfm_chain_triangulate(msfm_chain* C, int n_cams, const double* cam_R, const double* cam_t, const double* cam_c, const double* cam_fk,
                                    double th_error, double th_angle, int* n_accepted) {
  if (!C || !cam_R || !cam_t || !cam_c || !cam_fk) return MSFM_E_INVAL;
  msfm_ctx* ctx = C->ctx;
  if (!C->have_tracks) return msfm_set_error(ctx, MSFM_E_INVAL, "msfm_chain_triangulate: msfm_chain_build_tracks first");
  if (n_cams < C->n_images) return msfm_set_error(ctx, MSFM_E_INVAL, "msfm_chain_triangulate: %d cameras for %d images (image index = camera index)", n_cams, C->n_images);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const msfm_track_dev& T = C->tracks;
  const int nt = T.n_tracks, no = T.n_obs;
  DevBuf<double> dR, dt, dc, dfk;
  CH_TRY(dR.alloc(9 * (size_t)n_cams)); CH_TRY(dR.upload(cam_R, 9 * (size_t)n_cams, s));
  CH_TRY(dt.alloc(3 * (size_t)n_cams)); CH_TRY(dt.upload(cam_t, 3 * (size_t)n_cams, s));
  CH_TRY(dc.alloc(3 * (size_t)n_cams)); CH_TRY(dc.upload(cam_c, 3 * (size_t)n_cams, s));
  CH_TRY(dfk.alloc(3 * (size_t)n_cams)); CH_TRY(dfk.upload(cam_fk, 3 * (size_t)n_cams, s));
  CH_TRY(C->d_kp.from(C->kp, s));
  CH_TRY(C->xy.alloc(2 * (size_t)std::max(1, no)));
  CH_TRY(C->X.alloc(3 * (size_t)std::max(1, nt))); CH_TRY(C->mse.alloc(std::max(1, nt))); CH_TRY(C->tok.alloc(std::max(1, nt)));
  CH_TRY(hipMemsetAsync(C->X.p, 0, sizeof(double) * 3 * (size_t)std::max(1, nt), s));   // X is in / out: zeros where the 4x4 LLT fails
  if (no) hipLaunchKernelGGL(chn::k_track_xy, dim3(cdiv(no, 256)), dim3(256), 0, s, no, T.img.p, T.feat.p, C->d_kp.p, C->xy.p);
  const TrackPtrs P{nt, T.off.p, T.img.p, C->xy.p, dR.p, dt.p, dc.p, dfk.p};
  MSFM_TRY(tri_midpoint_dev(ctx, P, th_error, th_angle, C->X.p, C->mse.p, C->tok.p));
  CH_TRY(hipGetLastError());
  if (n_accepted) {
    std::vector<uint8_t> okh(std::max(1, nt));
    if (nt) CH_TRY(hipMemcpyAsync(okh.data(), C->tok.p, nt, hipMemcpyDeviceToHost, s));
    CH_TRY(hipStreamSynchronize(s));
    int n = 0;
    for (int t = 0; t < nt; t++) n += okh[t] != 0;
    *n_accepted = n;
  } else {
    CH_TRY(hipStreamSynchronize(s));   // the camera arrays above are released on return
  }
  C->triangulated = true;
  return MSFM_OK;
}

MSFM_API int msfm_chain_fetch_points(msfm_chain* C, double* X, double* mse, uint8_t* ok) {
  if (!C) return MSFM_E_INVAL;
  msfm_ctx* ctx = C->ctx;
  if (!C->triangulated) return msfm_set_error(ctx, MSFM_E_INVAL, "msfm_chain_fetch_points: msfm_chain_triangulate first");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int nt = C->tracks.n_tracks;
  if (nt) {
    if (X) CH_TRY(hipMemcpyAsync(X, C->X.p, sizeof(double) * 3 * (size_t)nt, hipMemcpyDeviceToHost, s));
    if (mse) CH_TRY(hipMemcpyAsync(mse, C->mse.p, sizeof(double) * (size_t)nt, hipMemcpyDeviceToHost, s));
    if (ok) CH_TRY(hipMemcpyAsync(ok, C->tok.p, (size_t)nt, hipMemcpyDeviceToHost, s));
  }
  CH_TRY(hipStreamSynchronize(s));
  return MSFM_OK;
}

MSFM_API int msfm_chain_ba_create(msfm_chain* C, int n_cams, int n_models, double* cam_pose, double* cam_model, const int32_t* cam_model_of_cam,
                                  int min_views, double weight_ge3, msfm_ba** out, int* n_points, int* n_observations) {
  if (!C || !out || !cam_pose || !cam_model || !cam_model_of_cam) return MSFM_E_INVAL;
  msfm_ctx* ctx = C->ctx;
  *out = nullptr;
  if (!C->triangulated) return msfm_set_error(ctx, MSFM_E_INVAL, "msfm_chain_ba_create: msfm_chain_triangulate first");
  if (n_cams < C->n_images) return msfm_set_error(ctx, MSFM_E_INVAL, "msfm_chain_ba_create: %d cameras for %d images", n_cams, C->n_images);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const msfm_track_dev& T = C->tracks;
  const int nt = T.n_tracks;
  DevBuf<int> keep, keep_obs, new_pt, new_obs;
  DevBuf<char> tmp;
  CH_TRY(keep.alloc((size_t)nt + 1)); CH_TRY(keep_obs.alloc((size_t)nt + 1)); CH_TRY(new_pt.alloc((size_t)nt + 1)); CH_TRY(new_obs.alloc((size_t)nt + 1));
  hipLaunchKernelGGL(chn::k_keep, dim3(cdiv(nt + 1, 256)), dim3(256), 0, s, nt, T.off.p, C->tok.p, min_views, keep.p, keep_obs.p);
  auto scan = [&](const int* in, int* o, size_t n) -> hipError_t {
    size_t bytes = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, bytes, in, o, 0, n, rocprim::plus<int>(), s);
    if (e != hipSuccess) return e;
    if (tmp.n < bytes) { e = tmp.alloc(bytes); if (e != hipSuccess) return e; }
    return rocprim::exclusive_scan(tmp.p, bytes, in, o, 0, n, rocprim::plus<int>(), s);
  };
  CH_TRY(scan(keep.p, new_pt.p, (size_t)nt + 1));
  CH_TRY(scan(keep_obs.p, new_obs.p, (size_t)nt + 1));
  int np = 0, no = 0;
  CH_TRY(hipMemcpyAsync(&np, new_pt.p + nt, sizeof(int), hipMemcpyDeviceToHost, s));
  CH_TRY(hipMemcpyAsync(&no, new_obs.p + nt, sizeof(int), hipMemcpyDeviceToHost, s));
  CH_TRY(hipStreamSynchronize(s));
  if (np == 0) return msfm_set_error(ctx, MSFM_E_INVAL, "msfm_chain_ba_create: no track qualifies (%d tracks)", nt);
  DevBuf<int> obs_cam, obs_pt;
  DevBuf<double> obs_xy, point, ptw;
  CH_TRY(obs_cam.alloc(no)); CH_TRY(obs_pt.alloc(no)); CH_TRY(obs_xy.alloc(2 * (size_t)no)); CH_TRY(point.alloc(3 * (size_t)np)); CH_TRY(ptw.alloc(np));
  CH_TRY(C->track_of_point.alloc(np));
  hipLaunchKernelGGL(chn::k_ba_arrays, dim3(cdiv(nt, 256)), dim3(256), 0, s, nt, T.off.p, T.img.p, C->xy.p, C->X.p, keep.p, new_pt.p, new_obs.p, weight_ge3,
                     obs_cam.p, obs_pt.p, obs_xy.p, point.p, ptw.p, C->track_of_point.p);
  CH_TRY(hipGetLastError());
  msfm_ba_problem P;
  memset(&P, 0, sizeof P);
  P.n_cams = n_cams; P.n_models = n_models; P.n_points = np; P.n_obs = no;
  P.cam_pose = cam_pose; P.cam_model = cam_model; P.cam_model_of_cam = cam_model_of_cam;
  P.point = point.p; P.obs_cam = obs_cam.p; P.obs_pt = obs_pt.p; P.obs_xy = obs_xy.p; P.pt_weight = ptw.p;
  MSFM_TRY(ba_create_impl(ctx, &P, /*bulk_on_device=*/true, out));
  CH_TRY(hipStreamSynchronize(s));
  C->n_ba_points = np; C->n_ba_obs = no;
  if (n_points) *n_points = np;
  if (n_observations) *n_observations = no;
  return MSFM_OK;
}

MSFM_API int msfm_chain_fetch_point_tracks(msfm_chain* C, int* track_of_point) {
  if (!C || !track_of_point) return MSFM_E_INVAL;
  msfm_ctx* ctx = C->ctx;
  if (C->n_ba_points == 0) return msfm_set_error(ctx, MSFM_E_INVAL, "msfm_chain_fetch_point_tracks: msfm_chain_ba_create first");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  CH_TRY(hipMemcpyAsync(track_of_point, C->track_of_point.p, sizeof(int) * (size_t)C->n_ba_points, hipMemcpyDeviceToHost, ctx->stream));
  CH_TRY(hipStreamSynchronize(ctx->stream));
  return MSFM_OK;
}

MSFM_API void msfm_chain_destroy(msfm_chain* C) {
  if (!C) return;
  msfm_ctx* ctx = C->ctx;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  delete C;
  msfm_ctx_child_released(ctx);
}
