// Track building between matching and triangulation: the data association of
//   SLAMGPS::Triangulation            SfM/src/slam_gps.cc:565-635
// restated on flat arrays, twice: msfm_tracks_build walks the match lists on the host exactly as the reference does,
// msfm_tracks_build_device produces the identical result on the GPU (below: the greedy walk is order dependent, but
// what it computes has a closed form over "first appearance" indices, which sorts and scans can evaluate).
// The reference keys a std::map<int,int> by `local + image * idx_max_per_image`; here feature -> track is a
// per-image array, and a track's observations are a (image -> feature) list with the map's semantics:
// std::map::insert keeps the FIRST value for a key, iteration is in ascending key order.
#include <algorithm>
#include <cstring>
#include <memory>

#include <rocprim/rocprim.hpp>

#include "common.h"

struct msfm_track_set {
  std::vector<int> off, img, feat;
};

MSFM_API int msfm_tracks_build(int n_images, const int* n_features, int n_pairs, const int* pair_img, const int* match_off,
                               const int* matches, msfm_track_set** out) {
  if (n_images < 0 || n_pairs < 0 || !out || (n_images && !n_features) || (n_pairs && (!pair_img || !match_off))) return MSFM_E_INVAL;
  if (n_pairs && match_off[n_pairs] > 0 && !matches) return MSFM_E_INVAL;
  std::vector<std::vector<int>> track_of(n_images);  // pts_points_map, per image
  for (int i = 0; i < n_images; i++) {
    if (n_features[i] < 0) return MSFM_E_INVAL;
    track_of[i].assign(n_features[i], -1);
  }
  std::vector<std::vector<std::pair<int, int>>> obs;  // per track: (image, feature), first insert per image wins
  auto add_obs = [&](int t, int image, int f) {
    for (auto& o : obs[t]) if (o.first == image) return;  // cams_.insert / pts2d_.insert on an existing key: no effect
    obs[t].push_back({image, f});
  };
  for (int p = 0; p < n_pairs; p++) {
    const int i1 = pair_img[2 * p], i2 = pair_img[2 * p + 1];
    if (i1 < 0 || i1 >= n_images || i2 < 0 || i2 >= n_images || match_off[p + 1] < match_off[p]) return MSFM_E_INVAL;
    for (int m = match_off[p]; m < match_off[p + 1]; m++) {
      const int f1 = matches[2 * m], f2 = matches[2 * m + 1];
      if (f1 < 0 || f1 >= n_features[i1] || f2 < 0 || f2 >= n_features[i2]) return MSFM_E_INVAL;
      int& t1 = track_of[i1][f1];
      int& t2 = track_of[i2][f2];
      if (t1 >= 0) {                 // slam_gps.cc:597-606: add the second feature to the first one's point
        add_obs(t1, i2, f2);
        if (t2 < 0) t2 = t1;         // pts_points_map.insert on an existing key: no effect
      } else if (t2 >= 0) {          // :607-616
        add_obs(t2, i1, f1);
        t1 = t2;
      } else {                       // :617-633: a new point with both observations
        const int t = (int)obs.size();
        obs.emplace_back();
        add_obs(t, i1, f1);
        add_obs(t, i2, f2);
        t1 = t;
        if (track_of[i2][f2] < 0) track_of[i2][f2] = t;  // (i1, f1) == (i2, f2) cannot happen for i1 != i2; kept for the insert semantics
      }
    }
  }
  msfm_track_set* S = new msfm_track_set();
  S->off.push_back(0);
  for (auto& o : obs) {
    std::sort(o.begin(), o.end());  // std::map iteration order: ascending image id
    for (auto& e : o) { S->img.push_back(e.first); S->feat.push_back(e.second); }
    S->off.push_back((int)S->img.size());
  }
  *out = S;
  return MSFM_OK;
}

MSFM_API int msfm_track_set_size(const msfm_track_set* S, int* n_tracks, int* n_obs) {
  if (!S) return MSFM_E_INVAL;
  if (n_tracks) *n_tracks = (int)S->off.size() - 1;
  if (n_obs) *n_obs = (int)S->img.size();
  return MSFM_OK;
}

MSFM_API int msfm_track_set_fetch(const msfm_track_set* S, int* track_off, int* obs_img, int* obs_feat) {
  if (!S) return MSFM_E_INVAL;
  if (track_off) std::copy(S->off.begin(), S->off.end(), track_off);
  if (obs_img) std::copy(S->img.begin(), S->img.end(), obs_img);
  if (obs_feat) std::copy(S->feat.begin(), S->feat.end(), obs_feat);
  return MSFM_OK;
}

MSFM_API void msfm_track_set_destroy(msfm_track_set* S) { delete S; }

// ======================================================================================================
// The same association on the device.
//
// The walk assigns every feature to a point at the FIRST match it appears in and never changes it again; two existing
// points are never merged.  With first(x) = index of the first match that names feature x, match m = (a, b) does this:
//   first(a) <  m                : a has its point already: b is observed by it; if first(b) == m, b joins it
//   first(a) == m,  first(b) < m : b has its point: a is observed by it and joins it
//   first(a) == m == first(b)    : a new point (numbered in match order) holding a and b
// so "joins" is a forest whose parent pointers go to strictly earlier first() and whose roots are the new points; the
// point of a feature is the number of its root.  Each match then contributes one observation (two for a new point) to a
// known point, a point keeps the earliest observation per image (std::map::insert), in ascending image order.
//   k_first (atomicMin) -> k_parent -> k_root (walk up; chains are as short as tracks) -> scan of the new-point flags
//   -> k_events (already in match order) -> stable radix sort by (point, image) -> heads -> compaction -> CSR.
// Integer work only: bit-identical to msfm_tracks_build by construction, compared in tests/test_gpu_tracks.py.
// ======================================================================================================
namespace trk {

__global__ __launch_bounds__(256) void k_first(int M, int n_pairs, int n_images, const int* __restrict__ match_off, const int* __restrict__ pair_img,
                                                const int* __restrict__ matches, const int* __restrict__ n_features,
                                                const int* __restrict__ feat_off, int* __restrict__ ga, int* __restrict__ gb,
                                                int* __restrict__ img_a, int* __restrict__ img_b, int* __restrict__ first, int* __restrict__ err) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  // the pair of match m: last p with match_off[p] <= m (empty pairs share their offset with the next one)
  int lo = 0, hi = n_pairs;   // invariant: match_off[lo] <= m < match_off[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (match_off[mid] <= m) lo = mid; else hi = mid;
  }
  const int i1 = pair_img[2 * lo], i2 = pair_img[2 * lo + 1];
  const int f1 = matches[2 * m], f2 = matches[2 * m + 1];
  if (f1 < 0 || f1 >= n_features[i1] || f2 < 0 || f2 >= n_features[i2]) { atomicMin(err, m); ga[m] = gb[m] = 0; img_a[m] = img_b[m] = 0; return; }
  const int a = feat_off[i1] + f1, b = feat_off[i2] + f2;
  ga[m] = a; gb[m] = b; img_a[m] = i1; img_b[m] = i2;
  atomicMin(&first[a], m);
  atomicMin(&first[b], m);
}

// parent[x] = x initially; every feature is written at most once (by its first match)
__global__ __launch_bounds__(256) void k_parent(int M, const int* __restrict__ ga, const int* __restrict__ gb, const int* __restrict__ first,
                                                 int* __restrict__ parent, int* __restrict__ is_new) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const int a = ga[m], b = gb[m];
  const bool fa = first[a] == m, fb = first[b] == m;
  int nw = 0;
  if (fa && fb) { if (b != a) parent[b] = a; nw = 1; }
  else if (fa) parent[a] = b;
  else if (fb) parent[b] = a;
  is_new[m] = nw;
}

__global__ __launch_bounds__(256) void k_iota_parent(int n, int* __restrict__ parent, int* __restrict__ first) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x < n) { parent[x] = x; first[x] = 0x7fffffff; }
}

// point of every feature that appears in a match: the number of the match that created its root
__global__ __launch_bounds__(256) void k_root(int n, const int* __restrict__ parent, const int* __restrict__ first, const int* __restrict__ new_before,
                                               int* __restrict__ point_of) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= n) return;
  if (first[x] == 0x7fffffff) { point_of[x] = -1; return; }
  int r = x;
  while (parent[r] != r) r = parent[r];
  point_of[x] = new_before[first[r]];   // the root's first match is the one that created the point
}

// observations in match order: slot m + (new points before m); a new point's second observation takes the next slot
__global__ __launch_bounds__(256) void k_events(int M, int n_images, const int* __restrict__ ga, const int* __restrict__ gb, const int* __restrict__ img_a,
                                                 const int* __restrict__ img_b, const int* __restrict__ matches, const int* __restrict__ first,
                                                 const int* __restrict__ new_before, const int* __restrict__ point_of,
                                                 unsigned long long* __restrict__ key, int* __restrict__ val) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const int a = ga[m], b = gb[m];
  const bool fa = first[a] == m, fb = first[b] == m;
  const long e = (long)m + new_before[m];
  const unsigned long long ni = (unsigned long long)n_images;
  if (!fa) {
    key[e] = (unsigned long long)point_of[a] * ni + img_b[m]; val[e] = matches[2 * m + 1];
  } else if (!fb) {
    key[e] = (unsigned long long)point_of[b] * ni + img_a[m]; val[e] = matches[2 * m];
  } else {
    const unsigned long long t = (unsigned long long)new_before[m];
    key[e] = t * ni + img_a[m]; val[e] = matches[2 * m];
    key[e + 1] = t * ni + img_b[m]; val[e + 1] = matches[2 * m + 1];
  }
}

__global__ __launch_bounds__(256) void k_heads(long E, const unsigned long long* __restrict__ key, int* __restrict__ head) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e < E) head[e] = (e == 0 || key[e] != key[e - 1]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_compact(long E, int n_images, const unsigned long long* __restrict__ key, const int* __restrict__ val,
                                                  const int* __restrict__ head, const int* __restrict__ slot, int* __restrict__ obs_img,
                                                  int* __restrict__ obs_feat, int* __restrict__ off) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= E || !head[e]) return;
  const unsigned long long ni = (unsigned long long)n_images;
  const int j = slot[e];
  obs_img[j] = (int)(key[e] % ni);
  obs_feat[j] = val[e];
  const unsigned long long t = key[e] / ni;
  if (e == 0 || key[e - 1] / ni != t) off[t] = j;   // first observation of point t (every point has at least one)
}

static hipError_t scan_excl(const int* in, int* out, size_t n, hipStream_t s, DevBuf<char>& tmp) {
  if (n == 0) return hipSuccess;
  size_t bytes = 0;
  hipError_t e = rocprim::exclusive_scan(nullptr, bytes, in, out, 0, n, rocprim::plus<int>(), s);
  if (e != hipSuccess) return e;
  if (tmp.n < bytes) { e = tmp.alloc(bytes); if (e != hipSuccess) return e; }
  return rocprim::exclusive_scan(tmp.p, bytes, in, out, 0, n, rocprim::plus<int>(), s);
}

}  // namespace trk

// The association on match lists that are already resident (d_pair [n_pairs][2], d_moff [n_pairs+1], d_match [M][2]; the
// per-image feature counts and their prefix sums also on the host: O(images)).  The CSR tracks stay on the device.
// msfm_tracks_build_device is this between an upload and a download; msfm_chain_build_tracks (chain.hip) calls it on the
// verified matches it holds.
int tracks_build_dev(msfm_ctx* ctx, int n_images, const std::vector<int>& feat_off, const int* d_nf, const int* d_fo, int n_pairs,
                     const int* d_pair, const int* d_moff, const int* d_match, int M, msfm_track_dev* out) {
  using namespace trk;
  hipStream_t s = ctx->stream;
  out->n_tracks = 0; out->n_obs = 0;
  if (M == 0) {
    HIP_TRY(ctx, out->off.alloc(1));
    HIP_TRY(ctx, hipMemsetAsync(out->off.p, 0, sizeof(int), s));
    return MSFM_OK;
  }
  const int NF = feat_off[n_images];
#define TTRY(e) HIP_TRY(ctx, (e))
  DevBuf<int> ga, gb, ia, ib, first, parent, is_new, new_before, point_of, err, val, val_s, head, slot;
  DevBuf<unsigned long long> key, key_s;
  DevBuf<char> tmp;
  TTRY(ga.alloc(M)); TTRY(gb.alloc(M)); TTRY(ia.alloc(M)); TTRY(ib.alloc(M));
  TTRY(first.alloc(std::max(1, NF))); TTRY(parent.alloc(std::max(1, NF))); TTRY(point_of.alloc(std::max(1, NF)));
  TTRY(is_new.alloc((size_t)M + 1)); TTRY(new_before.alloc((size_t)M + 1)); TTRY(err.alloc(1));
  const int big = 0x7fffffff;
  TTRY(hipMemcpyAsync(err.p, &big, sizeof(int), hipMemcpyHostToDevice, s));
  TTRY(hipMemsetAsync(is_new.p + M, 0, sizeof(int), s));
  if (NF) hipLaunchKernelGGL(k_iota_parent, dim3(cdiv(NF, 256)), dim3(256), 0, s, NF, parent.p, first.p);
  hipLaunchKernelGGL(k_first, dim3(cdiv(M, 256)), dim3(256), 0, s, M, n_pairs, n_images, d_moff, d_pair, d_match, d_nf, d_fo, ga.p, gb.p, ia.p, ib.p,
                     first.p, err.p);
  int bad = big;
  TTRY(hipMemcpyAsync(&bad, err.p, sizeof(int), hipMemcpyDeviceToHost, s));
  TTRY(hipStreamSynchronize(s));
  if (bad != big) return msfm_set_error(ctx, MSFM_E_INVAL, "msfm_tracks_build_device: match %d names a feature outside its image", bad);
  hipLaunchKernelGGL(k_parent, dim3(cdiv(M, 256)), dim3(256), 0, s, M, ga.p, gb.p, first.p, parent.p, is_new.p);
  TTRY(scan_excl(is_new.p, new_before.p, (size_t)M + 1, s, tmp));
  hipLaunchKernelGGL(k_root, dim3(cdiv(NF, 256)), dim3(256), 0, s, NF, parent.p, first.p, new_before.p, point_of.p);
  int n_tracks = 0;
  TTRY(hipMemcpyAsync(&n_tracks, new_before.p + M, sizeof(int), hipMemcpyDeviceToHost, s));
  TTRY(hipStreamSynchronize(s));
  const long E = (long)M + n_tracks;
  TTRY(key.alloc(E)); TTRY(key_s.alloc(E)); TTRY(val.alloc(E)); TTRY(val_s.alloc(E)); TTRY(head.alloc(E + 1)); TTRY(slot.alloc(E + 1));
  hipLaunchKernelGGL(k_events, dim3(cdiv(M, 256)), dim3(256), 0, s, M, n_images, ga.p, gb.p, ia.p, ib.p, d_match, first.p, new_before.p, point_of.p, key.p,
                     val.p);
  {
    // stable: within one (point, image) the earliest match stays first
    int bits = 1;
    while (bits < 64 && ((unsigned long long)n_tracks * (unsigned long long)std::max(1, n_images)) >> bits) bits++;
    size_t bytes = 0;
    TTRY(rocprim::radix_sort_pairs(nullptr, bytes, key.p, key_s.p, val.p, val_s.p, (size_t)E, 0, bits, s));
    if (tmp.n < bytes) TTRY(tmp.alloc(bytes));
    TTRY(rocprim::radix_sort_pairs(tmp.p, bytes, key.p, key_s.p, val.p, val_s.p, (size_t)E, 0, bits, s));
  }
  TTRY(hipMemsetAsync(head.p + E, 0, sizeof(int), s));
  hipLaunchKernelGGL(k_heads, dim3(cdiv(E, 256)), dim3(256), 0, s, E, key_s.p, head.p);
  TTRY(scan_excl(head.p, slot.p, (size_t)E + 1, s, tmp));
  int n_obs = 0;
  TTRY(hipMemcpyAsync(&n_obs, slot.p + E, sizeof(int), hipMemcpyDeviceToHost, s));
  TTRY(hipStreamSynchronize(s));
  TTRY(out->off.alloc((size_t)n_tracks + 1)); TTRY(out->img.alloc(std::max(1, n_obs))); TTRY(out->feat.alloc(std::max(1, n_obs)));
  hipLaunchKernelGGL(k_compact, dim3(cdiv(E, 256)), dim3(256), 0, s, E, n_images, key_s.p, val_s.p, head.p, slot.p, out->img.p, out->feat.p, out->off.p);
  TTRY(hipMemcpyAsync(out->off.p + n_tracks, &n_obs, sizeof(int), hipMemcpyHostToDevice, s));
  TTRY(hipStreamSynchronize(s));   // the scratch above is released on return
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) return msfm_set_error(ctx, MSFM_E_DEVICE, "msfm_tracks_build_device: %s", hipGetErrorString(le));
#undef TTRY
  out->n_tracks = n_tracks; out->n_obs = n_obs;
  return MSFM_OK;
}

MSFM_API int msfm_tracks_build_device(msfm_ctx* ctx, int n_images, const int* n_features, int n_pairs, const int* pair_img,
                                      const int* match_off, const int* matches, msfm_track_set** out) {
  if (!ctx) return MSFM_E_INVAL;
  if (n_images < 0 || n_pairs < 0 || !out || (n_images && !n_features) || (n_pairs && (!pair_img || !match_off)))
    return msfm_set_error(ctx, MSFM_E_INVAL, "msfm_tracks_build_device: null argument");
  *out = nullptr;
  // what is O(images + pairs) is checked on the host; the features of the matches on the device
  std::vector<int> feat_off(n_images + 1, 0);
  for (int i = 0; i < n_images; i++) {
    if (n_features[i] < 0) return msfm_set_error(ctx, MSFM_E_INVAL, "msfm_tracks_build_device: n_features[%d] < 0", i);
    if ((long)feat_off[i] + n_features[i] > 0x7fffffffL) return msfm_set_error(ctx, MSFM_E_INVAL, "msfm_tracks_build_device: more than 2^31 features");
    feat_off[i + 1] = feat_off[i] + n_features[i];
  }
  for (int p = 0; p < n_pairs; p++) {
    const int i1 = pair_img[2 * p], i2 = pair_img[2 * p + 1];
    if (i1 < 0 || i1 >= n_images || i2 < 0 || i2 >= n_images || match_off[p + 1] < match_off[p] || match_off[p] < 0)
      return msfm_set_error(ctx, MSFM_E_INVAL, "msfm_tracks_build_device: pair %d", p);
  }
  const int M = n_pairs ? match_off[n_pairs] : 0;
  if (n_pairs && match_off[0] != 0) return msfm_set_error(ctx, MSFM_E_INVAL, "msfm_tracks_build_device: match_off[0] != 0");
  if (M > 0 && !matches) return msfm_set_error(ctx, MSFM_E_INVAL, "msfm_tracks_build_device: null matches");
  msfm_track_set* S = new msfm_track_set();
  S->off.push_back(0);
  if (M == 0) { *out = S; return MSFM_OK; }
  std::unique_ptr<msfm_track_set> guard(S);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  DevBuf<int> d_nf, d_fo, d_pair, d_moff, d_match;
  HIP_TRY(ctx, d_nf.alloc(n_images)); HIP_TRY(ctx, d_nf.upload(n_features, n_images, s));
  HIP_TRY(ctx, d_fo.from(feat_off, s));
  HIP_TRY(ctx, d_pair.alloc(2 * (size_t)n_pairs)); HIP_TRY(ctx, d_pair.upload(pair_img, 2 * (size_t)n_pairs, s));
  HIP_TRY(ctx, d_moff.alloc((size_t)n_pairs + 1)); HIP_TRY(ctx, d_moff.upload(match_off, (size_t)n_pairs + 1, s));
  HIP_TRY(ctx, d_match.alloc(2 * (size_t)M)); HIP_TRY(ctx, d_match.upload(matches, 2 * (size_t)M, s));
  msfm_track_dev D;
  MSFM_TRY(tracks_build_dev(ctx, n_images, feat_off, d_nf.p, d_fo.p, n_pairs, d_pair.p, d_moff.p, d_match.p, M, &D));
  S->off.resize((size_t)D.n_tracks + 1); S->img.resize(D.n_obs); S->feat.resize(D.n_obs);
  HIP_TRY(ctx, hipMemcpyAsync(S->off.data(), D.off.p, sizeof(int) * ((size_t)D.n_tracks + 1), hipMemcpyDeviceToHost, s));
  if (D.n_obs) {
    HIP_TRY(ctx, hipMemcpyAsync(S->img.data(), D.img.p, sizeof(int) * (size_t)D.n_obs, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(S->feat.data(), D.feat.p, sizeof(int) * (size_t)D.n_obs, hipMemcpyDeviceToHost, s));
  }
  HIP_TRY(ctx, hipStreamSynchronize(s));
  *out = guard.release();
  return MSFM_OK;
}
