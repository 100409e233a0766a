// Track building between matching and triangulation: the data association of
//   SLAMGPS::Triangulation            SfM/src/slam_gps.cc:565-635
// restated on flat arrays.  Host code only (the association is an order-dependent greedy walk over the
// match lists; its result feeds msfm_triangulate_midpoint_batch, which is where the GPU work is).
// The reference keys a std::map<int,int> by `local + image * idx_max_per_image`; here feature -> track is a
// per-image array, and a track's observations are a (image -> feature) list with the map's semantics:
// std::map::insert keeps the FIRST value for a key, iteration is in ascending key order.
#include "common.h"

#include <algorithm>

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
