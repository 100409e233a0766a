// test_sfm — the reference's entry point (SfM/test/test_sfm/test_sfm.cc:22-70: set options,
// InitializeSystem(), Run()) reduced to the stages that sit on the GPU hot path, on BASELINE
// config 1 (10 synthetic pinhole cameras, 2k points, 128-D descriptors):
//   matching (all ordered pairs, ratio tests)  -> tracks -> Trianglate2 -> seed-style full bundle
//   adjustment with Normalize + Perturb (sfm_incremental.cc:393,1016-1026) -> outlier check.
// `int main`, no hard-coded Windows paths; exits non-zero if a stage misbehaves.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <numeric>
#include <random>

#include "objectsfm.h"

using namespace objectsfm;

int main(int argc, char** argv) {
  // --gpus N [--share-device]: N contexts in THIS process (the reference's entry point stays one `main`); --share-device puts
  // all of them on device 0, which is how the multi-GPU path runs on a one-GPU box
  int gpus = 1;
  bool share = false;
  for (int a = 1; a < argc; a++) {
    if (!std::strcmp(argv[a], "--gpus") && a + 1 < argc) gpus = std::atoi(argv[++a]);
    else if (!std::strcmp(argv[a], "--share-device")) share = true;
  }
  if (gpus > 1) { UseGpus(gpus, share); std::printf("contexts: %d%s\n", gpus, share ? " (one device)" : ""); }
  const int n_cams = 10, n_pts = 2000, W = 4000, H = 3000;
  const double f = 1.2 * W;  // f_hyp_ convention, basic_structs.h:56
  std::mt19937_64 gen(0x4D53464DULL + 1);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  std::normal_distribution<double> N(0.0, 1.0);
  // --- scene ---
  CameraModel model(0, H, W, 0.0, f, "synthetic", "pinhole");
  std::vector<Camera> cams(n_cams);
  for (int i = 0; i < n_cams; i++) {
    const double ang = 2 * M_PI * i / n_cams;
    Vec3 c; c[0] = 150 * std::cos(ang); c[1] = 150 * std::sin(ang); c[2] = 60;
    const double nc = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    double z[3] = {-c[0] / nc, -c[1] / nc, -c[2] / nc};
    double x[3] = {z[1], -z[0], 0};  // z x (0,0,1)
    const double nx = std::sqrt(x[0] * x[0] + x[1] * x[1]);
    for (double& v : x) v /= nx;
    double y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
    Mat3 R;
    for (int k = 0; k < 3; k++) { R(0, k) = x[k]; R(1, k) = y[k]; R(2, k) = z[k]; }
    Vec3 t = R * c;
    for (int k = 0; k < 3; k++) t[k] = -t[k];
    cams[i].AssociateImage(i);
    cams[i].AssociateCamereModel(&model);
    model.AddCamera(i);
    cams[i].SetRTPose(R, t);
  }
  std::vector<std::array<double, 3>> X(n_pts);
  std::vector<std::array<float, 128>> base(n_pts);
  std::gamma_distribution<double> G(0.6, 1.0);
  for (int p = 0; p < n_pts; p++) {
    X[p] = {30 * U(gen), 30 * U(gen), 10 * U(gen)};
    double nrm = 0, v[128];
    for (double& e : v) { e = G(gen); nrm += e * e; }
    for (int k = 0; k < 128; k++) base[p][k] = (float)std::min(255.0, std::floor(v[k] * 512.0 / std::sqrt(nrm) + 0.5));
  }
  // features per image: every point is seen by every camera (config 1), random feature order
  std::vector<std::vector<float>> desc(n_cams);
  std::vector<std::vector<Vec2>> kp(n_cams);
  std::vector<std::vector<int>> feat_pt(n_cams);
  std::uniform_int_distribution<int> noise(-4, 4);
  for (int i = 0; i < n_cams; i++) {
    std::vector<int> order(n_pts);
    std::iota(order.begin(), order.end(), 0);
    std::shuffle(order.begin(), order.end(), gen);
    desc[i].resize((size_t)n_pts * 128);
    kp[i].resize(n_pts);
    feat_pt[i] = order;
    for (int m = 0; m < n_pts; m++) {
      const int p = order[m];
      Vec3 xw; for (int k = 0; k < 3; k++) xw[k] = X[p][k];
      Vec3 pc = cams[i].pos_rt_.R * xw;
      for (int k = 0; k < 3; k++) pc[k] += cams[i].pos_rt_.t[k];
      kp[i][m].x = f * pc[0] / pc[2] + 0.5 * N(gen);
      kp[i][m].y = f * pc[1] / pc[2] + 0.5 * N(gen);
      for (int k = 0; k < 128; k++) desc[i][(size_t)m * 128 + k] = std::min(255.f, std::max(0.f, base[p][k] + noise(gen)));
    }
  }
  // --- stage 0: the extraction stage hands over <idx>_feature files (database.cc:490-541): write them, read them back ---
  char tmpl[] = "/tmp/msfm_test_sfm_XXXXXX";
  const std::string fold = mkdtemp(tmpl) ? std::string(tmpl) : std::string("/tmp");
  for (int i = 0; i < n_cams; i++) {
    ImageInfo info;
    info.rows = H; info.cols = W; info.f_pixel = (float)f; info.cam_maker = "synthetic"; info.cam_model = "pinhole";
    std::vector<Point2f> px(kp[i].size());
    for (size_t m = 0; m < px.size(); m++) { px[m].x = (float)(kp[i][m].x + W / 2.0); px[m].y = (float)(kp[i][m].y + H / 2.0); }
    if (!WriteoutImageFeature(fold, i, info, px, desc[i])) { std::printf("FAIL: feature file write\n"); return 1; }
    ImageInfo back; std::vector<Point2f> kc; std::vector<float> d; int dc = 0;
    if (!ReadinImageFeatures(fold, i, back, kc, d, dc) || dc != 128 || d != desc[i] || back.cols != W || back.cam_model != "pinhole" ||
        kc.size() != px.size() || std::fabs(kc[0].x - (float)kp[i][0].x) > 1e-3f) { std::printf("FAIL: feature file read\n"); return 1; }
    desc[i] = d;  // matching runs on what came back from disk
  }
  // --- stage 1: matching, matching_type = "all" (test_sfm.cc:50; initial_matching_graph.cc:55-63) ---
  std::vector<std::pair<int, int>> pairs;
  for (int i = 0; i < n_cams; i++) for (int j = 0; j < n_cams; j++) if (i != j) pairs.push_back({i, j});
  std::vector<PairMatches> matches = MatchImagePairs(desc, pairs);
  long good = 0, good_right = 0;
  for (auto& pm : matches)
    for (auto& m : pm.matches_good) { good++; good_right += feat_pt[pm.idx1][m.first] == feat_pt[pm.idx2][m.second]; }
  std::printf("matching: %zu pairs, %ld good matches, %.2f%% correct\n", pairs.size(), good, 100.0 * good_right / std::max(1L, good));
  if (good < 0.8 * pairs.size() * n_pts || good_right < 0.99 * good) { std::printf("FAIL: matching\n"); return 1; }
  // geometric verification (fine_matching_graph.cc:138-153): F by RANSAC on the good set, filter of the all set
  std::vector<std::vector<Point2f>> kpf(n_cams);
  for (int i = 0; i < n_cams; i++) {
    kpf[i].resize(kp[i].size());
    for (size_t m = 0; m < kp[i].size(); m++) { kpf[i][m].x = (float)kp[i][m].x; kpf[i][m].y = (float)kp[i][m].y; }
  }
  std::vector<std::vector<std::pair<int, int>>> verified = VerifyPairs(matches, kpf);
  long kept = 0, kept_right = 0, all_right = 0;
  for (size_t p = 0; p < matches.size(); p++) {
    for (auto& m : matches[p].matches_all) all_right += feat_pt[matches[p].idx1][m.first] == feat_pt[matches[p].idx2][m.second];
    for (auto& m : verified[p]) { kept++; kept_right += feat_pt[matches[p].idx1][m.first] == feat_pt[matches[p].idx2][m.second]; }
  }
  std::printf("verification: %ld matches kept, %.2f%% correct, %.2f%% of the correct ones kept\n", kept, 100.0 * kept_right / std::max(1L, kept),
              100.0 * kept_right / std::max(1L, all_right));
  if (kept_right < 0.95 * all_right || kept_right < 0.995 * kept) { std::printf("FAIL: geometric verification\n"); return 1; }
  {  // the single-pair members agree with the batch
    std::vector<Point2f> a, b;
    for (auto& m : matches[0].matches_good) { a.push_back(kpf[matches[0].idx1][m.first]); b.push_back(kpf[matches[0].idx2][m.second]); }
    std::vector<int> inl;
    Mat3 Fm;
    if (!GeoVerification::GeoVerificationFundamental(a, b, inl, Fm)) { std::printf("FAIL: GeoVerificationFundamental\n"); return 1; }
  }
  // the reference hands matches to the SfM stage through files: write the verified ones, read image 0's back
  std::vector<std::vector<int>> match_graph(n_cams, std::vector<int>(n_cams, 0));
  for (size_t p = 0; p < matches.size(); p++) {
    auto& pm = matches[p];
    pm.matches_good = verified[p];  // matches_inliers of fine_matching_graph.cc:153-156 feed the next stage
    WriteOutMatches(fold, pm.idx1, pm.idx2, pm.matches_good);
    match_graph[pm.idx1][pm.idx2] = (int)pm.matches_good.size();
  }
  WriteOutMatchGraph(fold, match_graph);
  std::vector<int> ids0;
  std::vector<std::vector<std::pair<int, int>>> m0;
  QueryMatch(fold, 0, ids0, m0);  // Graph::QueryMatch, graph.cc:92-121
  if ((int)ids0.size() != n_cams - 1) { std::printf("FAIL: match files\n"); return 1; }
  // --- stage 2: tracks from the whole match graph (SLAMGPS::Triangulation's data association, slam_gps.cc:565-635), Trianglate2 ---
  std::vector<Point3D> pts = BuildTracks(fold, match_graph, cams, kp);
  size_t pure = 0;
  for (auto& p : pts) {   // with verified matches a track never mixes two scene points
    int first = -1; bool same = true;
    for (auto& o : p.pts2d_) {
      int f = -1;
      for (size_t m = 0; m < kp[o.first].size(); m++) if (kp[o.first][m].x == o.second.x && kp[o.first][m].y == o.second.y) { f = (int)m; break; }
      const int sp = f >= 0 ? feat_pt[o.first][f] : -2;
      if (first < 0) first = sp; else same = same && sp == first;
    }
    pure += same;
  }
  std::printf("tracks: %zu points from the match graph, %zu pure\n", pts.size(), pure);
  if (pts.size() < (size_t)n_pts || pure < 0.99 * pts.size()) { std::printf("FAIL: track building\n"); return 1; }
  std::vector<Point3D*> pp;
  for (auto& p : pts) if (p.cams_.size() >= 3) pp.push_back(&p);  // fewer than 3 views -> bad (slam_gps.cc:642)
  std::vector<char> ok;
  TrianglateBatch(pp, 7.0, 3.0 / 180.0 * M_PI, /*dlt=*/false, &ok);  // thresholds of sfm_incremental.cc:780-784
  size_t n_ok = 0;
  for (size_t i = 0; i < pp.size(); i++) { n_ok += ok[i] != 0; pp[i]->is_bad_estimated_ = !ok[i]; }
  std::printf("triangulation: %zu tracks, %zu accepted\n", pp.size(), n_ok);
  if (n_ok < 0.9 * pp.size()) { std::printf("FAIL: triangulation\n"); return 1; }
  // --- stage 2b: the pose initialisers of the incremental loop ---
  // seed pair (sfm_incremental.cc:296-318): relative pose of image 1 against image 0 from their verified matches
  {
    size_t k1 = 0;
    while (k1 < ids0.size() && ids0[k1] != 1) k1++;
    if (k1 == ids0.size()) { std::printf("FAIL: no matches between images 0 and 1\n"); return 1; }
    std::vector<Vec2> pts1, pts2;
    for (auto& m : m0[k1]) { pts1.push_back(kp[0][m.first]); pts2.push_back(kp[1][m.second]); }
    RTPoseRelative rt21;
    if (!RelativePoseEstimation::RelativePoseWithFocalLength(pts1, pts2, f, f, rt21)) { std::printf("FAIL: RelativePoseWithFocalLength\n"); return 1; }
    Mat3 Rt;  // Xc1 = R1 R0^T Xc0 + ...
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) {
        Rt(r, c) = 0;
        for (int k = 0; k < 3; k++) Rt(r, c) += cams[1].pos_rt_.R(r, k) * cams[0].pos_rt_.R(c, k);
      }
    double dR = 0;
    for (int k = 0; k < 9; k++) dR = std::max(dR, std::fabs(rt21.R.m[k] - Rt.m[k]));
    std::printf("seed pair: %zu matches, relative rotation off by %.2e\n", pts1.size(), dR);
    if (!(dR < 2e-2)) { std::printf("FAIL: relative pose\n"); return 1; }
  }
  // localisation (sfm_incremental.cc:560-700): every camera against the triangulated points it observes, EPnP RANSAC
  {
    std::vector<std::vector<Vec3>> pw(n_cams);
    std::vector<std::vector<Vec2>> p2(n_cams);
    for (Point3D* p : pp) {
      if (p->is_bad_estimated_) continue;
      for (auto& o : p->pts2d_) {
        Vec3 xw; for (int k = 0; k < 3; k++) xw[k] = p->data[k];
        pw[o.first].push_back(xw);
        p2[o.first].push_back(o.second);
      }
    }
    std::vector<RTPose> poses;
    std::vector<std::vector<double>> errs;
    std::vector<double> avg;
    AbsolutePoseBatch(pw, p2, std::vector<double>(n_cams, f), poses, errs, avg);
    double worst_avg = 0, worst_R = 0;
    for (int i = 0; i < n_cams; i++) {
      worst_avg = std::max(worst_avg, avg[i]);
      for (int k = 0; k < 9; k++) worst_R = std::max(worst_R, std::fabs(poses[i].R.m[k] - cams[i].pos_rt_.R.m[k]));
    }
    std::printf("localisation: %d cameras, worst avg_error %.3f px (th_mse_localization 5.0), worst rotation error %.2e\n", n_cams, worst_avg, worst_R);
    if (!(worst_avg < 5.0) || !(worst_R < 2e-2)) { std::printf("FAIL: absolute pose\n"); return 1; }
    // the single-image entry point gives the same answer as its row of the batch
    RTPose one; std::vector<double> e1; double a1 = 0;
    AbsolutePoseEstimation::AbsolutePoseWithFocalLength(pw[0], p2[0], f, one, e1, a1);
    if (a1 != avg[0] || std::memcmp(one.R.m, poses[0].R.m, sizeof one.R.m) != 0) { std::printf("FAIL: batch / single mismatch\n"); return 1; }
  }
  // --- stage 3: full bundle adjustment as the seed reconstruction runs it (is_initial_run = true) ---
  std::vector<Camera*> cp;
  for (auto& c : cams) cp.push_back(&c);
  BundleAdjuster ba(cp, {&model}, pp);
  BundleAdjustOptions opt;
  opt.max_num_iterations = 100;  // th_max_iteration_full_bundle, test_sfm.cc:35
  opt.minimizer_progress_to_stdout = true;
  ba.SetOptions(opt);
  ba.RunOptimizetion(true, 1.0);
  ba.UpdateParameters();
  std::printf("BA: %d iterations, cost %.6e -> %.6e, termination %d, f = %.3f k1 = %.3e k2 = %.3e\n", ba.summary_.num_iterations,
              ba.summary_.initial_cost, ba.summary_.final_cost, ba.summary_.termination, model.f_, model.k1_, model.k2_);
  {
    // full-precision figures for the tests that compare runs with different numbers of contexts
    double sc = 0, sp = 0;
    for (auto& c : cams) for (int k = 0; k < 6; k++) sc += std::fabs(c.data[k]);
    for (Point3D* p : pp) for (int k = 0; k < 3; k++) sp += std::fabs(p->data[k]);
    std::printf("ba_final %.17g %.17g %.17g %.17g %.17g %d\n", ba.summary_.final_cost, model.f_, model.k1_, sc, sp, ba.summary_.num_iterations);
  }
  // --- stage 4: RemovePointOutliers (sfm_incremental.cc:1831-1863), th_mse_outliers = 3.0 ---
  std::vector<Point3D*> live;
  for (Point3D* p : pp) if (!p->is_bad_estimated_) live.push_back(p);
  ReprojectionBatch(live);
  double rms = 0; size_t outl = 0;
  for (Point3D* p : live) { rms += p->mse_; if (std::sqrt(p->mse_) > 3.0) { p->is_bad_estimated_ = true; outl++; } }
  rms = std::sqrt(rms / live.size());
  std::printf("after BA: rms reprojection error %.3f px over %zu points, %zu outliers\n", rms, live.size(), outl);
  if (!(rms < 1.5) || ba.summary_.termination > MSFM_BA_CONVERGENCE_PARAMETER) { std::printf("FAIL: bundle adjustment\n"); return 1; }
  std::printf("test_sfm ok\n");
  return 0;
}
