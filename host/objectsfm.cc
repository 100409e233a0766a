// See objectsfm.h.  Host-side glue only: gather -> C ABI -> scatter.
#include "objectsfm.h"

#include <cstdio>
#include <fstream>
#include <cstdlib>
#include <limits>
#include <stdexcept>

namespace objectsfm {

// ---- rotation (SfM/src/utils/basic_funcs.cc:25-158) --------------------------------------
void rotation::AngleAxisToRotationMatrix(const Vec3& aa, Mat3& R) {
  const double theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (theta2 > std::numeric_limits<double>::epsilon()) {
    const double theta = std::sqrt(theta2);
    const double wx = aa[0] / theta, wy = aa[1] / theta, wz = aa[2] / theta;
    const double c = std::cos(theta), s = std::sin(theta);
    R(0, 0) = c + wx * wx * (1 - c);       R(1, 0) = wz * s + wx * wy * (1 - c);  R(2, 0) = -wy * s + wx * wz * (1 - c);
    R(0, 1) = wx * wy * (1 - c) - wz * s;  R(1, 1) = c + wy * wy * (1 - c);       R(2, 1) = wx * s + wy * wz * (1 - c);
    R(0, 2) = wy * s + wx * wz * (1 - c);  R(1, 2) = -wx * s + wy * wz * (1 - c); R(2, 2) = c + wz * wz * (1 - c);
  } else {
    R(0, 0) = 1; R(1, 0) = aa[2]; R(2, 0) = -aa[1];
    R(0, 1) = -aa[2]; R(1, 1) = 1; R(2, 1) = aa[0];
    R(0, 2) = aa[1]; R(1, 2) = -aa[0]; R(2, 2) = 1;
  }
}

void rotation::RotationMatrixToAngleAxis(const Mat3& R, Vec3& axis) {
  double q[4];
  const double trace = R(0, 0) + R(1, 1) + R(2, 2);
  if (trace >= 0.0) {
    double t = std::sqrt(trace + 1.0);
    q[0] = 0.5 * t; t = 0.5 / t;
    q[1] = (R(2, 1) - R(1, 2)) * t; q[2] = (R(0, 2) - R(2, 0)) * t; q[3] = (R(1, 0) - R(0, 1)) * t;
  } else {
    int i = 0;
    if (R(1, 1) > R(0, 0)) i = 1;
    if (R(2, 2) > R(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = std::sqrt(R(i, i) - R(j, j) - R(k, k) + 1.0);
    q[i + 1] = 0.5 * t; t = 0.5 / t;
    q[0] = (R(k, j) - R(j, k)) * t; q[j + 1] = (R(j, i) + R(i, j)) * t; q[k + 1] = (R(k, i) + R(i, k)) * t;
  }
  const double s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  double k = 2.0;
  if (s2 > 0.0) {
    const double s = std::sqrt(s2);
    k = 2.0 * ((q[0] < 0.0) ? std::atan2(-s, -q[0]) : std::atan2(s, q[0])) / s;
  }
  axis[0] = q[1] * k; axis[1] = q[2] * k; axis[2] = q[3] * k;
}

// ---- CameraModel / Camera ------------------------------------------------------------------
CameraModel::CameraModel(int id, int h, int w, double f_mm, double f, std::string cam_maker, std::string cam_model) {
  f_mm_ = f_mm; f_ = f; f_hyp_ = (w > h ? w : h) * 1.2; w_ = w; h_ = h; px_ = w / 2.0; py_ = h / 2.0;
  id_ = id; cam_maker_ = cam_maker; cam_model_ = cam_model;
  UpdateDataFromModel();
}

void Camera::SetRTPose(const Mat3& R, const Vec3& t) {
  pos_rt_.R = R; pos_rt_.t = t;
  rotation::RotationMatrixToAngleAxis(pos_rt_.R, pos_ac_.a);
  const Vec3 c = transpose(R) * t;  // c = -R^-1 t
  for (int i = 0; i < 3; i++) pos_ac_.c[i] = -c[i];
  UpdateDataFromPose();
}

void Camera::SetACPose(const Vec3& a, const Vec3& c) {
  pos_ac_.a = a; pos_ac_.c = c;
  rotation::AngleAxisToRotationMatrix(pos_ac_.a, pos_rt_.R);
  const Vec3 t = pos_rt_.R * c;
  for (int i = 0; i < 3; i++) pos_rt_.t[i] = -t[i];
  UpdateDataFromPose();
}

void Camera::UpdateDataFromPose() {
  for (int i = 0; i < 3; i++) { data[i] = pos_ac_.a[i]; data[3 + i] = pos_rt_.t[i]; }
  for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) M[4 * r + c] = pos_rt_.R(r, c); M[4 * r + 3] = pos_rt_.t[r]; }
}

void Camera::UpdatePoseFromData() {
  for (int i = 0; i < 3; i++) { pos_ac_.a[i] = data[i]; pos_rt_.t[i] = data[3 + i]; }
  rotation::AngleAxisToRotationMatrix(pos_ac_.a, pos_rt_.R);
  const Vec3 c = transpose(pos_rt_.R) * pos_rt_.t;
  for (int i = 0; i < 3; i++) pos_ac_.c[i] = -c[i];
  for (int r = 0; r < 3; r++) { for (int cc = 0; cc < 3; cc++) M[4 * r + cc] = pos_rt_.R(r, cc); M[4 * r + 3] = pos_rt_.t[r]; }
}

// ---- context ---------------------------------------------------------------------------------
static msfm_ctx* g_ctx = nullptr;
static void destroy_ctx() { if (g_ctx) msfm_ctx_destroy(g_ctx); g_ctx = nullptr; }
msfm_ctx* Context() {
  if (!g_ctx) {
    if (msfm_ctx_create(-1, &g_ctx) != MSFM_OK) throw std::runtime_error("libmsfm: no usable MI355X (there is no CPU fallback)");
    std::atexit(destroy_ctx);
  }
  return g_ctx;
}
static void check(int rc, const char* what) {
  if (rc != MSFM_OK) throw std::runtime_error(std::string(what) + ": " + msfm_last_error(g_ctx));
}
// Several GPUs from the one process the reference is (test_sfm.cc:22-70 `main`): msfm_ctx_create_multi owns a context and a
// host thread per device and the communicator; the calls below that shard (matching, triangulation / reprojection, bundle
// adjustment) go through it, everything else through its rank-0 context.
static msfm_multi* g_multi = nullptr;
static void destroy_multi() { if (g_multi) msfm_multi_destroy(g_multi); g_multi = nullptr; g_ctx = nullptr; }
void UseGpus(int n_gpus, bool share_device_0) {
  if (g_multi || g_ctx) throw std::runtime_error("UseGpus: call before the first GPU call");
  if (n_gpus <= 1) return;
  std::vector<int> dev(n_gpus);
  for (int i = 0; i < n_gpus; i++) dev[i] = share_device_0 ? 0 : i;
  if (msfm_ctx_create_multi(n_gpus, dev.data(), &g_multi) != MSFM_OK) throw std::runtime_error("msfm_ctx_create_multi failed");
  g_ctx = msfm_multi_ctx(g_multi, 0);
  std::atexit(destroy_multi);
}
static void check_multi(int rc, const char* what) {
  if (rc != MSFM_OK) throw std::runtime_error(std::string(what) + ": " + msfm_multi_last_error(g_multi));
}

// ---- Point3D ------------------------------------------------------------------------------------
void Point3D::AddObservation(Camera* cam, double x, double y, int idx) {
  cams_.insert(std::make_pair(idx, cam));
  Vec2 p; p.x = x; p.y = y;
  pts2d_.insert(std::make_pair(idx, p));
}

namespace {
// Flatten a set of points into msfm_tracks (std::map key order, as every reference loop iterates).
struct Flat {
  std::vector<int32_t> off, cam;
  std::vector<double> xy, R, t, c, fk;
  std::map<Camera*, int> index;
  msfm_tracks tr;
  explicit Flat(const std::vector<Point3D*>& pts) {
    off.push_back(0);
    for (Point3D* p : pts) {
      auto ic = p->cams_.begin();
      auto ip = p->pts2d_.begin();
      for (; ic != p->cams_.end(); ++ic, ++ip) {
        auto f = index.find(ic->second);
        int id;
        if (f == index.end()) {
          id = (int)index.size();
          index[ic->second] = id;
          Camera* cm = ic->second;
          for (int k = 0; k < 9; k++) R.push_back(cm->pos_rt_.R.m[k]);
          for (int k = 0; k < 3; k++) { t.push_back(cm->pos_rt_.t[k]); c.push_back(cm->pos_ac_.c[k]); }
          fk.push_back(cm->cam_model_->f_); fk.push_back(cm->cam_model_->k1_); fk.push_back(cm->cam_model_->k2_);
        } else {
          id = f->second;
        }
        cam.push_back(id);
        xy.push_back(ip->second.x); xy.push_back(ip->second.y);
      }
      off.push_back((int32_t)cam.size());
    }
    if (index.empty()) { R.assign(9, 0); t.assign(3, 0); c.assign(3, 0); fk.assign(3, 1); }
    tr.n_tracks = (int)pts.size(); tr.n_cams = (int)std::max<size_t>(1, index.size());
    tr.track_off = off.data(); tr.track_cam = cam.data(); tr.track_xy = xy.data();
    tr.cam_R = R.data(); tr.cam_t = t.data(); tr.cam_c = c.data(); tr.cam_fk = fk.data();
  }
};
}  // namespace

void TrianglateBatch(const std::vector<Point3D*>& pts, double th_error, double th_angle, bool dlt, std::vector<char>* ok) {
  Flat F(pts);
  std::vector<double> X(3 * pts.size()), mse(pts.size());
  std::vector<uint8_t> okv(pts.size());
  for (size_t i = 0; i < pts.size(); i++) for (int k = 0; k < 3; k++) X[3 * i + k] = pts[i]->data[k];
  if (g_multi)
    check_multi(dlt ? msfm_multi_triangulate_dlt_batch(g_multi, &F.tr, th_error, th_angle, X.data(), mse.data(), okv.data())
                    : msfm_multi_triangulate_midpoint_batch(g_multi, &F.tr, th_error, th_angle, X.data(), mse.data(), okv.data()),
                "triangulate");
  else
    check(dlt ? msfm_triangulate_dlt_batch(Context(), &F.tr, th_error, th_angle, X.data(), mse.data(), okv.data())
              : msfm_triangulate_midpoint_batch(Context(), &F.tr, th_error, th_angle, X.data(), mse.data(), okv.data()),
          "triangulate");
  if (ok) ok->assign(pts.size(), 0);
  for (size_t i = 0; i < pts.size(); i++) {
    for (int k = 0; k < 3; k++) pts[i]->data[k] = X[3 * i + k];
    pts[i]->mse_ = mse[i];
    if (ok) (*ok)[i] = (char)okv[i];
  }
}

void ReprojectionBatch(const std::vector<Point3D*>& pts) {
  Flat F(pts);
  std::vector<double> X(3 * pts.size()), mse(pts.size());
  for (size_t i = 0; i < pts.size(); i++) for (int k = 0; k < 3; k++) X[3 * i + k] = pts[i]->data[k];
  if (g_multi) check_multi(msfm_multi_reproject_mse_batch(g_multi, &F.tr, X.data(), mse.data()), "reproject");
  else check(msfm_reproject_mse_batch(Context(), &F.tr, X.data(), mse.data()), "reproject");
  for (size_t i = 0; i < pts.size(); i++) pts[i]->mse_ = mse[i];
}

bool Point3D::Trianglate(double th_error, double th_angle) {
  std::vector<char> ok;
  TrianglateBatch(std::vector<Point3D*>(1, this), th_error, th_angle, true, &ok);
  return ok[0] != 0;
}
bool Point3D::Trianglate2(double th_error, double th_angle) {
  std::vector<char> ok;
  TrianglateBatch(std::vector<Point3D*>(1, this), th_error, th_angle, false, &ok);
  return ok[0] != 0;
}
void Point3D::Reprojection() { ReprojectionBatch(std::vector<Point3D*>(1, this)); }
bool Point3D::SufficientTriangulationAngle(double th) {
  // the acceptance test of the batch kernel with an unbounded error threshold isolates the angle gate
  Flat F(std::vector<Point3D*>(1, this));
  double X[3] = {data[0], data[1], data[2]}, mse = 0;
  uint8_t ok = 0;
  (void)X; (void)mse;
  std::vector<double> c(F.c);
  const int k = F.off[1];
  const double cos_min = std::cos(th);
  for (int i = 0; i + 1 < k; i++)
    for (int j = i + 1; j < k; j++) {
      double a[3], b[3], na = 0, nb = 0, d = 0;
      for (int q = 0; q < 3; q++) { a[q] = data[q] - c[3 * F.cam[i] + q]; b[q] = data[q] - c[3 * F.cam[j] + q]; na += a[q] * a[q]; nb += b[q] * b[q]; }
      for (int q = 0; q < 3; q++) d += a[q] / std::sqrt(na) * (b[q] / std::sqrt(nb));
      if (d < cos_min) ok = 1;
    }
  return ok != 0;
}

// ---- BundleAdjuster -----------------------------------------------------------------------------
BundleAdjuster::BundleAdjuster(std::vector<Camera*> cams, std::vector<CameraModel*> cam_models, std::vector<Point3D*> pts)
    : cams_(std::move(cams)), cam_models_(std::move(cam_models)), pts_(std::move(pts)) {
  msfm_ba_options_default(&options_);
  summary_ = msfm_ba_summary();
}

void BundleAdjuster::SetOptions(BundleAdjustOptions options) {
  options_.max_num_iterations = options.max_num_iterations;
  options_.progress_to_stdout = options.minimizer_progress_to_stdout ? 1 : 0;
  options_.num_threads = options.num_threads;  // linear solver: dense Schur, always (optimizer.cc:47)
}

void BundleAdjuster::RunOptimizetion(bool is_initial_run, double weight) {
  if (is_initial_run) { Normalize(); Perturb(); }
  // gather (optimizer.cc:59-129): points ascending, observations in std::map key order, bad points skipped
  std::map<Camera*, int> cam_id;
  std::map<CameraModel*, int> model_id;
  for (size_t i = 0; i < cams_.size(); i++) cam_id[cams_[i]] = (int)i;
  for (size_t i = 0; i < cam_models_.size(); i++) model_id[cam_models_[i]] = (int)i;
  std::vector<double> cam_pose(6 * cams_.size()), cam_model(3 * cam_models_.size()), point, obs_xy, pt_weight;
  std::vector<int32_t> model_of_cam(cams_.size()), obs_cam, obs_pt;
  std::vector<uint8_t> cam_mut(cams_.size()), model_mut(cam_models_.size()), pt_mut;
  std::vector<Point3D*> used;
  for (size_t i = 0; i < cams_.size(); i++) {
    for (int k = 0; k < 6; k++) cam_pose[6 * i + k] = cams_[i]->data[k];
    model_of_cam[i] = model_id.at(cams_[i]->cam_model_);
    cam_mut[i] = cams_[i]->is_mutable_;
  }
  for (size_t i = 0; i < cam_models_.size(); i++) {
    for (int k = 0; k < 3; k++) cam_model[3 * i + k] = cam_models_[i]->data[k];
    model_mut[i] = cam_models_[i]->is_mutable_;
  }
  for (Point3D* p : pts_) {
    if (p->is_bad_estimated_) continue;                 // optimizer.cc:64
    if (!keep_point_weights_) {
      if (p->cams_.size() == 2) p->weight = 1.0;        // optimizer.cc:69-78
      if (p->cams_.size() >= 3) p->weight = weight;
    }
    // a residual block exists only where the point or the camera is free (optimizer.cc:86-125): rows of a frozen point in
    // frozen cameras, and points left without any row, are not part of the problem (for the window of one camera out of
    // thousands that is nearly everything)
    const int pid = (int)used.size();
    bool any = false;
    auto ic = p->cams_.begin();
    auto ip = p->pts2d_.begin();
    for (; ic != p->cams_.end(); ++ic, ++ip) {
      if (!p->is_mutable_ && !ic->second->is_mutable_) continue;
      obs_cam.push_back(cam_id.at(ic->second));
      obs_pt.push_back(pid);
      obs_xy.push_back(ip->second.x); obs_xy.push_back(ip->second.y);
      any = true;
    }
    if (!any) continue;
    used.push_back(p);
    for (int k = 0; k < 3; k++) point.push_back(p->data[k]);
    pt_weight.push_back(p->weight);
    pt_mut.push_back(p->is_mutable_);
  }
  msfm_ba_problem P;
  P.n_cams = (int)cams_.size(); P.n_models = (int)cam_models_.size(); P.n_points = (int)used.size(); P.n_obs = (int)obs_cam.size();
  P.cam_pose = cam_pose.data(); P.cam_model = cam_model.data(); P.cam_model_of_cam = model_of_cam.data(); P.point = point.data();
  P.obs_cam = obs_cam.data(); P.obs_pt = obs_pt.data(); P.obs_xy = obs_xy.data(); P.pt_weight = pt_weight.data();
  P.cam_mutable = cam_mut.data(); P.model_mutable = model_mut.data(); P.pt_mutable = pt_mut.data();
  P.gps_xyz = nullptr; P.gps_weight = 0;
  std::vector<double> gps_xyz;
  if (!cams_gps_.empty()) {
    if (cams_gps_.size() != cams_.size()) throw std::runtime_error("SetGPS: one position per camera expected");
    // slam_gps.cc:818-830: `double weight = count1 / cams_.size();` - both int, so the division truncates; count1 is
    // the number of reprojection residual blocks added (rows whose camera and point are both frozen add none)
    long count1 = 0;
    for (size_t o = 0; o < obs_cam.size(); o++) count1 += (cam_mut[obs_cam[o]] || pt_mut[obs_pt[o]]) ? 1 : 0;
    gps_xyz.resize(3 * cams_.size());
    for (size_t i = 0; i < cams_.size(); i++) for (int k = 0; k < 3; k++) gps_xyz[3 * i + k] = cams_gps_[i][k];
    P.gps_xyz = gps_xyz.data();
    P.gps_weight = (double)(count1 / (long)cams_.size());
  }
  iterations_.assign((size_t)options_.max_num_iterations + 2, msfm_ba_iteration());
  summary_.iterations = iterations_.data();
  summary_.iterations_capacity = (int)iterations_.size();
  // == ceres::Solve, optimizer.cc:133 (several GPUs: the same call with the points split inside the library)
  if (g_multi) check_multi(msfm_multi_ba_solve(g_multi, &P, &options_, &summary_), "msfm_multi_ba_solve");
  else check(msfm_ba_solve(Context(), &P, &options_, &summary_), "msfm_ba_solve");
  // Ceres writes through the data blocks; so do we
  for (size_t i = 0; i < cams_.size(); i++) for (int k = 0; k < 6; k++) cams_[i]->data[k] = cam_pose[6 * i + k];
  for (size_t i = 0; i < cam_models_.size(); i++) for (int k = 0; k < 3; k++) cam_models_[i]->data[k] = cam_model[3 * i + k];
  for (size_t i = 0; i < used.size(); i++) for (int k = 0; k < 3; k++) used[i]->data[k] = point[3 * i + k];
}

void BundleAdjuster::UpdateParameters() {
  for (Camera* c : cams_) c->UpdatePoseFromData();
  for (CameraModel* m : cam_models_) m->UpdataModelFromData();
}

void BundleAdjuster::Normalize() {
  const int n = (int)pts_.size();
  double mid[3] = {0, 0, 0};
  for (Point3D* p : pts_) for (int k = 0; k < 3; k++) mid[k] += p->data[k];
  for (int k = 0; k < 3; k++) mid[k] /= n;
  double mad = 0;
  for (Point3D* p : pts_) mad += std::fabs(p->data[0] - mid[0]) + std::fabs(p->data[1] - mid[1]) + std::fabs(p->data[2] - mid[2]);
  mad /= n;
  const double scale = 100.0 / mad;
  for (Point3D* p : pts_) for (int k = 0; k < 3; k++) p->data[k] = scale * (p->data[k] - mid[k]);
  for (Camera* c : cams_) {
    Vec3 cc;
    for (int k = 0; k < 3; k++) cc[k] = scale * (c->pos_ac_.c[k] - mid[k]);
    c->SetACPose(c->pos_ac_.a, cc);
  }
}

void BundleAdjuster::Perturb() {
  std::mt19937_64 gen(perturb_seed_);
  std::normal_distribution<double> nrm(0.0, 1.0);
  const double rotation_sigma = 0.1, translation_sigma = 0.5, point_sigma = 0.5;
  for (Point3D* p : pts_) for (int k = 0; k < 3; k++) p->data[k] += nrm(gen) * point_sigma;
  for (Camera* c : cams_) {
    Vec3 a = c->pos_ac_.a;
    for (int k = 0; k < 3; k++) a[k] += nrm(gen) * rotation_sigma;
    c->SetACPose(a, c->pos_ac_.c);
    Vec3 t = c->pos_rt_.t;
    for (int k = 0; k < 3; k++) t[k] += nrm(gen) * translation_sigma;
    c->SetRTPose(c->pos_rt_.R, t);
  }
}

// ---- IncrementalSfM: window selection + the two adjustments ---------------------------------------
void IncrementalSfM::ImmutableCamsPoints() {
  for (Camera* c : cams_) {
    c->SetMutable(false);
    for (auto& kv : c->pts_) kv.second->SetMutable(false);
  }
}

void IncrementalSfM::MutableCamsPoints() {
  for (Camera* c : cams_) {
    c->SetMutable(true);
    for (auto& kv : c->pts_) kv.second->SetMutable(true);
  }
}

void IncrementalSfM::UpdateVisibleGraph(int idx_new_cam, std::vector<int> idxs_visible_cam) {
  cams_[idx_new_cam]->AddVisibleCamera(idx_new_cam);
  for (int v : idxs_visible_cam) {
    cams_[idx_new_cam]->AddVisibleCamera(v);
    cams_[v]->AddVisibleCamera(idx_new_cam);
  }
}

std::vector<int> IncrementalSfM::VisibleCameras(int idx_cam) const {
  // a 2D-3D match through camera j = a point of camera j (not bad) that the new camera observes too
  std::map<const Camera*, int> count;
  for (auto& kv : cams_[idx_cam]->pts_) {
    const Point3D* p = kv.second;
    if (p->is_bad_estimated_) continue;
    for (auto& pc : p->cams_) if (pc.second != cams_[idx_cam]) count[pc.second]++;
  }
  std::vector<int> vis;
  for (size_t j = 0; j < cams_.size(); j++) {
    auto it = count.find(cams_[j]);
    if (it != count.end() && it->second > options_.th_visible_matches) vis.push_back((int)j);
  }
  return vis;
}

void IncrementalSfM::PartialBundleAdjustment(int idx) {
  ImmutableCamsPoints();
  // optimize only the new camera (all cameras of its model) and its visible cameras, with their good points
  auto free_cam = [&](int idx_cam) {
    cams_[idx_cam]->SetMutable(true);
    for (auto& kv : cams_[idx_cam]->pts_) if (!kv.second->is_bad_estimated_) kv.second->SetMutable(true);
  };
  for (int idx_cam : cams_[idx]->cam_model_->idx_cams_) free_cam(idx_cam);
  for (int idx_cam : cams_[idx]->visible_cams_) free_cam(idx_cam);
  BundleAdjuster bundler(cams_, cam_models_, pts_);
  bundler.SetOptions(bundle_partial_options_);
  bundler.SetGPS(cams_gps_);
  bundler.RunOptimizetion(!found_seed_, 2.0);
  bundler.UpdateParameters();
  summary_ = bundler.summary_;
  iterations_ = bundler.iterations_;
  summary_.iterations = iterations_.data();
}

void IncrementalSfM::FullBundleAdjustment() {
  MutableCamsPoints();
  BundleAdjuster bundler(cams_, cam_models_, pts_);
  bundler.SetOptions(bundle_full_options_);
  bundler.SetGPS(cams_gps_);
  bundler.RunOptimizetion(!found_seed_, 1.0);
  bundler.UpdateParameters();
  summary_ = bundler.summary_;
  iterations_ = bundler.iterations_;
  summary_.iterations = iterations_.data();
}

void IncrementalSfM::RemovePointOutliers() {
  std::vector<Point3D*> live;
  for (Point3D* p : pts_) if (!p->is_bad_estimated_) live.push_back(p);
  ReprojectionBatch(live);   // pts_[i]->Reprojection() for every live point, one launch
  for (Point3D* p : live) {
    if (std::sqrt(p->mse_) > options_.th_mse_outliers) p->is_bad_estimated_ = true;
    p->is_new_added_ = false;
  }
}

void SLAMGPS::FullBundleAdjustment() {
  // slam_gps.cc:690-712 adds ReprojectionErrorPoseCamXYZ for every observation of every non-bad point with the point's
  // own weight: everything is free, and the weight rule of BundleAdjuster does not apply
  for (Camera* c : cams_) c->SetMutable(true);
  for (CameraModel* m : cam_models_) m->is_mutable_ = true;
  std::vector<double> keep;
  for (Point3D* p : pts_) { p->SetMutable(true); keep.push_back(p->weight); }
  BundleAdjuster bundler(cams_, cam_models_, pts_);
  BundleAdjustOptions o;
  o.max_num_iterations = 200; o.minimizer_progress_to_stdout = minimizer_progress_to_stdout_; o.num_threads = 8;   // slam_gps.cc:681-683
  bundler.SetOptions(o);
  bundler.SetGPS(cams_gps_);
  bundler.keep_point_weights_ = true;
  bundler.RunOptimizetion(false, 1.0);
  for (size_t i = 0; i < pts_.size(); i++) pts_[i]->weight = keep[i];
  bundler.UpdateParameters();   // slam_gps.cc:844-852
  summary_ = bundler.summary_;
  iterations_ = bundler.iterations_;
  summary_.iterations = iterations_.data();
}

// ---- matching -----------------------------------------------------------------------------------
std::vector<PairMatches> MatchImagePairs(const std::vector<std::vector<float>>& descriptors,
                                         const std::vector<std::pair<int, int>>& pairs, float thRatio_good, float thRatio_all) {
  if (g_multi) {
    // the pair list split over the contexts inside the library (fine_matching_graph.cc:87: the pairs are independent)
    const int n_img = (int)descriptors.size();
    std::vector<const float*> dp(n_img);
    std::vector<int> cnt(n_img), flat;
    for (int i = 0; i < n_img; i++) { dp[i] = descriptors[i].data(); cnt[i] = (int)(descriptors[i].size() / 128); }
    for (auto& p : pairs) { flat.push_back(p.first); flat.push_back(p.second); }
    std::vector<std::vector<int32_t>> codes(pairs.size());
    std::vector<int32_t*> cp(pairs.size());
    for (size_t p = 0; p < pairs.size(); p++) { codes[p].assign((size_t)std::max(1, cnt[pairs[p].second]), -1); cp[p] = codes[p].data(); }
    check_multi(msfm_multi_match_pairs(g_multi, n_img, dp.data(), cnt.data(), 128, flat.data(), (int)pairs.size(), thRatio_good, thRatio_all, cp.data(),
                                       nullptr, nullptr), "multi_match_pairs");
    std::vector<PairMatches> out(pairs.size());
    for (size_t p = 0; p < pairs.size(); p++) {
      out[p].idx1 = pairs[p].first; out[p].idx2 = pairs[p].second;
      for (int m = 0; m < cnt[pairs[p].second]; m++) {
        const int32_t code = codes[p][m];
        if (code < 0) continue;
        const int id1 = code & MSFM_MATCH_ID_MASK;
        if (code & MSFM_MATCH_GOOD) out[p].matches_good.push_back(std::make_pair(id1, m));
        if (!(code & MSFM_MATCH_NOT_ALL)) out[p].matches_all.push_back(std::make_pair(id1, m));
      }
    }
    return out;
  }
  msfm_descset* set = nullptr;
  check(msfm_descset_create(Context(), (int)descriptors.size(), 128, &set), "descset_create");
  for (size_t i = 0; i < descriptors.size(); i++)
    check(msfm_descset_upload(set, (int)i, descriptors[i].data(), (int)(descriptors[i].size() / 128)), "descset_upload");
  std::vector<int> flat;
  for (auto& p : pairs) { flat.push_back(p.first); flat.push_back(p.second); }
  msfm_match_result* res = nullptr;
  check(msfm_match_pairs(set, flat.data(), (int)pairs.size(), thRatio_good, thRatio_all, 0, &res), "match_pairs");
  std::vector<PairMatches> out(pairs.size());
  for (size_t p = 0; p < pairs.size(); p++) {
    out[p].idx1 = pairs[p].first; out[p].idx2 = pairs[p].second;
    const int n2 = (int)(descriptors[pairs[p].second].size() / 128);
    std::vector<int32_t> code((size_t)std::max(1, n2));
    check(msfm_match_result_fetch(res, (int)p, code.data(), nullptr, nullptr), "match_fetch");
    for (int m = 0; m < n2; m++) {  // the loop of fine_matching_graph.cc:116-133, decisions already made on the GPU
      if (code[m] < 0) continue;
      const int id1 = code[m] & MSFM_MATCH_ID_MASK;
      if (code[m] & MSFM_MATCH_GOOD) out[p].matches_good.push_back(std::make_pair(id1, m));
      if (!(code[m] & MSFM_MATCH_NOT_ALL)) out[p].matches_all.push_back(std::make_pair(id1, m));
    }
  }
  msfm_match_result_destroy(res);
  msfm_descset_destroy(set);
  return out;
}

// ---- geometric verification ----------------------------------------------------------------------
static void flatten(const std::vector<Point2f>& v, std::vector<float>& out) {
  out.resize(2 * v.size());
  for (size_t i = 0; i < v.size(); i++) { out[2 * i] = v[i].x; out[2 * i + 1] = v[i].y; }
}

bool GeoVerification::GeoVerificationFundamental(std::vector<Point2f>& pt1, std::vector<Point2f>& pt2, std::vector<int>& match_inliers,
                                                 Mat3& FMatrix) {
  if (pt1.size() < 30) return false;
  std::vector<float> a, b;
  flatten(pt1, a); flatten(pt2, b);
  const int off[2] = {0, (int)pt1.size()};
  msfm_fransac_options o;
  msfm_fransac_default_options(&o);
  std::vector<uint8_t> status(pt1.size());
  int nin = 0;
  uint8_t ok = 0;
  check(msfm_fundamental_ransac_batch(Context(), 1, off, a.data(), b.data(), &o, FMatrix.m, status.data(), &nin, &ok), "fundamental_ransac");
  for (size_t i = 0; i < status.size(); i++) if (status[i]) match_inliers.push_back((int)i);
  return match_inliers.size() >= 30;
}

bool GeoVerification::GeoVerificationFundamental(std::vector<Point2f>& pt1, std::vector<Point2f>& pt2, Mat3 FMatrix,
                                                 std::vector<int>& match_inliers) {
  match_inliers.clear();
  if (pt1.empty()) return true;
  std::vector<float> a, b;
  flatten(pt1, a); flatten(pt2, b);
  std::vector<uint8_t> in(pt1.size());
  check(msfm_epipolar_filter(Context(), a.data(), b.data(), (int)pt1.size(), FMatrix.m, 3.0, in.data()), "epipolar_filter");
  for (size_t i = 0; i < in.size(); i++) if (in[i]) match_inliers.push_back((int)i);
  return true;
}

// ---- pose initialisers ----
static const uint64_t kPoseSeed = 0x4D53464D50ull;

void AbsolutePoseBatch(const std::vector<std::vector<Vec3>>& pts_w, const std::vector<std::vector<Vec2>>& pts_2d, const std::vector<double>& f,
                       std::vector<RTPose>& poses, std::vector<std::vector<double>>& errors, std::vector<double>& avg_error) {
  const int n = (int)pts_w.size();
  std::vector<int> off(n + 1, 0);
  for (int p = 0; p < n; p++) off[p + 1] = off[p] + (int)pts_w[p].size();
  std::vector<double> X(3 * (size_t)std::max(1, off[n])), x(2 * (size_t)std::max(1, off[n])), R(9 * (size_t)std::max(1, n)),
      t(3 * (size_t)std::max(1, n)), err(std::max(1, off[n]));
  for (int p = 0; p < n; p++)
    for (size_t i = 0; i < pts_w[p].size(); i++) {
      const size_t e = off[p] + i;
      for (int k = 0; k < 3; k++) X[3 * e + k] = pts_w[p][i][k];
      x[2 * e] = pts_2d[p][i].x; x[2 * e + 1] = pts_2d[p][i].y;
    }
  avg_error.assign(std::max(1, n), 0.0);
  check(msfm_epnp_ransac_batch(Context(), n, off.data(), X.data(), x.data(), f.data(), 200, kPoseSeed, R.data(), t.data(), err.data(),
                               avg_error.data(), nullptr), "epnp_ransac_batch");
  avg_error.resize(n);
  poses.resize(n); errors.resize(n);
  for (int p = 0; p < n; p++) {
    for (int k = 0; k < 9; k++) poses[p].R.m[k] = R[9 * (size_t)p + k];
    for (int k = 0; k < 3; k++) poses[p].t[k] = t[3 * (size_t)p + k];
    errors[p].assign(err.begin() + off[p], err.begin() + off[p + 1]);
  }
}

bool AbsolutePoseEstimation::AbsolutePoseWithFocalLength(std::vector<Vec3>& pts_w, std::vector<Vec2>& pts_2d, double f, RTPose& pose_absolute,
                                                         std::vector<double>& errors, double& avg_error) {
  std::vector<RTPose> poses;
  std::vector<std::vector<double>> errs;
  std::vector<double> avg;
  AbsolutePoseBatch({pts_w}, {pts_2d}, {f}, poses, errs, avg);
  pose_absolute = poses[0];
  errors = errs[0];
  avg_error = avg[0];
  return true;  // the reference always returns true; the caller gates on avg_error (sfm_incremental.cc:648)
}

bool RelativePoseEstimation::RelativePoseWithFocalLength(std::vector<Vec2>& pts_ref, std::vector<Vec2>& pts_cur, double f_ref, double f_cur,
                                                         RTPoseRelative& pose_relative) {
  const int off[2] = {0, (int)pts_cur.size()};
  std::vector<double> a(2 * (size_t)std::max(1, off[1])), b(a.size());
  for (int i = 0; i < off[1]; i++) { a[2 * i] = pts_ref[i].x; a[2 * i + 1] = pts_ref[i].y; b[2 * i] = pts_cur[i].x; b[2 * i + 1] = pts_cur[i].y; }
  double E[9], R[9], t[3];
  uint8_t ok = 0;
  check(msfm_relpose_5pt_batch(Context(), 1, off, a.data(), b.data(), &f_ref, &f_cur, 100, kPoseSeed, E, R, t, &ok, nullptr), "relpose_5pt_batch");
  if (!ok) return false;
  for (int k = 0; k < 9; k++) pose_relative.R.m[k] = R[k];
  for (int k = 0; k < 3; k++) pose_relative.t[k] = t[k];
  return true;
}

std::vector<std::vector<std::pair<int, int>>> VerifyPairs(const std::vector<PairMatches>& matches,
                                                          const std::vector<std::vector<Point2f>>& keypoints) {
  const int np = (int)matches.size();
  std::vector<int> off_g(np + 1, 0), off_a(np + 1, 0);
  for (int p = 0; p < np; p++) {
    off_g[p + 1] = off_g[p] + (int)matches[p].matches_good.size();
    off_a[p + 1] = off_a[p] + (int)matches[p].matches_all.size();
  }
  std::vector<float> g1(2 * (size_t)off_g[np]), g2(g1.size()), a1(2 * (size_t)off_a[np]), a2(a1.size());
  for (int p = 0; p < np; p++) {
    const auto& k1 = keypoints[matches[p].idx1];
    const auto& k2 = keypoints[matches[p].idx2];
    size_t e = off_g[p];
    for (auto& m : matches[p].matches_good) { g1[2 * e] = k1[m.first].x; g1[2 * e + 1] = k1[m.first].y; g2[2 * e] = k2[m.second].x; g2[2 * e + 1] = k2[m.second].y; e++; }
    e = off_a[p];
    for (auto& m : matches[p].matches_all) { a1[2 * e] = k1[m.first].x; a1[2 * e + 1] = k1[m.first].y; a2[2 * e] = k2[m.second].x; a2[2 * e + 1] = k2[m.second].y; e++; }
  }
  msfm_fransac_options o;
  msfm_fransac_default_options(&o);
  std::vector<double> F(9 * (size_t)std::max(1, np));
  std::vector<uint8_t> in_g(std::max(1, off_g[np])), ok(std::max(1, np)), in_a(std::max(1, off_a[np]));
  std::vector<int> nin(std::max(1, np));
  check(msfm_fundamental_ransac_batch(Context(), np, off_g.data(), g1.data(), g2.data(), &o, F.data(), in_g.data(), nin.data(), ok.data()),
        "fundamental_ransac_batch");
  check(msfm_epipolar_filter_batch(Context(), np, off_a.data(), a1.data(), a2.data(), F.data(), ok.data(), 3.0, in_a.data()),
        "epipolar_filter_batch");
  std::vector<std::vector<std::pair<int, int>>> out(np);
  for (int p = 0; p < np; p++) {
    if (!ok[p]) continue;  // isOK == false: the pair writes nothing (fine_matching_graph.cc:182-186)
    for (int e = off_a[p]; e < off_a[p + 1]; e++)
      if (in_a[e]) out[p].push_back(matches[p].matches_all[e - off_a[p]]);
  }
  return out;
}

// ---- feature files ---------------------------------------------------------------------------------
bool WriteoutImageFeature(const std::string& fold, int idx, const ImageInfo& info, const std::vector<Point2f>& kp,
                          const std::vector<float>& desc, int desc_cols) {
  std::ofstream ofs(fold + "/" + std::to_string(idx) + "_feature", std::ios::binary);
  if (!ofs.is_open()) return false;
  ofs.write((const char*)&info.rows, sizeof(int));
  ofs.write((const char*)&info.cols, sizeof(int));
  for (const float* f : {&info.zoom_ratio, &info.f_mm, &info.f_pixel, &info.gps_latitude, &info.gps_longitude}) ofs.write((const char*)f, sizeof(float));
  for (const std::string* t : {&info.cam_maker, &info.cam_model}) {
    const int n = (int)t->length();
    ofs.write((const char*)&n, sizeof(int));
    ofs.write(t->data(), n);
  }
  const int num_pts = (int)kp.size();
  ofs.write((const char*)&num_pts, sizeof(int));
  std::vector<float> c(2 * (size_t)num_pts);
  for (int i = 0; i < num_pts; i++) {  // points are centralized (database.cc:522-527)
    c[2 * i] = (float)(kp[i].x - info.cols / 2.0);
    c[2 * i + 1] = (float)(kp[i].y - info.rows / 2.0);
  }
  ofs.write((const char*)c.data(), c.size() * sizeof(float));
  const int rows = desc_cols ? (int)(desc.size() / desc_cols) : 0, type = 5;  // CV_32FC1
  ofs.write((const char*)&rows, sizeof(int));
  ofs.write((const char*)&desc_cols, sizeof(int));
  ofs.write((const char*)&type, sizeof(int));
  ofs.write((const char*)desc.data(), desc.size() * sizeof(float));
  return true;
}

bool ReadinImageFeatures(const std::string& fold, int idx, ImageInfo& info, std::vector<Point2f>& kp, std::vector<float>& desc,
                         int& desc_cols) {
  std::ifstream ifs(fold + "/" + std::to_string(idx) + "_feature", std::ios::binary);
  if (!ifs.is_open()) return false;
  ifs.read((char*)&info.rows, sizeof(int));
  ifs.read((char*)&info.cols, sizeof(int));
  for (float* f : {&info.zoom_ratio, &info.f_mm, &info.f_pixel, &info.gps_latitude, &info.gps_longitude}) ifs.read((char*)f, sizeof(float));
  for (std::string* t : {&info.cam_maker, &info.cam_model}) {
    int n = 0;
    ifs.read((char*)&n, sizeof(int));
    t->assign((size_t)std::max(0, n), '\0');
    ifs.read(&(*t)[0], n);
  }
  int num_pts = 0;
  ifs.read((char*)&num_pts, sizeof(int));
  std::vector<float> c(2 * (size_t)std::max(0, num_pts));
  ifs.read((char*)c.data(), c.size() * sizeof(float));
  kp.resize(num_pts);
  for (int i = 0; i < num_pts; i++) { kp[i].x = c[2 * i]; kp[i].y = c[2 * i + 1]; }
  int rows = 0, type = 0;
  ifs.read((char*)&rows, sizeof(int));
  ifs.read((char*)&desc_cols, sizeof(int));
  ifs.read((char*)&type, sizeof(int));
  if (type != 5) return false;  // only CV_32FC1 descriptors (database.cc:412-418)
  desc.resize((size_t)rows * desc_cols);
  ifs.read((char*)desc.data(), desc.size() * sizeof(float));
  return (bool)ifs;
}

// ---- track building -------------------------------------------------------------------------------
std::vector<Point3D> BuildTracks(const std::string& fold, const std::vector<std::vector<int>>& match_graph, std::vector<Camera>& cams,
                                 const std::vector<std::vector<Vec2>>& keypoints) {
  const int n_img = (int)match_graph.size();
  std::vector<int> n_feat(n_img), pair_img, off{0}, flat;
  for (int i = 0; i < n_img; i++) n_feat[i] = (int)keypoints[i].size();
  for (int i = 0; i < n_img; i++) {
    std::vector<int> ids;
    std::vector<std::vector<std::pair<int, int>>> recs;
    QueryMatch(fold, i, ids, recs);  // the file of image i holds one record per matched image
    for (int j = 0; j < n_img; j++) {
      if (match_graph[i][j] <= 0) continue;  // slam_gps.cc:571-575
      for (size_t r = 0; r < ids.size(); r++) {
        if (ids[r] != j) continue;
        pair_img.push_back(i); pair_img.push_back(j);
        for (auto& m : recs[r]) { flat.push_back(m.first); flat.push_back(m.second); }
        off.push_back((int)flat.size() / 2);
        break;
      }
    }
  }
  msfm_track_set* set = nullptr;
  // the whole image set at once: the GPU form of the walk (identical result, tests/test_gpu_tracks.py)
  check(msfm_tracks_build_device(Context(), n_img, n_feat.data(), (int)pair_img.size() / 2, pair_img.data(), off.data(), flat.data(), &set), "tracks_build");
  int nt = 0, no = 0;
  msfm_track_set_size(set, &nt, &no);
  std::vector<int> toff(nt + 1), oi(std::max(1, no)), of(std::max(1, no));
  msfm_track_set_fetch(set, toff.data(), oi.data(), of.data());
  msfm_track_set_destroy(set);
  std::vector<Point3D> pts(nt);
  for (int t = 0; t < nt; t++)
    for (int e = toff[t]; e < toff[t + 1]; e++)
      pts[t].AddObservation(&cams[oi[e]], keypoints[oi[e]][of[e]].x, keypoints[oi[e]][of[e]].y, oi[e]);
  return pts;
}

// ---- match files ---------------------------------------------------------------------------------
void WriteOutMatches(const std::string& fold, int idx1, int idx2, const std::vector<std::pair<int, int>>& matches) {
  const int num_match = (int)matches.size();
  if (!num_match) return;
  std::vector<int> tmp(2 * (size_t)num_match);
  for (int m = 0; m < num_match; m++) { tmp[2 * m] = matches[m].first; tmp[2 * m + 1] = matches[m].second; }
  std::ofstream ofs(fold + "/" + std::to_string(idx1) + "_match", std::ios::out | std::ios::app | std::ios::binary);
  ofs.write((const char*)&idx2, sizeof(int));
  ofs.write((const char*)&num_match, sizeof(int));
  ofs.write((const char*)tmp.data(), tmp.size() * sizeof(int));
}

void WriteOutMatchGraph(const std::string& fold, const std::vector<std::vector<int>>& match_graph) {
  std::ofstream ofs(fold + "/graph_matching.txt", std::ios::binary);
  for (auto& row : match_graph) {
    for (int v : row) ofs << v << " ";
    ofs << std::endl;
  }
}

void QueryMatch(const std::string& fold, int idx, std::vector<int>& image_ids,
                std::vector<std::vector<std::pair<int, int>>>& match_pts) {
  std::ifstream ifs(fold + "/" + std::to_string(idx) + "_match", std::ios::in | std::ios::binary);
  int id, num_match;
  while (ifs.read((char*)&id, sizeof(int))) {
    ifs.read((char*)&num_match, sizeof(int));
    std::vector<int> tmp(2 * (size_t)num_match);
    ifs.read((char*)tmp.data(), tmp.size() * sizeof(int));
    image_ids.push_back(id);
    std::vector<std::pair<int, int>> mt(num_match);
    for (int i = 0; i < num_match; i++) mt[i] = std::make_pair(tmp[2 * i], tmp[2 * i + 1]);
    match_pts.push_back(mt);
  }
}

}  // namespace objectsfm
