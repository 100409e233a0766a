// C++ host side above the C ABI: the reference's camera / point / bundle-adjuster classes
// (same names, members and method spellings, typos included) re-implemented without Eigen, Ceres,
// OpenCV or FLANN, so that a maintainer can see exactly where libmsfm plugs in:
//   CameraModel         SfM/src/basic_structs.h:48-124
//   Camera              SfM/src/camera.h:34-85, camera.cc:43-137
//   Point3D             SfM/src/structure.h:29-72, structure.cc:163-355
//   BundleAdjuster      SfM/src/optimizer.h, optimizer.cc:31-232
//   FineMatchingGraph   SfM/src/graph/fine_matching_graph.cc:40-194 (kNN + ratio tests part)
// Everything numeric on the hot path goes through include/msfm.h; there is no CPU fallback.
#pragma once
#include <array>
#include <cmath>
#include <map>
#include <random>
#include <string>
#include <utility>
#include <vector>

#include "../include/msfm.h"

namespace objectsfm {

struct Vec2 { double x = 0, y = 0; double operator()(int i) const { return i ? y : x; } };
struct Vec3 {
  double v[3] = {0, 0, 0};
  double& operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
  double& operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
};
struct Mat3 {
  double m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // row-major
  double& operator()(int r, int c) { return m[3 * r + c]; }
  double operator()(int r, int c) const { return m[3 * r + c]; }
};
inline Vec3 operator*(const Mat3& A, const Vec3& x) {
  Vec3 y;
  for (int r = 0; r < 3; r++) y[r] = A(r, 0) * x[0] + A(r, 1) * x[1] + A(r, 2) * x[2];
  return y;
}
inline Mat3 transpose(const Mat3& A) { Mat3 T; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) T(r, c) = A(c, r); return T; }

namespace rotation {  // SfM/src/utils/basic_funcs.cc:25-158
void AngleAxisToRotationMatrix(const Vec3& angle_axis, Mat3& R);
void RotationMatrixToAngleAxis(const Mat3& R, Vec3& axis);
}  // namespace rotation

struct RTPose { Mat3 R; Vec3 t; };  // basic_structs.h:126-145
struct ACPose { Vec3 a, c; };

struct BundleAdjustOptions {  // basic_structs.h:229-235
  int max_num_iterations = 200;
  bool minimizer_progress_to_stdout = true;
  int num_threads = 1;
};

struct CameraModel {  // basic_structs.h:48-124
  CameraModel() {}
  CameraModel(int id, int h, int w, double f_mm, double f, std::string cam_maker, std::string cam_model);
  void SetFocalLength(double f) { f_ = f; UpdateDataFromModel(); }
  void UpdateDataFromModel() { data[0] = f_; data[1] = k1_; data[2] = k2_; data[3] = dcx_; data[4] = dcy_; }
  void UpdataModelFromData() { f_ = data[0]; k1_ = data[1]; k2_ = data[2]; dcx_ = data[3]; dcy_ = data[4]; px_ += dcx_; py_ += dcy_; }
  void AddCamera(int idx) { idx_cams_.push_back(idx); num_cams_++; }
  void SetImmutable() { is_mutable_ = false; }
  int id_ = 0;
  std::string cam_maker_, cam_model_;
  int w_ = 0, h_ = 0;
  double f_mm_ = 0, f_ = 0, f_hyp_ = 0, px_ = 0, py_ = 0;
  double k1_ = 0, k2_ = 0, dcx_ = 0, dcy_ = 0;
  double data[5] = {0, 0, 0, 0, 0};  // {f, k1, k2, dcx, dcy}; BA optimises the first three (optimizer.cc:90-92)
  int num_cams_ = 0;
  std::vector<int> idx_cams_;
  bool is_mutable_ = true;
};

class Point3D;
class Camera {  // camera.h:34-85
 public:
  void AssociateImage(int id_img) { id_img_ = id_img; }
  void AssociateCamereModel(CameraModel* cam_model) { cam_model_ = cam_model; }
  void SetRTPose(const Mat3& R, const Vec3& t);   // camera.cc:43-54
  void SetACPose(const Vec3& a, const Vec3& c);   // camera.cc:69-80
  void UpdateDataFromPose();                      // camera.cc:89-111
  void UpdatePoseFromData();                      // camera.cc:113-137
  void SetMutable(bool is_mutable) { is_mutable_ = is_mutable; }
  void AddPoints(Point3D* pt, int idx) { pts_.insert(std::make_pair(idx, pt)); }     // camera.cc:152-155
  void AddVisibleCamera(int id_visible_cam) { visible_cams_.push_back(id_visible_cam); }  // camera.cc:157-160
  void SetID(int id) { id_ = id; }
  int id_ = 0;
  int id_img_ = 0;
  std::map<int, Point3D*> pts_;      // global feature id -> 3-D point (camera.h:81)
  std::vector<int> visible_cams_;    // camera.h:82
  CameraModel* cam_model_ = nullptr;
  RTPose pos_rt_;
  ACPose pos_ac_;
  double data[6] = {0, 0, 0, 0, 0, 0};  // angle-axis, t
  double M[12] = {0};                   // [R|t] row-major 3x4
  bool is_mutable_ = true;
};

class Point3D {  // structure.h:29-72
 public:
  void AddObservation(Camera* cam, double x, double y, int idx);  // structure.cc:128-137
  bool Trianglate(double th_error, double th_angle);   // DLT, structure.cc:163-209
  bool Trianglate2(double th_error, double th_angle);  // ray midpoint, structure.cc:211-265
  void Reprojection();                                 // structure.cc:267-300
  bool SufficientTriangulationAngle(double th_angle_triangulation);  // structure.cc:325-355 (through the batch kernel)
  void SetMutable(bool is_mutable) { is_mutable_ = is_mutable; }
  int id_ = 0;
  double data[3] = {0, 0, 0};
  std::map<int, Camera*> cams_;
  std::map<int, Vec2> pts2d_;
  double weight = 1.0, mse_ = 0.0;
  bool is_mutable_ = true, is_bad_estimated_ = false, is_new_added_ = true;
};

// One GPU context per process; created on first use, destroyed at exit.
msfm_ctx* Context();
// `n_gpus` contexts in this one process (msfm_ctx_create_multi); share_device_0: all of them on device 0 (a one-GPU box).
// Call before the first GPU call.  Matching, triangulation / reprojection and the bundle adjustment then split inside the library.
void UseGpus(int n_gpus, bool share_device_0 = false);

// Batched forms the pipeline should prefer (IncrementalSfM::GenerateNew3DPoints /
// RemovePointOutliers, sfm_incremental.cc:755-915,1831-1863): one kernel launch for all points.
void TrianglateBatch(const std::vector<Point3D*>& pts, double th_error, double th_angle, bool dlt, std::vector<char>* ok);
void ReprojectionBatch(const std::vector<Point3D*>& pts);

class BundleAdjuster {  // optimizer.h / optimizer.cc:31-232
 public:
  BundleAdjuster(std::vector<Camera*> cams, std::vector<CameraModel*> cam_models, std::vector<Point3D*> pts);
  void SetOptions(BundleAdjustOptions options);                 // optimizer.cc:42-48
  void RunOptimizetion(bool is_initial_run, double weight);     // optimizer.cc:50-135 -> msfm_ba_solve
  void UpdateParameters();                                      // optimizer.cc:142-153
  void Normalize();                                             // optimizer.cc:155-195
  void Perturb();                                               // optimizer.cc:197-232 (seeded std::mt19937_64, not std::rand)
  // The absolute GPS rows SLAMGPS::FullBundleAdjustment adds after the reprojection rows (slam_gps.cc:714-832,
  // use_absolute_gps): one GPSErrorPoseAbsolute per camera on pose[3:6], Huber(1), weight = count1 / cams_.size()
  // (integer division, :824) with count1 = reprojection residual blocks added.  Empty = none.
  void SetGPS(const std::vector<Vec3>& cams_gps) { cams_gps_ = cams_gps; }
  msfm_ba_summary summary_;
  std::vector<msfm_ba_iteration> iterations_;
  unsigned long long perturb_seed_ = 0x4D53464DULL;
  bool keep_point_weights_ = false;   // SLAMGPS passes pts_[i]->weight as it is (slam_gps.cc:703)

 private:
  std::vector<Camera*> cams_;
  std::vector<CameraModel*> cam_models_;
  std::vector<Point3D*> pts_;
  std::vector<Vec3> cams_gps_;
  msfm_ba_options options_;
};

// The bundle-adjustment side of the incremental loop (SfM/src/sfm_incremental.h/.cc): which cameras and points a
// partial adjustment frees, the full adjustment, the outlier sweep.  Localisation, seed search and file handling stay
// with their own stages (pose initialisers / matching above).
struct IncrementalSfMOptions {       // basic_structs.h:147-227, the fields this part reads
  double th_mse_outliers = 3.0;      // test_sfm.cc:46 (UAV), 1.0 for WEB (:57)
  int th_visible_matches = 5;        // `count_2d3d_ij > 5`, sfm_incremental.cc:503
  bool use_same_camera = false;      // basic_structs.h:167
};
class IncrementalSfM {
 public:
  void ImmutableCamsPoints();                                            // sfm_incremental.cc:1865-1878
  void MutableCamsPoints();                                              // sfm_incremental.cc:1880-1893
  void UpdateVisibleGraph(int idx_new_cam, std::vector<int> idxs_visible_cam);  // sfm_incremental.cc:1895-1903
  // the counting loop of FindImageToLocalize (sfm_incremental.cc:455-506) for a camera whose points are attached:
  // cameras through which it has more than th_visible_matches 2D-3D matches (non-bad points), ascending
  std::vector<int> VisibleCameras(int idx_cam) const;
  void PartialBundleAdjustment(int idx);                                 // sfm_incremental.cc:917-1014
  void FullBundleAdjustment();                                           // sfm_incremental.cc:1016-1026
  void RemovePointOutliers();                                            // sfm_incremental.cc:1831-1863 (one batched reprojection)
  std::vector<Camera*> cams_;
  std::vector<CameraModel*> cam_models_;
  std::vector<Point3D*> pts_;
  // BASELINE config 5 ("incremental-window BA with GCP constraints"): when set, both adjustments attach the
  // absolute GPS rows of SLAMGPS::FullBundleAdjustment (slam_gps.cc:818-830) to the cameras they free
  std::vector<Vec3> cams_gps_;
  IncrementalSfMOptions options_;
  BundleAdjustOptions bundle_full_options_, bundle_partial_options_;
  bool found_seed_ = true;
  msfm_ba_summary summary_;                  // of the last adjustment
  std::vector<msfm_ba_iteration> iterations_;
};

// SLAMGPS::FullBundleAdjustment (SfM/src/slam_gps.cc:675-863): every non-bad point's observations as
// ReprojectionErrorPoseCamXYZ rows with the point's weight, the absolute GPS rows, max 200 iterations, 8 threads.
class SLAMGPS {
 public:
  void FullBundleAdjustment();
  std::vector<Camera*> cams_;
  std::vector<CameraModel*> cam_models_;
  std::vector<Point3D*> pts_;
  std::vector<Vec3> cams_gps_;               // cv::Point3d cams_gps_ (slam_gps.h)
  bool minimizer_progress_to_stdout_ = true; // slam_gps.cc:682
  msfm_ba_summary summary_;
  std::vector<msfm_ba_iteration> iterations_;
};

// The kNN + ratio-test part of FineMatchingGraph::BuildMatchGraph (fine_matching_graph.cc:87-133),
// one batched call for a whole pair list.  matches_good/all[p] = (ptid1 in idx1, ptid2 in idx2).
struct PairMatches { int idx1, idx2; std::vector<std::pair<int, int>> matches_good, matches_all; };
std::vector<PairMatches> MatchImagePairs(const std::vector<std::vector<float>>& descriptors /*[image][n*128]*/,
                                         const std::vector<std::pair<int, int>>& pairs, float thRatio_good = 0.6f,
                                         float thRatio_all = 0.85f);

// Geometric verification (SfM/src/utils/geo_verification.h/.cc:30-79).  cv::Point2f / cv::Mat become Point2f / Mat3.
struct Point2f { float x = 0, y = 0; };
class GeoVerification {
 public:
  // cv::findFundamentalMat(FM_RANSAC, 3.0) + the 30-point / 30-inlier gates            (geo_verification.cc:30-58)
  static bool GeoVerificationFundamental(std::vector<Point2f>& pt1, std::vector<Point2f>& pt2, std::vector<int>& match_inliers,
                                         Mat3& FMatrix);
  // closed-form filter of a second match set with a given F                            (geo_verification.cc:60-79)
  static bool GeoVerificationFundamental(std::vector<Point2f>& pt1, std::vector<Point2f>& pt2, Mat3 FMatrix,
                                         std::vector<int>& match_inliers);
};
// The verification half of FineMatchingGraph::BuildMatchGraph (fine_matching_graph.cc:138-187) for every pair in two
// batched calls: RANSAC on the "good" matches, then the F filter on the "all" matches of the pairs that passed.
// keypoints[image][feature] are the centred pixel coordinates of database.cc:522-527.
// Returns per pair the surviving matches_all entries (empty when the pair failed: nothing is written for it).
std::vector<std::vector<std::pair<int, int>>> VerifyPairs(const std::vector<PairMatches>& matches,
                                                          const std::vector<std::vector<Point2f>>& keypoints);

// Pose initialisers (SfM/src/orientation/{absolute,relative}_pose_estimation.h).  Eigen::Vector3d / Vector2d / Matrix3d
// become Vec3 / Vec2 / Mat3; RTPoseRelative has the fields of RTPose (basic_structs.h:126-138).
typedef RTPose RTPoseRelative;
class AbsolutePoseEstimation {
 public:
  // EPnP RANSAC (200 samples of 4 correspondences) + per-point reprojection errors   (absolute_pose_estimation.cc:42-58)
  static bool AbsolutePoseWithFocalLength(std::vector<Vec3>& pts_w, std::vector<Vec2>& pts_2d, double f, RTPose& pose_absolute,
                                          std::vector<double>& errors, double& avg_error);
};
class RelativePoseEstimation {
 public:
  // five-point RANSAC on points / f, then the pose from the best essential matrix     (relative_pose_estimation.cc:91-120)
  static bool RelativePoseWithFocalLength(std::vector<Vec2>& pts_ref, std::vector<Vec2>& pts_cur, double f_ref, double f_cur,
                                          RTPoseRelative& pose_relative);
};
// Many images / pairs in one call each (the batched form the GPU wants; same results as the per-item calls above).
void AbsolutePoseBatch(const std::vector<std::vector<Vec3>>& pts_w, const std::vector<std::vector<Vec2>>& pts_2d, const std::vector<double>& f,
                       std::vector<RTPose>& poses, std::vector<std::vector<double>>& errors, std::vector<double>& avg_error);

// The per-image feature file of the extraction stage (Database::WriteoutImageFeature / ReadinImageFeatures,
// SfM/src/database.cc:490-541, :352-423): header, centred keypoints, raw descriptors.  cv::Mat -> flat float rows.
struct ImageInfo {  // basic_structs.h ImageInfo
  int rows = 0, cols = 0;
  float zoom_ratio = 1.f, f_mm = 0.f, f_pixel = 0.f, gps_latitude = 0.f, gps_longitude = 0.f;
  std::string cam_maker, cam_model;
};
bool WriteoutImageFeature(const std::string& output_fold, int idx, const ImageInfo& info, const std::vector<Point2f>& keypoints_px,
                          const std::vector<float>& descriptors /*[n][cols]*/, int desc_cols = 128);
bool ReadinImageFeatures(const std::string& output_fold, int idx, ImageInfo& info, std::vector<Point2f>& keypoints_centred,
                         std::vector<float>& descriptors, int& desc_cols);

// Track building, the data association of SLAMGPS::Triangulation (slam_gps.cc:565-635): walk the match graph in the
// reference's order (idx1 ascending, idx2 ascending over match_graph[idx1][idx2] > 0, matches read back with
// QueryMatch) and grow points greedily - msfm_tracks_build_device keeps the std::map::insert semantics.  Returns the new
// points with their observations attached (AddObservation(cam, x, y, image id), as :600-603 keys them).
std::vector<Point3D> BuildTracks(const std::string& output_fold, const std::vector<std::vector<int>>& match_graph,
                                 std::vector<Camera>& cams, const std::vector<std::vector<Vec2>>& keypoints);

// The reference's stage boundary is a set of files: per image `<idx1>_match` (binary records
// int idx2, int n, int[2n]) and `graph_matching.txt`.  Same bytes as FineMatchingGraph::WriteOutMatches /
// WriteOutMatchGraph (fine_matching_graph.cc:247-292) and Graph::QueryMatch (graph.cc:92-137).
void WriteOutMatches(const std::string& output_fold, int idx1, int idx2, const std::vector<std::pair<int, int>>& matches);
void WriteOutMatchGraph(const std::string& output_fold, const std::vector<std::vector<int>>& match_graph);
void QueryMatch(const std::string& output_fold, int idx, std::vector<int>& image_ids,
                std::vector<std::vector<std::pair<int, int>>>& match_pts);

}  // namespace objectsfm
