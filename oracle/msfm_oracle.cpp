// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
//
// A single-threaded CPU restatement of the arithmetic on MetricSfM's matching +
// bundle-adjustment hot path, written from scratch for this repo.  It
// exists to check the HIP kernels and to be timed as the `cpu_baseline` leg of
// bench.py; nothing under metricsfm_amd/ may import, link or call it.
//
// "Parity unpinned": the reference ships no golden vectors, known-answer tests or
// fixtures and cannot be built here (Ceres 1.13, Eigen3, OpenCV 2.4
// and FLANN are absent; the tree is MSVC-only), so this restatement is pinned only by
// (a) analytic known-answer cases, (b) a forward-mode dual-number evaluation of the
// same residual code (what Ceres' AutoDiffCostFunction does with Jets) and (c) an
// independent numpy/scipy dense Levenberg–Marquardt in tests/ — not by outputs of the
// reference itself.
//
// Each function cites the reference lines it follows.  Solver semantics that live in
// un-vendored Ceres Solver 1.13 (README.md:12, SfM/CMakeLists.txt:52) are restated from
// its published algorithm (trust_region_minimizer.cc, levenberg_marquardt_strategy.cc,
// corrector.cc, loss_function.cc, schur_eliminator_impl.h of the 1.13 release) and
// anchored on the call sites SfM/src/optimizer.cc:42-48,84,133 and
// SfM/src/slam_gps.cc:681-684,822,841.
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <array>
#include <vector>

#include "../include/msfm.h"
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API extern "C" __attribute__((visibility("default")))

// Host threads for the timing legs of bench.py (SURVEY.md 8d "(b) all host cores": the reference runs its kNN loop under
// OpenMP, fine_matching_graph.cc:87, and SLAMGPS hands Ceres num_threads = 8, slam_gps.cc:683).  Threads only ever split
// work whose results do not depend on the split: every sum keeps the order of the single-thread code, so a run with N
// threads is bit-identical to the run with one (tests/test_oracle.py).
static int g_orc_threads = 1;
ORC_API void orc_set_num_threads(int n) { g_orc_threads = n < 1 ? 1 : n; }
ORC_API int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

// -------------------------------------------------------------------------------------
// Rotation helpers — SfM/src/utils/basic_funcs.cc:25-225 (in-tree copies of the Ceres
// rotation.h formulas, incl. the small-angle branch at theta2 <= DBL_EPSILON :122,165).
// -------------------------------------------------------------------------------------

// basic_funcs.cc:118-158.  R row-major.
ORC_API void orc_angle_axis_to_R(const double* aa, double* R) {
  const double theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (theta2 > std::numeric_limits<double>::epsilon()) {
    const double theta = std::sqrt(theta2);
    const double wx = aa[0] / theta, wy = aa[1] / theta, wz = aa[2] / theta;
    const double c = std::cos(theta), s = std::sin(theta);
    R[0] = c + wx * wx * (1.0 - c);
    R[3] = wz * s + wx * wy * (1.0 - c);
    R[6] = -wy * s + wx * wz * (1.0 - c);
    R[1] = wx * wy * (1.0 - c) - wz * s;
    R[4] = c + wy * wy * (1.0 - c);
    R[7] = wx * s + wy * wz * (1.0 - c);
    R[2] = wy * s + wx * wz * (1.0 - c);
    R[5] = -wx * s + wy * wz * (1.0 - c);
    R[8] = c + wz * wz * (1.0 - c);
  } else {
    R[0] = 1.0; R[3] = aa[2]; R[6] = -aa[1];
    R[1] = -aa[2]; R[4] = 1.0; R[7] = aa[0];
    R[2] = aa[1]; R[5] = -aa[0]; R[8] = 1.0;
  }
}

// basic_funcs.cc:25-59 (Shoemake) then :61-107.
ORC_API void orc_R_to_angle_axis(const double* R, double* aa) {
  double q[4];
  const double trace = R[0] + R[4] + R[8];
  if (trace >= 0.0) {
    double t = std::sqrt(trace + 1.0);
    q[0] = 0.5 * t;
    t = 0.5 / t;
    q[1] = (R[7] - R[5]) * t;
    q[2] = (R[2] - R[6]) * t;
    q[3] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 3 + i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = std::sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
    q[i + 1] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[k * 3 + j] - R[j * 3 + k]) * t;
    q[j + 1] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    q[k + 1] = (R[k * 3 + i] + R[i * 3 + k]) * t;
  }
  const double s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (s2 > 0.0) {
    const double s = std::sqrt(s2), c = q[0];
    const double two_theta = 2.0 * ((c < 0.0) ? std::atan2(-s, -c) : std::atan2(s, c));
    const double k = two_theta / s;
    aa[0] = q[1] * k; aa[1] = q[2] * k; aa[2] = q[3] * k;
  } else {
    aa[0] = q[1] * 2.0; aa[1] = q[2] * 2.0; aa[2] = q[3] * 2.0;
  }
}

// A tiny forward-mode dual number with N partials: the moral equivalent of ceres::Jet.
template <int N>
struct Dual {
  double a;
  double v[N];
  Dual() : a(0) { for (int i = 0; i < N; i++) v[i] = 0; }
  Dual(double x) : a(x) { for (int i = 0; i < N; i++) v[i] = 0; }
};
template <int N> Dual<N> operator+(const Dual<N>& x, const Dual<N>& y) { Dual<N> r; r.a = x.a + y.a; for (int i = 0; i < N; i++) r.v[i] = x.v[i] + y.v[i]; return r; }
template <int N> Dual<N> operator-(const Dual<N>& x, const Dual<N>& y) { Dual<N> r; r.a = x.a - y.a; for (int i = 0; i < N; i++) r.v[i] = x.v[i] - y.v[i]; return r; }
template <int N> Dual<N> operator-(const Dual<N>& x) { Dual<N> r; r.a = -x.a; for (int i = 0; i < N; i++) r.v[i] = -x.v[i]; return r; }
template <int N> Dual<N> operator*(const Dual<N>& x, const Dual<N>& y) { Dual<N> r; r.a = x.a * y.a; for (int i = 0; i < N; i++) r.v[i] = x.a * y.v[i] + x.v[i] * y.a; return r; }
template <int N> Dual<N> operator/(const Dual<N>& x, const Dual<N>& y) { Dual<N> r; const double inv = 1.0 / y.a; r.a = x.a * inv; for (int i = 0; i < N; i++) r.v[i] = (x.v[i] - r.a * y.v[i]) * inv; return r; }
template <int N> Dual<N> operator*(double s, const Dual<N>& y) { return Dual<N>(s) * y; }
template <int N> Dual<N> operator+(double s, const Dual<N>& y) { return Dual<N>(s) + y; }
template <int N> Dual<N> operator-(double s, const Dual<N>& y) { return Dual<N>(s) - y; }
template <int N> bool operator>(const Dual<N>& x, double y) { return x.a > y; }
template <int N> Dual<N> dsqrt(const Dual<N>& x) { Dual<N> r; r.a = std::sqrt(x.a); const double k = 0.5 / r.a; for (int i = 0; i < N; i++) r.v[i] = x.v[i] * k; return r; }
template <int N> Dual<N> dcos(const Dual<N>& x) { Dual<N> r; r.a = std::cos(x.a); const double k = -std::sin(x.a); for (int i = 0; i < N; i++) r.v[i] = x.v[i] * k; return r; }
template <int N> Dual<N> dsin(const Dual<N>& x) { Dual<N> r; r.a = std::sin(x.a); const double k = std::cos(x.a); for (int i = 0; i < N; i++) r.v[i] = x.v[i] * k; return r; }
static inline double dsqrt(double x) { return std::sqrt(x); }
static inline double dcos(double x) { return std::cos(x); }
static inline double dsin(double x) { return std::sin(x); }

// basic_funcs.cc:160-225 == ceres::AngleAxisRotatePoint (called at
// reprojection_error_pose_cam_xyz.h:41).  Templated so doubles and Duals share the code.
template <typename T>
static void angle_axis_rotate_point(const T* aa, const T* pt, T* result) {
  const T theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (theta2 > std::numeric_limits<double>::epsilon()) {
    const T theta = dsqrt(theta2);
    const T costheta = dcos(theta);
    const T sintheta = dsin(theta);
    const T theta_inverse = T(1.0) / theta;
    const T w[3] = {aa[0] * theta_inverse, aa[1] * theta_inverse, aa[2] * theta_inverse};
    const T w_cross_pt[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2],
                             w[0] * pt[1] - w[1] * pt[0]};
    const T tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (T(1.0) - costheta);
    result[0] = pt[0] * costheta + w_cross_pt[0] * sintheta + w[0] * tmp;
    result[1] = pt[1] * costheta + w_cross_pt[1] * sintheta + w[1] * tmp;
    result[2] = pt[2] * costheta + w_cross_pt[2] * sintheta + w[2] * tmp;
  } else {
    const T w_cross_pt[3] = {aa[1] * pt[2] - aa[2] * pt[1], aa[2] * pt[0] - aa[0] * pt[2],
                             aa[0] * pt[1] - aa[1] * pt[0]};
    result[0] = pt[0] + w_cross_pt[0];
    result[1] = pt[1] + w_cross_pt[1];
    result[2] = pt[2] + w_cross_pt[2];
  }
}

ORC_API void orc_angle_axis_rotate_point(const double* aa, const double* pt, double* out) {
  angle_axis_rotate_point<double>(aa, pt, out);
}

// The one projection model shared by all five functors
// (reprojection_error_pose_cam_xyz.h:33-70; _xyz.h:31-73; _pose_cam.h:32-70;
//  _pose_xyz.h:32-68; _pose.h:31-68): +z forward, no sign flip, radial distortion.
template <typename T>
static void reproj_functor(const T* pose, const T* cam, const T* xyz, double ox, double oy,
                           double weight, T* residuals) {
  T p[3];
  angle_axis_rotate_point<T>(pose, xyz, p);
  p[0] = p[0] + pose[3];
  p[1] = p[1] + pose[4];
  p[2] = p[2] + pose[5];
  const T xp = p[0] / p[2];
  const T yp = p[1] / p[2];
  const T& focal = cam[0];
  const T& l1 = cam[1];
  const T& l2 = cam[2];
  const T r2 = xp * xp + yp * yp;
  const T distortion = 1.0 + r2 * (l1 + l2 * r2);
  const T predicted_x = focal * distortion * xp;
  const T predicted_y = focal * distortion * yp;
  residuals[0] = weight * (predicted_x - T(ox));
  residuals[1] = weight * (predicted_y - T(oy));
}

// Residual + Jacobian the way Ceres produces them: one pass of Jets over all 12
// parameters.  J is 2x12 row-major, columns [pose 0..5 | cam 6..8 | xyz 9..11].
ORC_API void orc_reproj_dual(const double* pose, const double* cam, const double* xyz,
                             const double* obs, double weight, double* r, double* J) {
  typedef Dual<12> D;
  D P[6], C[3], X[3], res[2];
  for (int i = 0; i < 6; i++) { P[i] = D(pose[i]); P[i].v[i] = 1.0; }
  for (int i = 0; i < 3; i++) { C[i] = D(cam[i]); C[i].v[6 + i] = 1.0; }
  for (int i = 0; i < 3; i++) { X[i] = D(xyz[i]); X[i].v[9 + i] = 1.0; }
  reproj_functor<D>(P, C, X, obs[0], obs[1], weight, res);
  for (int k = 0; k < 2; k++) {
    r[k] = res[k].a;
    if (J) for (int i = 0; i < 12; i++) J[k * 12 + i] = res[k].v[i];
  }
}

// The same residual with closed-form derivatives (what the HIP kernel evaluates).
// d(R(w)X)/dw is the exact derivative of the formula above, branch included.
ORC_API void orc_reproj_analytic(const double* pose, const double* cam, const double* xyz,
                                 const double* obs, double weight, double* r, double* J) {
  const double* aa = pose;
  const double X0 = xyz[0], X1 = xyz[1], X2 = xyz[2];
  double p[3], dpdw[9];  // dpdw[i*3+j] = d p_i / d w_j
  double R[9];
  const double theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (theta2 > std::numeric_limits<double>::epsilon()) {
    const double theta = std::sqrt(theta2);
    const double c = std::cos(theta), s = std::sin(theta);
    const double ti = 1.0 / theta;
    const double w[3] = {aa[0] * ti, aa[1] * ti, aa[2] * ti};
    const double wx[3] = {w[1] * X2 - w[2] * X1, w[2] * X0 - w[0] * X2, w[0] * X1 - w[1] * X0};
    const double wdx = w[0] * X0 + w[1] * X1 + w[2] * X2;
    const double tmp = wdx * (1.0 - c);
    p[0] = X0 * c + wx[0] * s + w[0] * tmp;
    p[1] = X1 * c + wx[1] * s + w[1] * tmp;
    p[2] = X2 * c + wx[2] * s + w[2] * tmp;
    const double Xv[3] = {X0, X1, X2};
    for (int j = 0; j < 3; j++) {
      // dw/dw_j = (e_j - w w_j)/theta ; dtheta/dw_j = w_j
      double dw[3];
      for (int i = 0; i < 3; i++) dw[i] = ((i == j ? 1.0 : 0.0) - w[i] * w[j]) * ti;
      const double dwx[3] = {dw[1] * X2 - dw[2] * X1, dw[2] * X0 - dw[0] * X2,
                             dw[0] * X1 - dw[1] * X0};
      const double dwdx = dw[0] * X0 + dw[1] * X1 + dw[2] * X2;
      const double dc = -s * w[j], ds = c * w[j];
      const double dtmp = dwdx * (1.0 - c) - wdx * dc;
      for (int i = 0; i < 3; i++)
        dpdw[i * 3 + j] = Xv[i] * dc + dwx[i] * s + wx[i] * ds + dw[i] * tmp + w[i] * dtmp;
    }
    // dp/dX = R (Rodrigues), needed below
    R[0] = c + w[0] * w[0] * (1 - c);        R[1] = w[0] * w[1] * (1 - c) - w[2] * s; R[2] = w[1] * s + w[0] * w[2] * (1 - c);
    R[3] = w[2] * s + w[0] * w[1] * (1 - c); R[4] = c + w[1] * w[1] * (1 - c);        R[5] = -w[0] * s + w[1] * w[2] * (1 - c);
    R[6] = -w[1] * s + w[0] * w[2] * (1 - c); R[7] = w[0] * s + w[1] * w[2] * (1 - c); R[8] = c + w[2] * w[2] * (1 - c);
  } else {
    p[0] = X0 + (aa[1] * X2 - aa[2] * X1);
    p[1] = X1 + (aa[2] * X0 - aa[0] * X2);
    p[2] = X2 + (aa[0] * X1 - aa[1] * X0);
    // d(w x X)/dw_j = e_j x X
    dpdw[0] = 0;   dpdw[1] = X2;  dpdw[2] = -X1;
    dpdw[3] = -X2; dpdw[4] = 0;   dpdw[5] = X0;
    dpdw[6] = X1;  dpdw[7] = -X0; dpdw[8] = 0;
    R[0] = 1; R[1] = -aa[2]; R[2] = aa[1];
    R[3] = aa[2]; R[4] = 1; R[5] = -aa[0];
    R[6] = -aa[1]; R[7] = aa[0]; R[8] = 1;
  }
  p[0] += pose[3]; p[1] += pose[4]; p[2] += pose[5];
  // Ceres evaluates a residual block with Jets when Jacobians are asked for and with plain doubles otherwise
  // (AutoDiffCostFunction::Evaluate).  The functor's `p[0] / p[2]` (reprojection_error_pose_cam_xyz.h:52-53) is then
  // Jet / Jet, whose value part jet.h forms as f.a * (1.0 / g.a) (Ceres 1.13 jet.h, operator/ - the Dual<N> above does the
  // same), or a true division: the linearisation pass and the trial-cost pass differ in that one rounding, and so do we.
  const double iz = 1.0 / p[2];
  const double xp = J ? p[0] * iz : p[0] / p[2], yp = J ? p[1] * iz : p[1] / p[2];
  const double f = cam[0], l1 = cam[1], l2 = cam[2];
  const double r2 = xp * xp + yp * yp;
  const double dist = 1.0 + r2 * (l1 + l2 * r2);
  r[0] = weight * (f * dist * xp - obs[0]);
  r[1] = weight * (f * dist * yp - obs[1]);
  if (!J) return;
  const double dd = l1 + 2.0 * l2 * r2;  // d dist / d r2
  // d(u,v)/d(xp,yp)
  const double uxp = f * (dist + 2.0 * xp * xp * dd), uyp = f * 2.0 * xp * yp * dd;
  const double vxp = uyp, vyp = f * (dist + 2.0 * yp * yp * dd);
  // d(u,v)/dp
  const double up[3] = {uxp * iz, uyp * iz, -(uxp * xp + uyp * yp) * iz};
  const double vp[3] = {vxp * iz, vyp * iz, -(vxp * xp + vyp * yp) * iz};
  for (int j = 0; j < 3; j++) {
    J[0 * 12 + j] = weight * (up[0] * dpdw[0 + j] + up[1] * dpdw[3 + j] + up[2] * dpdw[6 + j]);
    J[1 * 12 + j] = weight * (vp[0] * dpdw[0 + j] + vp[1] * dpdw[3 + j] + vp[2] * dpdw[6 + j]);
    J[0 * 12 + 3 + j] = weight * up[j];
    J[1 * 12 + 3 + j] = weight * vp[j];
    J[0 * 12 + 9 + j] = weight * (up[0] * R[0 + j] + up[1] * R[3 + j] + up[2] * R[6 + j]);
    J[1 * 12 + 9 + j] = weight * (vp[0] * R[0 + j] + vp[1] * R[3 + j] + vp[2] * R[6 + j]);
  }
  J[0 * 12 + 6] = weight * dist * xp;           J[1 * 12 + 6] = weight * dist * yp;
  J[0 * 12 + 7] = weight * f * r2 * xp;         J[1 * 12 + 7] = weight * f * r2 * yp;
  J[0 * 12 + 8] = weight * f * r2 * r2 * xp;    J[1 * 12 + 8] = weight * f * r2 * r2 * yp;
}

// ceres::HuberLoss::Evaluate (loss_function.cc, Ceres 1.13), constructed at
// optimizer.cc:84 / slam_gps.cc:822 with a = 1.0.
static inline void huber(double a, double s, double rho[3]) {
  const double b = a * a;
  if (s > b) {
    const double r = std::sqrt(s);
    rho[0] = 2.0 * a * r - b;
    rho[1] = std::max(std::numeric_limits<double>::min(), a / r);
    rho[2] = -rho[1] / (2.0 * s);
  } else {
    rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
}
ORC_API void orc_huber(double a, double s, double* rho) { huber(a, s, rho); }


// =====================================================================================
// Bundle adjustment: the problem of optimizer.cc:59-129 solved as ceres::Solve does at
// optimizer.cc:133 with linear_solver_type = DENSE_SCHUR (optimizer.cc:47).
// =====================================================================================
namespace {

struct Obs {
  int src;     // index into the caller's obs arrays
  int cam, pt, model;
  int cslot;   // first reduced-system column of the camera block, or -1
  int mslot;   // first reduced-system column of the intrinsics block, or -1
  int pslot;   // index of the eliminated point block, or -1
  double r[2];     // corrected residual
  double Jc[12];   // 2x6   corrected (and, after scale_columns, column-scaled)
  double Jm[6];    // 2x3
  double Jp[6];    // 2x3
};

struct GpsRes { int cam, cslot; double r[3]; double J[3]; };  // J[i] = d r_i / d t_i

struct Ba {
  const msfm_ba_problem* P;
  msfm_ba_options opt;
  int Nc, Nm, Np, No;
  std::vector<int> cam_slot, model_slot, pt_slot;  // -1 = not a parameter block
  int n_cam_blocks = 0, n_model_blocks = 0, n_pt_blocks = 0;
  int nred = 0;  // 6*n_cam_blocks + 3*n_model_blocks
  std::vector<Obs> obs;       // active residual blocks; eliminated-point rows first
  std::vector<int> pt_first;  // [n_pt_blocks+1] ranges of obs
  size_t n_erows = 0;         // obs[0..n_erows) have an e-block
  std::vector<GpsRes> gps;
  std::vector<double> cam, model, pt;              // x_
  std::vector<double> scale_c, scale_m, scale_p;   // jacobian_scaling_
  std::vector<double> diag_c, diag_m, diag_p;      // LM strategy diagonal_
  std::vector<double> step_c, step_m, step_p;      // trust_region_step_
  std::vector<double> grad_c, grad_m, grad_p;      // gradient_
  std::vector<double> lhs, rhs;
  int threads = 1;
  std::unique_ptr<char[]> elim_store;   // per-point products of the threaded elimination
  size_t elim_bytes = 0;
  std::vector<double> scratch;   // per-observation terms of a threaded pass, summed in order afterwards
};

static bool is_mut(const uint8_t* m, int i) { return m == nullptr || m[i] != 0; }

static void ba_setup(Ba& B) {
  const msfm_ba_problem* P = B.P;
  B.Nc = P->n_cams; B.Nm = P->n_models; B.Np = P->n_points; B.No = P->n_obs;
  B.cam_slot.assign(B.Nc, -1); B.model_slot.assign(B.Nm, -1); B.pt_slot.assign(B.Np, -1);
  // A parameter block exists iff some residual block uses it (optimizer.cc:86-125 only
  // ever hands Ceres the blocks of the functor it picked).
  std::vector<char> cam_used(B.Nc, 0), model_used(B.Nm, 0), pt_used(B.Np, 0);
  for (int o = 0; o < B.No; o++) {
    const int c = P->obs_cam[o], p = P->obs_pt[o], m = P->cam_model_of_cam[c];
    const bool cm = is_mut(P->cam_mutable, c), pm = is_mut(P->pt_mutable, p);
    if (!cm && !pm) continue;
    if (pm) pt_used[p] = 1;
    if (cm) { cam_used[c] = 1; if (is_mut(P->model_mutable, m)) model_used[m] = 1; }
  }
  if (P->gps_xyz) for (int c = 0; c < B.Nc; c++) if (is_mut(P->cam_mutable, c)) cam_used[c] = 1;
  for (int c = 0; c < B.Nc; c++) if (cam_used[c]) B.cam_slot[c] = B.n_cam_blocks++;
  for (int m = 0; m < B.Nm; m++) if (model_used[m]) B.model_slot[m] = B.n_model_blocks++;
  for (int p = 0; p < B.Np; p++) if (pt_used[p]) B.pt_slot[p] = B.n_pt_blocks++;
  B.nred = 6 * B.n_cam_blocks + 3 * B.n_model_blocks;
  B.obs.clear();
  B.pt_first.assign(B.n_pt_blocks + 1, 0);
  for (int pass = 0; pass < 2; pass++) {
    for (int o = 0; o < B.No; o++) {
      const int c = P->obs_cam[o], p = P->obs_pt[o], m = P->cam_model_of_cam[c];
      const bool cm = is_mut(P->cam_mutable, c), pm = is_mut(P->pt_mutable, p);
      if (!cm && !pm) continue;
      if ((pass == 0) != pm) continue;
      Obs ob;
      ob.src = o; ob.cam = c; ob.pt = p; ob.model = m;
      ob.cslot = cm ? 6 * B.cam_slot[c] : -1;
      ob.mslot = (cm && is_mut(P->model_mutable, m)) ? 6 * B.n_cam_blocks + 3 * B.model_slot[m] : -1;
      ob.pslot = pm ? B.pt_slot[p] : -1;
      if (pm) B.pt_first[ob.pslot + 1]++;
      B.obs.push_back(ob);
    }
    if (pass == 0) B.n_erows = B.obs.size();
  }
  for (int i = 0; i < B.n_pt_blocks; i++) B.pt_first[i + 1] += B.pt_first[i];
  B.gps.clear();
  if (P->gps_xyz)
    for (int c = 0; c < B.Nc; c++)
      if (B.cam_slot[c] >= 0) { GpsRes g; g.cam = c; g.cslot = 6 * B.cam_slot[c]; B.gps.push_back(g); }
  B.cam.assign(P->cam_pose, P->cam_pose + 6 * (size_t)B.Nc);
  B.model.assign(P->cam_model, P->cam_model + 3 * (size_t)B.Nm);
  B.pt.assign(P->point, P->point + 3 * (size_t)B.Np);
  B.scale_c.assign(6 * (size_t)B.n_cam_blocks, 1.0);
  B.scale_m.assign(3 * (size_t)B.n_model_blocks, 1.0);
  B.scale_p.assign(3 * (size_t)B.n_pt_blocks, 1.0);
  B.diag_c.assign(B.scale_c.size(), 0.0);
  B.diag_m.assign(B.scale_m.size(), 0.0);
  B.diag_p.assign(B.scale_p.size(), 0.0);
}

// GPSErrorPoseAbsolute (gps_error_pose_absolute.h:31-44) under Jets: d|x|/dx = x<0 ? -1 : 1.
static void gps_residual(const double* pose, const double* g, double w, double r[3], double J[3]) {
  const double wz[3] = {w, w, w / 5.0};
  for (int i = 0; i < 3; i++) {
    const double d = pose[3 + i] - g[i];
    r[i] = wz[i] * std::fabs(d);
    J[i] = wz[i] * (d < 0.0 ? -1.0 : 1.0);
  }
}

// Evaluator::Evaluate -> ResidualBlock::Evaluate -> Corrector (Ceres 1.13):
// cost = sum 1/2 rho(|r|^2); since rho'' <= 0 for Huber, residual and Jacobian are both
// scaled by sqrt(rho') (corrector.cc, alpha = 0 branch).  With `lin` the corrected,
// not-yet-column-scaled Jacobians go to B.obs / B.gps and the gradient J^T r to B.grad_*.
static double ba_evaluate(Ba& B, const std::vector<double>& cam, const std::vector<double>& model,
                          const std::vector<double>& pt, bool lin) {
  const msfm_ba_problem* P = B.P;
  const double delta = B.opt.huber_delta;
  double cost = 0.0;
  if (lin) {
    B.grad_c.assign(6 * (size_t)B.n_cam_blocks, 0.0);
    B.grad_m.assign(3 * (size_t)B.n_model_blocks, 0.0);
    B.grad_p.assign(3 * (size_t)B.n_pt_blocks, 0.0);
  }
  const int nt = B.threads;
  // one residual block: returns its cost term; with `lin` the corrected Jacobian goes to ob
  auto eval_one = [&](Obs& ob) -> double {
    double r[2], J[24], rho[3];
    const double w = P->pt_weight ? P->pt_weight[ob.pt] : 1.0;
    orc_reproj_analytic(&cam[6 * (size_t)ob.cam], &model[3 * (size_t)ob.model],
                        &pt[3 * (size_t)ob.pt], &P->obs_xy[2 * (size_t)ob.src], w, r,
                        lin ? J : nullptr);
    const double s = r[0] * r[0] + r[1] * r[1];
    huber(delta, s, rho);
    if (lin) {
      const double sq = std::sqrt(rho[1]);
      ob.r[0] = sq * r[0]; ob.r[1] = sq * r[1];
      for (int k = 0; k < 2; k++) {
        for (int j = 0; j < 6; j++) ob.Jc[k * 6 + j] = sq * J[k * 12 + j];
        for (int j = 0; j < 3; j++) ob.Jm[k * 3 + j] = sq * J[k * 12 + 6 + j];
        for (int j = 0; j < 3; j++) ob.Jp[k * 3 + j] = sq * J[k * 12 + 9 + j];
      }
    }
    return 0.5 * rho[0];
  };
  auto grad_one = [&](const Obs& ob) {
    if (ob.cslot >= 0) for (int j = 0; j < 6; j++) B.grad_c[ob.cslot + j] += ob.Jc[j] * ob.r[0] + ob.Jc[6 + j] * ob.r[1];
    if (ob.mslot >= 0) for (int j = 0; j < 3; j++) B.grad_m[ob.mslot - 6 * B.n_cam_blocks + j] += ob.Jm[j] * ob.r[0] + ob.Jm[3 + j] * ob.r[1];
    if (ob.pslot >= 0) for (int j = 0; j < 3; j++) B.grad_p[3 * (size_t)ob.pslot + j] += ob.Jp[j] * ob.r[0] + ob.Jp[3 + j] * ob.r[1];
  };
  if (nt <= 1) {
    for (size_t i = 0; i < B.obs.size(); i++) {
      cost += eval_one(B.obs[i]);
      if (lin) grad_one(B.obs[i]);
    }
  } else {
    // the projections (sincos, divisions) run on all threads; the sums that follow keep the single-thread order
    B.scratch.resize(B.obs.size());
    const long n = (long)B.obs.size();
#pragma omp parallel for num_threads(nt) schedule(static)
    for (long i = 0; i < n; i++) B.scratch[i] = eval_one(B.obs[i]);
    for (long i = 0; i < n; i++) {
      cost += B.scratch[i];
      if (lin) grad_one(B.obs[i]);
    }
  }
  for (size_t i = 0; i < B.gps.size(); i++) {
    GpsRes& g = B.gps[i];
    double r[3], J[3], rho[3];
    gps_residual(&cam[6 * (size_t)g.cam], &P->gps_xyz[3 * (size_t)g.cam], P->gps_weight, r, J);
    const double s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    huber(delta, s, rho);
    cost += 0.5 * rho[0];
    if (!lin) continue;
    const double sq = std::sqrt(rho[1]);
    for (int k = 0; k < 3; k++) { g.r[k] = sq * r[k]; g.J[k] = sq * J[k]; B.grad_c[g.cslot + 3 + k] += g.J[k] * g.r[k]; }
  }
  return cost;
}

// TrustRegionMinimizer::EvaluateGradientAndJacobian, iteration 0 (Ceres 1.13):
// jacobian_scaling_[i] = 1 / (1 + sqrt(SquaredColumnNorm_i)).
static void ba_compute_jacobi_scaling(Ba& B) {
  std::vector<double> nc(B.scale_c.size(), 0.0), nm(B.scale_m.size(), 0.0), np(B.scale_p.size(), 0.0);
  const int mo = 6 * B.n_cam_blocks;
  for (const Obs& ob : B.obs) {
    if (ob.cslot >= 0) for (int j = 0; j < 6; j++) nc[ob.cslot + j] += ob.Jc[j] * ob.Jc[j] + ob.Jc[6 + j] * ob.Jc[6 + j];
    if (ob.mslot >= 0) for (int j = 0; j < 3; j++) nm[ob.mslot - mo + j] += ob.Jm[j] * ob.Jm[j] + ob.Jm[3 + j] * ob.Jm[3 + j];
    if (ob.pslot >= 0) for (int j = 0; j < 3; j++) np[3 * (size_t)ob.pslot + j] += ob.Jp[j] * ob.Jp[j] + ob.Jp[3 + j] * ob.Jp[3 + j];
  }
  for (const GpsRes& g : B.gps) for (int k = 0; k < 3; k++) nc[g.cslot + 3 + k] += g.J[k] * g.J[k];
  for (size_t i = 0; i < nc.size(); i++) B.scale_c[i] = 1.0 / (1.0 + std::sqrt(nc[i]));
  for (size_t i = 0; i < nm.size(); i++) B.scale_m[i] = 1.0 / (1.0 + std::sqrt(nm[i]));
  for (size_t i = 0; i < np.size(); i++) B.scale_p[i] = 1.0 / (1.0 + std::sqrt(np[i]));
}

// jacobian_->ScaleColumns(jacobian_scaling_)
static void ba_scale_columns(Ba& B) {
  const int mo = 6 * B.n_cam_blocks;
  for (Obs& ob : B.obs) {
    if (ob.cslot >= 0) for (int k = 0; k < 2; k++) for (int j = 0; j < 6; j++) ob.Jc[k * 6 + j] *= B.scale_c[ob.cslot + j];
    if (ob.mslot >= 0) for (int k = 0; k < 2; k++) for (int j = 0; j < 3; j++) ob.Jm[k * 3 + j] *= B.scale_m[ob.mslot - mo + j];
    if (ob.pslot >= 0) for (int k = 0; k < 2; k++) for (int j = 0; j < 3; j++) ob.Jp[k * 3 + j] *= B.scale_p[3 * (size_t)ob.pslot + j];
  }
  for (GpsRes& g : B.gps) for (int k = 0; k < 3; k++) g.J[k] *= B.scale_c[g.cslot + 3 + k];
}

// In-place Cholesky A = L L^T on the lower triangle of a row-major n x n matrix
// (what `lhs.selfadjointView<Upper>().llt()` does in DenseSchurComplementSolver).
// Row-oriented so the inner loop is a contiguous dot product.  Returns false if not PD.
static __attribute__((noinline)) double chol_dot(const double* a, const double* b, int n) {
  double s = 0.0;
#pragma omp simd reduction(+ : s)
  for (int k = 0; k < n; k++) s += a[k] * b[k];
  return s;
}
static bool dense_cholesky_lower(double* A, int n, int nt = 1) {
  if (nt <= 1) {
    for (int i = 0; i < n; i++) {
      double* Ai = A + (size_t)i * n;
      for (int j = 0; j <= i; j++) {
        const double* Aj = A + (size_t)j * n;
        const double s = chol_dot(Ai, Aj, j);
        if (i == j) {
          const double d = Ai[i] - s;
          if (!(d > 0.0)) return false;
          Ai[i] = std::sqrt(d);
        } else {
          Ai[j] = (Ai[j] - s) / Aj[j];
        }
      }
    }
    return true;
  }
  // The same row-oriented recurrence, every entry from the same dot product: a block of rows first takes its columns
  // left of the block on all threads (they only need finished rows), then the small triangle inside the block in order.
  const int RB = 48;
  for (int r0 = 0; r0 < n; r0 += RB) {
    const int r1 = std::min(n, r0 + RB);
#pragma omp parallel for num_threads(nt) schedule(static, 1)
    for (int i = r0; i < r1; i++) {
      double* Ai = A + (size_t)i * n;
      for (int j = 0; j < r0; j++) {
        const double* Aj = A + (size_t)j * n;
        Ai[j] = (Ai[j] - chol_dot(Ai, Aj, j)) / Aj[j];
      }
    }
    for (int i = r0; i < r1; i++) {
      double* Ai = A + (size_t)i * n;
      for (int j = r0; j <= i; j++) {
        const double* Aj = A + (size_t)j * n;
        const double s = chol_dot(Ai, Aj, j);
        if (i == j) {
          const double d = Ai[i] - s;
          if (!(d > 0.0)) return false;
          Ai[i] = std::sqrt(d);
        } else {
          Ai[j] = (Ai[j] - s) / Aj[j];
        }
      }
    }
  }
  return true;
}
static void dense_cholesky_solve(const double* L, int n, double* b) {
  for (int i = 0; i < n; i++) {
    const double* Li = L + (size_t)i * n;
    double s = b[i];
    for (int k = 0; k < i; k++) s -= Li[k] * b[k];
    b[i] = s / Li[i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = b[i];
    for (int k = i + 1; k < n; k++) s -= L[(size_t)k * n + i] * b[k];
    b[i] = s / L[(size_t)i * n + i];
  }
}

// 3x3 SPD: Cholesky factor (lower, row-major 3x3) — InvertPSDMatrix<3> in
// schur_eliminator_impl.h uses Eigen LLT on the 3x3 e-block.
static bool chol3(const double* V, double* L) {
  const double l00 = V[0]; if (!(l00 > 0)) return false;
  L[0] = std::sqrt(l00);
  L[3] = V[3] / L[0]; L[6] = V[6] / L[0];
  const double d1 = V[4] - L[3] * L[3]; if (!(d1 > 0)) return false;
  L[4] = std::sqrt(d1);
  L[7] = (V[7] - L[6] * L[3]) / L[4];
  const double d2 = V[8] - L[6] * L[6] - L[7] * L[7]; if (!(d2 > 0)) return false;
  L[8] = std::sqrt(d2);
  L[1] = L[2] = L[5] = 0;
  return true;
}
static void chol3_solve(const double* L, double* b) {  // solves (L L^T) x = b in place
  b[0] = b[0] / L[0];
  b[1] = (b[1] - L[3] * b[0]) / L[4];
  b[2] = (b[2] - L[6] * b[0] - L[7] * b[1]) / L[8];
  b[2] = b[2] / L[8];
  b[1] = (b[1] - L[7] * b[2]) / L[4];
  b[0] = (b[0] - L[3] * b[1] - L[6] * b[2]) / L[0];
}

// LevenbergMarquardtStrategy::ComputeStep + SchurComplementSolver::SolveImpl
// (SchurEliminator::Eliminate, dense LLT, BackSubstitute) of Ceres 1.13, then
// TrustRegionMinimizer::ComputeTrustRegionStep's model_cost_change.
// Returns false for LINEAR_SOLVER_FAILURE (step stays invalid).
static bool ba_compute_step(Ba& B, double radius, bool reuse_diagonal, double* model_cost_change,
                            bool assemble_only = false) {
  static const bool verbose = getenv("ORC_VERBOSE") != nullptr;
  auto tlast = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!verbose) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "orc: %-22s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - tlast).count());
    tlast = now;
  };
  const int n = B.nred, mo = 6 * B.n_cam_blocks;
  const size_t ncc = B.scale_c.size(), nmc = B.scale_m.size(), npc = B.scale_p.size();
  if (!reuse_diagonal) {
    std::fill(B.diag_c.begin(), B.diag_c.end(), 0.0);
    std::fill(B.diag_m.begin(), B.diag_m.end(), 0.0);
    std::fill(B.diag_p.begin(), B.diag_p.end(), 0.0);
    for (const Obs& ob : B.obs) {
      if (ob.cslot >= 0) for (int j = 0; j < 6; j++) B.diag_c[ob.cslot + j] += ob.Jc[j] * ob.Jc[j] + ob.Jc[6 + j] * ob.Jc[6 + j];
      if (ob.mslot >= 0) for (int j = 0; j < 3; j++) B.diag_m[ob.mslot - mo + j] += ob.Jm[j] * ob.Jm[j] + ob.Jm[3 + j] * ob.Jm[3 + j];
      if (ob.pslot >= 0) for (int j = 0; j < 3; j++) B.diag_p[3 * (size_t)ob.pslot + j] += ob.Jp[j] * ob.Jp[j] + ob.Jp[3 + j] * ob.Jp[3 + j];
    }
    for (const GpsRes& g : B.gps) for (int k = 0; k < 3; k++) B.diag_c[g.cslot + 3 + k] += g.J[k] * g.J[k];
    const double lo = B.opt.min_lm_diagonal, hi = B.opt.max_lm_diagonal;
    for (double& d : B.diag_c) d = std::min(std::max(d, lo), hi);
    for (double& d : B.diag_m) d = std::min(std::max(d, lo), hi);
    for (double& d : B.diag_p) d = std::min(std::max(d, lo), hi);
  }
  // lm_diagonal_ = sqrt(diagonal_ / radius_); the eliminator adds D.^2.
  auto d2 = [radius](double d) { const double q = std::sqrt(d / radius); return q * q; };
  B.lhs.assign((size_t)n * n, 0.0);
  B.rhs.assign(n, 0.0);
  double* S = B.lhs.data();
  for (size_t i = 0; i < ncc; i++) S[i * n + i] += d2(B.diag_c[i]);
  for (size_t i = 0; i < nmc; i++) S[(mo + i) * n + (mo + i)] += d2(B.diag_m[i]);

  auto add_ftf = [&](const Obs& ob) {  // F^T F and F^T b of one row block (upper triangle)
    if (ob.cslot >= 0) {
      for (int a = 0; a < 6; a++) {
        for (int b = a; b < 6; b++) S[(size_t)(ob.cslot + a) * n + ob.cslot + b] += ob.Jc[a] * ob.Jc[b] + ob.Jc[6 + a] * ob.Jc[6 + b];
        B.rhs[ob.cslot + a] += ob.Jc[a] * ob.r[0] + ob.Jc[6 + a] * ob.r[1];
      }
      if (ob.mslot >= 0)
        for (int a = 0; a < 6; a++) for (int b = 0; b < 3; b++)
          S[(size_t)(ob.cslot + a) * n + ob.mslot + b] += ob.Jc[a] * ob.Jm[b] + ob.Jc[6 + a] * ob.Jm[3 + b];
    }
    if (ob.mslot >= 0)
      for (int a = 0; a < 3; a++) {
        for (int b = a; b < 3; b++) S[(size_t)(ob.mslot + a) * n + ob.mslot + b] += ob.Jm[a] * ob.Jm[b] + ob.Jm[3 + a] * ob.Jm[3 + b];
        B.rhs[ob.mslot + a] += ob.Jm[a] * ob.r[0] + ob.Jm[3 + a] * ob.r[1];
      }
  };

  std::vector<double> Lp(9 * (size_t)B.n_pt_blocks), gp(3 * (size_t)B.n_pt_blocks);
  struct Ent { int slot, dim; double W[18]; };  // W = F_b^T E, dim x 3
  const int nt = B.threads;
  if (nt > 1) {
    // Threaded elimination with the single-thread arithmetic.  Phase 1 (all threads, one point each): V, g, the
    // per-f-block products W, the 3x3 factor, V^-1 g and Z = V^-1 W^T.  Phase 2: every thread walks ALL points in order
    // but applies only the updates whose row block it owns, so each entry of S / rhs receives its terms in exactly the
    // order of the single-thread loop.
    struct PEnt { int slot, dim; double W[18], Z[18]; };
    const int npb = B.n_pt_blocks;
    std::vector<long> ent_first(npb + 1, 0);
    for (int pb = 0; pb < npb; pb++) {
      const int k = B.pt_first[pb + 1] - B.pt_first[pb];
      ent_first[pb + 1] = ent_first[pb] + k + std::min(k, std::max(1, B.n_model_blocks));
    }
    // not zero-filled (first touched by the thread that fills it) and kept for the next linear solve
    const size_t pents_bytes = sizeof(PEnt) * (size_t)ent_first[npb];
    if (B.elim_bytes < pents_bytes) { B.elim_store.reset(new char[pents_bytes]); B.elim_bytes = pents_bytes; }
    PEnt* const pents_data = reinterpret_cast<PEnt*>(B.elim_store.get());
    std::vector<int> ent_count(npb, 0);
    std::vector<double> vgs(3 * (size_t)npb);
    int failed = 0;
#pragma omp parallel for num_threads(nt) schedule(static, 256)
    for (int pb = 0; pb < npb; pb++) {
      double V[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
      PEnt* ents = pents_data + ent_first[pb];
      int ne = 0;
      for (int i = B.pt_first[pb]; i < B.pt_first[pb + 1]; i++) {
        const Obs& ob = B.obs[i];
        for (int a = 0; a < 3; a++) {
          for (int b = 0; b < 3; b++) V[a * 3 + b] += ob.Jp[a] * ob.Jp[b] + ob.Jp[3 + a] * ob.Jp[3 + b];
          g[a] += ob.Jp[a] * ob.r[0] + ob.Jp[3 + a] * ob.r[1];
        }
        for (int which = 0; which < 2; which++) {
          const int slot = which == 0 ? ob.cslot : ob.mslot;
          if (slot < 0) continue;
          const int dim = which == 0 ? 6 : 3;
          const double* Jf = which == 0 ? ob.Jc : ob.Jm;
          PEnt* e = nullptr;
          for (int q = 0; q < ne; q++) if (ents[q].slot == slot) { e = &ents[q]; break; }
          if (!e) { e = &ents[ne++]; e->slot = slot; e->dim = dim; std::fill(e->W, e->W + 18, 0.0); }
          for (int a = 0; a < dim; a++) for (int b = 0; b < 3; b++) e->W[a * 3 + b] += Jf[a] * ob.Jp[b] + Jf[dim + a] * ob.Jp[3 + b];
        }
      }
      for (int a = 0; a < 3; a++) V[a * 3 + a] += d2(B.diag_p[3 * (size_t)pb + a]);
      double* L = &Lp[9 * (size_t)pb];
      if (!chol3(V, L)) {
#pragma omp atomic write
        failed = 1;
        continue;
      }
      for (int a = 0; a < 3; a++) gp[3 * (size_t)pb + a] = g[a];
      std::sort(ents, ents + ne, [](const PEnt& a, const PEnt& b) { return a.slot < b.slot; });
      double vg[3] = {g[0], g[1], g[2]};
      chol3_solve(L, vg);
      for (int a = 0; a < 3; a++) vgs[3 * (size_t)pb + a] = vg[a];
      for (int j = 0; j < ne; j++) {
        PEnt& ej = ents[j];
        for (int a = 0; a < ej.dim; a++) {
          double col[3] = {ej.W[a * 3 + 0], ej.W[a * 3 + 1], ej.W[a * 3 + 2]};
          chol3_solve(L, col);
          for (int b = 0; b < 3; b++) ej.Z[b * ej.dim + a] = col[b];
        }
      }
      ent_count[pb] = ne;
    }
    if (failed) return false;
    lap("eliminate: per point");
    // Row blocks are owned in contiguous ranges (cameras that see a point are neighbours in index on survey-like
    // scenes, so a point usually concerns one or two owners), balanced by the rows' observation counts; every owner gets
    // the ordered list of the points it has work for and never looks at the others.
    const int nblk = B.n_cam_blocks + B.n_model_blocks;
    std::vector<long> blk_work(nblk, 0);
    for (const Obs& ob : B.obs) {
      if (ob.cslot >= 0) blk_work[ob.cslot / 6] += 4;
      if (ob.mslot >= 0) blk_work[B.n_cam_blocks + (ob.mslot - mo) / 3] += 1;
    }
    long total_work = 0;
    for (long w : blk_work) total_work += w;
    std::vector<int> blk_owner(nblk, 0);
    {
      long run = 0;
      for (int b = 0; b < nblk; b++) { blk_owner[b] = (int)std::min<long>(nt - 1, run * nt / std::max<long>(1, total_work)); run += blk_work[b]; }
    }
    auto owner = [&](int slot) { return blk_owner[slot < mo ? slot / 6 : B.n_cam_blocks + (slot - mo) / 3]; };
    std::vector<std::vector<int>> my_points(nt);
    {
      std::vector<char> seen(nt);
      for (int pb = 0; pb < npb; pb++) {
        std::fill(seen.begin(), seen.end(), 0);
        const PEnt* ents = pents_data + ent_first[pb];
        for (int j = 0; j < ent_count[pb]; j++) seen[owner(ents[j].slot)] = 1;
        for (int t = 0; t < nt; t++) if (seen[t]) my_points[t].push_back(pb);
      }
    }
    lap("eliminate: work lists");
#pragma omp parallel num_threads(nt)
    {
#ifdef _OPENMP
      const int t = omp_get_thread_num();
#else
      const int t = 0;
#endif
      for (int pb : my_points[t]) {
        for (int i = B.pt_first[pb]; i < B.pt_first[pb + 1]; i++) {   // add_ftf, rows owned by this thread
          const Obs& ob = B.obs[i];
          if (ob.cslot >= 0 && owner(ob.cslot) == t) {
            for (int a = 0; a < 6; a++) {
              for (int b = a; b < 6; b++) S[(size_t)(ob.cslot + a) * n + ob.cslot + b] += ob.Jc[a] * ob.Jc[b] + ob.Jc[6 + a] * ob.Jc[6 + b];
              B.rhs[ob.cslot + a] += ob.Jc[a] * ob.r[0] + ob.Jc[6 + a] * ob.r[1];
            }
            if (ob.mslot >= 0)
              for (int a = 0; a < 6; a++) for (int b = 0; b < 3; b++)
                S[(size_t)(ob.cslot + a) * n + ob.mslot + b] += ob.Jc[a] * ob.Jm[b] + ob.Jc[6 + a] * ob.Jm[3 + b];
          }
          if (ob.mslot >= 0 && owner(ob.mslot) == t)
            for (int a = 0; a < 3; a++) {
              for (int b = a; b < 3; b++) S[(size_t)(ob.mslot + a) * n + ob.mslot + b] += ob.Jm[a] * ob.Jm[b] + ob.Jm[3 + a] * ob.Jm[3 + b];
              B.rhs[ob.mslot + a] += ob.Jm[a] * ob.r[0] + ob.Jm[3 + a] * ob.r[1];
            }
        }
        const PEnt* ents = pents_data + ent_first[pb];
        const int ne = ent_count[pb];
        const double* vg = &vgs[3 * (size_t)pb];
        for (int j = 0; j < ne; j++) {
          const PEnt& ej = ents[j];
          if (owner(ej.slot) == t)
            for (int a = 0; a < ej.dim; a++) B.rhs[ej.slot + a] -= ej.W[a * 3 + 0] * vg[0] + ej.W[a * 3 + 1] * vg[1] + ej.W[a * 3 + 2] * vg[2];
          for (int i = 0; i <= j; i++) {
            const PEnt& ei = ents[i];
            if (owner(ei.slot) != t) continue;
            for (int a = 0; a < ei.dim; a++) for (int b = 0; b < ej.dim; b++) {
              if (i == j && b < a) continue;  // upper triangle only
              S[(size_t)(ei.slot + a) * n + ej.slot + b] -= ei.W[a * 3 + 0] * ej.Z[0 * ej.dim + b] + ei.W[a * 3 + 1] * ej.Z[1 * ej.dim + b] + ei.W[a * 3 + 2] * ej.Z[2 * ej.dim + b];
            }
          }
        }
      }
    }
  } else {
  // per eliminated point: chunk elimination
  std::vector<Ent> ents;
  for (int pb = 0; pb < B.n_pt_blocks; pb++) {
    double V[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
    ents.clear();
    for (int i = B.pt_first[pb]; i < B.pt_first[pb + 1]; i++) {
      const Obs& ob = B.obs[i];
      for (int a = 0; a < 3; a++) {
        for (int b = 0; b < 3; b++) V[a * 3 + b] += ob.Jp[a] * ob.Jp[b] + ob.Jp[3 + a] * ob.Jp[3 + b];
        g[a] += ob.Jp[a] * ob.r[0] + ob.Jp[3 + a] * ob.r[1];
      }
      add_ftf(ob);
      // buffer = E^T F, accumulated per f-block (std::map keyed by block in Ceres)
      for (int which = 0; which < 2; which++) {
        const int slot = which == 0 ? ob.cslot : ob.mslot;
        if (slot < 0) continue;
        const int dim = which == 0 ? 6 : 3;
        const double* Jf = which == 0 ? ob.Jc : ob.Jm;
        Ent* e = nullptr;
        for (Ent& x : ents) if (x.slot == slot) { e = &x; break; }
        if (!e) { Ent ne; ne.slot = slot; ne.dim = dim; std::fill(ne.W, ne.W + 18, 0.0); ents.push_back(ne); e = &ents.back(); }
        for (int a = 0; a < dim; a++) for (int b = 0; b < 3; b++) e->W[a * 3 + b] += Jf[a] * ob.Jp[b] + Jf[dim + a] * ob.Jp[3 + b];
      }
    }
    for (int a = 0; a < 3; a++) V[a * 3 + a] += d2(B.diag_p[3 * (size_t)pb + a]);
    double* L = &Lp[9 * (size_t)pb];
    if (!chol3(V, L)) return false;
    for (int a = 0; a < 3; a++) gp[3 * (size_t)pb + a] = g[a];
    std::sort(ents.begin(), ents.end(), [](const Ent& a, const Ent& b) { return a.slot < b.slot; });
    // rhs -= W V^-1 g ; lhs -= W_i V^-1 W_j^T (i <= j)
    double vg[3] = {g[0], g[1], g[2]};
    chol3_solve(L, vg);
    // Z_j = V^-1 W_j^T (3 x dim_j), column by column
    for (size_t j = 0; j < ents.size(); j++) {
      const Ent& ej = ents[j];
      double Z[18];  // Z[b*dim + a] = (V^-1 W_j^T)[b][a]
      for (int a = 0; a < ej.dim; a++) {
        double col[3] = {ej.W[a * 3 + 0], ej.W[a * 3 + 1], ej.W[a * 3 + 2]};
        chol3_solve(L, col);
        for (int b = 0; b < 3; b++) Z[b * ej.dim + a] = col[b];
      }
      for (int a = 0; a < ej.dim; a++) B.rhs[ej.slot + a] -= ej.W[a * 3 + 0] * vg[0] + ej.W[a * 3 + 1] * vg[1] + ej.W[a * 3 + 2] * vg[2];
      for (size_t i = 0; i <= j; i++) {
        const Ent& ei = ents[i];
        for (int a = 0; a < ei.dim; a++) for (int b = 0; b < ej.dim; b++) {
          if (i == j && b < a) continue;  // upper triangle only
          S[(size_t)(ei.slot + a) * n + ej.slot + b] -= ei.W[a * 3 + 0] * Z[0 * ej.dim + b] + ei.W[a * 3 + 1] * Z[1 * ej.dim + b] + ei.W[a * 3 + 2] * Z[2 * ej.dim + b];
        }
      }
    }
  }
  }
  lap("eliminate: total");
  // rows without an e-block (SchurEliminator::NoEBlockRowsUpdate) and GPS rows
  for (size_t i = B.n_erows; i < B.obs.size(); i++) add_ftf(B.obs[i]);
  for (const GpsRes& g : B.gps) for (int k = 0; k < 3; k++) {
    S[(size_t)(g.cslot + 3 + k) * n + g.cslot + 3 + k] += g.J[k] * g.J[k];
    B.rhs[g.cslot + 3 + k] += g.J[k] * g.r[k];
  }
  if (assemble_only) return true;
  // dense LLT of the reduced system
  std::vector<double> z(B.rhs);
  if (n > 0) {
    for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) S[(size_t)i * n + j] = S[(size_t)j * n + i];
    lap("before cholesky");
    if (!dense_cholesky_lower(S, n, nt)) return false;
    lap("cholesky");
    dense_cholesky_solve(S, n, z.data());
  }
  // back substitution: y_p = V^-1 sum_rows Jp^T (r - Jc z_c - Jm z_m)
  B.step_c.assign(ncc, 0.0); B.step_m.assign(nmc, 0.0); B.step_p.assign(npc, 0.0);
  for (size_t i = 0; i < ncc; i++) B.step_c[i] = -z[i];
  for (size_t i = 0; i < nmc; i++) B.step_m[i] = -z[mo + i];
#pragma omp parallel for num_threads(nt) schedule(static, 256) if (nt > 1)
  for (int pb = 0; pb < B.n_pt_blocks; pb++) {
    double y[3] = {0, 0, 0};
    for (int i = B.pt_first[pb]; i < B.pt_first[pb + 1]; i++) {
      const Obs& ob = B.obs[i];
      double sj[2] = {ob.r[0], ob.r[1]};
      if (ob.cslot >= 0) for (int j = 0; j < 6; j++) { sj[0] -= ob.Jc[j] * z[ob.cslot + j]; sj[1] -= ob.Jc[6 + j] * z[ob.cslot + j]; }
      if (ob.mslot >= 0) for (int j = 0; j < 3; j++) { sj[0] -= ob.Jm[j] * z[ob.mslot + j]; sj[1] -= ob.Jm[3 + j] * z[ob.mslot + j]; }
      for (int a = 0; a < 3; a++) y[a] += ob.Jp[a] * sj[0] + ob.Jp[3 + a] * sj[1];
    }
    chol3_solve(&Lp[9 * (size_t)pb], y);
    for (int a = 0; a < 3; a++) B.step_p[3 * (size_t)pb + a] = -y[a];
  }
  for (double v : B.step_c) if (!std::isfinite(v)) return false;
  for (double v : B.step_m) if (!std::isfinite(v)) return false;
  for (double v : B.step_p) if (!std::isfinite(v)) return false;
  // model_cost_change = -(J step)^T (r + J step / 2)
  double mcc = 0.0;
  auto mcc_one = [&](const Obs& ob) {
    double m[2] = {0, 0};
    if (ob.cslot >= 0) for (int j = 0; j < 6; j++) { m[0] += ob.Jc[j] * B.step_c[ob.cslot + j]; m[1] += ob.Jc[6 + j] * B.step_c[ob.cslot + j]; }
    if (ob.mslot >= 0) for (int j = 0; j < 3; j++) { m[0] += ob.Jm[j] * B.step_m[ob.mslot - mo + j]; m[1] += ob.Jm[3 + j] * B.step_m[ob.mslot - mo + j]; }
    if (ob.pslot >= 0) for (int j = 0; j < 3; j++) { m[0] += ob.Jp[j] * B.step_p[3 * (size_t)ob.pslot + j]; m[1] += ob.Jp[3 + j] * B.step_p[3 * (size_t)ob.pslot + j]; }
    return m[0] * (ob.r[0] + m[0] / 2.0) + m[1] * (ob.r[1] + m[1] / 2.0);
  };
  if (nt <= 1) {
    for (const Obs& ob : B.obs) mcc -= mcc_one(ob);
  } else {
    B.scratch.resize(B.obs.size());
    const long no = (long)B.obs.size();
#pragma omp parallel for num_threads(nt) schedule(static)
    for (long i = 0; i < no; i++) B.scratch[i] = mcc_one(B.obs[i]);
    for (long i = 0; i < no; i++) mcc -= B.scratch[i];
  }
  for (const GpsRes& g : B.gps) for (int k = 0; k < 3; k++) {
    const double m = g.J[k] * B.step_c[g.cslot + 3 + k];
    mcc -= m * (g.r[k] + m / 2.0);
  }
  *model_cost_change = mcc;
  lap("backsub + mcc");
  return true;
}

static double sq(double x) { return x * x; }

}  // namespace

ORC_API void orc_ba_options_default(msfm_ba_options* o) {
  o->max_num_iterations = 200;
  o->num_threads = 1;
  o->progress_to_stdout = 0;
  o->huber_delta = 1.0;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->max_num_consecutive_invalid_steps = 5;
  o->jacobi_scaling = 1;
}

// TrustRegionMinimizer::Minimize (Ceres 1.13 control flow), monotonic steps, no bounds,
// no inner iterations — the defaults optimizer.cc:42-48 leaves untouched.
ORC_API int orc_ba_solve(msfm_ba_problem* P, const msfm_ba_options* options, msfm_ba_summary* sum) {
  if (!P || !options || !sum) return MSFM_E_INVAL;
  for (int o = 1; o < P->n_obs; o++) if (P->obs_pt[o] < P->obs_pt[o - 1]) return MSFM_E_INVAL;
  const auto t0 = std::chrono::steady_clock::now();
  Ba B;
  B.P = P; B.opt = *options;
  B.threads = std::max(1, std::min(options->num_threads, 256));   // Ceres: options.num_threads (optimizer.cc:46, slam_gps.cc:683)
  ba_setup(B);
  const auto t1 = std::chrono::steady_clock::now();
  sum->num_residuals = 2 * (int)B.obs.size() + 3 * (int)B.gps.size();
  sum->num_reduced_params = B.nred;
  sum->num_successful_steps = 0; sum->num_unsuccessful_steps = 0;
  int rows = 0;
  auto push = [&](const msfm_ba_iteration& it) {
    if (sum->iterations && rows < sum->iterations_capacity) sum->iterations[rows] = it;
    rows++;
  };
  const size_t ncc = B.scale_c.size(), nmc = B.scale_m.size(), npc = B.scale_p.size();
  auto active_x_norm = [&](const std::vector<double>& cam, const std::vector<double>& model, const std::vector<double>& pt) {
    double s = 0;
    for (int c = 0; c < B.Nc; c++) if (B.cam_slot[c] >= 0) for (int j = 0; j < 6; j++) s += sq(cam[6 * (size_t)c + j]);
    for (int m = 0; m < B.Nm; m++) if (B.model_slot[m] >= 0) for (int j = 0; j < 3; j++) s += sq(model[3 * (size_t)m + j]);
    for (int p = 0; p < B.Np; p++) if (B.pt_slot[p] >= 0) for (int j = 0; j < 3; j++) s += sq(pt[3 * (size_t)p + j]);
    return std::sqrt(s);
  };
  auto grad_max = [&]() {
    double g = 0;
    for (double v : B.grad_c) g = std::max(g, std::fabs(v));
    for (double v : B.grad_m) g = std::max(g, std::fabs(v));
    for (double v : B.grad_p) g = std::max(g, std::fabs(v));
    return g;
  };
  // LM strategy state
  double radius = B.opt.initial_trust_region_radius, decrease_factor = 2.0;
  bool reuse_diagonal = false;
  // IterationZero
  double x_cost = ba_evaluate(B, B.cam, B.model, B.pt, true);
  if (B.opt.jacobi_scaling) { ba_compute_jacobi_scaling(B); ba_scale_columns(B); }
  double x_norm = active_x_norm(B.cam, B.model, B.pt);
  msfm_ba_iteration it;
  memset(&it, 0, sizeof it);
  it.cost = x_cost; it.gradient_max_norm = grad_max(); it.trust_region_radius = radius;
  it.step_is_valid = 1; it.step_is_successful = 1;
  sum->initial_cost = x_cost;
  int iteration = 0, num_invalid = 0;
  int termination = 0;
  std::vector<double> ccam, cmodel, cpt;
  for (;;) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (it.step_is_successful && iteration > 0) sum->num_successful_steps++;
    it.trust_region_radius = radius;
    push(it);
    if (B.opt.progress_to_stdout)
      printf("%4d  cost %.6e  change %.3e  |grad| %.3e  |step| %.3e  rho %.3e  radius %.3e\n", iteration,
             it.cost, it.cost_change, it.gradient_max_norm, it.step_norm, it.relative_decrease, radius);
    if (iteration >= B.opt.max_num_iterations) { termination = MSFM_BA_NO_CONVERGENCE; break; }
    if (it.gradient_max_norm <= B.opt.gradient_tolerance) { termination = MSFM_BA_CONVERGENCE_GRADIENT; break; }
    if (radius <= B.opt.min_trust_region_radius) { termination = MSFM_BA_MIN_RADIUS; break; }
    const double prev_gmax = it.gradient_max_norm;
    memset(&it, 0, sizeof it);
    iteration++;
    // ComputeTrustRegionStep
    double model_cost_change = 0.0;
    const bool solved = ba_compute_step(B, radius, reuse_diagonal, &model_cost_change);
    reuse_diagonal = true;
    it.step_is_valid = solved && (model_cost_change > 0.0);
    if (!it.step_is_valid) {
      // HandleInvalidStep
      if (++num_invalid >= B.opt.max_num_consecutive_invalid_steps) {
        termination = MSFM_BA_FAILURE;  // Ceres returns without recording this iteration
        break;
      }
      radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;  // StepIsInvalid
      it.cost = x_cost; it.cost_change = 0; it.gradient_max_norm = prev_gmax; it.step_norm = 0; it.relative_decrease = 0;
      sum->num_unsuccessful_steps++;
      continue;
    }
    num_invalid = 0;
    // delta = step .* scaling ; candidate = x + delta
    ccam = B.cam; cmodel = B.model; cpt = B.pt;
    double dn = 0.0;
    for (int c = 0; c < B.Nc; c++) if (B.cam_slot[c] >= 0) for (int j = 0; j < 6; j++) {
      const size_t k = 6 * (size_t)B.cam_slot[c] + j; const double d = B.step_c[k] * B.scale_c[k];
      ccam[6 * (size_t)c + j] += d; dn += sq(ccam[6 * (size_t)c + j] - B.cam[6 * (size_t)c + j]);
    }
    for (int m = 0; m < B.Nm; m++) if (B.model_slot[m] >= 0) for (int j = 0; j < 3; j++) {
      const size_t k = 3 * (size_t)B.model_slot[m] + j; const double d = B.step_m[k] * B.scale_m[k];
      cmodel[3 * (size_t)m + j] += d; dn += sq(cmodel[3 * (size_t)m + j] - B.model[3 * (size_t)m + j]);
    }
    for (int p = 0; p < B.Np; p++) if (B.pt_slot[p] >= 0) for (int j = 0; j < 3; j++) {
      const size_t k = 3 * (size_t)B.pt_slot[p] + j; const double d = B.step_p[k] * B.scale_p[k];
      cpt[3 * (size_t)p + j] += d; dn += sq(cpt[3 * (size_t)p + j] - B.pt[3 * (size_t)p + j]);
    }
    const double cand_cost = ba_evaluate(B, ccam, cmodel, cpt, false);
    // ParameterToleranceReached
    it.step_norm = std::sqrt(dn);
    const double step_tol = B.opt.parameter_tolerance * (x_norm + B.opt.parameter_tolerance);
    if (it.step_norm <= step_tol) {
      termination = MSFM_BA_CONVERGENCE_PARAMETER; break;  // returns without recording the iteration
    }
    // FunctionToleranceReached
    it.cost_change = x_cost - cand_cost;
    if (std::fabs(it.cost_change) <= B.opt.function_tolerance * x_cost) {
      termination = MSFM_BA_CONVERGENCE_FUNCTION; break;
    }
    // IsStepSuccessful (monotonic TrustRegionStepEvaluator)
    it.relative_decrease = (x_cost - cand_cost) / model_cost_change;
    if (it.relative_decrease > B.opt.min_relative_decrease) {
      // HandleSuccessfulStep
      B.cam.swap(ccam); B.model.swap(cmodel); B.pt.swap(cpt);
      x_norm = active_x_norm(B.cam, B.model, B.pt);
      x_cost = ba_evaluate(B, B.cam, B.model, B.pt, true);
      if (B.opt.jacobi_scaling) ba_scale_columns(B);
      it.cost = x_cost; it.gradient_max_norm = grad_max(); it.step_is_successful = 1;
      // StepAccepted
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * it.relative_decrease - 1.0, 3));
      radius = std::min(B.opt.max_trust_region_radius, radius);
      decrease_factor = 2.0; reuse_diagonal = false;
    } else {
      // HandleUnsuccessfulStep + StepRejected
      it.step_is_successful = 0; it.cost = cand_cost; it.gradient_max_norm = prev_gmax;
      radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
      sum->num_unsuccessful_steps++;
    }
  }
  sum->termination = termination;
  sum->num_iterations = rows - 1;
  sum->final_cost = x_cost;
  memcpy(P->cam_pose, B.cam.data(), sizeof(double) * B.cam.size());
  memcpy(P->cam_model, B.model.data(), sizeof(double) * B.model.size());
  memcpy(P->point, B.pt.data(), sizeof(double) * B.pt.size());
  const auto t2 = std::chrono::steady_clock::now();
  sum->setup_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
  sum->solve_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
  (void)ncc; (void)nmc; (void)npc;
  return MSFM_OK;
}

// One linearisation + reduced system, exported for kernel-level parity tests:
// fills S (nred x nred, upper triangle valid, row-major), rhs (nred), and the point-space
// quantities; returns nred.  radius is the trust-region radius the LM diagonal uses.
ORC_API int orc_ba_reduced_system(msfm_ba_problem* P, const msfm_ba_options* options, double radius,
                                  double* S, double* rhs, int cap, double* cost, double* gmax) {
  Ba B; B.P = P; B.opt = *options;
  ba_setup(B);
  const double c = ba_evaluate(B, B.cam, B.model, B.pt, true);
  if (B.opt.jacobi_scaling) { ba_compute_jacobi_scaling(B); ba_scale_columns(B); }
  if (cost) *cost = c;
  if (gmax) {
    double g = 0;
    for (double v : B.grad_c) g = std::max(g, std::fabs(v));
    for (double v : B.grad_m) g = std::max(g, std::fabs(v));
    for (double v : B.grad_p) g = std::max(g, std::fabs(v));
    *gmax = g;
  }
  if (B.nred > cap) return -B.nred;
  double mcc;
  (void)ba_compute_step(B, radius, false, &mcc, /*assemble_only=*/true);
  const int n = B.nred;
  memcpy(S, B.lhs.data(), sizeof(double) * (size_t)n * n);
  for (int i = 0; i < n; i++) rhs[i] = B.rhs[i];
  return n;
}

// =====================================================================================
// Triangulation / reprojection — SfM/src/structure.cc
// =====================================================================================
namespace {

// Point3D::Reprojection, structure.cc:267-300.
double reprojection_mse(const msfm_tracks* T, int t, const double* X) {
  double mse = 0.0;
  int count = 0;
  for (int i = T->track_off[t]; i < T->track_off[t + 1]; i++) {
    const int c = T->track_cam[i];
    const double* R = T->cam_R + 9 * (size_t)c;
    const double* tt = T->cam_t + 3 * (size_t)c;
    const double* fk = T->cam_fk + 3 * (size_t)c;
    const double pc0 = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + tt[0];
    const double pc1 = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + tt[1];
    const double pc2 = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + tt[2];
    if (pc2 < 0) return 100000.0;  // :280-284
    const double x = pc0 / pc2, y = pc1 / pc2;
    const double r2 = x * x + y * y;
    const double distortion = 1.0 + r2 * (fk[1] + fk[2] * r2);
    const double u = fk[0] * distortion * x, v = fk[0] * distortion * y;
    const double du = u - T->track_xy[2 * (size_t)i], dv = v - T->track_xy[2 * (size_t)i + 1];
    mse += du * du + dv * dv;
    count++;
  }
  return mse / count;
}

// Point3D::SufficientTriangulationAngle, structure.cc:325-355.
bool sufficient_angle(const msfm_tracks* T, int t, const double* X, double th_angle) {
  const int b = T->track_off[t], e = T->track_off[t + 1], k = e - b;
  std::vector<double> d(3 * (size_t)k);
  for (int i = 0; i < k; i++) {
    const double* c = T->cam_c + 3 * (size_t)T->track_cam[b + i];
    double v[3] = {X[0] - c[0], X[1] - c[1], X[2] - c[2]};
    const double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    for (int j = 0; j < 3; j++) d[3 * i + j] = v[j] / n;
  }
  const double cos_min = std::cos(th_angle);
  for (int i = 0; i + 1 < k; i++)
    for (int j = i + 1; j < k; j++)
      if (d[3 * i] * d[3 * j] + d[3 * i + 1] * d[3 * j + 1] + d[3 * i + 2] * d[3 * j + 2] < cos_min) return true;
  return false;
}

// Eigen::LLT<Matrix4d>: fails (info != Success) when a pivot is not positive.
bool llt4_solve(const double* A, const double* b, double* x) {
  double L[16] = {0};
  for (int j = 0; j < 4; j++) {
    double d = A[j * 4 + j];
    for (int k = 0; k < j; k++) d -= L[j * 4 + k] * L[j * 4 + k];
    if (!(d > 0.0)) return false;
    L[j * 4 + j] = std::sqrt(d);
    for (int i = j + 1; i < 4; i++) {
      double s = A[i * 4 + j];
      for (int k = 0; k < j; k++) s -= L[i * 4 + k] * L[j * 4 + k];
      L[i * 4 + j] = s / L[j * 4 + j];
    }
  }
  double y[4];
  for (int i = 0; i < 4; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L[i * 4 + k] * y[k]; y[i] = s / L[i * 4 + i]; }
  for (int i = 3; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < 4; k++) s -= L[k * 4 + i] * x[k]; x[i] = s / L[i * 4 + i]; }
  return true;
}

}  // namespace

// Point3D::Trianglate2, structure.cc:211-265.
ORC_API int orc_triangulate_midpoint_batch(const msfm_tracks* T, double th_error, double th_angle,
                                           double* X, double* mse, uint8_t* ok) {
  for (int t = 0; t < T->n_tracks; t++) {
    double A[16] = {0}, b[4] = {0};
    for (int i = T->track_off[t]; i < T->track_off[t + 1]; i++) {
      const int c = T->track_cam[i];
      const double* R = T->cam_R + 9 * (size_t)c;
      const double* o = T->cam_c + 3 * (size_t)c;
      const double f = T->cam_fk[3 * (size_t)c];
      const double dc[3] = {T->track_xy[2 * (size_t)i], T->track_xy[2 * (size_t)i + 1], f};
      double dw[3] = {R[0] * dc[0] + R[3] * dc[1] + R[6] * dc[2], R[1] * dc[0] + R[4] * dc[1] + R[7] * dc[2],
                      R[2] * dc[0] + R[5] * dc[1] + R[8] * dc[2]};  // R^T dir_c
      const double n = std::sqrt(dw[0] * dw[0] + dw[1] * dw[1] + dw[2] * dw[2]);
      dw[0] /= n; dw[1] /= n; dw[2] /= n;
      const double dh[4] = {dw[0], dw[1], dw[2], 0.0};
      const double oh[4] = {o[0], o[1], o[2], 1.0};
      for (int r = 0; r < 4; r++) {
        double acc = 0.0;
        for (int q = 0; q < 4; q++) {
          const double at = (r == q ? 1.0 : 0.0) - dh[r] * dh[q];
          A[r * 4 + q] += at;
          acc += at * oh[q];
        }
        b[r] += acc;
      }
    }
    double tp[4];
    ok[t] = 0;
    mse[t] = 0.0;
    if (!llt4_solve(A, b, tp)) continue;  // :248-251 returns false, data untouched
    double* Xt = X + 3 * (size_t)t;
    Xt[0] = tp[0] / tp[3]; Xt[1] = tp[1] / tp[3]; Xt[2] = tp[2] / tp[3];
    mse[t] = reprojection_mse(T, t, Xt);
    ok[t] = !(std::sqrt(mse[t]) > th_error || !sufficient_angle(T, t, Xt, th_angle));
  }
  return MSFM_OK;
}

ORC_API int orc_reproject_mse_batch(const msfm_tracks* T, const double* X, double* mse) {
  for (int t = 0; t < T->n_tracks; t++) mse[t] = reprojection_mse(T, t, X + 3 * (size_t)t);
  return MSFM_OK;
}

// Point3D::Trianglate (DLT), structure.cc:163-209: rows :179-182, then the right
// singular vector of the smallest singular value (`A.jacobiSvd(ComputeFullV)` :187).
// Restated as a streaming Givens QR of the 2k x 4 design matrix followed by a one-sided
// Jacobi SVD of the 4x4 triangular factor (same V up to sign, which the
// dehomogenisation removes).
ORC_API int orc_triangulate_dlt_batch(const msfm_tracks* T, double th_error, double th_angle,
                                      double* X, double* mse, uint8_t* ok) {
  for (int t = 0; t < T->n_tracks; t++) {
    ok[t] = 0; mse[t] = 0.0;
    const int b = T->track_off[t], e = T->track_off[t + 1];
    if (e - b < 2) continue;  // :165-168
    double Rf[16] = {0};      // upper-triangular factor, row-major
    for (int i = b; i < e; i++) {
      const int c = T->track_cam[i];
      const double* R = T->cam_R + 9 * (size_t)c;
      const double* tt = T->cam_t + 3 * (size_t)c;
      const double f = T->cam_fk[3 * (size_t)c];
      const double x = T->track_xy[2 * (size_t)i], y = T->track_xy[2 * (size_t)i + 1];
      const double M0[4] = {R[0], R[1], R[2], tt[0]}, M1[4] = {R[3], R[4], R[5], tt[1]}, M2[4] = {R[6], R[7], R[8], tt[2]};
      double rows[2][4];
      for (int q = 0; q < 4; q++) { rows[0][q] = -M1[q] * f + M2[q] * y; rows[1][q] = M0[q] * f - M2[q] * x; }
      for (int rr = 0; rr < 2; rr++) {
        double* v = rows[rr];
        for (int j = 0; j < 4; j++) {  // annihilate v[j] against Rf[j][j]
          if (v[j] == 0.0) continue;
          const double a = Rf[j * 4 + j], bb = v[j];
          const double h = std::hypot(a, bb);
          const double cs = a / h, sn = bb / h;
          for (int q = j; q < 4; q++) {
            const double rj = Rf[j * 4 + q], vq = v[q];
            Rf[j * 4 + q] = cs * rj + sn * vq;
            v[q] = -sn * rj + cs * vq;
          }
        }
      }
    }
    // one-sided Jacobi on the columns of Rf; V accumulates the rotations
    double V[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int sweep = 0; sweep < 30; sweep++) {
      bool rotated = false;
      for (int p = 0; p < 3; p++)
        for (int q = p + 1; q < 4; q++) {
          double alpha = 0, beta = 0, gamma = 0;
          for (int r = 0; r < 4; r++) { alpha += Rf[r * 4 + p] * Rf[r * 4 + p]; beta += Rf[r * 4 + q] * Rf[r * 4 + q]; gamma += Rf[r * 4 + p] * Rf[r * 4 + q]; }
          if (std::fabs(gamma) <= 1e-15 * std::sqrt(alpha * beta) || gamma == 0.0) continue;
          rotated = true;
          const double zeta = (beta - alpha) / (2.0 * gamma);
          const double tn = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
          const double cs = 1.0 / std::sqrt(1.0 + tn * tn), sn = cs * tn;
          for (int r = 0; r < 4; r++) {
            const double rp = Rf[r * 4 + p], rq = Rf[r * 4 + q];
            Rf[r * 4 + p] = cs * rp - sn * rq; Rf[r * 4 + q] = sn * rp + cs * rq;
            const double vp = V[r * 4 + p], vq = V[r * 4 + q];
            V[r * 4 + p] = cs * vp - sn * vq; V[r * 4 + q] = sn * vp + cs * vq;
          }
        }
      if (!rotated) break;
    }
    int best = 0; double bn = 0;
    for (int q = 0; q < 4; q++) {
      double nn = 0; for (int r = 0; r < 4; r++) nn += Rf[r * 4 + q] * Rf[r * 4 + q];
      if (q == 0 || nn < bn) { bn = nn; best = q; }
    }
    double* Xt = X + 3 * (size_t)t;
    Xt[0] = V[0 * 4 + best] / V[3 * 4 + best]; Xt[1] = V[1 * 4 + best] / V[3 * 4 + best]; Xt[2] = V[2 * 4 + best] / V[3 * 4 + best];
    mse[t] = reprojection_mse(T, t, Xt);
    ok[t] = !(std::sqrt(mse[t]) > th_error || !sufficient_angle(T, t, Xt, th_angle));
  }
  return MSFM_OK;
}

// GeoVerification::GeoVerificationFundamental (closed form), geo_verification.cc:60-79.
ORC_API int orc_epipolar_filter(const float* pt1, const float* pt2, int n, const double* F, double th,
                                uint8_t* inlier) {
  for (int i = 0; i < n; i++) {
    const double x1 = pt1[2 * i], y1 = pt1[2 * i + 1], x2 = pt2[2 * i], y2 = pt2[2 * i + 1];
    double l0 = F[0] * x1 + F[1] * y1 + F[2], l1 = F[3] * x1 + F[4] * y1 + F[5], l2 = F[6] * x1 + F[7] * y1 + F[8];
    const double nn = std::sqrt(l0 * l0 + l1 * l1);
    l0 /= nn; l1 /= nn; l2 /= nn;
    const double dis = l0 * x2 + l1 * y2 + l2;
    inlier[i] = std::fabs(dis) < th;
  }
  return MSFM_OK;
}

// =====================================================================================
// GeoVerification::GeoVerificationFundamental, geo_verification.cc:30-58:
//   cv::findFundamentalMat(pt1, pt2, status, cv::FM_RANSAC, 3.0), then "fewer than 30 inliers -> false".
// OpenCV 2.4 is not in the tree; its FM_RANSAC is restated from the published algorithm
// (CvFMEstimator::run7Point / computeReprojError / CvModelEstimator2::runRANSAC / cvRANSACUpdateNumIters):
// a plain SEQUENTIAL loop here — draw 7 matches, solve, score every model, keep the best, shrink the
// iteration budget.  OpenCV's own random stream cannot be reproduced, so the sampler is the counter-based
// one documented in DESIGN.md (sample h of pair p depends only on (seed, p, h)).  Contraction is off in
// this section so that the arithmetic is the same sequence of IEEE operations everywhere.
// =====================================================================================
#pragma GCC push_options
#pragma GCC optimize("fp-contract=off")
namespace fr {
static uint64_t next64(uint64_t& s) {
  s += 0x9E3779B97F4A7C15ull;
  uint64_t z = s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static double det3(const double* m) {
  return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}
static void cofactors(const double* m, double* c) {
  c[0] = m[4] * m[8] - m[5] * m[7];    c[1] = -(m[3] * m[8] - m[5] * m[6]); c[2] = m[3] * m[7] - m[4] * m[6];
  c[3] = -(m[1] * m[8] - m[2] * m[7]); c[4] = m[0] * m[8] - m[2] * m[6];    c[5] = -(m[0] * m[7] - m[1] * m[6]);
  c[6] = m[1] * m[5] - m[2] * m[4];    c[7] = -(m[0] * m[5] - m[2] * m[3]); c[8] = m[0] * m[4] - m[1] * m[3];
}
static bool finite(double v) { return std::fabs(v) <= DBL_MAX; }
// one real root by bisection in the Cauchy bound + 2 Newton steps, the others from the deflated quadratic
static std::vector<double> cubic_roots(double c3, double c2, double c1, double c0) {
  std::vector<double> out;
  const double a = c2 / c3, b = c1 / c3, c = c0 / c3;
  if (!(finite(a) && finite(b) && finite(c))) {
    if (c2 != 0.0) {
      const double p = c1 / c2, q = c0 / c2, disc = p * p - 4.0 * q;
      if (!(disc >= 0.0)) return out;
      const double sq = std::sqrt(disc), t = -0.5 * (p + (p >= 0.0 ? sq : -sq));
      out.push_back(t);
      if (t != 0.0) out.push_back(q / t);
    } else if (c1 != 0.0) {
      out.push_back(-c0 / c1);
    }
    return out;
  }
  double R = std::fabs(a);
  if (std::fabs(b) > R) R = std::fabs(b);
  if (std::fabs(c) > R) R = std::fabs(c);
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
      if (finite(rn)) r = rn;
    }
  }
  out.push_back(r);
  const double p = a + r, q = b + r * p, disc = p * p - 4.0 * q;
  if (disc >= 0.0) {
    const double sq = std::sqrt(disc), t = -0.5 * (p + (p >= 0.0 ? sq : -sq));
    out.push_back(t);
    if (t != 0.0) out.push_back(q / t);
  }
  return out;
}
// models of sample h of pair `pair`
static std::vector<std::array<double, 9>> seven_point(uint64_t seed, int pair, int h, int N, const float* p1, const float* p2) {
  std::vector<std::array<double, 9>> models;
  uint64_t s = seed ^ ((uint64_t)pair * 0xD1342543DE82EF95ull) ^ ((uint64_t)h * 0xA24BAED4963EE407ull);
  int pick[7];
  for (int k = 0; k < 7; k++) {
    for (;;) {
      const int v = (int)(next64(s) % (uint64_t)N);
      bool dup = false;
      for (int j = 0; j < k; j++) dup = dup || pick[j] == v;
      if (!dup) { pick[k] = v; break; }
    }
  }
  double A[7][9];
  for (int k = 0; k < 7; k++) {
    const double x1 = p1[2 * pick[k]], y1 = p1[2 * pick[k] + 1], x2 = p2[2 * pick[k]], y2 = p2[2 * pick[k] + 1];
    const double row[9] = {x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.0};
    for (int c = 0; c < 9; c++) A[k][c] = row[c];
  }
  int perm[9];
  for (int c = 0; c < 9; c++) perm[c] = c;
  for (int i = 0; i < 7; i++) {  // Gauss-Jordan, full pivoting, first maximum in row-major order
    int pr = i, pc = i;
    double best = -1.0;
    for (int r = i; r < 7; r++)
      for (int c = i; c < 9; c++)
        if (std::fabs(A[r][c]) > best) { best = std::fabs(A[r][c]); pr = r; pc = c; }
    if (!(best > 0.0)) return models;
    if (pr != i) for (int c = 0; c < 9; c++) std::swap(A[i][c], A[pr][c]);
    if (pc != i) { for (int r = 0; r < 7; r++) std::swap(A[r][i], A[r][pc]); std::swap(perm[i], perm[pc]); }
    const double piv = A[i][i];
    for (int c = i; c < 9; c++) A[i][c] = A[i][c] / piv;
    for (int r = 0; r < 7; r++) {
      if (r == i) continue;
      const double f = A[r][i];
      for (int c = i; c < 9; c++) A[r][c] = A[r][c] - f * A[i][c];
    }
  }
  double f1[9], f2[9];
  for (int j = 0; j < 9; j++) {
    f1[perm[j]] = j < 7 ? -A[j][7] : (j == 7 ? 1.0 : 0.0);
    f2[perm[j]] = j < 7 ? -A[j][8] : (j == 8 ? 1.0 : 0.0);
  }
  double G[9], cf[9];
  for (int k = 0; k < 9; k++) G[k] = f1[k] - f2[k];
  const double c0 = det3(f2), c3 = det3(G);
  cofactors(f2, cf);
  double c1 = 0.0;
  for (int k = 0; k < 9; k++) c1 = c1 + cf[k] * G[k];
  cofactors(G, cf);
  double c2 = 0.0;
  for (int k = 0; k < 9; k++) c2 = c2 + cf[k] * f2[k];
  for (double lam : cubic_roots(c3, c2, c1, c0)) {
    std::array<double, 9> F;
    bool fin = true;
    for (int k = 0; k < 9; k++) { F[k] = f2[k] + lam * G[k]; fin = fin && finite(F[k]); }
    if (!fin) continue;
    const double mu = F[8];
    if (std::fabs(mu) > DBL_EPSILON) {
      const double inv = 1.0 / mu;
      for (int k = 0; k < 9; k++) F[k] = F[k] * inv;
    }
    models.push_back(F);
  }
  return models;
}
static bool is_inlier(const double* F, double x1, double y1, double x2, double y2, double th2) {
  double a = F[0] * x1 + F[1] * y1 + F[2], b = F[3] * x1 + F[4] * y1 + F[5], c = F[6] * x1 + F[7] * y1 + F[8];
  const double s2 = 1.0 / (a * a + b * b), d2 = x2 * a + y2 * b + c;
  a = F[0] * x2 + F[3] * y2 + F[6]; b = F[1] * x2 + F[4] * y2 + F[7]; c = F[2] * x2 + F[5] * y2 + F[8];
  const double s1 = 1.0 / (a * a + b * b), d1 = x1 * a + y1 * b + c;
  const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
  return (e1 > e2 ? e1 : e2) <= th2;
}
static int update_num_iters(double p, double ep, int model_points, int max_iters) {  // cvRANSACUpdateNumIters
  p = std::min(std::max(p, 0.0), 1.0);
  ep = std::min(std::max(ep, 0.0), 1.0);
  double num = std::max(1.0 - p, DBL_MIN);
  double denom = 1.0 - std::pow(1.0 - ep, model_points);
  if (denom < DBL_MIN) return 0;
  num = std::log(num);
  denom = std::log(denom);
  return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)std::lrint(num / denom);
}
}  // namespace fr

ORC_API int orc_fundamental_ransac(int n_pairs, const int* off, const float* pt1, const float* pt2, double threshold,
                                   double confidence, int max_iterations, int min_points, int min_inliers, uint64_t seed,
                                   double* Fout, uint8_t* inlier, int* n_inliers, uint8_t* ok) {
  const double th2 = threshold * threshold;
  for (int p = 0; p < n_pairs; p++) {
    const int o = off[p], N = off[p + 1] - o;
    const float* a = pt1 + 2 * (size_t)o;
    const float* b = pt2 + 2 * (size_t)o;
    for (int k = 0; k < 9; k++) Fout[9 * (size_t)p + k] = 0.0;
    for (int e = 0; e < N; e++) inlier[o + e] = 0;
    n_inliers[p] = 0;
    ok[p] = 0;
    if (N < min_points || N < 8) continue;  // pt1.size() < 30 -> return false
    int niters = max_iterations, best = 6;  // maxGoodCount starts below modelPoints
    std::array<double, 9> Fbest{};
    bool have = false;
    for (int h = 0; h < niters; h++) {
      for (const auto& F : fr::seven_point(seed, p, h, N, a, b)) {
        int good = 0;
        for (int e = 0; e < N; e++) good += fr::is_inlier(F.data(), a[2 * e], a[2 * e + 1], b[2 * e], b[2 * e + 1], th2);
        if (good > best) {
          best = good; Fbest = F; have = true;
          niters = std::min(niters, fr::update_num_iters(confidence, (double)(N - good) / N, 7, max_iterations));
        }
      }
    }
    if (!have) continue;
    int c = 0;
    for (int e = 0; e < N; e++) {
      const bool in = fr::is_inlier(Fbest.data(), a[2 * e], a[2 * e + 1], b[2 * e], b[2 * e + 1], th2);
      inlier[o + e] = in;
      c += in;
    }
    for (int k = 0; k < 9; k++) Fout[9 * (size_t)p + k] = Fbest[k];
    n_inliers[p] = c;
    ok[p] = c >= min_inliers;
  }
  return MSFM_OK;
}
#pragma GCC pop_options

// =====================================================================================
// Matching — exact brute-force 2-NN on squared L2 (the quantity FLANN's L2 functor
// returns to fine_matching_graph.cc:99) + the ratio tests of :116-133.
// The kd-tree of :72-81 is approximate (8 trees, 64 checks); "identical indices" is
// defined against the exact answer, ties by lower train index.
// =====================================================================================
template <typename Acc>
static void knn2_impl(const float* train, int n_train, const float* query, int n_query, int dim, int* ids,
                      float* sqd) {
  // queries are independent (the reference loops over them under OpenMP, fine_matching_graph.cc:87)
#pragma omp parallel for num_threads(g_orc_threads) schedule(static, 16) if (g_orc_threads > 1)
  for (int q = 0; q < n_query; q++) {
    const float* b = query + (size_t)q * dim;
    Acc d0 = std::numeric_limits<Acc>::infinity(), d1 = d0;
    int i0 = -1, i1 = -1;
    for (int t = 0; t < n_train; t++) {
      const float* a = train + (size_t)t * dim;
      Acc s = 0;
      if (sizeof(Acc) == sizeof(double)) {
        // the definition: binary64, k sequential, one fused multiply-add per term
        for (int k = 0; k < dim; k++) { const double d = (double)a[k] - (double)b[k]; s = (Acc)std::fma(d, d, (double)s); }
      } else {
#pragma omp simd reduction(+ : s)
        for (int k = 0; k < dim; k++) { const Acc d = (Acc)a[k] - (Acc)b[k]; s += d * d; }
      }
      if (s < d0) { d1 = d0; i1 = i0; d0 = s; i0 = t; }
      else if (s < d1) { d1 = s; i1 = t; }
    }
    ids[2 * (size_t)q] = i0; ids[2 * (size_t)q + 1] = i1;
    sqd[2 * (size_t)q] = (float)d0; sqd[2 * (size_t)q + 1] = (float)d1;
  }
}
// Definition: distances accumulated in binary64, rounded once to binary32.
ORC_API int orc_knn2_f32(const float* train, int n_train, const float* query, int n_query, int dim, int* ids,
                         float* sqdists) {
  if (n_train < 2) return MSFM_E_INVAL;
  knn2_impl<double>(train, n_train, query, n_query, dim, ids, sqdists);
  return MSFM_OK;
}
// binary32 accumulation as FLANN's L2<float> does — bit-identical to the definition on
// integer-valued descriptors in [0,255] (every partial sum < 2^24); used for CPU timing.
ORC_API int orc_knn2_f32_fast(const float* train, int n_train, const float* query, int n_query, int dim,
                              int* ids, float* sqdists) {
  if (n_train < 2) return MSFM_E_INVAL;
  knn2_impl<float>(train, n_train, query, n_query, dim, ids, sqdists);
  return MSFM_OK;
}
// fine_matching_graph.cc:116-133 on one pair: code[m] as documented in msfm.h.
ORC_API void orc_ratio_codes(const int* ids, const float* sqd, int n_query, float ratio_good, float ratio_all,
                             int32_t* code, int* n_all, int* n_good) {
  int na = 0, ng = 0;
  for (int m = 0; m < n_query; m++) {
    const float ratio = sqd[2 * (size_t)m] / sqd[2 * (size_t)m + 1];
    int32_t c = -1;
    // two independent tests (fine_matching_graph.cc:118-130)
    const bool good = ratio < ratio_good, all = ratio < ratio_all;
    if (good) ng++;
    if (all) na++;
    if (good || all) c = ids[2 * (size_t)m] | (good ? MSFM_MATCH_GOOD : 0) | (all ? 0 : MSFM_MATCH_NOT_ALL);
    code[m] = c;
  }
  *n_all = na; *n_good = ng;
}

// SLAMGPS::FeatureMatching step 2 on one pair (slam_gps.cc:466-503): the three checks behind the 2-NN search, on the arrays
// FLANN filled.  kp1 / kp2: positions of the features of id1 (train) / id2 (query) as cv::Point2f; F, H: the prior matrices
// of the pair, row-major.  code[m] = ids[2m] for a match that passes all three checks, -1 otherwise; *n_ratio = survivors
// of check1 (count - count1), *n_kept = matches[j].size() before the geo-verification (:509).
// cv::Mat products of doubles are plain sums from k = 0 (every product and every sum rounded on its own, no contraction);
// the thresholds are floats compared against binary64 distances (`40 * th_distance` is formed in float).
ORC_API __attribute__((optimize("fp-contract=off"))) void orc_slam_gate(const int* ids, const float* sqd, int n_query, const float* kp1,
                                                                      const float* kp2, const double* F, const double* H,
                                                                      float th_first_second_ratio, float th_epipolar, float th_distance,
                                                                      int32_t* code, int* n_ratio, int* n_kept) {
  int nr = 0, nk = 0;
  const float th_h = 40 * th_distance;   // slam_gps.cc:496
  for (int m = 0; m < n_query; m++) {
    code[m] = -1;
    const float ratio = sqd[2 * (size_t)m] / sqd[2 * (size_t)m + 1];   // :467
    if (ratio > th_first_second_ratio) continue;                        // check1 :470-473
    nr++;
    const int i0 = ids[2 * (size_t)m];
    const double pt1[3] = {(double)kp1[2 * (size_t)i0], (double)kp1[2 * (size_t)i0 + 1], 1.0};   // :476-478
    const double pt2[3] = {(double)kp2[2 * (size_t)m], (double)kp2[2 * (size_t)m + 1], 1.0};     // :479-481
    double l2[3], pt22[3];
    for (int r = 0; r < 3; r++) {                                       // cv::Mat l2 = Fs[i][j] * pt1  :482
      double s = 0.0;
      for (int k = 0; k < 3; k++) s = s + F[3 * r + k] * pt1[k];
      l2[r] = s;
    }
    double dot = 0.0;
    for (int k = 0; k < 3; k++) dot = dot + l2[k] * pt2[k];             // l2.dot(pt2)
    const double epi_dis = std::fabs(dot) / std::sqrt(l2[0] * l2[0] + l2[1] * l2[1]);   // :483 (pow(x, 2) = x * x)
    if (epi_dis > th_epipolar) continue;                                // check2 :484-487
    for (int r = 0; r < 3; r++) {                                       // cv::Mat pt22 = Hs[i][j] * pt1  :490
      double s = 0.0;
      for (int k = 0; k < 3; k++) s = s + H[3 * r + k] * pt1[k];
      pt22[r] = s;
    }
    const double sc = 1.0 / pt22[2];                                    // pt22 *= 1.0 / pt22(2)  :491
    const double dx = pt2[0] - pt22[0] * sc, dy = pt2[1] - pt22[1] * sc;   // :492-493
    const double homo_dis = std::sqrt(dx * dx + dy * dy);               // :494
    if (homo_dis > th_h) continue;                                      // check3 :495-498
    code[m] = i0;                                                       // matches[j].push_back({id0, m})  :503
    nk++;
  }
  *n_ratio = nr; *n_kept = nk;
}
