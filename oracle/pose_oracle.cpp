// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
//
// CPU restatement of the two pose initialisers of MetricSfM's incremental loop:
//
//   * AbsolutePoseEstimation::AbsolutePoseWithFocalLength   SfM/src/orientation/absolute_pose_estimation.cc:42-58
//       -> AbsolutePoseEPNP::EPNPRansac (absolute_pose_via_epnp.cc:103-139): `max_iter` minimal samples of 4
//          correspondences, EPnP on each (:142-185, compute_pose :472-519), the sample whose own mean-square
//          reprojection error is smallest wins; then AbsolutePoseEstimation::Error over all points (:67-103).
//   * RelativePoseEstimation::RelativePoseWithFocalLength   SfM/src/orientation/relative_pose_estimation.cc:91-120
//       -> EssentialMatrixFivePoints::FivePointEssentialMatrixRANSAC (essential_matrix_five_point.cc:30-92): 100
//          samples of 5 matches, Nister's solver through the 10x10 action matrix (:97-178), every real solution of
//          every sample scored by the Sampson sum over all matches (:333-349), smallest wins; then
//          RelativePoseFromEssentialMatrix::ReltivePoseFromEMatrix (relative_pose_from_essential_matrix.cc:33-79).
//
// Third-party arithmetic that is not in the tree is restated from its published algorithm and anchored on the
// reference's call sites: OpenCV 2.4 cvSVD / cvInvert(CV_SVD) / cvSolve(CV_SVD) / cvMulTransposed (one-sided Jacobi
// SVD of modules/core/src/lapack.cpp, JacobiSVDImpl_<double> and SVBkSb) for EPnP; Eigen 3 FullPivLU (kernel, solve)
// and the real eigenvalues of a 10x10 matrix (Householder Hessenberg + Francis double-shift QR, what EigenSolver
// runs) for the five-point solver.  The reference draws its samples with std::random_shuffle over std::rand
// (basic_funcs.cc:259-281), which is not reproducible outside its platform: as for the fundamental-matrix RANSAC the
// samples come from a counter-based generator keyed by (seed, problem, iteration), so parity with the reference is
// statistical while the HIP path and this file agree bit for bit (only + - * / sqrt on doubles, fixed order, no FMA).
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/msfm.h"

#define ORC_API extern "C" __attribute__((visibility("default")))

#pragma GCC push_options
#pragma GCC optimize("fp-contract=off")

namespace pose {

static inline uint64_t sm64(uint64_t& s) {
  s += 0x9E3779B97F4A7C15ull;
  uint64_t z = s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// K distinct indices of [0, n): the stand-in for math::RandVectorN (basic_funcs.cc:269-281).
static void sample(uint64_t seed, uint64_t salt, int problem, int iter, int n, int K, int* idx) {
  uint64_t s = seed ^ salt ^ ((uint64_t)problem * 0xD1342543DE82EF95ull) ^ ((uint64_t)iter * 0xA24BAED4963EE407ull);
  for (int k = 0; k < K; k++) {
    for (;;) {
      const int v = (int)(sm64(s) % (uint64_t)n);
      bool dup = false;
      for (int j = 0; j < k; j++) dup = dup || idx[j] == v;
      if (!dup) { idx[k] = v; break; }
    }
  }
}

struct Mat {  // row-major dense matrix
  int r, c;
  std::vector<double> d;
  Mat(int rows, int cols) : r(rows), c(cols), d((size_t)rows * cols, 0.0) {}
  double& operator()(int i, int j) { return d[(size_t)i * c + j]; }
  double operator()(int i, int j) const { return d[(size_t)i * c + j]; }
};

// -------------------------------------------------------------------------------------------------------------
// OpenCV 2.4 JacobiSVDImpl_<double>: one-sided (Hestenes) Jacobi on the ROWS of At (n rows of length m = the
// columns of A).  On return row i of At is the i-th left singular vector, W descending, Vt the right ones.
// `with_v` = OpenCV's Vt != NULL (always the case when cvSVD is asked for U or V).
// -------------------------------------------------------------------------------------------------------------
static void jacobi_svd(Mat& At, std::vector<double>& W, Mat* Vt, int m, int n) {
  const double eps = DBL_EPSILON * 10, minval = DBL_MIN;
  W.assign(n, 0.0);
  for (int i = 0; i < n; i++) {
    double sd = 0;
    for (int k = 0; k < m; k++) { const double t = At(i, k); sd += t * t; }
    W[i] = sd;
    if (Vt) { for (int k = 0; k < n; k++) (*Vt)(i, k) = 0; (*Vt)(i, i) = 1; }
  }
  const int max_iter = m > 30 ? m : 30;
  for (int iter = 0; iter < max_iter; iter++) {
    bool changed = false;
    for (int i = 0; i < n - 1; i++)
      for (int j = i + 1; j < n; j++) {
        double a = W[i], p = 0, b = W[j];
        for (int k = 0; k < m; k++) p += At(i, k) * At(j, k);
        if (std::fabs(p) <= eps * std::sqrt(a * b)) continue;
        p *= 2;
        const double beta = a - b, gamma = std::sqrt(p * p + beta * beta);  // hypot(p, beta)
        double c, s;
        if (beta < 0) {
          const double delta = (gamma - beta) * 0.5;
          s = std::sqrt(delta / gamma);
          c = p / (gamma * s * 2);
        } else {
          c = std::sqrt((gamma + beta) / (gamma * 2));
          s = p / (gamma * c * 2);
        }
        a = b = 0;
        for (int k = 0; k < m; k++) {
          const double t0 = c * At(i, k) + s * At(j, k);
          const double t1 = -s * At(i, k) + c * At(j, k);
          At(i, k) = t0; At(j, k) = t1;
          a += t0 * t0; b += t1 * t1;
        }
        W[i] = a; W[j] = b;
        changed = true;
        if (Vt)
          for (int k = 0; k < n; k++) {
            const double t0 = c * (*Vt)(i, k) + s * (*Vt)(j, k);
            const double t1 = -s * (*Vt)(i, k) + c * (*Vt)(j, k);
            (*Vt)(i, k) = t0; (*Vt)(j, k) = t1;
          }
      }
    if (!changed) break;
  }
  for (int i = 0; i < n; i++) {
    double sd = 0;
    for (int k = 0; k < m; k++) { const double t = At(i, k); sd += t * t; }
    W[i] = std::sqrt(sd);
  }
  for (int i = 0; i < n - 1; i++) {
    int j = i;
    for (int k = i + 1; k < n; k++)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      std::swap(W[i], W[j]);
      if (Vt) {
        for (int k = 0; k < m; k++) std::swap(At(i, k), At(j, k));
        for (int k = 0; k < n; k++) std::swap((*Vt)(i, k), (*Vt)(j, k));
      }
    }
  }
  if (!Vt) return;
  uint64_t rng = 0x12345678;  // cv::RNG, multiply-with-carry
  for (int i = 0; i < n; i++) {
    double sd = W[i];
    while (sd <= minval) {
      // a zero singular value: a +-1/m vector made orthogonal to the rows already there
      const double val0 = 1. / m;
      for (int k = 0; k < m; k++) {
        rng = (uint64_t)(unsigned)rng * 4164903690U + (unsigned)(rng >> 32);
        At(i, k) = ((unsigned)rng & 256) != 0 ? val0 : -val0;
      }
      for (int it = 0; it < 2; it++)
        for (int j = 0; j < i; j++) {
          sd = 0;
          for (int k = 0; k < m; k++) sd += At(i, k) * At(j, k);
          double asum = 0;
          for (int k = 0; k < m; k++) {
            const double t = At(i, k) - sd * At(j, k);
            At(i, k) = t;
            asum += std::fabs(t);
          }
          asum = asum ? 1 / asum : 0;
          for (int k = 0; k < m; k++) At(i, k) *= asum;
        }
      sd = 0;
      for (int k = 0; k < m; k++) { const double t = At(i, k); sd += t * t; }
      sd = std::sqrt(sd);
    }
    const double s = 1 / sd;
    for (int k = 0; k < m; k++) At(i, k) *= s;
  }
}

// cv::SVD::compute for m >= n: transposes, runs the Jacobi, hands back Ut (rows = left vectors) and Vt.
static void cv_svd(const Mat& A, std::vector<double>& W, Mat& Ut, Mat& Vt) {
  const int m = A.r, n = A.c;
  Ut = Mat(n, m);
  for (int i = 0; i < n; i++)
    for (int k = 0; k < m; k++) Ut(i, k) = A(k, i);
  Vt = Mat(n, n);
  jacobi_svd(Ut, W, &Vt, m, n);
}

// SVBkSb with one right-hand side: x = V diag(1/w) U^T b, singular values <= 2 eps sum(w) dropped.
static void cv_svd_solve(const Mat& A, const double* b, double* x) {
  std::vector<double> W;
  Mat Ut(0, 0), Vt(0, 0);
  cv_svd(A, W, Ut, Vt);
  const int m = A.r, n = A.c;
  for (int j = 0; j < n; j++) x[j] = 0;
  double threshold = 0;
  for (int i = 0; i < n; i++) threshold += W[i];
  threshold *= DBL_EPSILON * 2;
  for (int i = 0; i < n; i++) {
    double wi = W[i];
    if (std::fabs(wi) <= threshold) continue;
    wi = 1 / wi;
    double s = 0;
    for (int j = 0; j < m; j++) s += Ut(i, j) * b[j];
    s *= wi;
    for (int j = 0; j < n; j++) x[j] = x[j] + s * Vt(i, j);
  }
}

// cvInvert(CV_SVD) of a square matrix: the same back substitution with the identity as right-hand side.
static void cv_svd_invert(const Mat& A, Mat& inv) {
  std::vector<double> W;
  Mat Ut(0, 0), Vt(0, 0);
  cv_svd(A, W, Ut, Vt);
  const int n = A.c;
  inv = Mat(n, n);
  double threshold = 0;
  for (int i = 0; i < n; i++) threshold += W[i];
  threshold *= DBL_EPSILON * 2;
  std::vector<double> buf(n);
  for (int k = 0; k < n; k++) {
    double wi = W[k];
    if (std::fabs(wi) <= threshold) continue;
    wi = 1 / wi;
    for (int j = 0; j < n; j++) buf[j] = Ut(k, j) * wi;
    for (int i = 0; i < n; i++) {
      const double s = Vt(k, i);
      for (int j = 0; j < n; j++) inv(i, j) = inv(i, j) + s * buf[j];
    }
  }
}

// -------------------------------------------------------------------------------------------------------------
// EPnP (absolute_pose_via_epnp.cc:340-935), the classic Lepetit / Moreno-Noguer / Fua code the reference embeds.
// -------------------------------------------------------------------------------------------------------------
struct Epnp {
  double fu, fv, uc, vc;
  int n;
  std::vector<double> pws, us, alphas, pcs;
  double cws[4][3], ccs[4][3];

  static double dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
  static double dist2(const double* p1, const double* p2) {
    return (p1[0] - p2[0]) * (p1[0] - p2[0]) + (p1[1] - p2[1]) * (p1[1] - p2[1]) + (p1[2] - p2[2]) * (p1[2] - p2[2]);
  }

  void choose_control_points() {  // :340-378
    cws[0][0] = cws[0][1] = cws[0][2] = 0;
    for (int i = 0; i < n; i++)
      for (int j = 0; j < 3; j++) cws[0][j] += pws[3 * i + j];
    for (int j = 0; j < 3; j++) cws[0][j] /= n;
    Mat PW0(n, 3);
    for (int i = 0; i < n; i++)
      for (int j = 0; j < 3; j++) PW0(i, j) = pws[3 * i + j] - cws[0][j];
    Mat C(3, 3);  // cvMulTransposed(PW0, PW0tPW0, 1): upper triangle, then mirrored
    for (int i = 0; i < 3; i++)
      for (int j = i; j < 3; j++) {
        double s = 0;
        for (int k = 0; k < n; k++) s += PW0(k, i) * PW0(k, j);
        C(i, j) = s; C(j, i) = s;
      }
    std::vector<double> dc;
    Mat uct(0, 0), vt(0, 0);
    cv_svd(C, dc, uct, vt);
    for (int i = 1; i < 4; i++) {
      const double k = std::sqrt(dc[i - 1] / n);
      for (int j = 0; j < 3; j++) cws[i][j] = cws[0][j] + k * uct(i - 1, j);
    }
  }

  void compute_barycentric_coordinates() {  // :380-405
    Mat CC(3, 3), CCinv(0, 0);
    for (int i = 0; i < 3; i++)
      for (int j = 1; j < 4; j++) CC(i, j - 1) = cws[j][i] - cws[0][i];
    cv_svd_invert(CC, CCinv);
    for (int i = 0; i < n; i++) {
      const double* pi = &pws[3 * i];
      double* a = &alphas[4 * i];
      for (int j = 0; j < 3; j++)
        a[1 + j] = CCinv(j, 0) * (pi[0] - cws[0][0]) + CCinv(j, 1) * (pi[1] - cws[0][1]) + CCinv(j, 2) * (pi[2] - cws[0][2]);
      a[0] = 1.0 - a[1] - a[2] - a[3];
    }
  }

  void compute_L_6x10(const Mat& ut, double* l) {  // :775-817
    const double* v[4] = {&ut.d[12 * 11], &ut.d[12 * 10], &ut.d[12 * 9], &ut.d[12 * 8]};
    double dv[4][6][3];
    for (int i = 0; i < 4; i++) {
      int a = 0, b = 1;
      for (int j = 0; j < 6; j++) {
        dv[i][j][0] = v[i][3 * a] - v[i][3 * b];
        dv[i][j][1] = v[i][3 * a + 1] - v[i][3 * b + 1];
        dv[i][j][2] = v[i][3 * a + 2] - v[i][3 * b + 2];
        b++;
        if (b > 3) { a++; b = a + 1; }
      }
    }
    for (int i = 0; i < 6; i++) {
      double* row = l + 10 * i;
      row[0] = dot(dv[0][i], dv[0][i]);
      row[1] = 2.0 * dot(dv[0][i], dv[1][i]);
      row[2] = dot(dv[1][i], dv[1][i]);
      row[3] = 2.0 * dot(dv[0][i], dv[2][i]);
      row[4] = 2.0 * dot(dv[1][i], dv[2][i]);
      row[5] = dot(dv[2][i], dv[2][i]);
      row[6] = 2.0 * dot(dv[0][i], dv[3][i]);
      row[7] = 2.0 * dot(dv[1][i], dv[3][i]);
      row[8] = 2.0 * dot(dv[2][i], dv[3][i]);
      row[9] = dot(dv[3][i], dv[3][i]);
    }
  }

  void find_betas(const double* l, const double* rho, int which, double* betas) {  // :655-773
    static const int ncol[4] = {0, 4, 3, 5};
    static const int cols[4][5] = {{0, 0, 0, 0, 0}, {0, 1, 3, 6, 0}, {0, 1, 2, 0, 0}, {0, 1, 2, 3, 4}};
    const int nc = ncol[which];
    Mat L(6, nc);
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < nc; j++) L(i, j) = l[10 * i + cols[which][j]];
    double b[5];
    cv_svd_solve(L, rho, b);
    if (which == 1) {
      if (b[0] < 0) {
        betas[0] = std::sqrt(-b[0]);
        betas[1] = -b[1] / betas[0]; betas[2] = -b[2] / betas[0]; betas[3] = -b[3] / betas[0];
      } else {
        betas[0] = std::sqrt(b[0]);
        betas[1] = b[1] / betas[0]; betas[2] = b[2] / betas[0]; betas[3] = b[3] / betas[0];
      }
      return;
    }
    if (b[0] < 0) {
      betas[0] = std::sqrt(-b[0]);
      betas[1] = (b[2] < 0) ? std::sqrt(-b[2]) : 0.0;
    } else {
      betas[0] = std::sqrt(b[0]);
      betas[1] = (b[2] > 0) ? std::sqrt(b[2]) : 0.0;
    }
    if (b[1] < 0) betas[0] = -betas[0];
    betas[2] = which == 3 ? b[3] / betas[0] : 0.0;
    betas[3] = 0.0;
  }

  // Householder QR least squares of the 6x4 system, :870-960 (returns without touching X when a column is zero).
  static void qr_solve(double* A, double* b, double* X) {
    const int nr = 6, nc = 4;
    double A1[6], A2[6];
    for (int k = 0; k < nc; k++) {
      double eta = std::fabs(A[k * nc + k]);
      for (int i = k + 1; i < nr; i++) {  // the reference walks its pointer one row behind: rows k .. nr-2
        const double elt = std::fabs(A[(i - 1) * nc + k]);
        if (eta < elt) eta = elt;
      }
      if (eta == 0) return;
      const double inv_eta = 1. / eta;
      double sum = 0.0;
      for (int i = k; i < nr; i++) {
        A[i * nc + k] *= inv_eta;
        sum += A[i * nc + k] * A[i * nc + k];
      }
      double sigma = std::sqrt(sum);
      if (A[k * nc + k] < 0) sigma = -sigma;
      A[k * nc + k] += sigma;
      A1[k] = sigma * A[k * nc + k];
      A2[k] = -eta * sigma;
      for (int j = k + 1; j < nc; j++) {
        double s = 0;
        for (int i = k; i < nr; i++) s += A[i * nc + k] * A[i * nc + j];
        const double tau = s / A1[k];
        for (int i = k; i < nr; i++) A[i * nc + j] -= tau * A[i * nc + k];
      }
    }
    for (int j = 0; j < nc; j++) {
      double tau = 0;
      for (int i = j; i < nr; i++) tau += A[i * nc + j] * b[i];
      tau /= A1[j];
      for (int i = j; i < nr; i++) b[i] -= tau * A[i * nc + j];
    }
    X[nc - 1] = b[nc - 1] / A2[nc - 1];
    for (int i = nc - 2; i >= 0; i--) {
      double s = 0;
      for (int j = i + 1; j < nc; j++) s += A[i * nc + j] * X[j];
      X[i] = (b[i] - s) / A2[i];
    }
  }

  void gauss_newton(const double* l, const double* rho, double* betas) {  // :819-868
    double a[24], b[6], x[4] = {0, 0, 0, 0};
    for (int it = 0; it < 5; it++) {
      for (int i = 0; i < 6; i++) {
        const double* rowL = l + i * 10;
        double* rowA = a + i * 4;
        rowA[0] = 2 * rowL[0] * betas[0] + rowL[1] * betas[1] + rowL[3] * betas[2] + rowL[6] * betas[3];
        rowA[1] = rowL[1] * betas[0] + 2 * rowL[2] * betas[1] + rowL[4] * betas[2] + rowL[7] * betas[3];
        rowA[2] = rowL[3] * betas[0] + rowL[4] * betas[1] + 2 * rowL[5] * betas[2] + rowL[8] * betas[3];
        rowA[3] = rowL[6] * betas[0] + rowL[7] * betas[1] + rowL[8] * betas[2] + 2 * rowL[9] * betas[3];
        b[i] = rho[i] - (rowL[0] * betas[0] * betas[0] + rowL[1] * betas[0] * betas[1] + rowL[2] * betas[1] * betas[1] +
                         rowL[3] * betas[0] * betas[2] + rowL[4] * betas[1] * betas[2] + rowL[5] * betas[2] * betas[2] +
                         rowL[6] * betas[0] * betas[3] + rowL[7] * betas[1] * betas[3] + rowL[8] * betas[2] * betas[3] +
                         rowL[9] * betas[3] * betas[3]);
      }
      qr_solve(a, b, x);
      for (int i = 0; i < 4; i++) betas[i] += x[i];
    }
  }

  double compute_R_and_t(const Mat& ut, const double* betas, double R[3][3], double t[3]) {  // :636-653
    for (int i = 0; i < 4; i++) ccs[i][0] = ccs[i][1] = ccs[i][2] = 0.0;
    for (int i = 0; i < 4; i++) {
      const double* v = &ut.d[12 * (11 - i)];
      for (int j = 0; j < 4; j++)
        for (int k = 0; k < 3; k++) ccs[j][k] += betas[i] * v[3 * j + k];
    }
    for (int i = 0; i < n; i++) {
      const double* a = &alphas[4 * i];
      for (int j = 0; j < 3; j++) pcs[3 * i + j] = a[0] * ccs[0][j] + a[1] * ccs[1][j] + a[2] * ccs[2][j] + a[3] * ccs[3][j];
    }
    if (pcs[2] < 0.0) {  // solve_for_sign :620-634
      for (int i = 0; i < 4; i++)
        for (int j = 0; j < 3; j++) ccs[i][j] = -ccs[i][j];
      for (int i = 0; i < 3 * n; i++) pcs[i] = -pcs[i];
    }
    // estimate_R_and_t :555-611
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    for (int i = 0; i < n; i++)
      for (int j = 0; j < 3; j++) { pc0[j] += pcs[3 * i + j]; pw0[j] += pws[3 * i + j]; }
    for (int j = 0; j < 3; j++) { pc0[j] /= n; pw0[j] /= n; }
    Mat ABt(3, 3);
    for (int i = 0; i < n; i++) {
      const double* pc = &pcs[3 * i];
      const double* pw = &pws[3 * i];
      for (int j = 0; j < 3; j++) {
        ABt(j, 0) += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
        ABt(j, 1) += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
        ABt(j, 2) += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
      }
    }
    std::vector<double> D;
    Mat Ut(0, 0), Vt(0, 0);
    cv_svd(ABt, D, Ut, Vt);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) R[i][j] = Ut(0, i) * Vt(0, j) + Ut(1, i) * Vt(1, j) + Ut(2, i) * Vt(2, j);  // dot(U row i, V row j)
    const double det = R[0][0] * R[1][1] * R[2][2] + R[0][1] * R[1][2] * R[2][0] + R[0][2] * R[1][0] * R[2][1] -
                       R[0][2] * R[1][1] * R[2][0] - R[0][1] * R[1][0] * R[2][2] - R[0][0] * R[1][2] * R[2][1];
    if (det < 0) { R[2][0] = -R[2][0]; R[2][1] = -R[2][1]; R[2][2] = -R[2][2]; }
    t[0] = pc0[0] - dot(R[0], pw0);
    t[1] = pc0[1] - dot(R[1], pw0);
    t[2] = pc0[2] - dot(R[2], pw0);
    // reprojection_error :538-553
    double sum2 = 0.0;
    for (int i = 0; i < n; i++) {
      const double* pw = &pws[3 * i];
      const double Xc = dot(R[0], pw) + t[0], Yc = dot(R[1], pw) + t[1], inv_Zc = 1.0 / (dot(R[2], pw) + t[2]);
      const double ue = uc + fu * Xc * inv_Zc, ve = vc + fv * Yc * inv_Zc;
      const double u = us[2 * i], v = us[2 * i + 1];
      sum2 += std::sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    }
    return sum2 / n;
  }

  void compute_pose(double R[3][3], double t[3]) {  // :472-519
    choose_control_points();
    compute_barycentric_coordinates();
    Mat M(2 * n, 12);
    for (int i = 0; i < n; i++) {  // fill_M :407-424
      const double* as = &alphas[4 * i];
      const double u = us[2 * i], v = us[2 * i + 1];
      for (int k = 0; k < 4; k++) {
        M(2 * i, 3 * k) = as[k] * fu;
        M(2 * i, 3 * k + 1) = 0.0;
        M(2 * i, 3 * k + 2) = as[k] * (uc - u);
        M(2 * i + 1, 3 * k) = 0.0;
        M(2 * i + 1, 3 * k + 1) = as[k] * fv;
        M(2 * i + 1, 3 * k + 2) = as[k] * (vc - v);
      }
    }
    Mat MtM(12, 12);
    for (int i = 0; i < 12; i++)
      for (int j = i; j < 12; j++) {
        double s = 0;
        for (int k = 0; k < 2 * n; k++) s += M(k, i) * M(k, j);
        MtM(i, j) = s; MtM(j, i) = s;
      }
    std::vector<double> D;
    Mat ut(0, 0), vt(0, 0);
    cv_svd(MtM, D, ut, vt);
    double l[60], rho[6];
    compute_L_6x10(ut, l);
    rho[0] = dist2(cws[0], cws[1]); rho[1] = dist2(cws[0], cws[2]); rho[2] = dist2(cws[0], cws[3]);
    rho[3] = dist2(cws[1], cws[2]); rho[4] = dist2(cws[1], cws[3]); rho[5] = dist2(cws[2], cws[3]);
    double Betas[4][4], rep[4], Rs[4][3][3], ts[4][3];
    for (int w = 1; w <= 3; w++) {
      find_betas(l, rho, w, Betas[w]);
      gauss_newton(l, rho, Betas[w]);
      rep[w] = compute_R_and_t(ut, Betas[w], Rs[w], ts[w]);
    }
    int N = 1;
    if (rep[2] < rep[1]) N = 2;
    if (rep[3] < rep[N]) N = 3;
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) R[i][j] = Rs[N][i][j];
      t[i] = ts[N][i];
    }
  }
};

// AbsolutePoseEPNP::Error, :187-208: |K [R|t] X - x| per point with P = diag(f, f, 1) [R|t] formed first.
static double reproj_err(const double R[3][3], const double t[3], double f, const double* X, const double* x) {
  double P[3][4];
  for (int j = 0; j < 3; j++) { P[0][j] = f * R[0][j]; P[1][j] = f * R[1][j]; P[2][j] = 1.0 * R[2][j]; }
  P[0][3] = f * t[0]; P[1][3] = f * t[1]; P[2][3] = 1.0 * t[2];
  double pc[3];
  for (int i = 0; i < 3; i++) pc[i] = P[i][0] * X[0] + P[i][1] * X[1] + P[i][2] * X[2] + P[i][3] * 1.0;
  const double dx = pc[0] / pc[2] - x[0], dy = pc[1] / pc[2] - x[1];
  return std::sqrt(dx * dx + dy * dy);
}

// AbsolutePoseEPNP::EPNP, :142-185, on the 4 sampled correspondences.
static double epnp_minimal(const double* Xw, const double* x2d, const int* idx, double f, double R[3][3], double t[3]) {
  Epnp e;
  e.fu = e.fv = f; e.uc = e.vc = 0.0;
  e.n = 4;
  e.pws.resize(12); e.us.resize(8); e.alphas.resize(16); e.pcs.resize(12);
  for (int i = 0; i < 4; i++) {
    for (int j = 0; j < 3; j++) e.pws[3 * i + j] = Xw[3 * (size_t)idx[i] + j];
    for (int j = 0; j < 2; j++) e.us[2 * i + j] = x2d[2 * (size_t)idx[i] + j];
  }
  e.compute_pose(R, t);
  double mse = 0.0;
  int count = 0;
  for (int i = 0; i < 4; i++) {
    const double er = reproj_err(R, t, f, &e.pws[3 * i], &e.us[2 * i]);
    if (er < 10.0) { mse += er * er; count++; }
  }
  return count < 4 / 2 ? 100000.0 : std::sqrt(mse / count);
}

// -------------------------------------------------------------------------------------------------------------
// Five-point solver pieces (essential_matrix_five_point.cc:97-331).
// -------------------------------------------------------------------------------------------------------------
// Monomials of degree <= 3 in (x, y, z), graded reverse lexicographic as the reference lists them (:181-246):
// x^3 x^2y xy^2 y^3 x^2z xyz y^2z xz^2 yz^2 z^3 | x^2 xy y^2 xz yz z^2 | x y z | 1
static const int kMono[20][3] = {{3, 0, 0}, {2, 1, 0}, {1, 2, 0}, {0, 3, 0}, {2, 0, 1}, {1, 1, 1}, {0, 2, 1}, {1, 0, 2}, {0, 1, 2}, {0, 0, 3},
                                 {2, 0, 0}, {1, 1, 0}, {0, 2, 0}, {1, 0, 1}, {0, 1, 1}, {0, 0, 2}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}};
static int mono_index(int ex, int ey, int ez) {
  for (int i = 0; i < 20; i++)
    if (kMono[i][0] == ex && kMono[i][1] == ey && kMono[i][2] == ez) return i;
  return -1;
}
// A polynomial is its 20 coefficients.  Degree-1 operands occupy slots 16..19, degree-2 slots 10..19.
struct Poly {
  double c[20];
  Poly() { for (double& v : c) v = 0.0; }
};
static Poly lin(const double* a) { Poly p; for (int i = 0; i < 4; i++) p.c[16 + i] = a[i]; return p; }
// a * b, accumulated into the product slot in the order (i ascending, j ascending) over the non-trivial slots.
static Poly mul(const Poly& a, int a0, const Poly& b, int b0) {
  Poly o;
  for (int i = a0; i < 20; i++)
    for (int j = b0; j < 20; j++) {
      const int k = mono_index(kMono[i][0] + kMono[j][0], kMono[i][1] + kMono[j][1], kMono[i][2] + kMono[j][2]);
      o.c[k] = o.c[k] + a.c[i] * b.c[j];
    }
  return o;
}
static Poly add(const Poly& a, const Poly& b) { Poly o; for (int i = 0; i < 20; i++) o.c[i] = a.c[i] + b.c[i]; return o; }
static Poly sub(const Poly& a, const Poly& b) { Poly o; for (int i = 0; i < 20; i++) o.c[i] = a.c[i] - b.c[i]; return o; }
static Poly scale(double s, const Poly& a) { Poly o; for (int i = 0; i < 20; i++) o.c[i] = s * a.c[i]; return o; }

// Eigen::FullPivLU elimination on a row-major r x c array: pivot = first largest |a| in column-major scan order.
// Returns the number of non-zero pivots; perm_r / perm_c give the final row / column order.
static int fullpiv_lu(double* a, int r, int c, int ld, int* perm_r, int* perm_c, double* maxpivot) {
  const int size = r < c ? r : c;
  for (int i = 0; i < r; i++) perm_r[i] = i;
  for (int j = 0; j < c; j++) perm_c[j] = j;
  int nonzero = size;
  *maxpivot = 0.0;
  for (int k = 0; k < size; k++) {
    int pr = k, pc = k;
    double best = -1.0;
    for (int j = k; j < c; j++)
      for (int i = k; i < r; i++) {
        const double v = std::fabs(a[i * ld + j]);
        if (v > best) { best = v; pr = i; pc = j; }
      }
    if (best == 0.0) { nonzero = k; break; }
    if (best > *maxpivot) *maxpivot = best;
    if (pr != k) { for (int j = 0; j < c; j++) std::swap(a[k * ld + j], a[pr * ld + j]); std::swap(perm_r[k], perm_r[pr]); }
    if (pc != k) { for (int i = 0; i < r; i++) std::swap(a[i * ld + k], a[i * ld + pc]); std::swap(perm_c[k], perm_c[pc]); }
    if (k < r - 1)
      for (int i = k + 1; i < r; i++) a[i * ld + k] /= a[k * ld + k];
    if (k < size - 1)
      for (int i = k + 1; i < r; i++)
        for (int j = k + 1; j < c; j++) a[i * ld + j] -= a[i * ld + k] * a[k * ld + j];
  }
  return nonzero;
}

// Real eigenvalues of a 10x10 matrix, in the order of the diagonal of its real Schur form: Householder reduction to
// Hessenberg form, then the Francis double-shift QR iteration (EISPACK orthes + hqr, which Eigen's RealSchur
// follows, exceptional shifts at iterations 10 and 30).  Returns false when an eigenvalue needs more than 40 sweeps.
static bool real_eigenvalues10(const double* Ain, double* wr, double* wi) {
  const int nn = 10;
  double H[10][10], ort[10];
  for (int i = 0; i < nn; i++)
    for (int j = 0; j < nn; j++) H[i][j] = Ain[i * nn + j];
  const int low = 0, high = nn - 1;
  for (int m = low + 1; m <= high - 1; m++) {
    double scale = 0.0;
    for (int i = m; i <= high; i++) scale = scale + std::fabs(H[i][m - 1]);
    if (scale != 0.0) {
      double h = 0.0;
      for (int i = high; i >= m; i--) { ort[i] = H[i][m - 1] / scale; h += ort[i] * ort[i]; }
      double g = std::sqrt(h);
      if (ort[m] > 0) g = -g;
      h = h - ort[m] * g;
      ort[m] = ort[m] - g;
      for (int j = m; j < nn; j++) {
        double f = 0.0;
        for (int i = high; i >= m; i--) f += ort[i] * H[i][j];
        f = f / h;
        for (int i = m; i <= high; i++) H[i][j] -= f * ort[i];
      }
      for (int i = 0; i <= high; i++) {
        double f = 0.0;
        for (int j = high; j >= m; j--) f += ort[j] * H[i][j];
        f = f / h;
        for (int j = m; j <= high; j++) H[i][j] -= f * ort[j];
      }
      ort[m] = scale * ort[m];
      H[m][m - 1] = scale * g;
      for (int i = m + 1; i <= high; i++) H[i][m - 1] = 0.0;
    }
  }
  int n = nn - 1;
  const double eps = DBL_EPSILON;
  double exshift = 0.0, p = 0, q = 0, r = 0, s = 0, z = 0, w, x, y;
  double norm = 0.0;
  for (int i = 0; i < nn; i++)
    for (int j = (i - 1 > 0 ? i - 1 : 0); j < nn; j++) norm = norm + std::fabs(H[i][j]);
  int iter = 0;
  while (n >= low) {
    int l = n;
    while (l > low) {
      s = std::fabs(H[l - 1][l - 1]) + std::fabs(H[l][l]);
      if (s == 0.0) s = norm;
      if (std::fabs(H[l][l - 1]) < eps * s) break;
      l--;
    }
    if (l == n) {
      H[n][n] = H[n][n] + exshift;
      wr[n] = H[n][n]; wi[n] = 0.0;
      n--; iter = 0;
    } else if (l == n - 1) {
      w = H[n][n - 1] * H[n - 1][n];
      p = (H[n - 1][n - 1] - H[n][n]) / 2.0;
      q = p * p + w;
      z = std::sqrt(std::fabs(q));
      H[n][n] = H[n][n] + exshift;
      H[n - 1][n - 1] = H[n - 1][n - 1] + exshift;
      x = H[n][n];
      if (q >= 0) {
        z = p >= 0 ? p + z : p - z;
        wr[n - 1] = x + z;
        wr[n] = wr[n - 1];
        if (z != 0.0) wr[n] = x - w / z;
        wi[n - 1] = 0.0; wi[n] = 0.0;
      } else {
        wr[n - 1] = x + p; wr[n] = x + p;
        wi[n - 1] = z; wi[n] = -z;
      }
      n = n - 2; iter = 0;
    } else {
      x = H[n][n]; y = 0.0; w = 0.0;
      if (l < n) { y = H[n - 1][n - 1]; w = H[n][n - 1] * H[n - 1][n]; }
      if (iter == 10) {
        exshift += x;
        for (int i = low; i <= n; i++) H[i][i] -= x;
        s = std::fabs(H[n][n - 1]) + std::fabs(H[n - 1][n - 2]);
        x = y = 0.75 * s;
        w = -0.4375 * s * s;
      }
      if (iter == 30) {
        s = (y - x) / 2.0;
        s = s * s + w;
        if (s > 0) {
          s = std::sqrt(s);
          if (y < x) s = -s;
          s = x - w / ((y - x) / 2.0 + s);
          for (int i = low; i <= n; i++) H[i][i] -= s;
          exshift += s;
          x = y = w = 0.964;
        }
      }
      iter = iter + 1;
      if (iter > 40) return false;
      int m = n - 2;
      while (m >= l) {
        z = H[m][m];
        r = x - z; s = y - z;
        p = (r * s - w) / H[m + 1][m] + H[m][m + 1];
        q = H[m + 1][m + 1] - z - r - s;
        r = H[m + 2][m + 1];
        s = std::fabs(p) + std::fabs(q) + std::fabs(r);
        p = p / s; q = q / s; r = r / s;
        if (m == l) break;
        if (std::fabs(H[m][m - 1]) * (std::fabs(q) + std::fabs(r)) <
            eps * (std::fabs(p) * (std::fabs(H[m - 1][m - 1]) + std::fabs(z) + std::fabs(H[m + 1][m + 1]))))
          break;
        m--;
      }
      for (int i = m + 2; i <= n; i++) {
        H[i][i - 2] = 0.0;
        if (i > m + 2) H[i][i - 3] = 0.0;
      }
      for (int k = m; k <= n - 1; k++) {
        const bool notlast = (k != n - 1);
        if (k != m) {
          p = H[k][k - 1];
          q = H[k + 1][k - 1];
          r = notlast ? H[k + 2][k - 1] : 0.0;
          x = std::fabs(p) + std::fabs(q) + std::fabs(r);
          if (x == 0.0) continue;
          p = p / x; q = q / x; r = r / x;
        }
        s = std::sqrt(p * p + q * q + r * r);
        if (p < 0) s = -s;
        if (s != 0) {
          if (k != m) H[k][k - 1] = -s * x;
          else if (l != m) H[k][k - 1] = -H[k][k - 1];
          p = p + s;
          x = p / s; y = q / s; z = r / s;
          q = q / p; r = r / p;
          for (int j = k; j < nn; j++) {
            p = H[k][j] + q * H[k + 1][j];
            if (notlast) { p = p + r * H[k + 2][j]; H[k + 2][j] = H[k + 2][j] - p * z; }
            H[k][j] = H[k][j] - p * x;
            H[k + 1][j] = H[k + 1][j] - p * y;
          }
          const int imax = n < k + 3 ? n : k + 3;
          for (int i = 0; i <= imax; i++) {
            p = x * H[i][k] + y * H[i][k + 1];
            if (notlast) { p = p + z * H[i][k + 2]; H[i][k + 2] = H[i][k + 2] - p * r; }
            H[i][k] = H[i][k] - p;
            H[i][k + 1] = H[i][k + 1] - p * q;
          }
        }
      }
    }
  }
  return true;
}

// Null vector of (A - lambda I), A 10x10: nine steps of full-pivot elimination, the last unknown set to one,
// back substitution, unit length.  out[0..3] = its last four components (the x, y, z, 1 slots).
static void eigvec_tail(const double* A, double lambda, double* out) {
  double B[100];
  int pr[10], pc[10];
  for (int i = 0; i < 10; i++)
    for (int j = 0; j < 10; j++) B[i * 10 + j] = A[i * 10 + j] - (i == j ? lambda : 0.0);
  for (int i = 0; i < 10; i++) { pr[i] = i; pc[i] = i; }
  for (int k = 0; k < 9; k++) {
    int br = k, bc = k;
    double best = -1.0;
    for (int j = k; j < 10; j++)
      for (int i = k; i < 10; i++) {
        const double v = std::fabs(B[i * 10 + j]);
        if (v > best) { best = v; br = i; bc = j; }
      }
    if (br != k) for (int j = 0; j < 10; j++) std::swap(B[k * 10 + j], B[br * 10 + j]);
    if (bc != k) { for (int i = 0; i < 10; i++) std::swap(B[i * 10 + k], B[i * 10 + bc]); std::swap(pc[k], pc[bc]); }
    for (int i = k + 1; i < 10; i++) {
      const double f = B[i * 10 + k] / B[k * 10 + k];
      for (int j = k + 1; j < 10; j++) B[i * 10 + j] -= f * B[k * 10 + j];
    }
  }
  double y[10], v[10];
  y[9] = 1.0;
  for (int i = 8; i >= 0; i--) {
    double s = 0.0;
    for (int j = i + 1; j < 10; j++) s += B[i * 10 + j] * y[j];
    y[i] = -s / B[i * 10 + i];
  }
  for (int i = 0; i < 10; i++) v[pc[i]] = y[i];
  double nrm = 0.0;
  for (int i = 0; i < 10; i++) nrm += v[i] * v[i];
  nrm = std::sqrt(nrm);
  for (int i = 0; i < 4; i++) out[i] = v[6 + i] / nrm;
  (void)pr;
}

// FivePointEssentialMatrix, :97-178.  x1, x2: n >= 5 normalised points; appends the real solutions (E as its nine
// entries in column-major order, the layout of Eigen::Matrix3d::data()).  Returns their number.
static int five_point(const double* x1, const double* x2, int n, std::vector<double>& Es) {
  double null_space[9][4];
  {
    std::vector<double> A((size_t)n * 9);
    for (int i = 0; i < n; i++) {
      const double ax = x1[2 * i], ay = x1[2 * i + 1], bx = x2[2 * i], by = x2[2 * i + 1];
      double* r = &A[(size_t)i * 9];
      r[0] = bx * ax; r[1] = by * ax; r[2] = ax; r[3] = bx * ay; r[4] = by * ay; r[5] = ay; r[6] = bx; r[7] = by; r[8] = 1.0;
    }
    if (n == 5) {
      // FullPivLU::kernel(): with PAQ = LU and U = [U1 U2] (rank columns first), Ker = Q [-U1^-1 U2; I]
      int pr[5], pc[9];
      double maxpivot;
      const int nz = fullpiv_lu(A.data(), 5, 9, 9, pr, pc, &maxpivot);
      const double thr = maxpivot * (DBL_EPSILON * 5);
      int rank = 0;
      for (int i = 0; i < nz; i++) rank += std::fabs(A[i * 9 + i]) > thr;
      if (rank != 5) return 0;  // dimensionOfKernel() != 4
      for (int k = 0; k < 4; k++) {
        double y[5];
        for (int i = 4; i >= 0; i--) {  // upper-triangular solve U1 y = U2(:, k), column-oriented
          y[i] = A[i * 9 + 5 + k];
        }
        for (int i = 4; i >= 0; i--) {
          y[i] = y[i] / A[i * 9 + i];
          for (int j = 0; j < i; j++) y[j] -= A[j * 9 + i] * y[i];
        }
        for (int i = 0; i < 5; i++) null_space[pc[i]][k] = -y[i];
        for (int i = 5; i < 9; i++) null_space[pc[i]][k] = (i == 5 + k) ? 1.0 : 0.0;
      }
    } else {
      // JacobiSVD of A^T A, last four right singular vectors (restated with the one-sided Jacobi above)
      Mat AtA(9, 9);
      for (int i = 0; i < 9; i++)
        for (int j = 0; j < 9; j++) {
          double s = 0;
          for (int k = 0; k < n; k++) s += A[(size_t)k * 9 + i] * A[(size_t)k * 9 + j];
          AtA(i, j) = s;
        }
      std::vector<double> W;
      Mat Ut(0, 0), Vt(0, 0);
      cv_svd(AtA, W, Ut, Vt);
      for (int i = 0; i < 9; i++)
        for (int k = 0; k < 4; k++) null_space[i][k] = Vt(5 + k, i);
    }
  }
  // E(i, j) as a linear polynomial in (x, y, z, 1): row i + 3 j of the null space
  Poly e[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) e[i][j] = lin(null_space[i + 3 * j]);
  // 2 E E^T and its trace
  Poly eet[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      eet[i][j] = scale(2.0, add(add(mul(e[i][0], 16, e[j][0], 16), mul(e[i][1], 16, e[j][1], 16)), mul(e[i][2], 16, e[j][2], 16)));
  const Poly trace = add(add(eet[0][0], eet[1][1]), eet[2][2]);
  double C[10][20];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      const Poly row = sub(add(add(mul(eet[i][0], 10, e[0][j], 16), mul(eet[i][1], 10, e[1][j], 16)), mul(eet[i][2], 10, e[2][j], 16)),
                           scale(0.5, mul(trace, 10, e[i][j], 16)));
      for (int k = 0; k < 20; k++) C[3 * i + j][k] = row.c[k];
    }
  {
    const Poly det = add(add(mul(sub(mul(e[0][1], 16, e[1][2], 16), mul(e[0][2], 16, e[1][1], 16)), 10, e[2][0], 16),
                             mul(sub(mul(e[0][2], 16, e[1][0], 16), mul(e[0][0], 16, e[1][2], 16)), 10, e[2][1], 16)),
                         mul(sub(mul(e[0][0], 16, e[1][1], 16), mul(e[0][1], 16, e[1][0], 16)), 10, e[2][2], 16));
    for (int k = 0; k < 20; k++) C[9][k] = det.c[k];
  }
  // eliminated = C(:, 0:10)^-1 C(:, 10:20) by full-pivot LU (FullPivLU::solve)
  double LU[100], G[10][10];
  int pr[10], pc[10];
  for (int i = 0; i < 10; i++)
    for (int j = 0; j < 10; j++) LU[i * 10 + j] = C[i][j];
  double maxpivot;
  const int nz = fullpiv_lu(LU, 10, 10, 10, pr, pc, &maxpivot);
  const double thr = maxpivot * (DBL_EPSILON * 10);
  int rank = 0;
  for (int i = 0; i < nz; i++) rank += std::fabs(LU[i * 10 + i]) > thr;
  for (int col = 0; col < 10; col++) {
    double cvec[10];
    for (int i = 0; i < 10; i++) cvec[i] = C[pr[i]][10 + col];
    for (int k = 0; k < 10; k++)  // unit lower, column-oriented
      for (int i = k + 1; i < 10; i++) cvec[i] -= LU[i * 10 + k] * cvec[k];
    for (int k = rank - 1; k >= 0; k--) {
      cvec[k] = cvec[k] / LU[k * 10 + k];
      for (int i = 0; i < k; i++) cvec[i] -= LU[i * 10 + k] * cvec[k];
    }
    for (int i = 0; i < rank; i++) G[pc[i]][col] = cvec[i];
    for (int i = rank; i < 10; i++) G[pc[i]][col] = 0.0;
  }
  double Act[100];
  for (double& v : Act) v = 0.0;
  static const int src_row[6] = {0, 1, 2, 4, 5, 7};
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 10; j++) Act[i * 10 + j] = G[src_row[i]][j];
  Act[6 * 10 + 0] = -1.0; Act[7 * 10 + 1] = -1.0; Act[8 * 10 + 3] = -1.0; Act[9 * 10 + 6] = -1.0;
  double wr[10], wi[10];
  if (!real_eigenvalues10(Act, wr, wi)) return 0;
  int count = 0;
  for (int i = 0; i < 10; i++) {
    if (wi[i] != 0) continue;
    double tail[4];
    eigvec_tail(Act, wr[i], tail);
    for (int k = 0; k < 9; k++)
      Es.push_back(null_space[k][0] * tail[0] + null_space[k][1] * tail[1] + null_space[k][2] * tail[2] + null_space[k][3] * tail[3]);
    count++;
  }
  return count;
}

// EssentialMatrixFivePoints::Error, :333-349 — the Sampson sum.  E column-major.
static double sampson_sum(const double* E, const double* x1, const double* x2, int n) {
  double total = 0.0;
  for (int i = 0; i < n; i++) {
    const double ax = x1[2 * i], ay = x1[2 * i + 1], bx = x2[2 * i], by = x2[2 * i + 1];
    const double l0 = E[0] * ax + E[3] * ay + E[6] * 1.0, l1 = E[1] * ax + E[4] * ay + E[7] * 1.0, l2 = E[2] * ax + E[5] * ay + E[8] * 1.0;
    const double num = bx * l0 + by * l1 + 1.0 * l2;
    const double d0 = bx * E[0] + by * E[1] + 1.0 * E[2], d1 = bx * E[3] + by * E[4] + 1.0 * E[5];
    const double den = d0 * d0 + d1 * d1 + l0 * l0 + l1 * l1;
    total += num * num / den;
  }
  return total;
}

// DecomposeEssentialMatrix :81-104 and the four-hypothesis cheirality vote :33-79.  E column-major in, R row-major out.
static bool pose_from_E(const double* E, const double* x1, const double* x2, int n, double* Rout, double* tout) {
  Mat Em(3, 3);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Em(i, j) = E[i + 3 * j];
  std::vector<double> W;
  Mat Ut(0, 0), Vt(0, 0);
  cv_svd(Em, W, Ut, Vt);
  double U[3][3], V[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) { U[i][j] = Ut(j, i); V[i][j] = Vt(j, i); }
  auto det3 = [](double M[3][3]) {
    return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
           M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
  };
  if (det3(U) < 0) for (int i = 0; i < 3; i++) U[i][2] *= -1.0;
  if (det3(V) < 0) for (int i = 0; i < 3; i++) V[i][2] *= -1.0;
  // U w = [-u1 u0 u2], U w^T = [u1 -u0 u2]
  double Rh[4][3][3], th[4][3], t[3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      const double r1 = -U[i][1] * V[j][0] + U[i][0] * V[j][1] + U[i][2] * V[j][2];
      const double r2 = U[i][1] * V[j][0] + -U[i][0] * V[j][1] + U[i][2] * V[j][2];
      Rh[0][i][j] = r1; Rh[1][i][j] = r1; Rh[2][i][j] = r2; Rh[3][i][j] = r2;
    }
  const double tn = std::sqrt(U[0][2] * U[0][2] + U[1][2] * U[1][2] + U[2][2] * U[2][2]);
  for (int i = 0; i < 3; i++) t[i] = U[i][2] / tn;
  for (int h = 0; h < 4; h++) {
    const double sg = (h & 1) ? -1.0 : 1.0;
    for (int i = 0; i < 3; i++) th[h][i] = -(Rh[h][0][i] * (sg * t[0]) + Rh[h][1][i] * (sg * t[1]) + Rh[h][2][i] * (sg * t[2]));
  }
  int votes[4] = {0, 0, 0, 0};
  for (int p = 0; p < n; p++)
    for (int h = 0; h < 4; h++) {
      const double (*R)[3] = Rh[h];
      const double* tt = th[h];
      double c[3], d2[3];
      const double d1[3] = {x1[2 * p], x1[2 * p + 1], 1.0}, q[3] = {x2[2 * p], x2[2 * p + 1], 1.0};
      for (int i = 0; i < 3; i++) {
        c[i] = -(R[0][i] * tt[0] + R[1][i] * tt[1] + R[2][i] * tt[2]);
        d2[i] = R[0][i] * q[0] + R[1][i] * q[1] + R[2][i] * q[2];
      }
      const double d1sq = d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2], d2sq = d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2];
      const double d12 = d1[0] * d2[0] + d1[1] * d2[1] + d1[2] * d2[2];
      const double d1p = d1[0] * c[0] + d1[1] * c[1] + d1[2] * c[2], d2p = d2[0] * c[0] + d2[1] * c[1] + d2[2] * c[2];
      if (d2sq * d1p - d12 * d2p > 0 && d12 * d1p - d1sq * d2p > 0) { votes[h]++; break; }
    }
  int mx = votes[0];
  for (int h = 1; h < 4; h++) mx = votes[h] > mx ? votes[h] : mx;
  for (int h = 0; h < 4; h++)
    if (votes[h] == mx) {
      for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) Rout[3 * i + j] = Rh[h][i][j];
        tout[i] = th[h][i];
      }
      return true;
    }
  return false;
}

}  // namespace pose

// AbsolutePoseWithFocalLength for a batch of images.  off[n+1] delimits each image's 2D-3D correspondences.
// R row-major [n][9], t [n][3], errors [total] (1000.0 where the reprojection is >= 10 px), avg_error [n].
ORC_API int orc_epnp_ransac(int n_problems, const int* off, const double* pts_w, const double* pts_2d, const double* f, int max_iter,
                            uint64_t seed, double* Rout, double* tout, double* errors, double* avg_error, int* best_iter) {
  for (int p = 0; p < n_problems; p++) {
    const int o = off[p], N = off[p + 1] - o;
    const double* Xw = pts_w + 3 * (size_t)o;
    const double* x2 = pts_2d + 2 * (size_t)o;
    double R[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, t[3] = {0, 0, 0};
    double error = 1000000000.0;
    int best = -1;
    if (N >= 4)
      for (int it = 0; it < max_iter; it++) {  // EPNPRansac :103-139
        int idx[4];
        pose::sample(seed, 0x45506E50ull, p, it, N, 4, idx);
        double Rt[3][3], tt[3];
        const double e = pose::epnp_minimal(Xw, x2, idx, f[p], Rt, tt);
        if (e < error) {
          error = e; best = it;
          std::memcpy(R, Rt, sizeof R); std::memcpy(t, tt, sizeof t);
        }
      }
    // AbsolutePoseEstimation::Error :67-103
    double sum = 0.0;
    int count = 0;
    for (int i = 0; i < N; i++) {
      errors[o + i] = 1000.0;
      const double e = pose::reproj_err(R, t, f[p], Xw + 3 * (size_t)i, x2 + 2 * (size_t)i);
      if (std::fabs(e) < 10.0) { errors[o + i] = e; sum += e * e; count++; }
    }
    avg_error[p] = count == 0 ? 10000.0 : std::sqrt(sum / count);
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) Rout[9 * (size_t)p + 3 * i + j] = R[i][j];
      tout[3 * (size_t)p + i] = t[i];
    }
    if (best_iter) best_iter[p] = best;
  }
  return MSFM_OK;
}

// RelativePoseWithFocalLength for a batch of image pairs.  E row-major [n][9] (x_cur^T E x_ref = 0 on points / f),
// R row-major, t, ok.
ORC_API int orc_relpose_5pt(int n_pairs, const int* off, const double* pts_ref, const double* pts_cur, const double* f_ref,
                            const double* f_cur, int ransac_times, uint64_t seed, double* Eout, double* Rout, double* tout, uint8_t* ok,
                            int* n_candidates) {
  for (int p = 0; p < n_pairs; p++) {
    const int o = off[p], N = off[p + 1] - o;
    ok[p] = 0;
    for (int k = 0; k < 9; k++) { Eout[9 * (size_t)p + k] = 0.0; Rout[9 * (size_t)p + k] = 0.0; }
    for (int k = 0; k < 3; k++) tout[3 * (size_t)p + k] = 0.0;
    if (n_candidates) n_candidates[p] = 0;
    std::vector<double> x1(2 * (size_t)N), x2(2 * (size_t)N);
    for (int i = 0; i < 2 * N; i++) {
      x1[i] = pts_ref[2 * (size_t)o + i] / f_ref[p];
      x2[i] = pts_cur[2 * (size_t)o + i] / f_cur[p];
    }
    std::vector<double> Es;
    if (N < 5) continue;
    if (N < 10) {
      if (!pose::five_point(x1.data(), x2.data(), N, Es)) continue;
    } else {
      for (int it = 0; it < ransac_times; it++) {
        int idx[5];
        pose::sample(seed, 0x35707445ull, p, it, N, 5, idx);
        double a[10], b[10];
        for (int k = 0; k < 5; k++) { a[2 * k] = x1[2 * idx[k]]; a[2 * k + 1] = x1[2 * idx[k] + 1]; b[2 * k] = x2[2 * idx[k]]; b[2 * k + 1] = x2[2 * idx[k] + 1]; }
        pose::five_point(a, b, 5, Es);
      }
    }
    const int nE = (int)(Es.size() / 9);
    if (n_candidates) n_candidates[p] = nE;
    if (nE < 4) continue;
    double error_min = 1000000.0;
    int idx_min = 0;
    for (int i = 0; i < nE; i++) {
      const double e = pose::sampson_sum(&Es[9 * (size_t)i], x1.data(), x2.data(), N);
      if (e < error_min) { error_min = e; idx_min = i; }
    }
    const double* E = &Es[9 * (size_t)idx_min];
    if (!pose::pose_from_E(E, x1.data(), x2.data(), N, Rout + 9 * (size_t)p, tout + 3 * (size_t)p)) continue;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Eout[9 * (size_t)p + 3 * i + j] = E[i + 3 * j];
    ok[p] = 1;
  }
  return MSFM_OK;
}

// Test hooks: the linear-algebra restatements against numpy.
ORC_API void orc_test_jacobi_svd(const double* A, int m, int n, double* W, double* Ut, double* Vt) {
  pose::Mat a(m, n), ut(0, 0), vt(0, 0);
  for (int i = 0; i < m * n; i++) a.d[i] = A[i];
  std::vector<double> w;
  pose::cv_svd(a, w, ut, vt);
  for (int i = 0; i < n; i++) W[i] = w[i];
  for (int i = 0; i < n * m; i++) Ut[i] = ut.d[i];
  for (int i = 0; i < n * n; i++) Vt[i] = vt.d[i];
}
ORC_API int orc_test_eig10(const double* A, double* wr, double* wi) { return pose::real_eigenvalues10(A, wr, wi) ? 1 : 0; }
ORC_API int orc_test_five_point(const double* x1, const double* x2, int n, double* Es) {
  std::vector<double> v;
  const int c = pose::five_point(x1, x2, n, v);
  for (size_t i = 0; i < v.size(); i++) Es[i] = v[i];
  return c;
}
ORC_API void orc_test_epnp4(const double* Xw, const double* x2d, double f, double* R, double* t, double* err) {
  const int idx[4] = {0, 1, 2, 3};
  double Rm[3][3], tm[3];
  *err = pose::epnp_minimal(Xw, x2d, idx, f, Rm, tm);
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R[3 * i + j] = Rm[i][j]; t[i] = tm[i]; }
}
// EPnP on all n correspondences (the solver itself is size-generic; the reference's RANSAC only feeds it 4)
ORC_API double orc_test_epnp_n(const double* Xw, const double* x2d, int n, double f, double* R, double* t) {
  pose::Epnp e;
  e.fu = e.fv = f; e.uc = e.vc = 0.0;
  e.n = n;
  e.pws.assign(Xw, Xw + 3 * n); e.us.assign(x2d, x2d + 2 * n); e.alphas.resize(4 * n); e.pcs.resize(3 * n);
  double Rm[3][3], tm[3];
  e.compute_pose(Rm, tm);
  double s = 0;
  for (int i = 0; i < n; i++) s += pose::reproj_err(Rm, tm, f, Xw + 3 * i, x2d + 2 * i);
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R[3 * i + j] = Rm[i][j]; t[i] = tm[i]; }
  return s / n;
}
#pragma GCC pop_options
