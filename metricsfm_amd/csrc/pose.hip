// Batched pose initialisers for gfx950 — the step of the reference's incremental loop that precedes each bundle
// adjustment:
//   AbsolutePoseEstimation::AbsolutePoseWithFocalLength   SfM/src/orientation/absolute_pose_estimation.cc:42-58
//     -> AbsolutePoseEPNP::EPNPRansac                      SfM/src/orientation/absolute_pose_via_epnp.cc:103-139
//        (max_iter samples of 4 correspondences, EPnP on each :142-185 / compute_pose :472-519, the sample with the
//         smallest error over its own four points wins), then AbsolutePoseEstimation::Error over all points (:67-103);
//     called when an image is localised against the model, sfm_incremental.cc:646.
//   RelativePoseEstimation::RelativePoseWithFocalLength    SfM/src/orientation/relative_pose_estimation.cc:91-120
//     -> EssentialMatrixFivePoints::FivePointEssentialMatrixRANSAC  essential_matrix_five_point.cc:30-92
//        (100 samples of 5 matches, every real solution of every sample scored by the Sampson sum :333-349), then
//        RelativePoseFromEssentialMatrix::ReltivePoseFromEMatrix    relative_pose_from_essential_matrix.cc:33-104;
//     called for the seed pair, sfm_incremental.cc:309.
// The linear algebra behind them lives in OpenCV 2.4 (cvSVD, cvInvert, cvSolve: one-sided Jacobi SVD) and Eigen 3
// (FullPivLU, EigenSolver), neither in the tree: restated from their published algorithms.  std::random_shuffle is
// replaced by a counter-based sampler keyed by (seed, problem, iteration), so parity with the reference is
// statistical and parity with oracle/pose_oracle.cpp is exact: this file uses only + - * / sqrt on doubles, in a
// fixed order, with contraction off.
//
// One GPU thread = one minimal sample.  A sample's solver state (a 12x12 Jacobi SVD for EPnP, the 10x20 constraint
// matrix and a 10x10 QR iteration for the five-point solver) is a few KB of thread-private memory; all samples of
// all problems of a batch run side by side, a second kernel per problem replays the sequential selection.
#include "common.h"

#include <cfloat>
#include <cmath>

#pragma clang fp contract(off)

#define POSE_WAVE 64

__device__ static inline uint64_t pose_sm64(uint64_t& s) {
  s += 0x9E3779B97F4A7C15ull;
  uint64_t z = s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// K distinct indices of [0, n): stands in for math::RandVectorN (utils/basic_funcs.cc:269-281).
template <int K>
__device__ static inline void pose_sample(uint64_t seed, uint64_t salt, int problem, int iter, int n, int* idx) {
  uint64_t s = seed ^ salt ^ ((uint64_t)problem * 0xD1342543DE82EF95ull) ^ ((uint64_t)iter * 0xA24BAED4963EE407ull);
  for (int k = 0; k < K; k++) {
    for (;;) {
      const int v = (int)(pose_sm64(s) % (uint64_t)n);
      bool dup = false;
      for (int j = 0; j < k; j++) dup = dup || (idx[j] == v);
      if (!dup) { idx[k] = v; break; }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// One-sided Jacobi SVD on the rows of At[N][M] (OpenCV 2.4 JacobiSVDImpl_<double>).  Row i of At becomes the i-th
// left singular vector, W is sorted descending, Vt[N][N] (ACCV) receives the right singular vectors.
// ---------------------------------------------------------------------------------------------------------------
template <int M, int N, bool ACCV>
__device__ static void pose_jsvd(double* At, double* W, double* Vt) {
  const double eps = DBL_EPSILON * 10, minval = DBL_MIN;
  for (int i = 0; i < N; i++) {
    double sd = 0;
    for (int k = 0; k < M; k++) { const double t = At[i * M + k]; sd += t * t; }
    W[i] = sd;
    if (ACCV) { for (int k = 0; k < N; k++) Vt[i * N + k] = 0; Vt[i * N + i] = 1; }
  }
  const int max_iter = M > 30 ? M : 30;
  for (int iter = 0; iter < max_iter; iter++) {
    bool changed = false;
    for (int i = 0; i < N - 1; i++)
      for (int j = i + 1; j < N; j++) {
        double* Ai = At + i * M;
        double* Aj = At + j * M;
        double a = W[i], p = 0, b = W[j];
        for (int k = 0; k < M; k++) p += Ai[k] * Aj[k];
        if (fabs(p) <= eps * sqrt(a * b)) continue;
        p *= 2;
        const double beta = a - b, gamma = sqrt(p * p + beta * beta);
        double c, s;
        if (beta < 0) {
          const double delta = (gamma - beta) * 0.5;
          s = sqrt(delta / gamma);
          c = p / (gamma * s * 2);
        } else {
          c = sqrt((gamma + beta) / (gamma * 2));
          s = p / (gamma * c * 2);
        }
        a = b = 0;
        for (int k = 0; k < M; k++) {
          const double t0 = c * Ai[k] + s * Aj[k];
          const double t1 = -s * Ai[k] + c * Aj[k];
          Ai[k] = t0; Aj[k] = t1;
          a += t0 * t0; b += t1 * t1;
        }
        W[i] = a; W[j] = b;
        changed = true;
        if (ACCV) {
          double* Vi = Vt + i * N;
          double* Vj = Vt + j * N;
          for (int k = 0; k < N; k++) {
            const double t0 = c * Vi[k] + s * Vj[k];
            const double t1 = -s * Vi[k] + c * Vj[k];
            Vi[k] = t0; Vj[k] = t1;
          }
        }
      }
    if (!changed) break;
  }
  for (int i = 0; i < N; i++) {
    double sd = 0;
    for (int k = 0; k < M; k++) { const double t = At[i * M + k]; sd += t * t; }
    W[i] = sqrt(sd);
  }
  for (int i = 0; i < N - 1; i++) {
    int j = i;
    for (int k = i + 1; k < N; k++)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      { const double t = W[i]; W[i] = W[j]; W[j] = t; }
      for (int k = 0; k < M; k++) { const double t = At[i * M + k]; At[i * M + k] = At[j * M + k]; At[j * M + k] = t; }
      if (ACCV)
        for (int k = 0; k < N; k++) { const double t = Vt[i * N + k]; Vt[i * N + k] = Vt[j * N + k]; Vt[j * N + k] = t; }
    }
  }
  uint64_t rng = 0x12345678;
  for (int i = 0; i < N; i++) {
    double sd = W[i];
    int guard = 0;
    while (sd <= minval && guard++ < 64) {
      const double val0 = 1. / M;
      for (int k = 0; k < M; k++) {
        rng = (uint64_t)(unsigned)rng * 4164903690U + (unsigned)(rng >> 32);
        At[i * M + k] = ((unsigned)rng & 256) != 0 ? val0 : -val0;
      }
      for (int it = 0; it < 2; it++)
        for (int j = 0; j < i; j++) {
          sd = 0;
          for (int k = 0; k < M; k++) sd += At[i * M + k] * At[j * M + k];
          double asum = 0;
          for (int k = 0; k < M; k++) {
            const double t = At[i * M + k] - sd * At[j * M + k];
            At[i * M + k] = t;
            asum += fabs(t);
          }
          asum = asum ? 1 / asum : 0;
          for (int k = 0; k < M; k++) At[i * M + k] *= asum;
        }
      sd = 0;
      for (int k = 0; k < M; k++) { const double t = At[i * M + k]; sd += t * t; }
      sd = sqrt(sd);
    }
    const double s = 1 / sd;
    for (int k = 0; k < M; k++) At[i * M + k] *= s;
  }
}

// x = V diag(1/w) U^T b with singular values <= 2 eps sum(w) dropped (cvSolve(CV_SVD), one right-hand side).
// A is 6 x NC, row-major.
template <int NC>
__device__ static void pose_svd_solve6(const double* A, const double* b, double* x) {
  double At[NC * 6], W[NC], Vt[NC * NC];
  for (int i = 0; i < NC; i++)
    for (int k = 0; k < 6; k++) At[i * 6 + k] = A[k * NC + i];
  pose_jsvd<6, NC, true>(At, W, Vt);
  for (int j = 0; j < NC; j++) x[j] = 0;
  double threshold = 0;
  for (int i = 0; i < NC; i++) threshold += W[i];
  threshold *= DBL_EPSILON * 2;
  for (int i = 0; i < NC; i++) {
    double wi = W[i];
    if (fabs(wi) <= threshold) continue;
    wi = 1 / wi;
    double s = 0;
    for (int j = 0; j < 6; j++) s += At[i * 6 + j] * b[j];
    s *= wi;
    for (int j = 0; j < NC; j++) x[j] = x[j] + s * Vt[i * NC + j];
  }
}

__device__ static inline double pose_dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ static inline double pose_dist2(const double* p1, const double* p2) {
  return (p1[0] - p2[0]) * (p1[0] - p2[0]) + (p1[1] - p2[1]) * (p1[1] - p2[1]) + (p1[2] - p2[2]) * (p1[2] - p2[2]);
}

// |K [R|t] X - x| with P = diag(f, f, 1) [R|t] formed first (AbsolutePoseEPNP::Error :187-208, AbsolutePoseEstimation::Error :67-103)
__device__ static inline double pose_reproj_err(const double* R, const double* t, double f, const double* X, const double* x) {
  double P[3][4];
  for (int j = 0; j < 3; j++) { P[0][j] = f * R[j]; P[1][j] = f * R[3 + j]; P[2][j] = 1.0 * R[6 + j]; }
  P[0][3] = f * t[0]; P[1][3] = f * t[1]; P[2][3] = 1.0 * t[2];
  double pc[3];
  for (int i = 0; i < 3; i++) pc[i] = P[i][0] * X[0] + P[i][1] * X[1] + P[i][2] * X[2] + P[i][3] * 1.0;
  const double dx = pc[0] / pc[2] - x[0], dy = pc[1] / pc[2] - x[1];
  return sqrt(dx * dx + dy * dy);
}

// Householder QR least squares of the 6x4 Gauss-Newton system (qr_solve :870-960, incl. its one-row-late max search)
__device__ static void pose_qr_solve(double* A, double* b, double* X) {
  const int nr = 6, nc = 4;
  double A1[6], A2[6];
  for (int k = 0; k < nc; k++) {
    double eta = fabs(A[k * nc + k]);
    for (int i = k + 1; i < nr; i++) {
      const double elt = fabs(A[(i - 1) * nc + k]);
      if (eta < elt) eta = elt;
    }
    if (eta == 0) return;
    const double inv_eta = 1. / eta;
    double sum = 0.0;
    for (int i = k; i < nr; i++) {
      A[i * nc + k] *= inv_eta;
      sum += A[i * nc + k] * A[i * nc + k];
    }
    double sigma = sqrt(sum);
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

struct EpnpState {
  double pws[12], us[8], alphas[16], pcs[12], cws[4][3], ccs[4][3];
  double ut[144];  // rows = left singular vectors of M^T M
  double l[60], rho[6];
  double fu;
};

__device__ static void epnp_find_betas(const EpnpState& S, int which, double* betas) {
  double b[5] = {0, 0, 0, 0, 0};
  if (which == 1) {
    double L[24];
    for (int i = 0; i < 6; i++) { L[i * 4] = S.l[10 * i]; L[i * 4 + 1] = S.l[10 * i + 1]; L[i * 4 + 2] = S.l[10 * i + 3]; L[i * 4 + 3] = S.l[10 * i + 6]; }
    pose_svd_solve6<4>(L, S.rho, b);
    if (b[0] < 0) {
      betas[0] = sqrt(-b[0]);
      betas[1] = -b[1] / betas[0]; betas[2] = -b[2] / betas[0]; betas[3] = -b[3] / betas[0];
    } else {
      betas[0] = sqrt(b[0]);
      betas[1] = b[1] / betas[0]; betas[2] = b[2] / betas[0]; betas[3] = b[3] / betas[0];
    }
    return;
  }
  if (which == 2) {
    double L[18];
    for (int i = 0; i < 6; i++) { L[i * 3] = S.l[10 * i]; L[i * 3 + 1] = S.l[10 * i + 1]; L[i * 3 + 2] = S.l[10 * i + 2]; }
    pose_svd_solve6<3>(L, S.rho, b);
  } else {
    double L[30];
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 5; j++) L[i * 5 + j] = S.l[10 * i + j];
    pose_svd_solve6<5>(L, S.rho, b);
  }
  if (b[0] < 0) {
    betas[0] = sqrt(-b[0]);
    betas[1] = (b[2] < 0) ? sqrt(-b[2]) : 0.0;
  } else {
    betas[0] = sqrt(b[0]);
    betas[1] = (b[2] > 0) ? sqrt(b[2]) : 0.0;
  }
  if (b[1] < 0) betas[0] = -betas[0];
  betas[2] = which == 3 ? b[3] / betas[0] : 0.0;
  betas[3] = 0.0;
}

__device__ static void epnp_gauss_newton(const EpnpState& S, double* betas) {
  double a[24], b[6], x[4] = {0, 0, 0, 0};
  for (int it = 0; it < 5; it++) {
    for (int i = 0; i < 6; i++) {
      const double* rowL = S.l + i * 10;
      double* rowA = a + i * 4;
      rowA[0] = 2 * rowL[0] * betas[0] + rowL[1] * betas[1] + rowL[3] * betas[2] + rowL[6] * betas[3];
      rowA[1] = rowL[1] * betas[0] + 2 * rowL[2] * betas[1] + rowL[4] * betas[2] + rowL[7] * betas[3];
      rowA[2] = rowL[3] * betas[0] + rowL[4] * betas[1] + 2 * rowL[5] * betas[2] + rowL[8] * betas[3];
      rowA[3] = rowL[6] * betas[0] + rowL[7] * betas[1] + rowL[8] * betas[2] + 2 * rowL[9] * betas[3];
      b[i] = S.rho[i] - (rowL[0] * betas[0] * betas[0] + rowL[1] * betas[0] * betas[1] + rowL[2] * betas[1] * betas[1] +
                         rowL[3] * betas[0] * betas[2] + rowL[4] * betas[1] * betas[2] + rowL[5] * betas[2] * betas[2] +
                         rowL[6] * betas[0] * betas[3] + rowL[7] * betas[1] * betas[3] + rowL[8] * betas[2] * betas[3] +
                         rowL[9] * betas[3] * betas[3]);
    }
    pose_qr_solve(a, b, x);
    for (int i = 0; i < 4; i++) betas[i] += x[i];
  }
}

__device__ static double epnp_R_and_t(EpnpState& S, const double* betas, double* R, double* t) {
  for (int i = 0; i < 4; i++) S.ccs[i][0] = S.ccs[i][1] = S.ccs[i][2] = 0.0;
  for (int i = 0; i < 4; i++) {
    const double* v = S.ut + 12 * (11 - i);
    for (int j = 0; j < 4; j++)
      for (int k = 0; k < 3; k++) S.ccs[j][k] += betas[i] * v[3 * j + k];
  }
  for (int i = 0; i < 4; i++) {
    const double* a = S.alphas + 4 * i;
    for (int j = 0; j < 3; j++) S.pcs[3 * i + j] = a[0] * S.ccs[0][j] + a[1] * S.ccs[1][j] + a[2] * S.ccs[2][j] + a[3] * S.ccs[3][j];
  }
  if (S.pcs[2] < 0.0) {
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 3; j++) S.ccs[i][j] = -S.ccs[i][j];
    for (int i = 0; i < 12; i++) S.pcs[i] = -S.pcs[i];
  }
  double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 3; j++) { pc0[j] += S.pcs[3 * i + j]; pw0[j] += S.pws[3 * i + j]; }
  for (int j = 0; j < 3; j++) { pc0[j] /= 4; pw0[j] /= 4; }
  double ABt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    const double* pc = S.pcs + 3 * i;
    const double* pw = S.pws + 3 * i;
    for (int j = 0; j < 3; j++) {
      ABt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
      ABt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
      ABt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
    }
  }
  double Ut[9], D[3], Vt[9];
  for (int i = 0; i < 3; i++)
    for (int k = 0; k < 3; k++) Ut[i * 3 + k] = ABt[k * 3 + i];
  pose_jsvd<3, 3, true>(Ut, D, Vt);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R[3 * i + j] = Ut[i] * Vt[j] + Ut[3 + i] * Vt[3 + j] + Ut[6 + i] * Vt[6 + j];
  const double det = R[0] * R[4] * R[8] + R[1] * R[5] * R[6] + R[2] * R[3] * R[7] - R[2] * R[4] * R[6] - R[1] * R[3] * R[8] - R[0] * R[5] * R[7];
  if (det < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
  t[0] = pc0[0] - pose_dot3(R, pw0);
  t[1] = pc0[1] - pose_dot3(R + 3, pw0);
  t[2] = pc0[2] - pose_dot3(R + 6, pw0);
  double sum2 = 0.0;
  for (int i = 0; i < 4; i++) {
    const double* pw = S.pws + 3 * i;
    const double Xc = pose_dot3(R, pw) + t[0], Yc = pose_dot3(R + 3, pw) + t[1], inv_Zc = 1.0 / (pose_dot3(R + 6, pw) + t[2]);
    const double ue = 0.0 + S.fu * Xc * inv_Zc, ve = 0.0 + S.fu * Yc * inv_Zc;
    const double u = S.us[2 * i], v = S.us[2 * i + 1];
    sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
  }
  return sum2 / 4;
}

// AbsolutePoseEPNP::EPNP on four correspondences (:142-185): pose (R row-major, t) and the error over the four points.
__device__ static double epnp_minimal(EpnpState& S, double f, double* R, double* t) {
  S.fu = f;
  // choose_control_points :340-378
  S.cws[0][0] = S.cws[0][1] = S.cws[0][2] = 0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 3; j++) S.cws[0][j] += S.pws[3 * i + j];
  for (int j = 0; j < 3; j++) S.cws[0][j] /= 4;
  {
    double PW0[12], C[9], uct[9], dc[3];
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 3; j++) PW0[3 * i + j] = S.pws[3 * i + j] - S.cws[0][j];
    for (int i = 0; i < 3; i++)
      for (int j = i; j < 3; j++) {
        double s = 0;
        for (int k = 0; k < 4; k++) s += PW0[3 * k + i] * PW0[3 * k + j];
        C[3 * i + j] = s; C[3 * j + i] = s;
      }
    for (int i = 0; i < 3; i++)
      for (int k = 0; k < 3; k++) uct[i * 3 + k] = C[k * 3 + i];
    pose_jsvd<3, 3, false>(uct, dc, nullptr);
    for (int i = 1; i < 4; i++) {
      const double k = sqrt(dc[i - 1] / 4);
      for (int j = 0; j < 3; j++) S.cws[i][j] = S.cws[0][j] + k * uct[3 * (i - 1) + j];
    }
  }
  // compute_barycentric_coordinates :380-405 (cvInvert CV_SVD)
  {
    double Ut[9], W[3], Vt[9], inv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, buf[3];
    for (int i = 0; i < 3; i++)      // Ut = CC^T with CC(i, j-1) = cws[j][i] - cws[0][i]
      for (int k = 0; k < 3; k++) Ut[i * 3 + k] = S.cws[i + 1][k] - S.cws[0][k];
    pose_jsvd<3, 3, true>(Ut, W, Vt);
    double threshold = 0;
    for (int i = 0; i < 3; i++) threshold += W[i];
    threshold *= DBL_EPSILON * 2;
    for (int k = 0; k < 3; k++) {
      double wi = W[k];
      if (fabs(wi) <= threshold) continue;
      wi = 1 / wi;
      for (int j = 0; j < 3; j++) buf[j] = Ut[k * 3 + j] * wi;
      for (int i = 0; i < 3; i++) {
        const double s = Vt[k * 3 + i];
        for (int j = 0; j < 3; j++) inv[3 * i + j] = inv[3 * i + j] + s * buf[j];
      }
    }
    for (int i = 0; i < 4; i++) {
      const double* pi = S.pws + 3 * i;
      double* a = S.alphas + 4 * i;
      for (int j = 0; j < 3; j++)
        a[1 + j] = inv[3 * j] * (pi[0] - S.cws[0][0]) + inv[3 * j + 1] * (pi[1] - S.cws[0][1]) + inv[3 * j + 2] * (pi[2] - S.cws[0][2]);
      a[0] = 1.0 - a[1] - a[2] - a[3];
    }
  }
  // M (8 x 12), M^T M, its SVD :476-492
  {
    double M[96], D[12];
    for (int i = 0; i < 4; i++) {
      const double* as = S.alphas + 4 * i;
      const double u = S.us[2 * i], v = S.us[2 * i + 1];
      double* M1 = M + 24 * i;
      double* M2 = M1 + 12;
      for (int k = 0; k < 4; k++) {
        M1[3 * k] = as[k] * f; M1[3 * k + 1] = 0.0; M1[3 * k + 2] = as[k] * (0.0 - u);
        M2[3 * k] = 0.0; M2[3 * k + 1] = as[k] * f; M2[3 * k + 2] = as[k] * (0.0 - v);
      }
    }
    for (int i = 0; i < 12; i++)
      for (int j = i; j < 12; j++) {
        double s = 0;
        for (int k = 0; k < 8; k++) s += M[12 * k + i] * M[12 * k + j];
        S.ut[12 * i + j] = s; S.ut[12 * j + i] = s;
      }
    pose_jsvd<12, 12, false>(S.ut, D, nullptr);
  }
  // compute_L_6x10 :775-817, compute_rho
  {
    const double* v[4] = {S.ut + 12 * 11, S.ut + 12 * 10, S.ut + 12 * 9, S.ut + 12 * 8};
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
      double* row = S.l + 10 * i;
      row[0] = pose_dot3(dv[0][i], dv[0][i]);
      row[1] = 2.0 * pose_dot3(dv[0][i], dv[1][i]);
      row[2] = pose_dot3(dv[1][i], dv[1][i]);
      row[3] = 2.0 * pose_dot3(dv[0][i], dv[2][i]);
      row[4] = 2.0 * pose_dot3(dv[1][i], dv[2][i]);
      row[5] = pose_dot3(dv[2][i], dv[2][i]);
      row[6] = 2.0 * pose_dot3(dv[0][i], dv[3][i]);
      row[7] = 2.0 * pose_dot3(dv[1][i], dv[3][i]);
      row[8] = 2.0 * pose_dot3(dv[2][i], dv[3][i]);
      row[9] = pose_dot3(dv[3][i], dv[3][i]);
    }
    S.rho[0] = pose_dist2(S.cws[0], S.cws[1]); S.rho[1] = pose_dist2(S.cws[0], S.cws[2]); S.rho[2] = pose_dist2(S.cws[0], S.cws[3]);
    S.rho[3] = pose_dist2(S.cws[1], S.cws[2]); S.rho[4] = pose_dist2(S.cws[1], S.cws[3]); S.rho[5] = pose_dist2(S.cws[2], S.cws[3]);
  }
  double rep[4], Rs[4][9], ts[4][3];
  for (int w = 1; w <= 3; w++) {
    double betas[4];
    epnp_find_betas(S, w, betas);
    epnp_gauss_newton(S, betas);
    rep[w] = epnp_R_and_t(S, betas, Rs[w], ts[w]);
  }
  int N = 1;
  if (rep[2] < rep[1]) N = 2;
  if (rep[3] < rep[N]) N = 3;
  for (int i = 0; i < 9; i++) R[i] = Rs[N][i];
  for (int i = 0; i < 3; i++) t[i] = ts[N][i];
  double mse = 0.0;
  int count = 0;
  for (int i = 0; i < 4; i++) {
    const double er = pose_reproj_err(R, t, f, S.pws + 3 * i, S.us + 2 * i);
    if (er < 10.0) { mse += er * er; count++; }
  }
  return count < 2 ? 100000.0 : sqrt(mse / count);
}

// grid (ceil(H / 64), problems): thread = sample `it` of problem blockIdx.y.  hyp[(p * H + it) * 13] = {err, R[9], t[3]}
__global__ __launch_bounds__(POSE_WAVE) void k_epnp_hyp(int H, const int* __restrict__ off, const double* __restrict__ pts_w,
                                                        const double* __restrict__ pts_2d, const double* __restrict__ f, uint64_t seed,
                                                        double* __restrict__ hyp) {
  const int p = blockIdx.y, it = blockIdx.x * POSE_WAVE + threadIdx.x;
  if (it >= H) return;
  const int o = off[p], N = off[p + 1] - o;
  double* out = hyp + ((size_t)p * H + it) * 13;
  if (N < 4) { out[0] = 2000000000.0; return; }
  int idx[4];
  pose_sample<4>(seed, 0x45506E50ull, p, it, N, idx);
  EpnpState S;
  for (int i = 0; i < 4; i++) {
    for (int j = 0; j < 3; j++) S.pws[3 * i + j] = pts_w[3 * ((size_t)o + idx[i]) + j];
    for (int j = 0; j < 2; j++) S.us[2 * i + j] = pts_2d[2 * ((size_t)o + idx[i]) + j];
  }
  double R[9], t[3];
  const double e = epnp_minimal(S, f[p], R, t);
  out[0] = e;
  for (int i = 0; i < 9; i++) out[1 + i] = R[i];
  for (int i = 0; i < 3; i++) out[10 + i] = t[i];
}

// One workgroup per problem: the sample the sequential loop `if (error_temp < error)` from error = 1e9 keeps (the first
// smallest), then AbsolutePoseEstimation::Error over all points; the sum of squares is taken in point order.
__global__ __launch_bounds__(256) void k_epnp_select(int H, const int* __restrict__ off, const double* __restrict__ pts_w,
                                                     const double* __restrict__ pts_2d, const double* __restrict__ f,
                                                     const double* __restrict__ hyp, double* __restrict__ Rout, double* __restrict__ tout,
                                                     double* __restrict__ errors, double* __restrict__ avg_error, int* __restrict__ best_iter) {
  __shared__ double s_e[256];
  __shared__ int s_i[256];
  __shared__ double s_pose[12];
  const int p = blockIdx.x, tid = threadIdx.x;
  const int o = off[p], N = off[p + 1] - o;
  double be = 1000000000.0;
  int bi = -1;
  for (int it = tid; it < H; it += 256) {
    const double e = hyp[((size_t)p * H + it) * 13];
    if (e < be) { be = e; bi = it; }
  }
  s_e[tid] = be; s_i[tid] = bi;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) {
      const double e2 = s_e[tid + st];
      const int i2 = s_i[tid + st];
      if (i2 >= 0 && (s_i[tid] < 0 || e2 < s_e[tid] || (e2 == s_e[tid] && i2 < s_i[tid]))) { s_e[tid] = e2; s_i[tid] = i2; }
    }
    __syncthreads();
  }
  const int best = s_i[0];
  if (tid < 12) s_pose[tid] = best >= 0 ? hyp[((size_t)p * H + best) * 13 + 1 + tid] : 0.0;
  __syncthreads();
  double R[9], t[3];
  for (int i = 0; i < 9; i++) R[i] = s_pose[i];
  for (int i = 0; i < 3; i++) t[i] = s_pose[9 + i];
  const double fp = f[p];
  for (int i = tid; i < N; i += 256) {
    const double e = pose_reproj_err(R, t, fp, pts_w + 3 * ((size_t)o + i), pts_2d + 2 * ((size_t)o + i));
    errors[o + i] = fabs(e) < 10.0 ? e : 1000.0;
  }
  __syncthreads();
  if (tid == 0) {
    double sum = 0.0;
    int count = 0;
    for (int i = 0; i < N; i++) {
      const double e = errors[o + i];
      if (e < 10.0) { sum += e * e; count++; }
    }
    avg_error[p] = count == 0 ? 10000.0 : sqrt(sum / count);
    for (int i = 0; i < 9; i++) Rout[9 * (size_t)p + i] = R[i];
    for (int i = 0; i < 3; i++) tout[3 * (size_t)p + i] = t[i];
    best_iter[p] = best;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Five-point solver (essential_matrix_five_point.cc:97-331)
// ---------------------------------------------------------------------------------------------------------------
// Monomials of degree <= 3 in (x, y, z) in the reference's order (:181-246):
//   x^3 x^2y xy^2 y^3 x^2z xyz y^2z xz^2 yz^2 z^3 | x^2 xy y^2 xz yz z^2 | x y z | 1
// i.e. by degree (descending), then by the power of z, then by the power of y.
__device__ static inline int p5_slot(int ey, int ez, int d) {
  const int base = d == 3 ? 0 : d == 2 ? 10 : d == 1 ? 16 : 19;
  return base + ez * (d + 1) - (ez * (ez - 1)) / 2 + ey;
}
__device__ static inline void p5_exps(int slot, int& ey, int& ez, int& d) {
  int r;
  if (slot < 10) { d = 3; r = slot; } else if (slot < 16) { d = 2; r = slot - 10; } else if (slot < 19) { d = 1; r = slot - 16; } else { d = 0; r = 0; }
  ez = 0;
  while (r >= d - ez + 1) { r -= d - ez + 1; ez++; }
  ey = r;
}
// o = a * b; a non-zero from slot a0 on, b from b0 on; products accumulated in (i ascending, j ascending) order
__device__ static void p5_mul(const double* a, int a0, const double* b, int b0, double* o) {
  for (int k = 0; k < 20; k++) o[k] = 0.0;
  for (int i = a0; i < 20; i++) {
    int eyi, ezi, di;
    p5_exps(i, eyi, ezi, di);
    for (int j = b0; j < 20; j++) {
      int eyj, ezj, dj;
      p5_exps(j, eyj, ezj, dj);
      const int k = p5_slot(eyi + eyj, ezi + ezj, di + dj);
      o[k] = o[k] + a[i] * b[j];
    }
  }
}
__device__ static inline void p5_lin(const double* a4, double* o) { for (int k = 0; k < 16; k++) o[k] = 0.0; for (int k = 0; k < 4; k++) o[16 + k] = a4[k]; }
__device__ static inline void p5_add(const double* a, const double* b, double* o) { for (int k = 0; k < 20; k++) o[k] = a[k] + b[k]; }
__device__ static inline void p5_sub(const double* a, const double* b, double* o) { for (int k = 0; k < 20; k++) o[k] = a[k] - b[k]; }
__device__ static inline void p5_scale(double s, const double* a, double* o) { for (int k = 0; k < 20; k++) o[k] = s * a[k]; }

// Eigen::FullPivLU elimination, row-major r x c (ld): pivot = first largest |a| in column-major scan order
__device__ static int p5_fullpiv_lu(double* a, int r, int c, int ld, int* perm_r, int* perm_c, double* maxpivot) {
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
        const double v = fabs(a[i * ld + j]);
        if (v > best) { best = v; pr = i; pc = j; }
      }
    if (best == 0.0) { nonzero = k; break; }
    if (best > *maxpivot) *maxpivot = best;
    if (pr != k) {
      for (int j = 0; j < c; j++) { const double t = a[k * ld + j]; a[k * ld + j] = a[pr * ld + j]; a[pr * ld + j] = t; }
      const int t = perm_r[k]; perm_r[k] = perm_r[pr]; perm_r[pr] = t;
    }
    if (pc != k) {
      for (int i = 0; i < r; i++) { const double t = a[i * ld + k]; a[i * ld + k] = a[i * ld + pc]; a[i * ld + pc] = t; }
      const int t = perm_c[k]; perm_c[k] = perm_c[pc]; perm_c[pc] = t;
    }
    if (k < r - 1)
      for (int i = k + 1; i < r; i++) a[i * ld + k] /= a[k * ld + k];
    if (k < size - 1)
      for (int i = k + 1; i < r; i++)
        for (int j = k + 1; j < c; j++) a[i * ld + j] -= a[i * ld + k] * a[k * ld + j];
  }
  return nonzero;
}

// Real eigenvalues of a 10x10 matrix in Schur-diagonal order: Householder Hessenberg + Francis double-shift QR
// (EISPACK orthes / hqr, exceptional shifts at sweeps 10 and 30, at most 40 sweeps per eigenvalue).
#define HH(i, j) H[(i) * 10 + (j)]
__device__ static bool p5_real_eigenvalues10(const double* Ain, double* wr, double* wi) {
  const int nn = 10;
  double H[100], ort[10];
  for (int i = 0; i < 100; i++) H[i] = Ain[i];
  const int low = 0, high = nn - 1;
  for (int m = low + 1; m <= high - 1; m++) {
    double scale = 0.0;
    for (int i = m; i <= high; i++) scale = scale + fabs(HH(i, m - 1));
    if (scale != 0.0) {
      double h = 0.0;
      for (int i = high; i >= m; i--) { ort[i] = HH(i, m - 1) / scale; h += ort[i] * ort[i]; }
      double g = sqrt(h);
      if (ort[m] > 0) g = -g;
      h = h - ort[m] * g;
      ort[m] = ort[m] - g;
      for (int j = m; j < nn; j++) {
        double f = 0.0;
        for (int i = high; i >= m; i--) f += ort[i] * HH(i, j);
        f = f / h;
        for (int i = m; i <= high; i++) HH(i, j) -= f * ort[i];
      }
      for (int i = 0; i <= high; i++) {
        double f = 0.0;
        for (int j = high; j >= m; j--) f += ort[j] * HH(i, j);
        f = f / h;
        for (int j = m; j <= high; j++) HH(i, j) -= f * ort[j];
      }
      ort[m] = scale * ort[m];
      HH(m, m - 1) = scale * g;
      for (int i = m + 1; i <= high; i++) HH(i, m - 1) = 0.0;
    }
  }
  int n = nn - 1;
  const double eps = DBL_EPSILON;
  double exshift = 0.0, p = 0, q = 0, r = 0, s = 0, z = 0, w, x, y;
  double norm = 0.0;
  for (int i = 0; i < nn; i++)
    for (int j = (i - 1 > 0 ? i - 1 : 0); j < nn; j++) norm = norm + fabs(HH(i, j));
  int iter = 0;
  while (n >= low) {
    int l = n;
    while (l > low) {
      s = fabs(HH(l - 1, l - 1)) + fabs(HH(l, l));
      if (s == 0.0) s = norm;
      if (fabs(HH(l, l - 1)) < eps * s) break;
      l--;
    }
    if (l == n) {
      HH(n, n) = HH(n, n) + exshift;
      wr[n] = HH(n, n); wi[n] = 0.0;
      n--; iter = 0;
    } else if (l == n - 1) {
      w = HH(n, n - 1) * HH(n - 1, n);
      p = (HH(n - 1, n - 1) - HH(n, n)) / 2.0;
      q = p * p + w;
      z = sqrt(fabs(q));
      HH(n, n) = HH(n, n) + exshift;
      HH(n - 1, n - 1) = HH(n - 1, n - 1) + exshift;
      x = HH(n, n);
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
      x = HH(n, n); y = 0.0; w = 0.0;
      if (l < n) { y = HH(n - 1, n - 1); w = HH(n, n - 1) * HH(n - 1, n); }
      if (iter == 10) {
        exshift += x;
        for (int i = low; i <= n; i++) HH(i, i) -= x;
        s = fabs(HH(n, n - 1)) + fabs(HH(n - 1, n - 2));
        x = y = 0.75 * s;
        w = -0.4375 * s * s;
      }
      if (iter == 30) {
        s = (y - x) / 2.0;
        s = s * s + w;
        if (s > 0) {
          s = sqrt(s);
          if (y < x) s = -s;
          s = x - w / ((y - x) / 2.0 + s);
          for (int i = low; i <= n; i++) HH(i, i) -= s;
          exshift += s;
          x = y = w = 0.964;
        }
      }
      iter = iter + 1;
      if (iter > 40) return false;
      int m = n - 2;
      while (m >= l) {
        z = HH(m, m);
        r = x - z; s = y - z;
        p = (r * s - w) / HH(m + 1, m) + HH(m, m + 1);
        q = HH(m + 1, m + 1) - z - r - s;
        r = HH(m + 2, m + 1);
        s = fabs(p) + fabs(q) + fabs(r);
        p = p / s; q = q / s; r = r / s;
        if (m == l) break;
        if (fabs(HH(m, m - 1)) * (fabs(q) + fabs(r)) < eps * (fabs(p) * (fabs(HH(m - 1, m - 1)) + fabs(z) + fabs(HH(m + 1, m + 1))))) break;
        m--;
      }
      for (int i = m + 2; i <= n; i++) {
        HH(i, i - 2) = 0.0;
        if (i > m + 2) HH(i, i - 3) = 0.0;
      }
      for (int k = m; k <= n - 1; k++) {
        const bool notlast = (k != n - 1);
        if (k != m) {
          p = HH(k, k - 1);
          q = HH(k + 1, k - 1);
          r = notlast ? HH(k + 2, k - 1) : 0.0;
          x = fabs(p) + fabs(q) + fabs(r);
          if (x == 0.0) continue;
          p = p / x; q = q / x; r = r / x;
        }
        s = sqrt(p * p + q * q + r * r);
        if (p < 0) s = -s;
        if (s != 0) {
          if (k != m) HH(k, k - 1) = -s * x;
          else if (l != m) HH(k, k - 1) = -HH(k, k - 1);
          p = p + s;
          x = p / s; y = q / s; z = r / s;
          q = q / p; r = r / p;
          for (int j = k; j < nn; j++) {
            p = HH(k, j) + q * HH(k + 1, j);
            if (notlast) { p = p + r * HH(k + 2, j); HH(k + 2, j) = HH(k + 2, j) - p * z; }
            HH(k, j) = HH(k, j) - p * x;
            HH(k + 1, j) = HH(k + 1, j) - p * y;
          }
          const int imax = n < k + 3 ? n : k + 3;
          for (int i = 0; i <= imax; i++) {
            p = x * HH(i, k) + y * HH(i, k + 1);
            if (notlast) { p = p + z * HH(i, k + 2); HH(i, k + 2) = HH(i, k + 2) - p * r; }
            HH(i, k) = HH(i, k) - p;
            HH(i, k + 1) = HH(i, k + 1) - p * q;
          }
        }
      }
    }
  }
  return true;
}
#undef HH

// Null vector of (A - lambda I): nine full-pivot elimination steps, last unknown = 1, back substitution, unit length;
// out = its last four components.
__device__ static void p5_eigvec_tail(const double* A, double lambda, double* out) {
  double B[100];
  int pc[10];
  for (int i = 0; i < 10; i++)
    for (int j = 0; j < 10; j++) B[i * 10 + j] = A[i * 10 + j] - (i == j ? lambda : 0.0);
  for (int i = 0; i < 10; i++) pc[i] = i;
  for (int k = 0; k < 9; k++) {
    int br = k, bc = k;
    double best = -1.0;
    for (int j = k; j < 10; j++)
      for (int i = k; i < 10; i++) {
        const double v = fabs(B[i * 10 + j]);
        if (v > best) { best = v; br = i; bc = j; }
      }
    if (br != k)
      for (int j = 0; j < 10; j++) { const double t = B[k * 10 + j]; B[k * 10 + j] = B[br * 10 + j]; B[br * 10 + j] = t; }
    if (bc != k) {
      for (int i = 0; i < 10; i++) { const double t = B[i * 10 + k]; B[i * 10 + k] = B[i * 10 + bc]; B[i * 10 + bc] = t; }
      const int t = pc[k]; pc[k] = pc[bc]; pc[bc] = t;
    }
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
  nrm = sqrt(nrm);
  for (int i = 0; i < 4; i++) out[i] = v[6 + i] / nrm;
}

// FivePointEssentialMatrix (:97-178) on n in [5, 9] normalised matches x1[2n], x2[2n].  Writes up to 10 solutions
// (9 entries each, column-major like Eigen::Matrix3d::data()) to Es; returns their number.
__device__ static int p5_five_point(const double* x1, const double* x2, int n, double* Es) {
  double ns[36];  // null_space[9][4]
  {
    double A[81];
    for (int i = 0; i < n; i++) {
      const double ax = x1[2 * i], ay = x1[2 * i + 1], bx = x2[2 * i], by = x2[2 * i + 1];
      double* r = A + 9 * i;
      r[0] = bx * ax; r[1] = by * ax; r[2] = ax; r[3] = bx * ay; r[4] = by * ay; r[5] = ay; r[6] = bx; r[7] = by; r[8] = 1.0;
    }
    if (n == 5) {
      int pr[5], pc[9];
      double maxpivot;
      const int nz = p5_fullpiv_lu(A, 5, 9, 9, pr, pc, &maxpivot);
      const double thr = maxpivot * (DBL_EPSILON * 5);
      int rank = 0;
      for (int i = 0; i < nz; i++) rank += fabs(A[i * 9 + i]) > thr;
      if (rank != 5) return 0;
      for (int k = 0; k < 4; k++) {
        double y[5];
        for (int i = 4; i >= 0; i--) y[i] = A[i * 9 + 5 + k];
        for (int i = 4; i >= 0; i--) {
          y[i] = y[i] / A[i * 9 + i];
          for (int j = 0; j < i; j++) y[j] -= A[j * 9 + i] * y[i];
        }
        for (int i = 0; i < 5; i++) ns[pc[i] * 4 + k] = -y[i];
        for (int i = 5; i < 9; i++) ns[pc[i] * 4 + k] = (i == 5 + k) ? 1.0 : 0.0;
      }
    } else {
      double Ut[81], W[9], Vt[81];
      for (int i = 0; i < 9; i++)
        for (int j = 0; j < 9; j++) {
          double s = 0;
          for (int k = 0; k < n; k++) s += A[k * 9 + i] * A[k * 9 + j];
          Ut[j * 9 + i] = s;  // transposed copy (cv::SVD::compute); A^T A is symmetric up to the order of the products
        }
      pose_jsvd<9, 9, true>(Ut, W, Vt);
      for (int i = 0; i < 9; i++)
        for (int k = 0; k < 4; k++) ns[i * 4 + k] = Vt[(5 + k) * 9 + i];
    }
  }
  double C[200];  // 10 x 20
  {
    double e[9][20], eet[9][20], trace[20], t0[20], t1[20], t2[20], t3[20];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) p5_lin(ns + (i + 3 * j) * 4, e[3 * i + j]);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        p5_mul(e[3 * i], 16, e[3 * j], 16, t0);
        p5_mul(e[3 * i + 1], 16, e[3 * j + 1], 16, t1);
        p5_add(t0, t1, t2);
        p5_mul(e[3 * i + 2], 16, e[3 * j + 2], 16, t0);
        p5_add(t2, t0, t1);
        p5_scale(2.0, t1, eet[3 * i + j]);
      }
    p5_add(eet[0], eet[4], t0);
    p5_add(t0, eet[8], trace);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        p5_mul(eet[3 * i], 10, e[j], 16, t0);
        p5_mul(eet[3 * i + 1], 10, e[3 + j], 16, t1);
        p5_add(t0, t1, t2);
        p5_mul(eet[3 * i + 2], 10, e[6 + j], 16, t0);
        p5_add(t2, t0, t1);
        p5_mul(trace, 10, e[3 * i + j], 16, t0);
        p5_scale(0.5, t0, t2);
        p5_sub(t1, t2, C + 20 * (3 * i + j));
      }
    // determinant: (e01 e12 - e02 e11) e20 + (e02 e10 - e00 e12) e21 + (e00 e11 - e01 e10) e22
    p5_mul(e[1], 16, e[5], 16, t0); p5_mul(e[2], 16, e[4], 16, t1); p5_sub(t0, t1, t2); p5_mul(t2, 10, e[6], 16, t3);
    p5_mul(e[2], 16, e[3], 16, t0); p5_mul(e[0], 16, e[5], 16, t1); p5_sub(t0, t1, t2); p5_mul(t2, 10, e[7], 16, t0);
    p5_add(t3, t0, t1);  // t1 = first + second
    p5_mul(e[0], 16, e[4], 16, t0); p5_mul(e[1], 16, e[3], 16, t2); p5_sub(t0, t2, t3); p5_mul(t3, 10, e[8], 16, t0);
    p5_add(t1, t0, C + 180);
  }
  double Act[100];
  {
    double LU[100], G[100];
    int pr[10], pc[10];
    for (int i = 0; i < 10; i++)
      for (int j = 0; j < 10; j++) LU[i * 10 + j] = C[i * 20 + j];
    double maxpivot;
    const int nz = p5_fullpiv_lu(LU, 10, 10, 10, pr, pc, &maxpivot);
    const double thr = maxpivot * (DBL_EPSILON * 10);
    int rank = 0;
    for (int i = 0; i < nz; i++) rank += fabs(LU[i * 10 + i]) > thr;
    for (int col = 0; col < 10; col++) {
      double cv[10];
      for (int i = 0; i < 10; i++) cv[i] = C[pr[i] * 20 + 10 + col];
      for (int k = 0; k < 10; k++)
        for (int i = k + 1; i < 10; i++) cv[i] -= LU[i * 10 + k] * cv[k];
      for (int k = rank - 1; k >= 0; k--) {
        cv[k] = cv[k] / LU[k * 10 + k];
        for (int i = 0; i < k; i++) cv[i] -= LU[i * 10 + k] * cv[k];
      }
      for (int i = 0; i < rank; i++) G[pc[i] * 10 + col] = cv[i];
      for (int i = rank; i < 10; i++) G[pc[i] * 10 + col] = 0.0;
    }
    for (int i = 0; i < 100; i++) Act[i] = 0.0;
    const int src_row[6] = {0, 1, 2, 4, 5, 7};
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 10; j++) Act[i * 10 + j] = G[src_row[i] * 10 + j];
    Act[60] = -1.0; Act[71] = -1.0; Act[83] = -1.0; Act[96] = -1.0;
  }
  double wr[10], wi[10];
  if (!p5_real_eigenvalues10(Act, wr, wi)) return 0;
  int count = 0;
  for (int i = 0; i < 10; i++) {
    if (wi[i] != 0) continue;
    double tail[4];
    p5_eigvec_tail(Act, wr[i], tail);
    for (int k = 0; k < 9; k++) Es[9 * count + k] = ns[k * 4] * tail[0] + ns[k * 4 + 1] * tail[1] + ns[k * 4 + 2] * tail[2] + ns[k * 4 + 3] * tail[3];
    count++;
  }
  return count;
}

// grid (ceil(T / 64), pairs): thread = sample `it` of pair blockIdx.y.  cand_E[(p * T + it) * 90], cand_n[p * T + it].
// A pair with 5..9 matches has one "sample": all of them (essential_matrix_five_point.cc:41-48).
__global__ __launch_bounds__(POSE_WAVE) void k_e5_hyp(int T, const int* __restrict__ off, const double* __restrict__ pts_ref,
                                                      const double* __restrict__ pts_cur, const double* __restrict__ f_ref,
                                                      const double* __restrict__ f_cur, uint64_t seed, double* __restrict__ cand_E,
                                                      int* __restrict__ cand_n) {
  const int p = blockIdx.y, it = blockIdx.x * POSE_WAVE + threadIdx.x;
  if (it >= T) return;
  const int o = off[p], N = off[p + 1] - o;
  int* cn = cand_n + (size_t)p * T + it;
  if (N < 5 || (N < 10 && it > 0)) { *cn = 0; return; }
  int idx[9];
  int n = 5;
  if (N < 10) { n = N; for (int k = 0; k < N; k++) idx[k] = k; }
  else pose_sample<5>(seed, 0x35707445ull, p, it, N, idx);
  double a[18], b[18];
  const double f1 = f_ref[p], f2 = f_cur[p];
  for (int k = 0; k < n; k++) {
    a[2 * k] = pts_ref[2 * ((size_t)o + idx[k])] / f1; a[2 * k + 1] = pts_ref[2 * ((size_t)o + idx[k]) + 1] / f1;
    b[2 * k] = pts_cur[2 * ((size_t)o + idx[k])] / f2; b[2 * k + 1] = pts_cur[2 * ((size_t)o + idx[k]) + 1] / f2;
  }
  double Es[90];
  const int c = p5_five_point(a, b, n, Es);
  double* out = cand_E + ((size_t)p * T + it) * 90;
  for (int k = 0; k < 9 * c; k++) out[k] = Es[k];
  *cn = c;
}

// thread = candidate (pair, sample, slot): the Sampson sum over all matches in match order (Error :333-349)
__global__ __launch_bounds__(256) void k_e5_score(int T, const int* __restrict__ off, const double* __restrict__ pts_ref,
                                                  const double* __restrict__ pts_cur, const double* __restrict__ f_ref,
                                                  const double* __restrict__ f_cur, const double* __restrict__ cand_E,
                                                  const int* __restrict__ cand_n, double* __restrict__ cand_err) {
  const int p = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (c >= T * 10) return;
  const int it = c / 10, slot = c - 10 * it;
  if (slot >= cand_n[(size_t)p * T + it]) return;
  const int o = off[p], N = off[p + 1] - o;
  const double* Ep = cand_E + ((size_t)p * T + it) * 90 + 9 * slot;
  double E[9];
  for (int k = 0; k < 9; k++) E[k] = Ep[k];
  const double f1 = f_ref[p], f2 = f_cur[p];
  double total = 0.0;
  for (int i = 0; i < N; i++) {
    const double ax = pts_ref[2 * ((size_t)o + i)] / f1, ay = pts_ref[2 * ((size_t)o + i) + 1] / f1;
    const double bx = pts_cur[2 * ((size_t)o + i)] / f2, by = pts_cur[2 * ((size_t)o + i) + 1] / f2;
    const double l0 = E[0] * ax + E[3] * ay + E[6] * 1.0, l1 = E[1] * ax + E[4] * ay + E[7] * 1.0, l2 = E[2] * ax + E[5] * ay + E[8] * 1.0;
    const double num = bx * l0 + by * l1 + 1.0 * l2;
    const double d0 = bx * E[0] + by * E[1] + 1.0 * E[2], d1 = bx * E[3] + by * E[4] + 1.0 * E[5];
    const double den = d0 * d0 + d1 * d1 + l0 * l0 + l1 * l1;
    total += num * num / den;
  }
  cand_err[((size_t)p * T + it) * 10 + slot] = total;
}

// One workgroup per pair: replay `if (error_i < error_min)` from 1e6 over the candidates in (sample, slot) order, then
// DecomposeEssentialMatrix (:81-104) and the cheirality vote over all matches (:33-79).
__global__ __launch_bounds__(256) void k_e5_select(int T, const int* __restrict__ off, const double* __restrict__ pts_ref,
                                                   const double* __restrict__ pts_cur, const double* __restrict__ f_ref,
                                                   const double* __restrict__ f_cur, const double* __restrict__ cand_E,
                                                   const int* __restrict__ cand_n, const double* __restrict__ cand_err,
                                                   double* __restrict__ Eout, double* __restrict__ Rout, double* __restrict__ tout,
                                                   uint8_t* __restrict__ ok, int* __restrict__ n_candidates) {
  __shared__ double s_e[256];
  __shared__ int s_k[256], s_first[256], s_cnt[256];
  __shared__ double s_R[4][9], s_t[4][3];
  __shared__ int s_votes[4];
  const int p = blockIdx.x, tid = threadIdx.x;
  const int o = off[p], N = off[p + 1] - o;
  double be = 1000000.0;
  int bk = -1, first = 0x7fffffff, cnt = 0;
  for (int it = tid; it < T; it += 256) {
    const int c = cand_n[(size_t)p * T + it];
    cnt += c;
    for (int sl = 0; sl < c; sl++) {
      const int key = it * 10 + sl;
      if (key < first) first = key;
      const double e = cand_err[((size_t)p * T + it) * 10 + sl];
      if (e < be) { be = e; bk = key; }
    }
  }
  s_e[tid] = be; s_k[tid] = bk; s_first[tid] = first; s_cnt[tid] = cnt;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) {
      const double e2 = s_e[tid + st];
      const int k2 = s_k[tid + st];
      if (k2 >= 0 && (s_k[tid] < 0 || e2 < s_e[tid] || (e2 == s_e[tid] && k2 < s_k[tid]))) { s_e[tid] = e2; s_k[tid] = k2; }
      if (s_first[tid + st] < s_first[tid]) s_first[tid] = s_first[tid + st];
      s_cnt[tid] += s_cnt[tid + st];
    }
    __syncthreads();
  }
  const int nE = s_cnt[0];
  const int key = s_k[0] >= 0 ? s_k[0] : s_first[0];
  if (tid < 4) s_votes[tid] = 0;
  if (nE < 4) {
    if (tid == 0) {
      for (int k = 0; k < 9; k++) { Eout[9 * (size_t)p + k] = 0.0; Rout[9 * (size_t)p + k] = 0.0; }
      for (int k = 0; k < 3; k++) tout[3 * (size_t)p + k] = 0.0;
      ok[p] = 0;
      n_candidates[p] = nE;
    }
    return;
  }
  const double* E = cand_E + ((size_t)p * T + key / 10) * 90 + 9 * (key % 10);
  if (tid == 0) {
    double Ut[9], W[3], Vt[9], U[3][3], V[3][3];
    for (int i = 0; i < 3; i++)      // Ut = Em^T with Em(i, j) = E[i + 3 j]
      for (int k = 0; k < 3; k++) Ut[i * 3 + k] = E[k + 3 * i];
    pose_jsvd<3, 3, true>(Ut, W, Vt);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) { U[i][j] = Ut[j * 3 + i]; V[i][j] = Vt[j * 3 + i]; }
    const double detU = U[0][0] * (U[1][1] * U[2][2] - U[1][2] * U[2][1]) - U[0][1] * (U[1][0] * U[2][2] - U[1][2] * U[2][0]) +
                        U[0][2] * (U[1][0] * U[2][1] - U[1][1] * U[2][0]);
    if (detU < 0) for (int i = 0; i < 3; i++) U[i][2] *= -1.0;
    const double detV = V[0][0] * (V[1][1] * V[2][2] - V[1][2] * V[2][1]) - V[0][1] * (V[1][0] * V[2][2] - V[1][2] * V[2][0]) +
                        V[0][2] * (V[1][0] * V[2][1] - V[1][1] * V[2][0]);
    if (detV < 0) for (int i = 0; i < 3; i++) V[i][2] *= -1.0;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        const double r1 = -U[i][1] * V[j][0] + U[i][0] * V[j][1] + U[i][2] * V[j][2];
        const double r2 = U[i][1] * V[j][0] + -U[i][0] * V[j][1] + U[i][2] * V[j][2];
        s_R[0][3 * i + j] = r1; s_R[1][3 * i + j] = r1; s_R[2][3 * i + j] = r2; s_R[3][3 * i + j] = r2;
      }
    const double tn = sqrt(U[0][2] * U[0][2] + U[1][2] * U[1][2] + U[2][2] * U[2][2]);
    double t[3];
    for (int i = 0; i < 3; i++) t[i] = U[i][2] / tn;
    for (int h = 0; h < 4; h++) {
      const double sg = (h & 1) ? -1.0 : 1.0;
      for (int i = 0; i < 3; i++) s_t[h][i] = -(s_R[h][i] * (sg * t[0]) + s_R[h][3 + i] * (sg * t[1]) + s_R[h][6 + i] * (sg * t[2]));
    }
  }
  __syncthreads();
  const double f1 = f_ref[p], f2 = f_cur[p];
  for (int i = tid; i < N; i += 256) {
    const double d1[3] = {pts_ref[2 * ((size_t)o + i)] / f1, pts_ref[2 * ((size_t)o + i) + 1] / f1, 1.0};
    const double q[3] = {pts_cur[2 * ((size_t)o + i)] / f2, pts_cur[2 * ((size_t)o + i) + 1] / f2, 1.0};
    for (int h = 0; h < 4; h++) {
      const double* R = s_R[h];
      const double* tt = s_t[h];
      double c[3], d2[3];
      for (int k = 0; k < 3; k++) {
        c[k] = -(R[k] * tt[0] + R[3 + k] * tt[1] + R[6 + k] * tt[2]);
        d2[k] = R[k] * q[0] + R[3 + k] * q[1] + R[6 + k] * q[2];
      }
      const double d1sq = d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2], d2sq = d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2];
      const double d12 = d1[0] * d2[0] + d1[1] * d2[1] + d1[2] * d2[2];
      const double d1p = d1[0] * c[0] + d1[1] * c[1] + d1[2] * c[2], d2p = d2[0] * c[0] + d2[1] * c[1] + d2[2] * c[2];
      if (d2sq * d1p - d12 * d2p > 0 && d12 * d1p - d1sq * d2p > 0) { atomicAdd(&s_votes[h], 1); break; }
    }
  }
  __syncthreads();
  if (tid == 0) {
    int mx = s_votes[0];
    for (int h = 1; h < 4; h++) mx = s_votes[h] > mx ? s_votes[h] : mx;
    int h = 0;
    while (s_votes[h] != mx) h++;
    for (int i = 0; i < 9; i++) Rout[9 * (size_t)p + i] = s_R[h][i];
    for (int i = 0; i < 3; i++) tout[3 * (size_t)p + i] = s_t[h][i];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Eout[9 * (size_t)p + 3 * i + j] = E[i + 3 * j];
    ok[p] = 1;
    n_candidates[p] = nE;
  }
}

static int pose_check_offsets(msfm_ctx* ctx, const char* who, int n, const int* offsets) {
  if (offsets[0] != 0) return msfm_set_error(ctx, MSFM_E_INVAL, "%s: offsets[0] must be 0", who);
  for (int p = 0; p < n; p++)
    if (offsets[p + 1] < offsets[p]) return msfm_set_error(ctx, MSFM_E_INVAL, "%s: offsets must be non-decreasing", who);
  return MSFM_OK;
}

MSFM_API int msfm_epnp_ransac_batch(msfm_ctx* ctx, int n_problems, const int* offsets, const double* pts_w, const double* pts_2d,
                                    const double* f, int max_iter, uint64_t seed, double* R, double* t, double* errors, double* avg_error,
                                    int* best_iter) {
  if (!ctx || n_problems < 0 || !offsets || !f || !R || !t || !avg_error) return MSFM_E_INVAL;
  if (max_iter < 1 || max_iter > 65536) return msfm_set_error(ctx, MSFM_E_INVAL, "epnp: max_iter out of range");
  if (n_problems == 0) return MSFM_OK;
  if (n_problems > 65535) return msfm_set_error(ctx, MSFM_E_INVAL, "epnp: at most 65535 problems per call");
  MSFM_TRY(pose_check_offsets(ctx, "epnp", n_problems, offsets));
  const int total = offsets[n_problems];
  if (total > 0 && (!pts_w || !pts_2d || !errors)) return MSFM_E_INVAL;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  DevBuf<int> d_off, d_best;
  DevBuf<double> d_w, d_2d, d_f, d_hyp, d_R, d_t, d_err, d_avg;
  HIP_TRY(ctx, d_off.alloc((size_t)n_problems + 1));
  HIP_TRY(ctx, d_off.upload(offsets, (size_t)n_problems + 1, s));
  HIP_TRY(ctx, d_w.alloc(3 * (size_t)std::max(1, total)));
  HIP_TRY(ctx, d_2d.alloc(2 * (size_t)std::max(1, total)));
  HIP_TRY(ctx, d_w.upload(pts_w, 3 * (size_t)total, s));
  HIP_TRY(ctx, d_2d.upload(pts_2d, 2 * (size_t)total, s));
  HIP_TRY(ctx, d_f.alloc(n_problems));
  HIP_TRY(ctx, d_f.upload(f, n_problems, s));
  HIP_TRY(ctx, d_hyp.alloc((size_t)n_problems * max_iter * 13));
  HIP_TRY(ctx, d_R.alloc(9 * (size_t)n_problems));
  HIP_TRY(ctx, d_t.alloc(3 * (size_t)n_problems));
  HIP_TRY(ctx, d_err.alloc((size_t)std::max(1, total)));
  HIP_TRY(ctx, d_avg.alloc(n_problems));
  HIP_TRY(ctx, d_best.alloc(n_problems));
  {
    KTimer tm(ctx, "pose_epnp_hyp");
    hipLaunchKernelGGL(k_epnp_hyp, dim3(cdiv(max_iter, POSE_WAVE), n_problems), dim3(POSE_WAVE), 0, s, max_iter, d_off.p, d_w.p, d_2d.p,
                       d_f.p, seed, d_hyp.p);
  }
  {
    KTimer tm(ctx, "pose_epnp_select");
    hipLaunchKernelGGL(k_epnp_select, dim3(n_problems), dim3(256), 0, s, max_iter, d_off.p, d_w.p, d_2d.p, d_f.p, d_hyp.p, d_R.p, d_t.p,
                       d_err.p, d_avg.p, d_best.p);
  }
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(R, d_R.p, sizeof(double) * 9 * (size_t)n_problems, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(t, d_t.p, sizeof(double) * 3 * (size_t)n_problems, hipMemcpyDeviceToHost, s));
  if (total) HIP_TRY(ctx, hipMemcpyAsync(errors, d_err.p, sizeof(double) * (size_t)total, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(avg_error, d_avg.p, sizeof(double) * (size_t)n_problems, hipMemcpyDeviceToHost, s));
  if (best_iter) HIP_TRY(ctx, hipMemcpyAsync(best_iter, d_best.p, sizeof(int) * (size_t)n_problems, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return MSFM_OK;
}

MSFM_API int msfm_relpose_5pt_batch(msfm_ctx* ctx, int n_pairs, const int* offsets, const double* pts_ref, const double* pts_cur,
                                    const double* f_ref, const double* f_cur, int ransac_times, uint64_t seed, double* E, double* R, double* t,
                                    uint8_t* ok, int* n_candidates) {
  if (!ctx || n_pairs < 0 || !offsets || !f_ref || !f_cur || !E || !R || !t || !ok) return MSFM_E_INVAL;
  if (ransac_times < 1 || ransac_times > 65536) return msfm_set_error(ctx, MSFM_E_INVAL, "relpose: ransac_times out of range");
  if (n_pairs == 0) return MSFM_OK;
  if (n_pairs > 65535) return msfm_set_error(ctx, MSFM_E_INVAL, "relpose: at most 65535 pairs per call");
  MSFM_TRY(pose_check_offsets(ctx, "relpose", n_pairs, offsets));
  const int total = offsets[n_pairs];
  if (total > 0 && (!pts_ref || !pts_cur)) return MSFM_E_INVAL;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int T = ransac_times;
  DevBuf<int> d_off, d_cn, d_nc;
  DevBuf<double> d_a, d_b, d_f1, d_f2, d_cE, d_ce, d_E, d_R, d_t;
  DevBuf<uint8_t> d_ok;
  HIP_TRY(ctx, d_off.alloc((size_t)n_pairs + 1));
  HIP_TRY(ctx, d_off.upload(offsets, (size_t)n_pairs + 1, s));
  HIP_TRY(ctx, d_a.alloc(2 * (size_t)std::max(1, total)));
  HIP_TRY(ctx, d_b.alloc(2 * (size_t)std::max(1, total)));
  HIP_TRY(ctx, d_a.upload(pts_ref, 2 * (size_t)total, s));
  HIP_TRY(ctx, d_b.upload(pts_cur, 2 * (size_t)total, s));
  HIP_TRY(ctx, d_f1.alloc(n_pairs)); HIP_TRY(ctx, d_f1.upload(f_ref, n_pairs, s));
  HIP_TRY(ctx, d_f2.alloc(n_pairs)); HIP_TRY(ctx, d_f2.upload(f_cur, n_pairs, s));
  HIP_TRY(ctx, d_cE.alloc((size_t)n_pairs * T * 90));
  HIP_TRY(ctx, d_ce.alloc((size_t)n_pairs * T * 10));
  HIP_TRY(ctx, d_cn.alloc((size_t)n_pairs * T));
  HIP_TRY(ctx, d_E.alloc(9 * (size_t)n_pairs)); HIP_TRY(ctx, d_R.alloc(9 * (size_t)n_pairs)); HIP_TRY(ctx, d_t.alloc(3 * (size_t)n_pairs));
  HIP_TRY(ctx, d_ok.alloc(n_pairs)); HIP_TRY(ctx, d_nc.alloc(n_pairs));
  {
    KTimer tm(ctx, "pose_e5_hyp");
    hipLaunchKernelGGL(k_e5_hyp, dim3(cdiv(T, POSE_WAVE), n_pairs), dim3(POSE_WAVE), 0, s, T, d_off.p, d_a.p, d_b.p, d_f1.p, d_f2.p, seed,
                       d_cE.p, d_cn.p);
  }
  {
    KTimer tm(ctx, "pose_e5_score");
    hipLaunchKernelGGL(k_e5_score, dim3(cdiv(T * 10, 256), n_pairs), dim3(256), 0, s, T, d_off.p, d_a.p, d_b.p, d_f1.p, d_f2.p, d_cE.p,
                       d_cn.p, d_ce.p);
  }
  {
    KTimer tm(ctx, "pose_e5_select");
    hipLaunchKernelGGL(k_e5_select, dim3(n_pairs), dim3(256), 0, s, T, d_off.p, d_a.p, d_b.p, d_f1.p, d_f2.p, d_cE.p, d_cn.p, d_ce.p, d_E.p,
                       d_R.p, d_t.p, d_ok.p, d_nc.p);
  }
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(E, d_E.p, sizeof(double) * 9 * (size_t)n_pairs, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(R, d_R.p, sizeof(double) * 9 * (size_t)n_pairs, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(t, d_t.p, sizeof(double) * 3 * (size_t)n_pairs, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(ok, d_ok.p, (size_t)n_pairs, hipMemcpyDeviceToHost, s));
  if (n_candidates) HIP_TRY(ctx, hipMemcpyAsync(n_candidates, d_nc.p, sizeof(int) * (size_t)n_pairs, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return MSFM_OK;
}
