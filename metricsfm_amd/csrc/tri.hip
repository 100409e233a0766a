// Batched triangulation / reprojection / epipolar filter (FP64).
// Round 4: the midpoint triangulation and the reprojection run in two kinds of phases inside one workgroup of 64 tracks - what
// belongs to ONE observation (its ray, its reprojection error, its ray for the angle gate: all the loads, divisions and square
// roots) is computed with lane = observation from coalesced reads and parked in LDS; what the reference SUMS over a track's
// observations in their order is then done by one thread per track from LDS.  Every number is formed by the same operations
// in the same order as in the thread-per-track form (which remains: DLT, and workgroups whose observations do not fit the
// LDS area), so results are unchanged; the kernel no longer waits for a chain of dependent loads per observation.
// Reference: Point3D::Trianglate2 SfM/src/structure.cc:211-265, Point3D::Trianglate (DLT)
// :163-209, Point3D::Reprojection :267-300, Point3D::SufficientTriangulationAngle :325-355,
// GeoVerification::GeoVerificationFundamental (closed form) SfM/src/utils/geo_verification.cc:60-79.
#include "common.h"


// structure.cc:267-300
__device__ double track_mse(const TrackPtrs& T, int b, int e, const double* X) {
  double mse = 0.0;
  int count = 0;
  for (int i = b; i < e; i++) {
    const int c = T.cam[i];
    const double* R = T.R + 9 * (size_t)c;
    const double* tt = T.t + 3 * (size_t)c;
    const double* fk = T.fk + 3 * (size_t)c;
    const double pc0 = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + tt[0];
    const double pc1 = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + tt[1];
    const double pc2 = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + tt[2];
    if (pc2 < 0) return 100000.0;
    const double x = pc0 / pc2, y = pc1 / pc2;
    const double r2 = x * x + y * y;
    const double distortion = 1.0 + r2 * (fk[1] + fk[2] * r2);
    const double u = fk[0] * distortion * x, v = fk[0] * distortion * y;
    const double du = u - T.xy[2 * (size_t)i], dv = v - T.xy[2 * (size_t)i + 1];
    mse += du * du + dv * dv;
    count++;
  }
  return mse / count;
}

// structure.cc:325-355
__device__ bool track_angle_ok(const TrackPtrs& T, int b, int e, const double* X, double cos_min) {
  for (int i = b; i + 1 < e; i++) {
    const double* ci = T.c + 3 * (size_t)T.cam[i];
    double a[3] = {X[0] - ci[0], X[1] - ci[1], X[2] - ci[2]};
    const double na = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    a[0] /= na; a[1] /= na; a[2] /= na;
    for (int j = i + 1; j < e; j++) {
      const double* cj = T.c + 3 * (size_t)T.cam[j];
      double d[3] = {X[0] - cj[0], X[1] - cj[1], X[2] - cj[2]};
      const double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      d[0] /= nd; d[1] /= nd; d[2] /= nd;
      if (a[0] * d[0] + a[1] * d[1] + a[2] * d[2] < cos_min) return true;
    }
  }
  return false;
}

__device__ __forceinline__ void tri_midpoint_track(const TrackPtrs& T, int t, double th_error, double cos_min, double* __restrict__ Xo,
                                                   double* __restrict__ mse, uint8_t* __restrict__ ok) {
  const int b = T.off[t], e = T.off[t + 1];
  double A[16], bv[4] = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 16; k++) A[k] = 0.0;
  for (int i = b; i < e; i++) {
    const int c = T.cam[i];
    const double* R = T.R + 9 * (size_t)c;
    const double* o = T.c + 3 * (size_t)c;
    const double f = T.fk[3 * (size_t)c];
    const double d0 = T.xy[2 * (size_t)i], d1 = T.xy[2 * (size_t)i + 1];
    double dw[3] = {R[0] * d0 + R[3] * d1 + R[6] * f, R[1] * d0 + R[4] * d1 + R[7] * f, R[2] * d0 + R[5] * d1 + R[8] * f};
    const double n = sqrt(dw[0] * dw[0] + dw[1] * dw[1] + dw[2] * dw[2]);
    dw[0] /= n; dw[1] /= n; dw[2] /= n;
    const double dh[4] = {dw[0], dw[1], dw[2], 0.0};
    const double oh[4] = {o[0], o[1], o[2], 1.0};
#pragma unroll
    for (int r = 0; r < 4; r++) {
      double acc = 0.0;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const double at = (r == q ? 1.0 : 0.0) - dh[r] * dh[q];
        A[r * 4 + q] += at;
        acc += at * oh[q];
      }
      bv[r] += acc;
    }
  }
  ok[t] = 0;
  mse[t] = 0.0;
  // Eigen::LLT<Matrix4d>: fail on a non-positive pivot (structure.cc:247-251)
  double L[16];
#pragma unroll
  for (int k = 0; k < 16; k++) L[k] = 0.0;
  bool pd = true;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    double d = A[j * 4 + j];
#pragma unroll
    for (int k = 0; k < j; k++) d -= L[j * 4 + k] * L[j * 4 + k];
    if (!(d > 0.0)) pd = false;
    L[j * 4 + j] = sqrt(d);
#pragma unroll
    for (int i = j + 1; i < 4; i++) {
      double s = A[i * 4 + j];
#pragma unroll
      for (int k = 0; k < j; k++) s -= L[i * 4 + k] * L[j * 4 + k];
      L[i * 4 + j] = s / L[j * 4 + j];
    }
  }
  if (!pd) return;
  double y[4], x[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    double s = bv[i];
#pragma unroll
    for (int k = 0; k < i; k++) s -= L[i * 4 + k] * y[k];
    y[i] = s / L[i * 4 + i];
  }
#pragma unroll
  for (int i = 3; i >= 0; i--) {
    double s = y[i];
#pragma unroll
    for (int k = i + 1; k < 4; k++) s -= L[k * 4 + i] * x[k];
    x[i] = s / L[i * 4 + i];
  }
  double X[3] = {x[0] / x[3], x[1] / x[3], x[2] / x[3]};
  Xo[3 * (size_t)t] = X[0]; Xo[3 * (size_t)t + 1] = X[1]; Xo[3 * (size_t)t + 2] = X[2];
  const double m = track_mse(T, b, e, X);
  mse[t] = m;
  ok[t] = !(sqrt(m) > th_error || !track_angle_ok(T, b, e, X, cos_min));
}
__global__ __launch_bounds__(256) void k_tri_midpoint(TrackPtrs T, double th_error, double cos_min, double* __restrict__ Xo,
                                                       double* __restrict__ mse, uint8_t* __restrict__ ok) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T.n_tracks) return;
  tri_midpoint_track(T, t, th_error, cos_min, Xo, mse, ok);
}

// ---- the staged form: 64 tracks per workgroup, observations through LDS ----
#define TRI_TPB 64      // tracks per workgroup
#define TRI_CAP 768     // observations per workgroup that fit the LDS area (else: thread per track)
#define TRI_REC 7       // doubles per observation: ray (3), centre (3), squared reprojection error
__global__ __launch_bounds__(256) void k_tri_midpoint_staged(TrackPtrs T, double th_error, double cos_min, double* __restrict__ Xo,
                                                              double* __restrict__ mse, uint8_t* __restrict__ ok) {
  __shared__ double sd[TRI_CAP * TRI_REC];
  __shared__ double Xs[TRI_TPB * 3];
  __shared__ int offs[TRI_TPB + 1];
  __shared__ short tk[TRI_CAP];
  __shared__ unsigned char alive[TRI_TPB];
  const int tid = threadIdx.x, t0 = blockIdx.x * TRI_TPB;
  const int nt = min(TRI_TPB, T.n_tracks - t0);
  if (tid <= nt) offs[tid] = T.off[t0 + tid];
  __syncthreads();
  const int ob = offs[0], nobs = offs[nt] - ob;
  if (nobs > TRI_CAP) {   // (uniform) long tracks: the thread-per-track form for this workgroup
    if (tid < nt) tri_midpoint_track(T, t0 + tid, th_error, cos_min, Xo, mse, ok);
    return;
  }
  // -- per observation: the unit ray through the image point in world coordinates, the camera centre (structure.cc:224-236)
  for (int j = tid; j < nobs; j += 256) {
    const int i = ob + j;
    const int c = T.cam[i];
    const double* R = T.R + 9 * (size_t)c;
    const double* o = T.c + 3 * (size_t)c;
    const double f = T.fk[3 * (size_t)c];
    const double d0 = T.xy[2 * (size_t)i], d1 = T.xy[2 * (size_t)i + 1];
    double dw[3] = {R[0] * d0 + R[3] * d1 + R[6] * f, R[1] * d0 + R[4] * d1 + R[7] * f, R[2] * d0 + R[5] * d1 + R[8] * f};
    const double n = sqrt(dw[0] * dw[0] + dw[1] * dw[1] + dw[2] * dw[2]);
    dw[0] /= n; dw[1] /= n; dw[2] /= n;
    double* rec = sd + (size_t)j * TRI_REC;
    rec[0] = dw[0]; rec[1] = dw[1]; rec[2] = dw[2]; rec[3] = o[0]; rec[4] = o[1]; rec[5] = o[2];
  }
  __syncthreads();
  // -- per track: the sums over its observations in their order, the 4 x 4 LLT (structure.cc:237-258)
  if (tid < nt) {
    const int t = t0 + tid, b = offs[tid] - ob, e = offs[tid + 1] - ob;
    double A[16], bv[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 16; k++) A[k] = 0.0;
    for (int j = b; j < e; j++) {
      const double* rec = sd + (size_t)j * TRI_REC;
      tk[j] = (short)tid;
      const double dh[4] = {rec[0], rec[1], rec[2], 0.0};
      const double oh[4] = {rec[3], rec[4], rec[5], 1.0};
#pragma unroll
      for (int r = 0; r < 4; r++) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const double at = (r == q ? 1.0 : 0.0) - dh[r] * dh[q];
          A[r * 4 + q] += at;
          acc += at * oh[q];
        }
        bv[r] += acc;
      }
    }
    ok[t] = 0;
    mse[t] = 0.0;
    double L[16];
#pragma unroll
    for (int k = 0; k < 16; k++) L[k] = 0.0;
    bool pd = true;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      double d = A[j * 4 + j];
#pragma unroll
      for (int k = 0; k < j; k++) d -= L[j * 4 + k] * L[j * 4 + k];
      if (!(d > 0.0)) pd = false;
      L[j * 4 + j] = sqrt(d);
#pragma unroll
      for (int i = j + 1; i < 4; i++) {
        double sacc = A[i * 4 + j];
#pragma unroll
        for (int k = 0; k < j; k++) sacc -= L[i * 4 + k] * L[j * 4 + k];
        L[i * 4 + j] = sacc / L[j * 4 + j];
      }
    }
    alive[tid] = pd ? 1 : 0;
    if (pd) {
      double y[4], x[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        double sacc = bv[i];
#pragma unroll
        for (int k = 0; k < i; k++) sacc -= L[i * 4 + k] * y[k];
        y[i] = sacc / L[i * 4 + i];
      }
#pragma unroll
      for (int i = 3; i >= 0; i--) {
        double sacc = y[i];
#pragma unroll
        for (int k = i + 1; k < 4; k++) sacc -= L[k * 4 + i] * x[k];
        x[i] = sacc / L[i * 4 + i];
      }
      const double X0 = x[0] / x[3], X1 = x[1] / x[3], X2 = x[2] / x[3];
      Xo[3 * (size_t)t] = X0; Xo[3 * (size_t)t + 1] = X1; Xo[3 * (size_t)t + 2] = X2;
      Xs[3 * tid] = X0; Xs[3 * tid + 1] = X1; Xs[3 * tid + 2] = X2;
    }
  }
  __syncthreads();
  // -- per observation: squared reprojection error (structure.cc:267-300; negative depth is flagged) and the unit ray from the
  //    camera centre to the point (:325-355)
  for (int j = tid; j < nobs; j += 256) {
    const int trk = tk[j];
    if (!alive[trk]) continue;
    const int i = ob + j;
    const int c = T.cam[i];
    const double X[3] = {Xs[3 * trk], Xs[3 * trk + 1], Xs[3 * trk + 2]};
    const double* R = T.R + 9 * (size_t)c;
    const double* tt = T.t + 3 * (size_t)c;
    const double* fk = T.fk + 3 * (size_t)c;
    double* rec = sd + (size_t)j * TRI_REC;
    const double pc0 = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + tt[0];
    const double pc1 = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + tt[1];
    const double pc2 = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + tt[2];
    double err = -1.0;   // pc2 < 0: the track's error is the constant 100000 (structure.cc:283)
    if (!(pc2 < 0)) {
      const double x = pc0 / pc2, y = pc1 / pc2;
      const double r2 = x * x + y * y;
      const double distortion = 1.0 + r2 * (fk[1] + fk[2] * r2);
      const double u = fk[0] * distortion * x, v = fk[0] * distortion * y;
      const double du = u - T.xy[2 * (size_t)i], dv = v - T.xy[2 * (size_t)i + 1];
      err = du * du + dv * dv;
    }
    double a[3] = {X[0] - rec[3], X[1] - rec[4], X[2] - rec[5]};
    const double na = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    a[0] /= na; a[1] /= na; a[2] /= na;
    rec[0] = a[0]; rec[1] = a[1]; rec[2] = a[2]; rec[6] = err;
  }
  __syncthreads();
  // -- per track: the error sum in observation order, the angle gate over the pairs of rays
  if (tid < nt && alive[tid]) {
    const int t = t0 + tid, b = offs[tid] - ob, e = offs[tid + 1] - ob;
    double m = 0.0;
    int count = 0;
    bool behind = false;
    for (int j = b; j < e; j++) {
      const double er = sd[(size_t)j * TRI_REC + 6];
      if (er < 0.0) { behind = true; break; }
      m += er;
      count++;
    }
    m = behind ? 100000.0 : m / count;
    mse[t] = m;
    bool wide = false;
    for (int i = b; i + 1 < e && !wide; i++) {
      const double* ai = sd + (size_t)i * TRI_REC;
      for (int j = i + 1; j < e; j++) {
        const double* dj = sd + (size_t)j * TRI_REC;
        if (ai[0] * dj[0] + ai[1] * dj[1] + ai[2] * dj[2] < cos_min) { wide = true; break; }
      }
    }
    ok[t] = !(sqrt(m) > th_error || !wide);
  }
}

// Reprojection only (RemovePointOutliers, sfm_incremental.cc:1831-1863): the errors with lane = observation, the sums per track.
__global__ __launch_bounds__(256) void k_reproject_staged(TrackPtrs T, const double* __restrict__ X, double* __restrict__ mse) {
  __shared__ double er_s[TRI_CAP];
  __shared__ int offs[TRI_TPB + 1];
  const int tid = threadIdx.x, t0 = blockIdx.x * TRI_TPB;
  const int nt = min(TRI_TPB, T.n_tracks - t0);
  if (tid <= nt) offs[tid] = T.off[t0 + tid];
  __syncthreads();
  const int ob = offs[0], nobs = offs[nt] - ob;
  if (nobs > TRI_CAP) {
    if (tid < nt) {
      const int t = t0 + tid;
      const double Xt[3] = {X[3 * (size_t)t], X[3 * (size_t)t + 1], X[3 * (size_t)t + 2]};
      mse[t] = track_mse(T, T.off[t], T.off[t + 1], Xt);
    }
    return;
  }
  for (int j = tid; j < nobs; j += 256) {
    const int i = ob + j;
    int lo = 0, hi = nt - 1;   // the track of observation i: the last one whose first observation is <= i
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (offs[mid] <= i) lo = mid; else hi = mid - 1; }
    const size_t t = (size_t)(t0 + lo);
    const double Xt[3] = {X[3 * t], X[3 * t + 1], X[3 * t + 2]};
    const int c = T.cam[i];
    const double* R = T.R + 9 * (size_t)c;
    const double* tt = T.t + 3 * (size_t)c;
    const double* fk = T.fk + 3 * (size_t)c;
    const double pc0 = R[0] * Xt[0] + R[1] * Xt[1] + R[2] * Xt[2] + tt[0];
    const double pc1 = R[3] * Xt[0] + R[4] * Xt[1] + R[5] * Xt[2] + tt[1];
    const double pc2 = R[6] * Xt[0] + R[7] * Xt[1] + R[8] * Xt[2] + tt[2];
    double err = -1.0;
    if (!(pc2 < 0)) {
      const double x = pc0 / pc2, y = pc1 / pc2;
      const double r2 = x * x + y * y;
      const double distortion = 1.0 + r2 * (fk[1] + fk[2] * r2);
      const double u = fk[0] * distortion * x, v = fk[0] * distortion * y;
      const double du = u - T.xy[2 * (size_t)i], dv = v - T.xy[2 * (size_t)i + 1];
      err = du * du + dv * dv;
    }
    er_s[j] = err;
  }
  __syncthreads();
  if (tid < nt) {
    const int b = offs[tid] - ob, e = offs[tid + 1] - ob;
    double m = 0.0;
    int count = 0;
    bool behind = false;
    for (int j = b; j < e; j++) {
      const double er = er_s[j];
      if (er < 0.0) { behind = true; break; }
      m += er;
      count++;
    }
    mse[t0 + tid] = behind ? 100000.0 : m / count;
  }
}

// DLT: streaming Givens QR of the 2k x 4 design matrix, one-sided Jacobi SVD of the 4x4 factor.
__global__ __launch_bounds__(256) void k_tri_dlt(TrackPtrs T, double th_error, double cos_min, double* __restrict__ Xo,
                                                  double* __restrict__ mse, uint8_t* __restrict__ ok) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T.n_tracks) return;
  const int b = T.off[t], e = T.off[t + 1];
  ok[t] = 0;
  mse[t] = 0.0;
  if (e - b < 2) return;
  double Rf[16];
#pragma unroll
  for (int k = 0; k < 16; k++) Rf[k] = 0.0;
  for (int i = b; i < e; i++) {
    const int c = T.cam[i];
    const double* R = T.R + 9 * (size_t)c;
    const double* tt = T.t + 3 * (size_t)c;
    const double f = T.fk[3 * (size_t)c];
    const double x = T.xy[2 * (size_t)i], y = T.xy[2 * (size_t)i + 1];
    const double M0[4] = {R[0], R[1], R[2], tt[0]}, M1[4] = {R[3], R[4], R[5], tt[1]}, M2[4] = {R[6], R[7], R[8], tt[2]};
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
      double v[4];
#pragma unroll
      for (int q = 0; q < 4; q++) v[q] = rr == 0 ? (-M1[q] * f + M2[q] * y) : (M0[q] * f - M2[q] * x);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (v[j] != 0.0) {
          const double a = Rf[j * 4 + j], bb = v[j];
          const double h = hypot(a, bb);
          const double cs = a / h, sn = bb / h;
#pragma unroll
          for (int q = j; q < 4; q++) {
            const double rj = Rf[j * 4 + q], vq = v[q];
            Rf[j * 4 + q] = cs * rj + sn * vq;
            v[q] = -sn * rj + cs * vq;
          }
        }
      }
    }
  }
  double V[16];
#pragma unroll
  for (int k = 0; k < 16; k++) V[k] = (k % 5 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 30; sweep++) {
    bool rotated = false;
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
      for (int q = p + 1; q < 4; q++) {
        double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          alpha += Rf[r * 4 + p] * Rf[r * 4 + p];
          beta += Rf[r * 4 + q] * Rf[r * 4 + q];
          gamma += Rf[r * 4 + p] * Rf[r * 4 + q];
        }
        if (!(fabs(gamma) <= 1e-15 * sqrt(alpha * beta) || gamma == 0.0)) {
          rotated = true;
          const double zeta = (beta - alpha) / (2.0 * gamma);
          const double tn = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          const double cs = 1.0 / sqrt(1.0 + tn * tn), sn = cs * tn;
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const double rp = Rf[r * 4 + p], rq = Rf[r * 4 + q];
            Rf[r * 4 + p] = cs * rp - sn * rq;
            Rf[r * 4 + q] = sn * rp + cs * rq;
            const double vp = V[r * 4 + p], vq = V[r * 4 + q];
            V[r * 4 + p] = cs * vp - sn * vq;
            V[r * 4 + q] = sn * vp + cs * vq;
          }
        }
      }
    if (!rotated) break;
  }
  double bn = 0.0, v0 = 0, v1 = 0, v2 = 0, v3 = 1;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    double nn = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) nn += Rf[r * 4 + q] * Rf[r * 4 + q];
    if (q == 0 || nn < bn) { bn = nn; v0 = V[q]; v1 = V[4 + q]; v2 = V[8 + q]; v3 = V[12 + q]; }
  }
  double X[3] = {v0 / v3, v1 / v3, v2 / v3};
  Xo[3 * (size_t)t] = X[0]; Xo[3 * (size_t)t + 1] = X[1]; Xo[3 * (size_t)t + 2] = X[2];
  const double m = track_mse(T, b, e, X);
  mse[t] = m;
  ok[t] = !(sqrt(m) > th_error || !track_angle_ok(T, b, e, X, cos_min));
}

__global__ __launch_bounds__(256) void k_reproject(TrackPtrs T, const double* __restrict__ X, double* __restrict__ mse) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T.n_tracks) return;
  const double Xt[3] = {X[3 * (size_t)t], X[3 * (size_t)t + 1], X[3 * (size_t)t + 2]};
  mse[t] = track_mse(T, T.off[t], T.off[t + 1], Xt);
}

struct F9 { double f[9]; };
__global__ __launch_bounds__(256) void k_epipolar(const float* __restrict__ pt1, const float* __restrict__ pt2, int n, F9 F, double th,
                                                   uint8_t* __restrict__ inlier) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double x1 = pt1[2 * (size_t)i], y1 = pt1[2 * (size_t)i + 1], x2 = pt2[2 * (size_t)i], y2 = pt2[2 * (size_t)i + 1];
  double l0 = F.f[0] * x1 + F.f[1] * y1 + F.f[2], l1 = F.f[3] * x1 + F.f[4] * y1 + F.f[5], l2 = F.f[6] * x1 + F.f[7] * y1 + F.f[8];
  const double nn = sqrt(l0 * l0 + l1 * l1);
  l0 /= nn; l1 /= nn; l2 /= nn;
  inlier[i] = fabs(l0 * x2 + l1 * y2 + l2) < th;
}

// ---- host ----
static bool tri_staged() {   // MSFM_TRI_STAGED=0: the thread-per-track kernels (comparison)
  static const bool on = [] { const char* e = getenv("MSFM_TRI_STAGED"); return !(e && atoi(e) == 0); }();
  return on;
}
struct TrackDev {
  DevBuf<int> off, cam;
  DevBuf<double> xy, R, t, c, fk;
  TrackPtrs ptrs;
};

static int upload_tracks(msfm_ctx* ctx, const msfm_tracks* T, TrackDev& D) {
  if (!ctx || !T || T->n_tracks < 0 || T->n_cams <= 0 || !T->track_off || !T->cam_R || !T->cam_t || !T->cam_c || !T->cam_fk)
    return msfm_set_error(ctx, MSFM_E_INVAL, "tracks: null arrays");
  const int n = T->n_tracks;
  if (T->track_off[0] != 0) return msfm_set_error(ctx, MSFM_E_INVAL, "track_off[0] != 0");
  for (int i = 0; i < n; i++) if (T->track_off[i + 1] < T->track_off[i]) return msfm_set_error(ctx, MSFM_E_INVAL, "track_off not monotone");
  const int no = T->track_off[n];
  for (int i = 0; i < no; i++) if (T->track_cam[i] < 0 || T->track_cam[i] >= T->n_cams) return msfm_set_error(ctx, MSFM_E_INVAL, "track_cam[%d] out of range", i);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  HIP_TRY(ctx, D.off.alloc(n + 1)); HIP_TRY(ctx, D.off.upload(T->track_off, n + 1, s));
  HIP_TRY(ctx, D.cam.alloc(std::max(1, no))); HIP_TRY(ctx, D.cam.upload(T->track_cam, no, s));
  HIP_TRY(ctx, D.xy.alloc(std::max(1, 2 * no))); HIP_TRY(ctx, D.xy.upload(T->track_xy, 2 * (size_t)no, s));
  HIP_TRY(ctx, D.R.alloc(9 * (size_t)T->n_cams)); HIP_TRY(ctx, D.R.upload(T->cam_R, 9 * (size_t)T->n_cams, s));
  HIP_TRY(ctx, D.t.alloc(3 * (size_t)T->n_cams)); HIP_TRY(ctx, D.t.upload(T->cam_t, 3 * (size_t)T->n_cams, s));
  HIP_TRY(ctx, D.c.alloc(3 * (size_t)T->n_cams)); HIP_TRY(ctx, D.c.upload(T->cam_c, 3 * (size_t)T->n_cams, s));
  HIP_TRY(ctx, D.fk.alloc(3 * (size_t)T->n_cams)); HIP_TRY(ctx, D.fk.upload(T->cam_fk, 3 * (size_t)T->n_cams, s));
  D.ptrs = TrackPtrs{n, D.off.p, D.cam.p, D.xy.p, D.R.p, D.t.p, D.c.p, D.fk.p};
  return MSFM_OK;
}

static int triangulate(msfm_ctx* ctx, const msfm_tracks* T, double th_error, double th_angle, double* X, double* mse, uint8_t* ok,
                       bool dlt) {
  if (!X || !mse || !ok) return MSFM_E_INVAL;
  TrackDev D;
  MSFM_TRY(upload_tracks(ctx, T, D));
  const int n = T->n_tracks;
  if (n == 0) return MSFM_OK;
  hipStream_t s = ctx->stream;
  DevBuf<double> dX, dm;
  DevBuf<uint8_t> dok;
  HIP_TRY(ctx, dX.alloc(3 * (size_t)n)); HIP_TRY(ctx, dm.alloc(n)); HIP_TRY(ctx, dok.alloc(n));
  HIP_TRY(ctx, dX.upload(X, 3 * (size_t)n, s));  // X is in/out: untouched on LLT failure
  {
    KTimer t(ctx, dlt ? "tri_dlt" : "tri_midpoint");
    if (dlt) hipLaunchKernelGGL(k_tri_dlt, dim3(cdiv(n, 256)), dim3(256), 0, s, D.ptrs, th_error, cos(th_angle), dX.p, dm.p, dok.p);
    else if (tri_staged()) hipLaunchKernelGGL(k_tri_midpoint_staged, dim3(cdiv(n, TRI_TPB)), dim3(256), 0, s, D.ptrs, th_error, cos(th_angle), dX.p, dm.p, dok.p);
    else hipLaunchKernelGGL(k_tri_midpoint, dim3(cdiv(n, 256)), dim3(256), 0, s, D.ptrs, th_error, cos(th_angle), dX.p, dm.p, dok.p);
  }
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(X, dX.p, sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(mse, dm.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipMemcpyAsync(ok, dok.p, (size_t)n, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return MSFM_OK;
}

// Point3D::Trianglate2 + Reprojection + the angle gate on tracks that are already resident (all pointers device memory);
// X is in/out.  msfm_chain_triangulate (chain.hip) calls it on the tracks it built.
int tri_midpoint_dev(msfm_ctx* ctx, const TrackPtrs& T, double th_error, double th_angle, double* dX, double* dmse, uint8_t* dok) {
  if (T.n_tracks == 0) return MSFM_OK;
  KTimer t(ctx, "tri_midpoint");
  if (tri_staged()) hipLaunchKernelGGL(k_tri_midpoint_staged, dim3(cdiv(T.n_tracks, TRI_TPB)), dim3(256), 0, ctx->stream, T, th_error, cos(th_angle), dX, dmse, dok);
  else hipLaunchKernelGGL(k_tri_midpoint, dim3(cdiv(T.n_tracks, 256)), dim3(256), 0, ctx->stream, T, th_error, cos(th_angle), dX, dmse, dok);
  HIP_TRY(ctx, hipGetLastError());
  return MSFM_OK;
}

MSFM_API int msfm_triangulate_midpoint_batch(msfm_ctx* ctx, const msfm_tracks* T, double th_error, double th_angle, double* X,
                                             double* mse, uint8_t* ok) {
  if (!ctx) return MSFM_E_INVAL;
  return triangulate(ctx, T, th_error, th_angle, X, mse, ok, false);
}

MSFM_API int msfm_triangulate_dlt_batch(msfm_ctx* ctx, const msfm_tracks* T, double th_error, double th_angle, double* X,
                                        double* mse, uint8_t* ok) {
  if (!ctx) return MSFM_E_INVAL;
  return triangulate(ctx, T, th_error, th_angle, X, mse, ok, true);
}

MSFM_API int msfm_reproject_mse_batch(msfm_ctx* ctx, const msfm_tracks* T, const double* X, double* mse) {
  if (!ctx || !X || !mse) return MSFM_E_INVAL;
  TrackDev D;
  MSFM_TRY(upload_tracks(ctx, T, D));
  const int n = T->n_tracks;
  if (n == 0) return MSFM_OK;
  hipStream_t s = ctx->stream;
  DevBuf<double> dX, dm;
  HIP_TRY(ctx, dX.alloc(3 * (size_t)n)); HIP_TRY(ctx, dm.alloc(n));
  HIP_TRY(ctx, dX.upload(X, 3 * (size_t)n, s));
  {
    KTimer t(ctx, "tri_reproject");
    if (tri_staged()) hipLaunchKernelGGL(k_reproject_staged, dim3(cdiv(n, TRI_TPB)), dim3(256), 0, s, D.ptrs, dX.p, dm.p);
    else hipLaunchKernelGGL(k_reproject, dim3(cdiv(n, 256)), dim3(256), 0, s, D.ptrs, dX.p, dm.p);
  }
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(mse, dm.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return MSFM_OK;
}

MSFM_API int msfm_epipolar_filter(msfm_ctx* ctx, const float* pt1, const float* pt2, int n, const double F[9], double th,
                                  uint8_t* inlier) {
  if (!ctx || n < 0 || (n > 0 && (!pt1 || !pt2 || !inlier)) || !F) return MSFM_E_INVAL;
  if (n == 0) return MSFM_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  DevBuf<float> d1, d2;
  DevBuf<uint8_t> di;
  HIP_TRY(ctx, d1.alloc(2 * (size_t)n)); HIP_TRY(ctx, d2.alloc(2 * (size_t)n)); HIP_TRY(ctx, di.alloc(n));
  HIP_TRY(ctx, d1.upload(pt1, 2 * (size_t)n, s)); HIP_TRY(ctx, d2.upload(pt2, 2 * (size_t)n, s));
  F9 f;
  for (int k = 0; k < 9; k++) f.f[k] = F[k];
  hipLaunchKernelGGL(k_epipolar, dim3(cdiv(n, 256)), dim3(256), 0, s, d1.p, d2.p, n, f, th, di.p);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(inlier, di.p, (size_t)n, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return MSFM_OK;
}
