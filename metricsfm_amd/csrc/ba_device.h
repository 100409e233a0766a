// Device-side arithmetic shared by the BA kernels: the reference's projection model with
// closed-form derivatives, the Huber corrector and small dense helpers.  FP64 throughout.
#pragma once
#include <hip/hip_runtime.h>

// AngleAxisRotatePoint (SfM/src/utils/basic_funcs.cc:160-225 == ceres/rotation.h, called from
// reprojection_error_pose_cam_xyz.h:41) + translation + pinhole with radial distortion
// (reprojection_error_pose_cam_xyz.h:44-63; the other four functors share the model).
// J (if non-null) is 2x12 row-major: [d pose(6) | d cam(3) | d xyz(3)], already times weight.
// The rotation derivative is the exact derivative of the evaluated formula (including the
// first-order branch at theta^2 <= DBL_EPSILON), i.e. what Ceres' Jets propagate.
__device__ __forceinline__ void msfm_reproj(const double* __restrict__ pose, const double* __restrict__ cam,
                                            const double* __restrict__ xyz, double ox, double oy, double weight,
                                            double* r, double* J) {
  const double a0 = pose[0], a1 = pose[1], a2 = pose[2];
  const double X0 = xyz[0], X1 = xyz[1], X2 = xyz[2];
  double p0, p1, p2;
  double dpdw[9];
  double R[9];
  const double theta2 = a0 * a0 + a1 * a1 + a2 * a2;
  if (theta2 > 2.220446049250313e-16) {
    const double theta = sqrt(theta2);
    double s, c;
    sincos(theta, &s, &c);
    const double ti = 1.0 / theta;
    const double w0 = a0 * ti, w1 = a1 * ti, w2 = a2 * ti;
    const double wx0 = w1 * X2 - w2 * X1, wx1 = w2 * X0 - w0 * X2, wx2 = w0 * X1 - w1 * X0;
    const double wdx = w0 * X0 + w1 * X1 + w2 * X2;
    const double omc = 1.0 - c;
    const double tmp = wdx * omc;
    p0 = X0 * c + wx0 * s + w0 * tmp;
    p1 = X1 * c + wx1 * s + w1 * tmp;
    p2 = X2 * c + wx2 * s + w2 * tmp;
    if (J) {
      const double w[3] = {w0, w1, w2};
      const double wx[3] = {wx0, wx1, wx2};
      const double Xv[3] = {X0, X1, X2};
#pragma unroll
      for (int j = 0; j < 3; j++) {
        double dw[3];
#pragma unroll
        for (int i = 0; i < 3; i++) dw[i] = ((i == j ? 1.0 : 0.0) - w[i] * w[j]) * ti;
        const double dwx[3] = {dw[1] * X2 - dw[2] * X1, dw[2] * X0 - dw[0] * X2, dw[0] * X1 - dw[1] * X0};
        const double dwdx = dw[0] * X0 + dw[1] * X1 + dw[2] * X2;
        const double dc = -s * w[j], ds = c * w[j];
        const double dtmp = dwdx * omc - wdx * dc;
#pragma unroll
        for (int i = 0; i < 3; i++) dpdw[i * 3 + j] = Xv[i] * dc + dwx[i] * s + wx[i] * ds + dw[i] * tmp + w[i] * dtmp;
      }
      R[0] = c + w0 * w0 * omc;      R[1] = w0 * w1 * omc - w2 * s; R[2] = w1 * s + w0 * w2 * omc;
      R[3] = w2 * s + w0 * w1 * omc; R[4] = c + w1 * w1 * omc;      R[5] = -w0 * s + w1 * w2 * omc;
      R[6] = -w1 * s + w0 * w2 * omc; R[7] = w0 * s + w1 * w2 * omc; R[8] = c + w2 * w2 * omc;
    }
  } else {
    p0 = X0 + (a1 * X2 - a2 * X1);
    p1 = X1 + (a2 * X0 - a0 * X2);
    p2 = X2 + (a0 * X1 - a1 * X0);
    if (J) {
      dpdw[0] = 0;   dpdw[1] = X2;  dpdw[2] = -X1;
      dpdw[3] = -X2; dpdw[4] = 0;   dpdw[5] = X0;
      dpdw[6] = X1;  dpdw[7] = -X0; dpdw[8] = 0;
      R[0] = 1;   R[1] = -a2; R[2] = a1;
      R[3] = a2;  R[4] = 1;   R[5] = -a0;
      R[6] = -a1; R[7] = a0;  R[8] = 1;
    }
  }
  p0 += pose[3]; p1 += pose[4]; p2 += pose[5];
  const double iz = 1.0 / p2;
  const double xp = p0 * iz, yp = p1 * iz;
  const double f = cam[0], l1 = cam[1], l2 = cam[2];
  const double r2 = xp * xp + yp * yp;
  const double dist = 1.0 + r2 * (l1 + l2 * r2);
  r[0] = weight * (f * dist * xp - ox);
  r[1] = weight * (f * dist * yp - oy);
  if (!J) return;
  const double dd = l1 + 2.0 * l2 * r2;
  const double uxp = f * (dist + 2.0 * xp * xp * dd), uyp = f * 2.0 * xp * yp * dd;
  const double vxp = uyp, vyp = f * (dist + 2.0 * yp * yp * dd);
  const double up[3] = {uxp * iz, uyp * iz, -(uxp * xp + uyp * yp) * iz};
  const double vp[3] = {vxp * iz, vyp * iz, -(vxp * xp + vyp * yp) * iz};
#pragma unroll
  for (int j = 0; j < 3; j++) {
    J[j] = weight * (up[0] * dpdw[j] + up[1] * dpdw[3 + j] + up[2] * dpdw[6 + j]);
    J[12 + j] = weight * (vp[0] * dpdw[j] + vp[1] * dpdw[3 + j] + vp[2] * dpdw[6 + j]);
    J[3 + j] = weight * up[j];
    J[12 + 3 + j] = weight * vp[j];
    J[9 + j] = weight * (up[0] * R[j] + up[1] * R[3 + j] + up[2] * R[6 + j]);
    J[12 + 9 + j] = weight * (vp[0] * R[j] + vp[1] * R[3 + j] + vp[2] * R[6 + j]);
  }
  J[6] = weight * dist * xp;        J[12 + 6] = weight * dist * yp;
  J[7] = weight * f * r2 * xp;      J[12 + 7] = weight * f * r2 * yp;
  J[8] = weight * f * r2 * r2 * xp; J[12 + 8] = weight * f * r2 * r2 * yp;
}

// ceres::HuberLoss(a) (constructed at optimizer.cc:84): rho0 = rho(s), rho1 = rho'(s).
__device__ __forceinline__ void msfm_huber(double a, double s, double& rho0, double& rho1) {
  const double b = a * a;
  if (s > b) {
    const double r = sqrt(s);
    rho0 = 2.0 * a * r - b;
    rho1 = fmax(2.2250738585072014e-308, a / r);
  } else {
    rho0 = s;
    rho1 = 1.0;
  }
}

// Deterministic wave / block sums (fixed butterfly, fixed wave order).
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}
// blockDim.x must be 256.  Result valid in thread 0.
__device__ __forceinline__ double block_sum256(double v, double* sh /*[4]*/) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) t = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  return t;
}
__device__ __forceinline__ double block_max256(double v, double* sh) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) t = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
  __syncthreads();
  return t;
}
