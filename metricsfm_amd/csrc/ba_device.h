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
// What depends on the camera's angle-axis vector alone: which branch of the rotation formula applies, and sin, cos and
// 1 / |a|.  A camera is seen by thousands of observations per evaluation; msfm_rot_prepare computes these four numbers
// once per camera and parameter set (rot[4 c ..]: flag, sin, cos, 1/theta) with the operations msfm_reproj would use.
__device__ __forceinline__ void msfm_rot_prepare(const double* __restrict__ pose, double* __restrict__ rc) {
  const double a0 = pose[0], a1 = pose[1], a2 = pose[2];
  const double theta2 = a0 * a0 + a1 * a1 + a2 * a2;
  double s = 0.0, c = 1.0, ti = 0.0, flag = 0.0;
  if (theta2 > 2.220446049250313e-16) {
    const double theta = sqrt(theta2);
    sincos(theta, &s, &c);
    ti = 1.0 / theta;
    flag = 1.0;
  }
  rc[0] = flag; rc[1] = s; rc[2] = c; rc[3] = ti;
}

__device__ __forceinline__ void msfm_reproj(const double* __restrict__ pose, const double* __restrict__ rc, const double* __restrict__ cam,
                                            const double* __restrict__ xyz, double ox, double oy, double weight,
                                            double* r, double* J) {
  const double a0 = pose[0], a1 = pose[1], a2 = pose[2];
  const double X0 = xyz[0], X1 = xyz[1], X2 = xyz[2];
  double p0, p1, p2;
  double dpdw[9];
  double R[9];
  if (rc[0] != 0.0) {
    const double s = rc[1], c = rc[2], ti = rc[3];
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
  // (with Jacobians: Ceres' Jet division, value part f.a * (1 / g.a); without: the functor's plain division - one rounding
  //  apart, as in the reference's two evaluation modes)
  const double iz = 1.0 / p2;
  const double xp = J ? p0 * iz : p0 / p2, yp = J ? p1 * iz : p1 / p2;
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

// The row as the BACK SUBSTITUTION needs it (round 5): the residual, the point block of the Jacobian and the row's product with a
// GIVEN camera / intrinsics step - J_c zeta_c + J_m zeta_m as the directional derivative of the residual along (zeta_w, zeta_t,
// zeta_m) instead of the twelve + six Jacobian columns contracted afterwards.  The rotation part is the derivative of the
// evaluated formula along zeta_w (the same terms msfm_reproj forms per column j, with e_j replaced by zeta_w: about a third of
// the operations), the rest is shared with msfm_reproj line for line.  r, Jp (2 x 3, times weight) and d = J_c zeta_c + J_m zeta_m
// (2, times weight) come out unscaled and uncorrected like msfm_reproj's.
__device__ __forceinline__ void msfm_reproj_dir(const double* __restrict__ pose, const double* __restrict__ rc, const double* __restrict__ cam,
                                                const double* __restrict__ xyz, double ox, double oy, double weight, const double (&zw)[3],
                                                const double (&zt)[3], const double (&zm)[3], double* r, double* Jp, double* d) {
  const double a0 = pose[0], a1 = pose[1], a2 = pose[2];
  const double X0 = xyz[0], X1 = xyz[1], X2 = xyz[2];
  double p0, p1, p2, dp0, dp1, dp2;
  double R[9];
  if (rc[0] != 0.0) {
    const double s = rc[1], c = rc[2], ti = rc[3];
    const double w0 = a0 * ti, w1 = a1 * ti, w2 = a2 * ti;
    const double wx0 = w1 * X2 - w2 * X1, wx1 = w2 * X0 - w0 * X2, wx2 = w0 * X1 - w1 * X0;
    const double wdx = w0 * X0 + w1 * X1 + w2 * X2;
    const double omc = 1.0 - c;
    const double tmp = wdx * omc;
    p0 = X0 * c + wx0 * s + w0 * tmp;
    p1 = X1 * c + wx1 * s + w1 * tmp;
    p2 = X2 * c + wx2 * s + w2 * tmp;
    // along zeta_w: d theta = w . zeta, d w = (zeta - w (w . zeta)) / theta
    const double dth = w0 * zw[0] + w1 * zw[1] + w2 * zw[2];
    const double dw0 = (zw[0] - w0 * dth) * ti, dw1 = (zw[1] - w1 * dth) * ti, dw2 = (zw[2] - w2 * dth) * ti;
    const double dwx0 = dw1 * X2 - dw2 * X1, dwx1 = dw2 * X0 - dw0 * X2, dwx2 = dw0 * X1 - dw1 * X0;
    const double dwdx = dw0 * X0 + dw1 * X1 + dw2 * X2;
    const double dc = -s * dth, ds = c * dth;
    const double dtmp = dwdx * omc - wdx * dc;
    dp0 = X0 * dc + dwx0 * s + wx0 * ds + dw0 * tmp + w0 * dtmp;
    dp1 = X1 * dc + dwx1 * s + wx1 * ds + dw1 * tmp + w1 * dtmp;
    dp2 = X2 * dc + dwx2 * s + wx2 * ds + dw2 * tmp + w2 * dtmp;
    R[0] = c + w0 * w0 * omc;      R[1] = w0 * w1 * omc - w2 * s; R[2] = w1 * s + w0 * w2 * omc;
    R[3] = w2 * s + w0 * w1 * omc; R[4] = c + w1 * w1 * omc;      R[5] = -w0 * s + w1 * w2 * omc;
    R[6] = -w1 * s + w0 * w2 * omc; R[7] = w0 * s + w1 * w2 * omc; R[8] = c + w2 * w2 * omc;
  } else {
    p0 = X0 + (a1 * X2 - a2 * X1);
    p1 = X1 + (a2 * X0 - a0 * X2);
    p2 = X2 + (a0 * X1 - a1 * X0);
    dp0 = zw[1] * X2 - zw[2] * X1;     // d(w x X) along zeta_w
    dp1 = zw[2] * X0 - zw[0] * X2;
    dp2 = zw[0] * X1 - zw[1] * X0;
    R[0] = 1;   R[1] = -a2; R[2] = a1;
    R[3] = a2;  R[4] = 1;   R[5] = -a0;
    R[6] = -a1; R[7] = a0;  R[8] = 1;
  }
  p0 += pose[3]; p1 += pose[4]; p2 += pose[5];
  dp0 += zt[0]; dp1 += zt[1]; dp2 += zt[2];
  const double iz = 1.0 / p2;
  const double xp = p0 * iz, yp = p1 * iz;   // (the linearisation pass: Jet division, as msfm_reproj with J)
  const double f = cam[0], l1 = cam[1], l2 = cam[2];
  const double r2 = xp * xp + yp * yp;
  const double dist = 1.0 + r2 * (l1 + l2 * r2);
  r[0] = weight * (f * dist * xp - ox);
  r[1] = weight * (f * dist * yp - oy);
  const double dd = l1 + 2.0 * l2 * r2;
  const double uxp = f * (dist + 2.0 * xp * xp * dd), uyp = f * 2.0 * xp * yp * dd;
  const double vxp = uyp, vyp = f * (dist + 2.0 * yp * yp * dd);
  const double up[3] = {uxp * iz, uyp * iz, -(uxp * xp + uyp * yp) * iz};
  const double vp[3] = {vxp * iz, vyp * iz, -(vxp * xp + vyp * yp) * iz};
#pragma unroll
  for (int j = 0; j < 3; j++) {
    Jp[j] = weight * (up[0] * R[j] + up[1] * R[3 + j] + up[2] * R[6 + j]);
    Jp[3 + j] = weight * (vp[0] * R[j] + vp[1] * R[3 + j] + vp[2] * R[6 + j]);
  }
  const double fr2 = f * r2;
  d[0] = weight * ((up[0] * dp0 + up[1] * dp1 + up[2] * dp2) + xp * (dist * zm[0] + fr2 * zm[1] + fr2 * r2 * zm[2]));
  d[1] = weight * ((vp[0] * dp0 + vp[1] * dp1 + vp[2] * dp2) + yp * (dist * zm[0] + fr2 * zm[1] + fr2 * r2 * zm[2]));
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
// The butterfly's partner values travel by DPP (inside a row of 16 lanes) and by the gfx950 row / half-wave swaps - VALU
// instructions - instead of ds_bpermute through the LDS pipe (two per step and double: 936 of them in one k_ftf wave).
// Every step adds the same two numbers as `v += __shfl_xor(v, off)` did, so results are bit-identical to that form.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
#define MSFM_DPP_XOR1 0xB1          // quad_perm [1,0,3,2]
#define MSFM_DPP_XOR2 0x4E          // quad_perm [2,3,0,1]
#define MSFM_DPP_ROR8 0x128         // row_ror:8 = lane ^ 8 inside a row
#define MSFM_DPP_HALF_MIRROR 0x141  // lane -> 7 - lane inside each half row
__device__ __forceinline__ int dpp_xor4_b32(int x) {
  // lane ^ 4: quads 0 and 2 take from the quad above (row_shl:4), quads 1 and 3 from the quad below (row_shr:4)
  const int t = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xf, 0x5, false);
  return __builtin_amdgcn_update_dpp(t, x, 0x114, 0xf, 0xa, false);
}
__device__ __forceinline__ double dpp_xor4_f64(double v) {
  return __hiloint2double(dpp_xor4_b32(__double2hiint(v)), dpp_xor4_b32(__double2loint(v)));
}
// the two halves of the row pairs / of the wave side by side: a = [R0 R0 R2 R2], b = [R1 R1 R3 R3] (rows of 16 lanes),
// resp. a = [lower 32, lower 32], b = [upper 32, upper 32]; a + b is the xor-16 (xor-32) step in every lane
__device__ __forceinline__ void swap16_f64(double v, double& a, double& b) {
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  a = __hiloint2double(h[0], l[0]);
  b = __hiloint2double(h[1], l[1]);
}
__device__ __forceinline__ void swap32_f64(double v, double& a, double& b) {
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  a = __hiloint2double(h[0], l[0]);
  b = __hiloint2double(h[1], l[1]);
}
__device__ __forceinline__ double wave_sum(double v) {
  double a, b;
  swap32_f64(v, a, b); v = a + b;
  swap16_f64(v, a, b); v = a + b;
  v += dpp_f64<MSFM_DPP_ROR8>(v);
  v += dpp_xor4_f64(v);
  v += dpp_f64<MSFM_DPP_XOR2>(v);
  v += dpp_f64<MSFM_DPP_XOR1>(v);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
  double a, b;
  swap32_f64(v, a, b); v = fmax(a, b);
  swap16_f64(v, a, b); v = fmax(a, b);
  v = fmax(v, dpp_f64<MSFM_DPP_ROR8>(v));
  v = fmax(v, dpp_xor4_f64(v));
  v = fmax(v, dpp_f64<MSFM_DPP_XOR2>(v));
  v = fmax(v, dpp_f64<MSFM_DPP_XOR1>(v));
  return v;
}
// Sums of N per-lane values over the wave, every sum with the pairing tree of wave_sum (xor 32, 16, 8, 4, 2, 1), but as a
// reduce-scatter: at every step a lane keeps only half of its values (the lower half of the index range if its bit of the
// step is clear, the upper half otherwise) and adds its partner's copies of those - N + N/2 + N/4 + ... additions instead
// of 6 N, and after the last step every value sits in exactly one lane.  The two widest steps are one
// v_permlane{32,16}_swap per dword and pair of values.  Returns this lane's value and its index (-1: none).
// Bit-identical to wave_sum of each value (missing partners of an odd count are +0.0).
template <int D>
__device__ __forceinline__ double partner_f64(double t) {
  if constexpr (D == 8) return dpp_f64<MSFM_DPP_ROR8>(t);
  else if constexpr (D == 4) return dpp_xor4_f64(t);
  else if constexpr (D == 2) return dpp_f64<MSFM_DPP_XOR2>(t);
  else return dpp_f64<MSFM_DPP_XOR1>(t);
}
template <int D, int n, int N>
__device__ __forceinline__ void reduce_scatter_step(double (&v)[N], int lane) {
  constexpr int h = (n + 1) / 2;
  const bool low = (lane & D) == 0;
#pragma unroll
  for (int j = 0; j < h; j++) {
    const double a = v[j], b = (h + j < n) ? v[h + j] : 0.0;
    if constexpr (D == 32) { double x, y; const unsigned al = __double2loint(a), ah = __double2hiint(a), bl = __double2loint(b), bh = __double2hiint(b);
      const auto l = __builtin_amdgcn_permlane32_swap(al, bl, false, false); const auto hh = __builtin_amdgcn_permlane32_swap(ah, bh, false, false);
      x = __hiloint2double(hh[0], l[0]); y = __hiloint2double(hh[1], l[1]); v[j] = x + y;
    } else if constexpr (D == 16) { double x, y; const unsigned al = __double2loint(a), ah = __double2hiint(a), bl = __double2loint(b), bh = __double2hiint(b);
      const auto l = __builtin_amdgcn_permlane16_swap(al, bl, false, false); const auto hh = __builtin_amdgcn_permlane16_swap(ah, bh, false, false);
      x = __hiloint2double(hh[0], l[0]); y = __hiloint2double(hh[1], l[1]); v[j] = x + y;
    } else {
      const double t = low ? b : a;            // what the partner keeps
      const double mine = low ? a : b;
      v[j] = mine + partner_f64<D>(t);
    }
  }
}
template <int N>
__device__ __forceinline__ void wave_reduce_scatter(double (&v)[N], int lane, double& val, int& idx) {
  constexpr int n1 = (N + 1) / 2, n2 = (n1 + 1) / 2, n3 = (n2 + 1) / 2, n4 = (n3 + 1) / 2, n5 = (n4 + 1) / 2;
  static_assert(N <= 64 && (n5 + 1) / 2 == 1, "at most 64 values");
  reduce_scatter_step<32, N>(v, lane);
  reduce_scatter_step<16, n1>(v, lane);
  reduce_scatter_step<8, n2>(v, lane);
  reduce_scatter_step<4, n3>(v, lane);
  reduce_scatter_step<2, n4>(v, lane);
  reduce_scatter_step<1, n5>(v, lane);
  val = v[0];
  // which value that is: every lane has the same number of slots n (the upper half of an odd count is padded with one
  // empty slot), of which the first nv hold values lo, lo + 1, ...
  int lo = 0, nv = N, n = N;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const int h = (n + 1) / 2;
    if (lane & d) { lo += h; nv = max(0, nv - h); } else nv = min(nv, h);
    n = h;
  }
  idx = nv > 0 ? lo : -1;
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
