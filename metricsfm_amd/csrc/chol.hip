// Dense FP64 Cholesky of the reduced camera system + the two triangular solves, for gfx950.
//
// Replaces `lhs.selfadjointView<Upper>().llt()` + `solve` inside Ceres' DENSE_SCHUR solver,
// reached from ceres::Solve at SfM/src/optimizer.cc:133 (options :47) and slam_gps.cc:841.
//
// Layout: M is npad x npad row-major (ld = npad, npad a multiple of 64), lower triangle.
// Rows/cols [0,n) hold S; row n holds rhs^T (the forward substitution L w = rhs then falls out
// of the factorisation: row n of the factor is w^T); indices > n are zero padding.
// Right-looking by 64-column panels:  potrf64 (1 workgroup) -> trsm (thread per row,
// substitution against the 64x64 factor in LDS) -> syrk (64x64 output tiles, K = 64,
// v_mfma_f64_16x16x4_f64 from padded LDS tiles).  The back substitution L^T z = w runs block
// by block with explicitly inverted diagonal blocks (k_trinv) so each step is a mat-vec.
#include "common.h"

#define NB 64
#define LDT 66  // LDS row stride in doubles: 132 dwords = 4 mod 64 -> conflict-free ds_read_b64 fragments

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------
// potrf of the 64x64 diagonal block at (j0,j0).  Columns with global index >= n are left
// alone (padding / rhs row).  fail[0] is set when a pivot is not positive (Eigen LLT:
// info() != Success -> Ceres LINEAR_SOLVER_FAILURE).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_potrf64(double* __restrict__ M, int ld, int j0, int n, int* fail) {
  __shared__ double a[NB][NB + 1];
  const int tid = threadIdx.x;
  for (int e = tid; e < NB * NB; e += 256) {
    const int r = e >> 6, c = e & 63;
    a[r][c] = (c <= r) ? M[(size_t)(j0 + r) * ld + j0 + c] : 0.0;
  }
  __syncthreads();
  const int ncol = min(NB, n - j0);
  for (int c = 0; c < ncol; c++) {
    const double d = a[c][c];
    if (!(d > 0.0)) {
      if (tid == 0) atomicOr(fail, 1);
      return;  // uniform: every thread sees the same d
    }
    const double rs = 1.0 / sqrt(d);
    __syncthreads();  // everyone has read a[c][c]
    if (tid < NB) {
      if (tid > c) a[tid][c] *= rs;
      else if (tid == c) a[c][c] = d * rs;
    }
    __syncthreads();
    // trailing update, lower triangle: a[r][q] -= a[r][c] * a[q][c], c < q <= r
    const int m = NB - 1 - c;  // rows/cols c+1 .. 63
    for (int e = tid; e < m * m; e += 256) {
      const int rr = e / m, qq = e - rr * m;
      if (qq <= rr) {
        const int r = c + 1 + rr, q = c + 1 + qq;
        a[r][q] -= a[r][c] * a[q][c];
      }
    }
    // next iteration's first barrier orders these writes before the reads
    __syncthreads();
  }
  for (int e = tid; e < NB * NB; e += 256) {
    const int r = e >> 6, c = e & 63;
    if (c <= r) M[(size_t)(j0 + r) * ld + j0 + c] = a[r][c];
  }
}

// ---------------------------------------------------------------------------------------
// trsm: rows r in [j0+64, nrows): X L11^T = A21  ->  forward substitution per row.
// One thread per row; L11 (64x64) in LDS, read as broadcasts.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_trsm64(double* __restrict__ M, int ld, int j0, int n, int nrows) {
  __shared__ double L[NB][NB + 1];
  __shared__ double rdiag[NB];
  __shared__ double tile[64][NB + 1];  // staging of 64 rows x 64 cols, coalesced <-> per-thread rows
  const int tid = threadIdx.x;
  for (int e = tid; e < NB * NB; e += 256) {
    const int r = e >> 6, c = e & 63;
    L[r][c] = (c <= r) ? M[(size_t)(j0 + r) * ld + j0 + c] : 0.0;
  }
  __syncthreads();
  const int ncol = min(NB, n - j0);
  if (tid < NB) rdiag[tid] = (tid < ncol) ? 1.0 / L[tid][tid] : 1.0;
  // this workgroup owns 64 rows; only wave 0 substitutes, all four waves move data
  const int r0 = j0 + NB + blockIdx.x * 64;
  for (int e = tid; e < 64 * NB; e += 256) {
    const int r = e >> 6, c = e & 63;
    tile[r][c] = (r0 + r < nrows) ? M[(size_t)(r0 + r) * ld + j0 + c] : 0.0;
  }
  __syncthreads();
  if (tid < 64) {
    double x[NB];
#pragma unroll
    for (int c = 0; c < NB; c++) x[c] = tile[tid][c];
#pragma unroll
    for (int c = 0; c < NB; c++) {
      if (c < ncol) {
        const double xc = x[c] * rdiag[c];
        x[c] = xc;
#pragma unroll
        for (int q = c + 1; q < NB; q++) x[q] -= xc * L[q][c];
      }
    }
#pragma unroll
    for (int c = 0; c < NB; c++) tile[tid][c] = x[c];
  }
  __syncthreads();
  for (int e = tid; e < 64 * NB; e += 256) {
    const int r = e >> 6, c = e & 63;
    if (r0 + r < nrows) M[(size_t)(r0 + r) * ld + j0 + c] = tile[r][c];
  }
}

// ---------------------------------------------------------------------------------------
// syrk: for every 64x64 tile (I >= J) of the trailing matrix, C_IJ -= P_I * P_J^T where
// P = the 64-column panel just solved.  4 waves, each a 32x32 quadrant = 2x2 MFMA tiles.
// blockIdx.x enumerates the lower-triangular tile pairs.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_syrk64(double* __restrict__ M, int ld, int j0, int ntile) {
  __shared__ double As[64 * LDT];
  __shared__ double Bs[64 * LDT];
  // tile pair from linear index: I = floor((sqrt(8b+1)-1)/2), J = b - I(I+1)/2
  const int b = blockIdx.x;
  int I = (int)((sqrt(8.0 * b + 1.0) - 1.0) * 0.5);
  while ((I + 1) * (I + 2) / 2 <= b) I++;
  while (I * (I + 1) / 2 > b) I--;
  const int J = b - I * (I + 1) / 2;
  (void)ntile;
  const int t0 = j0 + NB;  // first trailing row
  const int ri = t0 + I * 64, rj = t0 + J * 64;
  const int tid = threadIdx.x;
  // coalesced 16-byte loads: 64 rows x 64 cols
  for (int e = tid; e < 64 * 32; e += 256) {
    const int r = e >> 5, c2 = (e & 31) * 2;
    const d2 va = *reinterpret_cast<const d2*>(&M[(size_t)(ri + r) * ld + j0 + c2]);
    As[r * LDT + c2] = va.x;
    As[r * LDT + c2 + 1] = va.y;
    const d2 vb = *reinterpret_cast<const d2*>(&M[(size_t)(rj + r) * ld + j0 + c2]);
    Bs[r * LDT + c2] = vb.x;
    Bs[r * LDT + c2 + 1] = vb.y;
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  const int wr = wave >> 1, wc = wave & 1;
  const int lr = lane & 15, lk = lane >> 4;
  d4 acc00 = {0, 0, 0, 0}, acc01 = {0, 0, 0, 0}, acc10 = {0, 0, 0, 0}, acc11 = {0, 0, 0, 0};
  const double* ap0 = &As[(32 * wr + lr) * LDT + lk];
  const double* ap1 = ap0 + 16 * LDT;
  const double* bp0 = &Bs[(32 * wc + lr) * LDT + lk];
  const double* bp1 = bp0 + 16 * LDT;
#pragma unroll
  for (int k0 = 0; k0 < NB; k0 += 4) {
    const double a0 = ap0[k0], a1 = ap1[k0], b0 = bp0[k0], b1 = bp1[k0];
    acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc00, 0, 0, 0);
    acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc01, 0, 0, 0);
    acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc10, 0, 0, 0);
    acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc11, 0, 0, 0);
  }
  // f64 C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg
  const int crow = ri + 32 * wr + lk, ccol = rj + 32 * wc + lr;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    double* c0 = &M[(size_t)(crow + 4 * i) * ld + ccol];
    double* c1 = &M[(size_t)(crow + 16 + 4 * i) * ld + ccol];
    c0[0] -= acc00[i];
    c0[16] -= acc01[i];
    c1[0] -= acc10[i];
    c1[16] -= acc11[i];
  }
}

// ---------------------------------------------------------------------------------------
// Inverse of every 64x64 diagonal block of the factor (one workgroup per block), used by the
// back substitution.  Padding columns (>= n) are treated as identity.  Linv[b] row-major.
// Thread t < 64 solves L x = e_t.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_trinv64(const double* __restrict__ M, int ld, int n, double* __restrict__ Linv) {
  __shared__ double L[NB][NB + 1];
  const int b = blockIdx.x, j0 = b * NB, tid = threadIdx.x;
  for (int e = tid; e < NB * NB; e += 64) {
    const int r = e >> 6, c = e & 63;
    double v = 0.0;
    if (j0 + r < n && j0 + c < n) v = (c <= r) ? M[(size_t)(j0 + r) * ld + j0 + c] : 0.0;
    else if (r == c) v = 1.0;
    L[r][c] = v;
  }
  __syncthreads();
  double x[NB];
#pragma unroll
  for (int r = 0; r < NB; r++) x[r] = (r == tid) ? 1.0 : 0.0;
#pragma unroll
  for (int c = 0; c < NB; c++) {
    const double xc = x[c] / L[c][c];
    x[c] = xc;
#pragma unroll
    for (int q = c + 1; q < NB; q++) x[q] -= xc * L[q][c];
  }
  double* out = Linv + (size_t)b * NB * NB;
#pragma unroll
  for (int r = 0; r < NB; r++) out[r * NB + tid] = x[r];  // column tid of the inverse
}

// ---------------------------------------------------------------------------------------
// Back substitution step for block jb (from the last block down):
//   z_j = Linv_j^T w_j ;  w_i -= L_ji^T z_j for every block i < j.
// Grid = jb workgroups: workgroup i < jb updates w_i.  Every workgroup recomputes z_j itself
// (64x64 mat-vec) so there is no in-launch dependency; k_backsolve_final stores z_j.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_backsolve_step(const double* __restrict__ M, int ld, int n, int jb,
                                                       const double* __restrict__ Linv, double* __restrict__ w) {
  __shared__ double zj[NB];
  const int tid = threadIdx.x, j0 = jb * NB;
  const double* Li = Linv + (size_t)jb * NB * NB;
  double s = 0.0;
  for (int k = 0; k < NB; k++) s += Li[k * NB + tid] * w[j0 + k];  // (Linv^T w)_tid, coalesced over tid
  zj[tid] = s;
  __syncthreads();
  const int i = blockIdx.x;  // grid = jb workgroups, i < jb: nobody writes w_j in this launch
  // w_i[tid] -= sum_k L[j0+k][i*64+tid] * z_j[k]
  const int i0 = i * NB;
  double acc = 0.0;
  for (int k = 0; k < NB; k++) {
    const int r = j0 + k;
    if (r < n) acc += M[(size_t)r * ld + i0 + tid] * zj[k];
  }
  w[i0 + tid] -= acc;
}

// Separate tiny kernel that finalises z_j (avoids the read/write race on w_j inside one launch).
__global__ __launch_bounds__(64) void k_backsolve_final(int jb, const double* __restrict__ Linv,
                                                        const double* __restrict__ w, double* __restrict__ z) {
  const int tid = threadIdx.x, j0 = jb * NB;
  const double* Li = Linv + (size_t)jb * NB * NB;
  double s = 0.0;
  for (int k = 0; k < NB; k++) s += Li[k * NB + tid] * w[j0 + k];
  z[j0 + tid] = s;
}

__global__ void k_copy_row(const double* __restrict__ M, int ld, int row, int n, double* __restrict__ w, int npad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npad) w[i] = (i < n) ? M[(size_t)row * ld + i] : 0.0;
}

// Host driver.  M: npad x npad, row n = rhs.  On return z[0..n) solves S z = rhs.
// `fail` (device int) is OR-ed with 1 when S is not positive definite.
int msfm_chol_factor_solve(msfm_ctx* ctx, double* M, int npad, int n, double* Linv, double* w, double* z,
                           int* fail) {
  hipStream_t s = ctx->stream;
  const int nblk = npad / NB;
  const int nrows = n + 1;  // rows that carry data (S plus the rhs row)
  for (int jb = 0; jb < nblk; jb++) {
    const int j0 = jb * NB;
    if (j0 >= n) break;
    {
      KTimer t(ctx, "chol_potrf64");
      hipLaunchKernelGGL(k_potrf64, dim3(1), dim3(256), 0, s, M, npad, j0, n, fail);
    }
    const int rows_below = nrows - (j0 + NB);
    if (rows_below <= 0) continue;
    {
      KTimer t(ctx, "chol_trsm64");
      hipLaunchKernelGGL(k_trsm64, dim3(cdiv(rows_below, 64)), dim3(256), 0, s, M, npad, j0, n, nrows);
    }
    const int nt = cdiv(rows_below, 64);
    {
      KTimer t(ctx, "chol_syrk64_mfma");
      hipLaunchKernelGGL(k_syrk64, dim3(nt * (nt + 1) / 2), dim3(256), 0, s, M, npad, j0, nt);
    }
  }
  {
    KTimer t(ctx, "chol_trinv64");
    hipLaunchKernelGGL(k_trinv64, dim3(cdiv(n, NB)), dim3(64), 0, s, M, npad, n, Linv);
  }
  {
    KTimer t(ctx, "chol_backsolve");
    hipLaunchKernelGGL(k_copy_row, dim3(cdiv(npad, 256)), dim3(256), 0, s, M, npad, n, n, w, npad);
    for (int jb = cdiv(n, NB) - 1; jb >= 0; jb--) {
      hipLaunchKernelGGL(k_backsolve_final, dim3(1), dim3(64), 0, s, jb, Linv, w, z);
      if (jb > 0) hipLaunchKernelGGL(k_backsolve_step, dim3(jb), dim3(64), 0, s, M, npad, n, jb, Linv, w);
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return msfm_set_error(ctx, MSFM_E_DEVICE, "cholesky launch: %s", hipGetErrorString(e));
  return MSFM_OK;
}
