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
// potrf of the 64x64 diagonal block at (j0,j0), plus the inverses of its four 16x16 diagonal
// sub-blocks (Dinv[blk][4][16][16], row-major) that trsm and the back substitution use.
// Columns with global index >= n are left alone (padding / rhs row).  fail[0] |= 1 when a pivot
// is not positive (Eigen LLT: info() != Success -> Ceres LINEAR_SOLVER_FAILURE).
//
// Blocked by 16 columns.  The 64x16 panel is factored by wave 0 alone, one row per lane, the 16
// panel entries in registers: pivots and multipliers travel by v_readlane (no LDS round trip, no
// barrier inside the panel).  The trailing update (K = 16) is 16x16 f64 MFMA tiles on waves 0-2
// while wave 3 inverts the 16x16 diagonal block by substitution.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

template <bool FULL>
__device__ __forceinline__ void potrf64_lds(double* __restrict__ a /*[64][LDT]*/, double* __restrict__ dinv /*[4][16][17]*/,
                                            double* __restrict__ rdiag /*[64]*/, int ncol, int tid, int* fail) {
  const int wave = tid >> 6, lane = tid & 63;
  for (int jb = 0; jb < 4; jb++) {
    const int c0 = 16 * jb;
    if (!FULL && c0 >= ncol) break;
    if (wave == 0) {
      double p[16];
#pragma unroll
      for (int k = 0; k < 16; k++) p[k] = a[lane * LDT + c0 + k];
      bool bad = false;
#pragma unroll
      for (int c = 0; c < 16; c++) {
        if (FULL || c0 + c < ncol) {  // uniform
          const int piv = c0 + c;
          const double d = readlane_f64(p[c], piv);
          bad |= !(d > 0.0);
          const double rs = rsqrt(d);
          p[c] = lane > piv ? p[c] * rs : (lane == piv ? d * rs : 0.0);
          if (lane == piv) rdiag[piv] = rs;  // 1 / l_cc
#pragma unroll
          for (int q = c + 1; q < 16; q++) {
            const double s = readlane_f64(p[c], c0 + q);
            p[q] -= p[c] * s;
          }
        }
      }
      if (bad && lane == 0) atomicOr(fail, 1);
#pragma unroll
      for (int k = 0; k < 16; k++) a[lane * LDT + c0 + k] = p[k];
    }
    __syncthreads();
    const int tr = 3 - jb;  // 16-row tiles below the panel's diagonal block
    if (wave < 3) {
      // tiles (I >= J) of the trailing block, dealt round-robin to waves 0..2
      const int li = lane & 15, lq = lane >> 4;
      int t = 0;
      for (int I = 0; I < tr; I++)
        for (int J = 0; J <= I; J++, t++) {
          if (t % 3 != wave) continue;
          const int R0 = c0 + 16 + 16 * I, C0 = c0 + 16 + 16 * J;
          d4 acc = {0, 0, 0, 0};
#pragma unroll
          for (int s4 = 0; s4 < 4; s4++) {
            const int k = c0 + 4 * s4 + lq;
            double av = a[(R0 + li) * LDT + k], bv = a[(C0 + li) * LDT + k];
            if (!FULL && k >= ncol) { av = 0.0; bv = 0.0; }
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
          }
#pragma unroll
          for (int i = 0; i < 4; i++) a[(R0 + lq + 4 * i) * LDT + C0 + li] -= acc[i];
        }
    } else if (lane < 16) {
      // inverse of the 16x16 diagonal block: lane = column of the inverse, forward substitution
      double x[16];
#pragma unroll
      for (int r = 0; r < 16; r++) x[r] = (r == lane) ? 1.0 : 0.0;
#pragma unroll
      for (int c = 0; c < 16; c++) {
        const bool real = FULL || c0 + c < ncol;
        const double xc = real ? x[c] * rdiag[c0 + c] : x[c];
        x[c] = xc;
#pragma unroll
        for (int q = c + 1; q < 16; q++) x[q] -= (real && (FULL || c0 + q < ncol)) ? xc * a[(c0 + q) * LDT + c0 + c] : 0.0;
      }
#pragma unroll
      for (int r = 0; r < 16; r++) dinv[(jb * 16 + r) * 17 + lane] = x[r];
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_potrf64(double* __restrict__ M, int ld, int j0, int n, double* __restrict__ Dinv,
                                                  double* __restrict__ Ldiag, int* fail) {
  __shared__ double a[NB * LDT];
  __shared__ double dinv[4 * 16 * 17];
  __shared__ double rdiag[NB];
  const int tid = threadIdx.x;
  // 64 x 64 block, 16-byte loads; the strict upper triangle is cleared
  for (int e = tid; e < NB * 32; e += 256) {
    const int r = e >> 5, c2 = (e & 31) * 2;
    d2 v = *reinterpret_cast<const d2*>(&M[(size_t)(j0 + r) * ld + j0 + c2]);
    a[r * LDT + c2] = (c2 <= r) ? v.x : 0.0;
    a[r * LDT + c2 + 1] = (c2 + 1 <= r) ? v.y : 0.0;
  }
  for (int e = tid; e < 4 * 16 * 17; e += 256) dinv[e] = ((e % 17) == ((e / 17) & 15)) ? 1.0 : 0.0;  // identity for padding
  __syncthreads();
  const int ncol = min(NB, n - j0);
  if (ncol == NB) potrf64_lds<true>(a, dinv, rdiag, ncol, tid, fail);
  else potrf64_lds<false>(a, dinv, rdiag, ncol, tid, fail);
  for (int e = tid; e < NB * 32; e += 256) {
    const int r = e >> 5, c2 = (e & 31) * 2;
    if (c2 + 1 <= r) {
      d2 v = {a[r * LDT + c2], a[r * LDT + c2 + 1]};
      *reinterpret_cast<d2*>(&M[(size_t)(j0 + r) * ld + j0 + c2]) = v;
    } else if (c2 <= r) {
      M[(size_t)(j0 + r) * ld + j0 + c2] = a[r * LDT + c2];
    }
  }
  double* out = Dinv + (size_t)(j0 / NB) * 1024;
  for (int e = tid; e < 1024; e += 256) out[e] = dinv[(e >> 4) * 17 + (e & 15)];
  double* lo = Ldiag + (size_t)(j0 / NB) * NB * NB;
  for (int e = tid; e < NB * NB; e += 256) lo[e] = ((e & 63) <= (e >> 6)) ? a[(e >> 6) * LDT + (e & 63)] : 0.0;
}

// ---------------------------------------------------------------------------------------
// trsm: rows r in [j0+64, ..): X L11^T = A21.  One workgroup = 64 rows, each wave owns 16 rows and
// runs the 16-column-blocked forward substitution on its own with f64 MFMA 16x16 tiles held
// TRANSPOSED in the accumulator layout (columns of L on the accumulator rows), so that every
// product has the LDS matrix on the left and the register tile on the right:
//   T_jb^T = A_jb^T - sum_{i<jb} L_{jb,i} X_i^T ;  X_jb^T = Dinv_jb T_jb^T
// (an accumulator tile is directly the B operand of the next MFMA when the k index of slice s is
// taken as (lane >> 4) + 4 s).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_trsm64(double* __restrict__ M, int ld, int j0, const double* __restrict__ Dinv) {
  __shared__ double L[NB * LDT];
  __shared__ double dv[4 * 16 * 18];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int lj = lane & 15, lq = lane >> 4;
  const size_t row = (size_t)(j0 + NB + blockIdx.x * 64 + 16 * wave + lj);
  double* mrow = M + row * ld + j0;
  // this lane's 16 values of the 16 x 64 strip first: their latency hides behind the staging of
  // L11, and the stores at the end must not be allowed to fence them
  d4 Tin[4];
#pragma unroll
  for (int jb = 0; jb < 4; jb++)
#pragma unroll
    for (int i = 0; i < 4; i++) Tin[jb][i] = mrow[16 * jb + lq + 4 * i];
  for (int e = tid; e < NB * 32; e += 256) {
    const int r = e >> 5, c2 = (e & 31) * 2;
    const d2 v = *reinterpret_cast<const d2*>(&M[(size_t)(j0 + r) * ld + j0 + c2]);
    L[r * LDT + c2] = v.x;
    L[r * LDT + c2 + 1] = v.y;
  }
  const double* Di = Dinv + (size_t)(j0 / NB) * 1024;
  for (int e = tid; e < 1024; e += 256) dv[(e >> 4) * 18 + (e & 15)] = Di[e];
  __syncthreads();
  d4 X[4];
#pragma unroll
  for (int jb = 0; jb < 4; jb++) {
    d4 T = Tin[jb];
#pragma unroll
    for (int i2 = 0; i2 < jb; i2++) {
#pragma unroll
      for (int s4 = 0; s4 < 4; s4++) {
        const double av = -L[(16 * jb + lj) * LDT + 16 * i2 + lq + 4 * s4];
        T = __builtin_amdgcn_mfma_f64_16x16x4f64(av, X[i2][s4], T, 0, 0, 0);
      }
    }
    d4 Y = {0, 0, 0, 0};
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) {
      const double av = dv[(16 * jb + lj) * 18 + lq + 4 * s4];
      Y = __builtin_amdgcn_mfma_f64_16x16x4f64(av, T[s4], Y, 0, 0, 0);
    }
    X[jb] = Y;
  }
#pragma unroll
  for (int jb = 0; jb < 4; jb++)
#pragma unroll
    for (int i = 0; i < 4; i++) mrow[16 * jb + lq + 4 * i] = X[jb][i];
}

// ---------------------------------------------------------------------------------------
// One launch per panel j:  trailing update  C_IJ -= P_I P_J^T  (64x64 tiles, K = 64, f64 MFMA),
// with the next panel's factorisation and triangular solve folded in and NO cross-workgroup
// dependency: every workgroup that owns a column-0 tile (I,0) also forms the updated diagonal tile
// (0,0) for itself (P_0 is already in LDS as its B operand), factors it redundantly (potrf64_lds —
// bit-identical in every workgroup) and then solves its own tile against it.  Workgroup (0,0)
// publishes the factor (Ldiag, Dinv).  The diagonal tiles of M itself are never overwritten with L,
// so late-starting column-0 workgroups always read the pre-factorisation values.
// blockIdx.x < nt : tile (blockIdx.x, 0);  the rest enumerate (I, J >= 1) pairs.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void load_tile_pair(const double* __restrict__ M, int ld, int ri, int rj, int j0, double* As, double* Bs, int tid) {
  for (int e = tid; e < 64 * 32; e += 256) {
    const int r = e >> 5, c2 = (e & 31) * 2;
    const d2 va = *reinterpret_cast<const d2*>(&M[(size_t)(ri + r) * ld + j0 + c2]);
    As[r * LDT + c2] = va.x;
    As[r * LDT + c2 + 1] = va.y;
    const d2 vb = *reinterpret_cast<const d2*>(&M[(size_t)(rj + r) * ld + j0 + c2]);
    Bs[r * LDT + c2] = vb.x;
    Bs[r * LDT + c2 + 1] = vb.y;
  }
}

// acc (2x2 MFMA tiles of this wave's 32x32 quadrant) = X_rows(32 wr..) * Y_rows(32 wc..)^T over K = 64
__device__ __forceinline__ void quad_abt(const double* X, const double* Y, int wr, int wc, int lr, int lk, d4& a00, d4& a01, d4& a10, d4& a11) {
  const double* ap0 = &X[(32 * wr + lr) * LDT + lk];
  const double* ap1 = ap0 + 16 * LDT;
  const double* bp0 = &Y[(32 * wc + lr) * LDT + lk];
  const double* bp1 = bp0 + 16 * LDT;
#pragma unroll
  for (int k0 = 0; k0 < NB; k0 += 4) {
    const double a0 = ap0[k0], a1 = ap1[k0], b0 = bp0[k0], b1 = bp1[k0];
    a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, a00, 0, 0, 0);
    a01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, a01, 0, 0, 0);
    a10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, a10, 0, 0, 0);
    a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, a11, 0, 0, 0);
  }
}

__global__ __launch_bounds__(256) void k_panel64(double* __restrict__ M, int ld, int j0, int nt, int n, double* __restrict__ Dinv,
                                                  double* __restrict__ Ldiag, int* fail) {
  __shared__ double As[64 * LDT];
  __shared__ double Bs[64 * LDT];
  __shared__ double Cs[64 * LDT];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wr = wave >> 1, wc = wave & 1;
  const int lr = lane & 15, lk = lane >> 4;
  const int t0 = j0 + NB;  // first trailing row
  int I, J;
  if ((int)blockIdx.x < nt) {
    I = blockIdx.x; J = 0;
  } else {
    // pairs (I, J) with 1 <= J <= I < nt, enumerated row by row: index q = I(I-1)/2 + (J-1)
    const int q = blockIdx.x - nt;
    I = (int)((sqrt(8.0 * q + 1.0) + 1.0) * 0.5);
    while (I * (I - 1) / 2 > q) I--;
    while ((I + 1) * I / 2 <= q) I++;
    J = q - I * (I - 1) / 2 + 1;
  }
  const int ri = t0 + I * 64, rj = t0 + J * 64;
  load_tile_pair(M, ld, ri, rj, j0, As, Bs, tid);
  __syncthreads();
  d4 acc00 = {0, 0, 0, 0}, acc01 = {0, 0, 0, 0}, acc10 = {0, 0, 0, 0}, acc11 = {0, 0, 0, 0};
  quad_abt(As, Bs, wr, wc, lr, lk, acc00, acc01, acc10, acc11);
  // f64 C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg
  const int qrow = 32 * wr + lk, qcol = 32 * wc + lr;  // quadrant-local origin of this lane's values
  const int ncol_next = min(NB, n - t0);
  if (J != 0 || ncol_next <= 0) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      double* c0 = &M[(size_t)(ri + qrow + 4 * i) * ld + rj + qcol];
      double* c1 = &M[(size_t)(ri + qrow + 16 + 4 * i) * ld + rj + qcol];
      c0[0] -= acc00[i];
      c0[16] -= acc01[i];
      c1[0] -= acc10[i];
      c1[16] -= acc11[i];
    }
    return;
  }
  // ---- column-0 workgroup: updated diagonal tile -> As, own updated tile -> Cs ----
  d4 d00 = acc00, d01 = acc01, d10 = acc10, d11 = acc11;
  if (I != 0) {
    d00 = d4{0, 0, 0, 0}; d01 = d00; d10 = d00; d11 = d00;
    quad_abt(Bs, Bs, wr, wc, lr, lk, d00, d01, d10, d11);  // P_0 P_0^T
  }
  __syncthreads();  // every wave is done reading As / Bs
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int r0 = qrow + 4 * i, r1 = r0 + 16;
    As[r0 * LDT + qcol] = M[(size_t)(t0 + r0) * ld + t0 + qcol] - d00[i];
    As[r0 * LDT + qcol + 16] = M[(size_t)(t0 + r0) * ld + t0 + qcol + 16] - d01[i];
    As[r1 * LDT + qcol] = M[(size_t)(t0 + r1) * ld + t0 + qcol] - d10[i];
    As[r1 * LDT + qcol + 16] = M[(size_t)(t0 + r1) * ld + t0 + qcol + 16] - d11[i];
    if (I != 0) {
      Cs[r0 * LDT + qcol] = M[(size_t)(ri + r0) * ld + t0 + qcol] - acc00[i];
      Cs[r0 * LDT + qcol + 16] = M[(size_t)(ri + r0) * ld + t0 + qcol + 16] - acc01[i];
      Cs[r1 * LDT + qcol] = M[(size_t)(ri + r1) * ld + t0 + qcol] - acc10[i];
      Cs[r1 * LDT + qcol + 16] = M[(size_t)(ri + r1) * ld + t0 + qcol + 16] - acc11[i];
    }
  }
  double* dinv = Bs;  // 4*16*17 doubles
  double* rdiag = Bs + 4 * 16 * 17;
  for (int e = tid; e < 4 * 16 * 17; e += 256) dinv[e] = ((e % 17) == ((e / 17) & 15)) ? 1.0 : 0.0;
  __syncthreads();
  if (ncol_next == NB) potrf64_lds<true>(As, dinv, rdiag, ncol_next, tid, fail);
  else potrf64_lds<false>(As, dinv, rdiag, ncol_next, tid, fail);
  if (I == 0) {
    double* lo = Ldiag + (size_t)(t0 / NB) * NB * NB;
    for (int e = tid; e < NB * NB; e += 256) lo[e] = ((e & 63) <= (e >> 6)) ? As[(e >> 6) * LDT + (e & 63)] : 0.0;
    double* out = Dinv + (size_t)(t0 / NB) * 1024;
    for (int e = tid; e < 1024; e += 256) out[e] = dinv[(e >> 4) * 17 + (e & 15)];
    if (ncol_next < NB) {
      // last, partial block: it also holds the rhs row (row n), whose entries are the tail of
      // w = L^-1 rhs that the back substitution reads from M.  This workgroup is the only one of the
      // launch in that case (nt == 1), so writing the tile back cannot race with a reader.
      for (int e = tid; e < NB * NB; e += 256) {
        const int r = e >> 6, c = e & 63;
        if (c <= r) M[(size_t)(t0 + r) * ld + t0 + c] = As[r * LDT + c];
      }
    }
    return;
  }
  // ---- X = C_I0 L11^-T for this workgroup's 64 rows: wave-local blocked substitution (see k_trsm64) ----
  const int lj = lane & 15, lq = lane >> 4;
  const double* crow = &Cs[(16 * wave + lj) * LDT];
  double* mrow = M + (size_t)(ri + 16 * wave + lj) * ld + t0;
  d4 X[4];
#pragma unroll
  for (int jb = 0; jb < 4; jb++) {
    d4 T;
#pragma unroll
    for (int i = 0; i < 4; i++) T[i] = crow[16 * jb + lq + 4 * i];
#pragma unroll
    for (int i2 = 0; i2 < jb; i2++) {
#pragma unroll
      for (int s4 = 0; s4 < 4; s4++) {
        const double av = -As[(16 * jb + lj) * LDT + 16 * i2 + lq + 4 * s4];
        T = __builtin_amdgcn_mfma_f64_16x16x4f64(av, X[i2][s4], T, 0, 0, 0);
      }
    }
    d4 Y = {0, 0, 0, 0};
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) {
      const double av = dinv[(16 * jb + lj) * 17 + lq + 4 * s4];
      Y = __builtin_amdgcn_mfma_f64_16x16x4f64(av, T[s4], Y, 0, 0, 0);
    }
    X[jb] = Y;
  }
#pragma unroll
  for (int jb = 0; jb < 4; jb++)
#pragma unroll
    for (int i = 0; i < 4; i++) mrow[16 * jb + lq + 4 * i] = X[jb][i];
}

// ---------------------------------------------------------------------------------------
// Full inverses of all 64x64 diagonal blocks of the factor (one workgroup per block, after the
// factorisation, off the critical path): from the 16x16 inverses by two doubling steps
//   inv([A 0; B C]) = [A^-1 0; -C^-1 B A^-1  C^-1].   Linv[blk] row-major 64x64.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_trinv64_full(const double* __restrict__ Ldiag, int n, const double* __restrict__ Dinv,
                                                       double* __restrict__ Linv) {
  __shared__ double L[NB][NB + 1];
  __shared__ double v[NB][NB + 1];
  __shared__ double t[NB][NB + 1];
  const int blk = blockIdx.x, j0 = blk * NB, tid = threadIdx.x;
  const int ncol = min(NB, n - j0);
  for (int e = tid; e < NB * NB; e += 256) {
    const int r = e >> 6, c = e & 63;
    L[r][c] = (c <= r && r < ncol) ? Ldiag[(size_t)blk * NB * NB + r * NB + c] : 0.0;
    const bool diag16 = (r >> 4) == (c >> 4);
    v[r][c] = diag16 ? Dinv[(size_t)blk * 1024 + ((r >> 4) * 16 + (r & 15)) * 16 + (c & 15)] : 0.0;
  }
  __syncthreads();
  for (int s = 16; s < 64; s *= 2) {
    const int npair = 64 / (2 * s), ss = s * s;
    for (int e = tid; e < npair * ss; e += 256) {
      const int pr = e / ss, rem = e - pr * ss, r = rem / s, c = rem - r * s;
      const int o = pr * 2 * s;
      double acc = 0.0;
      for (int k = c; k < s; k++) acc += L[o + s + r][o + k] * v[o + k][o + c];
      t[o + s + r][o + c] = acc;
    }
    __syncthreads();
    for (int e = tid; e < npair * ss; e += 256) {
      const int pr = e / ss, rem = e - pr * ss, r = rem / s, c = rem - r * s;
      const int o = pr * 2 * s;
      double acc = 0.0;
      for (int k = 0; k <= r; k++) acc += v[o + s + r][o + s + k] * t[o + s + k][o + c];
      v[o + s + r][o + c] = -acc;
    }
    __syncthreads();
  }
  double* out = Linv + (size_t)blk * NB * NB;
  for (int e = tid; e < NB * NB; e += 256) out[e] = v[e >> 6][e & 63];
}

// ---------------------------------------------------------------------------------------
// Back substitution step for block jb (from the last block down):
//   z_j = Linv_j^T w_j ;  w_i -= L_ji^T z_j for every block i < j.
// Grid = max(jb, 1) workgroups of 256 threads; every workgroup recomputes z_j itself (64x64
// mat-vec against the explicit inverse) so there is no in-launch dependency; workgroup 0 stores
// z_j; workgroup i < jb updates w_i.  Nobody writes w_j in this launch.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_backsolve_step(const double* __restrict__ M, int ld, int n, int jb,
                                                         const double* __restrict__ Linv, double* __restrict__ w,
                                                         double* __restrict__ z) {
  __shared__ double part[4][NB];
  __shared__ double zj[NB];
  const int tid = threadIdx.x, j0 = jb * NB;
  const int col = tid & 63, kq = tid >> 6;
  const double* Li = Linv + (size_t)jb * NB * NB;
  // issue the loads of both phases up front: they do not depend on each other
  double lv[16], mv[16];
  const int i0 = blockIdx.x * NB;
#pragma unroll
  for (int k = 0; k < 16; k++) lv[k] = Li[(16 * kq + k) * NB + col];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int r = j0 + 16 * kq + k;
    mv[k] = (jb > 0 && r < n) ? M[(size_t)r * ld + i0 + col] : 0.0;
  }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 16; k++) s += lv[k] * w[j0 + 16 * kq + k];
  part[kq][col] = s;
  __syncthreads();
  if (tid < NB) {
    const double zz = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    zj[tid] = zz;
    if (blockIdx.x == 0) z[j0 + tid] = zz;
  }
  __syncthreads();
  if (jb == 0) return;
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < 16; k++) acc += mv[k] * zj[16 * kq + k];
  __syncthreads();
  part[kq][col] = acc;
  __syncthreads();
  if (tid < NB) w[i0 + tid] -= (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
}

__global__ void k_copy_row(const double* __restrict__ M, int ld, int row, int n, double* __restrict__ w, int npad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npad) w[i] = (i < n) ? M[(size_t)row * ld + i] : 0.0;
}

// Host driver.  M: npad x npad, row n = rhs.  On return z[0..n) solves S z = rhs.
// `fail` (device int) is OR-ed with 1 when S is not positive definite.
// work: npad*16 doubles (16x16 inverses) + npad*64 (full block inverses) + npad*64 (diagonal blocks of L).
int msfm_chol_factor_solve(msfm_ctx* ctx, double* M, int npad, int n, double* work, double* w, double* z,
                           int* fail) {
  if (!M || !work || !w || !z || !fail || npad % NB != 0 || n < 1 || n + 1 > npad)
    return msfm_set_error(ctx, MSFM_E_INVAL, "cholesky: bad workspace (null buffer or size)");
  hipStream_t s = ctx->stream;
  const int nblk = npad / NB;
  const int nrows = n + 1;  // rows that carry data (S plus the rhs row)
  double* Dinv = work;
  double* Linv = work + (size_t)npad * 16;
  double* Ldiag = work + (size_t)npad * 80;
  {
    KTimer t(ctx, "chol_potrf64");
    hipLaunchKernelGGL(k_potrf64, dim3(1), dim3(256), 0, s, M, npad, 0, n, Dinv, Ldiag, fail);
  }
  if (nrows > NB) {
    KTimer t(ctx, "chol_trsm_mfma");
    hipLaunchKernelGGL(k_trsm64, dim3(cdiv(nrows - NB, 64)), dim3(256), 0, s, M, npad, 0, Dinv);
  }
  for (int jb = 0; jb < nblk; jb++) {
    const int j0 = jb * NB;
    if (j0 >= n) break;
    const int rows_below = nrows - (j0 + NB);
    if (rows_below <= 0) continue;
    const int nt = cdiv(rows_below, 64);
    KTimer t(ctx, "chol_panel_mfma");  // trailing update + next panel's potrf + trsm, one launch
    hipLaunchKernelGGL(k_panel64, dim3(nt * (nt + 1) / 2), dim3(256), 0, s, M, npad, j0, nt, n, Dinv, Ldiag, fail);
  }
  {
    KTimer t(ctx, "chol_backsolve");
    hipLaunchKernelGGL(k_trinv64_full, dim3(cdiv(n, NB)), dim3(256), 0, s, Ldiag, n, Dinv, Linv);
    hipLaunchKernelGGL(k_copy_row, dim3(cdiv(npad, 256)), dim3(256), 0, s, M, npad, n, n, w, npad);
    for (int jb = cdiv(n, NB) - 1; jb >= 0; jb--)
      hipLaunchKernelGGL(k_backsolve_step, dim3(jb > 0 ? jb : 1), dim3(256), 0, s, M, npad, n, jb, Linv, w, z);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return msfm_set_error(ctx, MSFM_E_DEVICE, "cholesky launch: %s", hipGetErrorString(e));
  return MSFM_OK;
}
