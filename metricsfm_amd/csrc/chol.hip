// Dense FP64 Cholesky of the reduced camera system + the two triangular solves, for gfx950.
//
// Replaces `lhs.selfadjointView<Upper>().llt()` + `solve` inside Ceres' DENSE_SCHUR solver,
// reached from ceres::Solve at SfM/src/optimizer.cc:133 (options :47) and slam_gps.cc:841.
//
// Layout: M is npad x npad row-major (ld = npad, npad a multiple of 64), lower triangle.
// Rows/cols [0,n) hold S; row n holds rhs^T (the forward substitution L w = rhs then falls out
// of the factorisation: row n of the factor is w^T); indices > n are zero padding.
// Right-looking by 64-column panels, ONE launch per panel (k_panel_v2): the trailing update with
// panel j0 as 64x64 f64-MFMA tiles, with the factorisation (potrf + trsm) of the next panel folded
// into the same launch.  The back substitution L^T z = w runs block by block with explicitly
// inverted diagonal blocks (k_trinv64_full) so each step is a mat-vec.
#include "common.h"
#include <atomic>
#include <algorithm>
#include <cstdlib>
#include <memory>

#define NB 64
#define MSFM_Z_PENDING 0xFFF85A5A5A5A5A5Aull   // "not solved yet" in the solution vector of k_backsolve_chain
#define LDT 66  // LDS row stride in doubles: 132 dwords = 4 mod 64 -> conflict-free ds_read_b64 fragments

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// cycle-counter probes for scripts/chol_probe.hip (compiled out of the library)
#ifndef MSFM_PROBE
#define MSFM_PROBE(i)
#define MSFM_PROBE_ARM(j0)
#endif

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
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

// =======================================================================================
// Panel kernel: right-looking, one launch per 64-column panel, the next panel's potrf + trsm folded
// into the launch with no cross-workgroup dependency (every column-0 workgroup factors the updated
// diagonal block redundantly, bit-identically), organised around the only thing that bounds the
// launch: the dependent chain of a column-0 workgroup.  Measured on MI355X (scripts/chol_probe.hip)
// that chain is ~36 k cycles; a first version with one 64-row tile per workgroup, right-looking
// potrf16 and the triangular solve after the last pivot took ~65 k.
//
//  * A column-0 workgroup owns THREE 16-row tiles of the new panel (waves 1..3, one tile each);
//    wave 0 owns no rows and only runs the pivot chain.  All global loads are issued up front.
//  * The own rows live in registers as transposed accumulator tiles from the first load to the
//    final store (P_R fragments loaded straight from global in MFMA operand layout) - no third LDS
//    tile, so two workgroups fit on a CU (2 x 76.3 KB).
//  * Only the lower 16x16 tiles of the updated diagonal block are formed; column 0 first (all four
//    waves), the rest while wave 0 already factors.
//  * potrf16 is left-looking on UNSCALED columns (S = U D^-1 U^T): the multipliers of all but the
//    newest column come back from LDS as uniform-address (broadcast) reads issued one column ahead;
//    only the newest column's multiplier and the pivot travel by v_readlane.  The reciprocal (rcp +
//    one cubic correction) is the only transcendental on the chain, rsqrt is off it.  The inverse
//    of the 16x16 diagonal block falls out of the same column operations applied to identity rows
//    (spare lanes; a second register set for the first sub-panel, which has no spare lanes).
//  * The triangular solve of the own rows is pipelined behind the sub-panels: when sub-panel jb
//    is being factored the helpers already compute X_(jb-1) and subtract it from the later blocks,
//    so only X_3 = T_3 Dinv_3^T (4 MFMAs) and the store follow the last pivot.
// j0 = -64 means "no previous panel" (first block: t0 = 0, nothing to subtract).
// =======================================================================================
#define DV 17  // row stride of the 16x16 inverse blocks in LDS

__device__ __forceinline__ double rcp3(double d) {  // v_rcp_f64 is good to 2^-24; one cubic step -> < 2^-60
  const double x = __builtin_amdgcn_rcp(d);
  const double e = fma(-d, x, 1.0);
  const double t = fma(e, e, e);
  return fma(x, t, x);
}
__device__ __forceinline__ double rsq3(double d) {
  const double y = __builtin_amdgcn_rsq(d);
  const double e = fma(-d * y, y, 1.0);
  const double h = fma(e, 0.375, 0.5);
  return fma(y * e, h, y);
}

// Factor sub-panel JB (columns 16 JB .. 16 JB + 15, rows >= 16 JB) of the 64 x 64 block in Ls, in
// place, and write the inverse of its 16 x 16 diagonal block to dinv.  ONE wave; lane = row.
template <int JB, bool FULL>
__device__ __forceinline__ void potrf16_v2(double* Ls, double* dinv, double* dvec /*[64 + 16]*/, int ncol, int lane, int* fail) {
  constexpr int c0 = 16 * JB;
  const bool isrow = lane >= c0;
  double p[16], u[16], keep[16], rs[16];
  {
    const d2* prow = reinterpret_cast<const d2*>(&Ls[lane * LDT + c0]);
#pragma unroll
    for (int k2 = 0; k2 < 8; k2++) {
      const d2 v = prow[k2];
      p[2 * k2] = v.x;
      p[2 * k2 + 1] = v.y;
    }
  }
  if (JB > 0) {
    // lanes 0..15 carry the identity rows that end up as L_jj^-T; lanes 16..c0-1 idle (zero rows)
#pragma unroll
    for (int k = 0; k < 16; k++) p[k] = isrow ? p[k] : ((lane == k) ? 1.0 : 0.0);
  } else {
#pragma unroll
    for (int k = 0; k < 16; k++) u[k] = (lane == k) ? 1.0 : 0.0;
  }
  double inv_prev = 0.0;
  // Software pipeline.  Column q receives the contributions of columns k < q from three places:
  //   k <= q-3 : at step q-1, multipliers m_k[row q] fetched from LDS (uniform address = broadcast)
  //              at the end of step q-3, i.e. a full step before they are used;
  //   k  = q-2 : at step q-1, v_readlane of m_(q-2) (known since step q-2; off the chain);
  //   k  = q-1 : at step q, v_readlane of the unscaled entry times 1/d_(q-1)  (the chain).
  // (All LDS stores of this function are unpredicated - lanes above the sub-panel write into the
  // never-read upper triangle - so that the whole factorisation is one basic block to schedule.)
  double b0[16], b1[16], b2[16];
#pragma unroll
  for (int k = 0; k < 16; k++) b0[k] = b1[k] = b2[k] = 0.0;
  double m_prev = 0.0;
  // One step is hand-interleaved (the wave is alone on its SIMD and issues in order): after each
  // instruction of the dependent chain  v -> d -> rcp -> e1 -> e2 -> inv  come a few independent
  // terms of the next column's running sums.  sched_barrier pins that order.
#define MSFM_SB() __builtin_amdgcn_sched_barrier(0)
#define MSFM_CT(k)                                                            \
  do {                                                                        \
    if (cterms && (k) + 2 <= c) {                                             \
      if ((k) & 1) { s0 = fma(p[k], b0[k], s0); if (JB == 0) w0 = fma(u[k], b0[k], w0); } \
      else { p[c + 1] = fma(-p[k], b0[k], p[c + 1]); if (JB == 0) u[c + 1] = fma(-u[k], b0[k], u[c + 1]); } \
    }                                                                         \
  } while (0)
#pragma unroll
  for (int c = 0; c < 16; c++) {
    const int piv = c0 + c;
    const bool real = FULL || piv < ncol;  // uniform
    const bool cterms = c >= 1 && c < 15;  // p[c+1] -= p[c-1] m_(c-1)[piv+1] + sum_{k <= c-2} p[k] m_k[piv+1]
    // The pivot is formed on uniform values (the same arithmetic as the vector update of p[c], so
    // it is bit-identical to p[c] in lane piv): every readlane is off the chain.
    const double q = readlane_f64(p[c], piv);
    const double t = c >= 1 ? readlane_f64(p[c - 1], piv) : 0.0;
    const double e = cterms ? readlane_f64(m_prev, piv + 1) : 0.0;
    double s0 = 0.0, w0 = 0.0;
    MSFM_SB();
    const double v = t * inv_prev;
    if (cterms) { s0 = p[c - 1] * e; if (JB == 0) w0 = u[c - 1] * e; }
    MSFM_CT(0);
    MSFM_SB();
    double d = c >= 1 ? fma(-t, v, q) : q;
    if (!real) d = 1.0;
    MSFM_CT(1); MSFM_CT(2);
    MSFM_SB();
    const double x = __builtin_amdgcn_rcp(d);
    MSFM_CT(3); MSFM_CT(4);
    if (c >= 1) {
      p[c] = fma(-p[c - 1], v, p[c]);
      if (JB == 0) u[c] = fma(-u[c - 1], v, u[c]);
    }
    MSFM_SB();
    const double e1 = fma(-d, x, 1.0);
    MSFM_CT(5); MSFM_CT(6);
    MSFM_SB();
    const double e2 = fma(e1, e1, e1);
    MSFM_CT(7); MSFM_CT(8); MSFM_CT(9);
    MSFM_SB();
    const double inv = fma(x, e2, x);
    MSFM_CT(10); MSFM_CT(11); MSFM_CT(12);
    MSFM_SB();
    if (cterms) {
      p[c + 1] -= s0;
      if (JB == 0) u[c + 1] -= w0;
    }
    dvec[piv] = d;  // every lane, same value: the scaling 1/sqrt(d) is computed for all 16 columns at once below
    if (!FULL) {
      keep[c] = p[c];
      if (!real) p[c] = 0.0;  // padding / rhs columns are left alone and feed nothing
    }
    m_prev = p[c] * inv;
    Ls[lane * LDT + piv] = m_prev;  // broadcast source for the later columns
    inv_prev = inv;
    if (c < 13) {
      // multipliers m_k[piv + 3], k <= c, used during step c+2
      const double* mrow = &Ls[(piv + 3) * LDT + c0];
#pragma unroll
      for (int k = 0; k + 1 <= c; k += 2) {
        const d2 v2 = *reinterpret_cast<const d2*>(&mrow[k]);
        b2[k] = v2.x;
        b2[k + 1] = v2.y;
      }
      if (!(c & 1)) b2[c] = mrow[c];
    }
    MSFM_SB();  // keep the fetch here, a full step ahead of its use
#pragma unroll
    for (int k = 0; k < 16; k++) { b0[k] = b1[k]; b1[k] = b2[k]; }
  }
#undef MSFM_CT
#undef MSFM_SB
  // 1/sqrt(d) of the 16 pivots in one vector operation, then back as uniform values
  {
    const double dl = dvec[c0 + (lane & 15)];
    if (!(dl > 0.0)) atomicOr(fail, 1);  // Eigen LLT: info() != Success
    dvec[64 + (lane & 15)] = rsq3(dl);
    const d2* rv = reinterpret_cast<const d2*>(&dvec[64]);
#pragma unroll
    for (int k2 = 0; k2 < 8; k2++) {
      const d2 r2 = rv[k2];
      rs[2 * k2] = r2.x;
      rs[2 * k2 + 1] = r2.y;
    }
  }
  // L = U D^-1/2 (the entries above the diagonal of the 16x16 block are never read by anyone)
#pragma unroll
  for (int k2 = 0; k2 < 8; k2++) {
    d2 o;
    double l0 = p[2 * k2] * rs[2 * k2], l1 = p[2 * k2 + 1] * rs[2 * k2 + 1];
    if (!FULL) {
      if (c0 + 2 * k2 >= ncol) l0 = keep[2 * k2];
      if (c0 + 2 * k2 + 1 >= ncol) l1 = keep[2 * k2 + 1];
    }
    o.x = l0; o.y = l1;
    *reinterpret_cast<d2*>(&Ls[lane * LDT + c0 + 2 * k2]) = o;
  }
  if (lane < 16) {
#pragma unroll
    for (int r = 0; r < 16; r++) dinv[(c0 + r) * DV + lane] = ((JB == 0) ? u[r] : p[r]) * rs[r];
  }
}

// A sub-panel of the system's last block that holds padding columns only: nothing to factor (potrf16_v2 would walk sixteen
// pivots d = 1), the block of the inverse is the identity.
template <int JB>
__device__ __forceinline__ void potrf16_skip(double* dinv, int lane) {
  if (lane < 16) {
#pragma unroll
    for (int r = 0; r < 16; r++) dinv[(16 * JB + r) * DV + lane] = lane == r ? 1.0 : 0.0;
  }
}

// acc += sum_{s < NS} A_s B_s with A[m][k] = -X[(ra + m) * LDT + k0 + 4 s + k], B[k][n] = Y[(rb + n) * LDT + k0 + 4 s + k]
template <int NS>
__device__ __forceinline__ d4 mm_nt_neg(const double* X, int ra, const double* Y, int rb, int k0, int lr, int lk, d4 acc) {
  const double* ap = &X[(ra + lr) * LDT + k0 + lk];
  const double* bp = &Y[(rb + lr) * LDT + k0 + lk];
#pragma unroll
  for (int s = 0; s < NS; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-ap[4 * s], bp[4 * s], acc, 0, 0, 0);
  return acc;
}
// 16x16 tile (a, b) of the block in Ls, accumulator layout
__device__ __forceinline__ d4 tile_ld(const double* Ls, int a, int b, int lr, int lk) {
  d4 v;
#pragma unroll
  for (int i = 0; i < 4; i++) v[i] = Ls[(16 * a + lk + 4 * i) * LDT + 16 * b + lr];
  return v;
}
__device__ __forceinline__ void tile_st(double* Ls, int a, int b, int lr, int lk, d4 v) {
#pragma unroll
  for (int i = 0; i < 4; i++) Ls[(16 * a + lk + 4 * i) * LDT + 16 * b + lr] = v[i];
}

// One launch can carry several independent panel steps ("jobs"): with an elimination order that puts
// K mutually uncoupled camera domains first and their separator last, the K domain chains advance in
// the same launch.  A job's rows are two ranges: A = [a0, a0 + 16 na16) (rest of its own domain,
// starting with the 64 x 64 block to factor) and B = [b0, b0 + 16 nb16) (separator + rhs row); the
// B x B tiles, which every job's trailing update would write, are left out of the chains (defer_corner)
// and formed afterwards by one SYRK over all panels (k_corner_syrk, then k_merge_corners).  The plain dense
// factorisation is one job with B empty.
struct PanelJob {
  int j0;          // panel to apply, < 0: none (first block of a chain)
  int t0;          // block to factor, < 0: update only
  int a0, na16;    // range A, na16 a multiple of 4
  int b0, nb16;    // range B: nb16 16-row tiles that carry data, in up to four segments (a node's ancestors, one per level,
  int nseg;        // then the root with the rhs row): segment g starts at row sb0[g] and holds sn16[g] tiles, every segment but
  int sb0[4], sn16[4];   // the last a multiple of four.  nseg == 0: one segment starting at b0
  int nrt;         // 16-row tiles that carry data, over A then B
  int ncw;         // column-0 workgroups
  int wg0, nwg;    // workgroups [wg0, wg0 + nwg) of the launch: ncw column-0 ones, then the bulk ones
  int ntile;       // 64x64 tiles of the trailing update, dealt round-robin to the nwg - ncw bulk workgroups
  int ldc;         // leading dimension of corner
  double* corner;  // destination of B x B tiles, nullptr: M itself
  int defer_corner;  // the B x B tiles are left out here: one SYRK over all panels forms them after the domain chains
};
struct PanelJobs {
  int count;
  PanelJob job[8];
};
__device__ __forceinline__ int job_row16(const PanelJob& jb, int rt) {
  if (rt < jb.na16) return jb.a0 + 16 * rt;
  int r = rt - jb.na16;
  if (jb.nseg == 0) return jb.b0 + 16 * r;
  int g = 0;
  while (g + 1 < jb.nseg && r >= jb.sn16[g]) { r -= jb.sn16[g]; g++; }
  return jb.sb0[g] + 16 * r;
}
__device__ __forceinline__ int job_row64(const PanelJob& jb, int ti) { return job_row16(jb, 4 * ti); }

__device__ __forceinline__ void p0_load(const double* __restrict__ M, int ld, int t0, int j0, int tid, d2 (&pv)[8]) {
#pragma unroll
  for (int it = 0; it < 8; it++) {
    const int e = tid + 256 * it, r = e >> 5, c2 = (e & 31) * 2;
    pv[it] = *reinterpret_cast<const d2*>(&M[(size_t)(t0 + r) * ld + j0 + c2]);
  }
}
__device__ __forceinline__ void p0_store(double* Bs, int tid, const d2 (&pv)[8]) {
#pragma unroll
  for (int it = 0; it < 8; it++) {
    const int e = tid + 256 * it, r = e >> 5, c2 = (e & 31) * 2;
    Bs[r * LDT + c2] = pv[it].x;
    Bs[r * LDT + c2 + 1] = pv[it].y;
  }
}

// Workgroup 0 of a job publishes the factor of the diagonal block and the inverses of its 16x16 diagonal blocks.
// Sub-panel jbp (16 columns) is final once potrf16_v2<jbp> has passed its barrier: the helper waves write it out while
// wave 0 factors the next sub-panel, so that only the last quarter is left for the tail of the launch (the launch ends
// with workgroup 0: 33.7 k cycles against 31.4 k for the other column-0 workgroups before this).
__device__ __forceinline__ void publish_subpanel(const double* Ls, const double* dinv, double* __restrict__ lo, double* __restrict__ out, int jbp,
                                                 int h, int nh) {
  for (int e = h; e < 1024; e += nh) {
    const int r = e >> 4, c = 16 * jbp + (e & 15);
    lo[r * NB + c] = (c <= r) ? Ls[r * LDT + c] : 0.0;
  }
  for (int e = h; e < 256; e += nh) out[256 * jbp + e] = dinv[(16 * jbp + (e >> 4)) * DV + (e & 15)];
}

// Column-0 workgroup.  Wave 0 and the helper waves run two different programs with the same number
// of barriers (the branch is wave-uniform), so that the register file of a wave holds either the
// pivot chain's state or a helper's tiles, never both.
template <bool FULL>
__device__ __forceinline__ void panel_col0(double* __restrict__ M, int ld, const PanelJob& jb, int b, int n, double* __restrict__ Dinv,
                                           double* __restrict__ Ldiag, int* fail, double* Bs, double* Ls, double* dinv, double* dvec) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 15, lk = lane >> 4;
  const int j0 = jb.j0, t0 = jb.t0, nrt = jb.nrt;
  const bool upd = j0 >= 0;
  const int ncol = min(NB, n - t0);
  if (wave == 0) {
    // ================= pivot-chain wave =================
    d2 pv[8];
    if (upd) p0_load(M, ld, t0, j0, tid, pv);
    d4 D0;
#pragma unroll
    for (int i = 0; i < 4; i++) D0[i] = M[(size_t)(t0 + lk + 4 * i) * ld + t0 + lr];
    if (upd) p0_store(Bs, tid, pv);
    __syncthreads();
    MSFM_PROBE(1);
    if (upd) D0 = mm_nt_neg<16>(Bs, 0, Bs, 0, 0, lr, lk, D0);
    tile_st(Ls, 0, 0, lr, lk, D0);
    __syncthreads();
    MSFM_PROBE(2);
    potrf16_v2<0, FULL>(Ls, dinv, dvec, ncol, lane, fail);
    __syncthreads();
    MSFM_PROBE(3);
    __syncthreads();  // A1
    MSFM_PROBE(4);
    if (FULL || 16 < ncol) potrf16_v2<1, FULL>(Ls, dinv, dvec, ncol, lane, fail);
    else potrf16_skip<1>(dinv, lane);   // nothing but padding from here on (last block of the system)
    __syncthreads();
    MSFM_PROBE(5);
    __syncthreads();  // A2
    MSFM_PROBE(6);
    if (FULL || 32 < ncol) potrf16_v2<2, FULL>(Ls, dinv, dvec, ncol, lane, fail);
    else potrf16_skip<2>(dinv, lane);   // nothing but padding from here on (last block of the system)
    __syncthreads();
    MSFM_PROBE(7);
    __syncthreads();  // A3
    MSFM_PROBE(8);
    if (FULL || 48 < ncol) potrf16_v2<3, FULL>(Ls, dinv, dvec, ncol, lane, fail);
    else potrf16_skip<3>(dinv, lane);   // nothing but padding from here on (last block of the system)
    __syncthreads();
    MSFM_PROBE(9);
  } else {
    // ================= helper waves: one 16-row tile of the new panel each =================
    d2 pv[8];
    if (upd) p0_load(M, ld, t0, j0, tid, pv);
    const int rt = 4 + 3 * b + wave - 1;
    const bool own = rt < nrt;
    double* const pub_lo = Ldiag + (size_t)(t0 / NB) * NB * NB;
    double* const pub_out = Dinv + (size_t)(t0 / NB) * 1024;
    const size_t r0 = (size_t)job_row16(jb, own ? rt : 0);
    // The own rows are read and written as 32 contiguous bytes per lane (one row's 128 bytes across the four lanes lk of a
    // row): a lane's four doubles of a 16-column block are the columns 4 lk + i, i.e. accumulator register i stands for row
    // pl(4 i + lk) = 4 lk + i of the transposed tile instead of row 4 i + lk, and step s of a K = 64 product takes the panel
    // column kc(s) = 16 (s >> 2) + 4 lk + (s & 3).  The permutations cost nothing: they only move the LDS addresses of the
    // other operand (row pl(lr), column kc).  (Eight-byte accesses in the MFMA's natural order touch sixteen rows with 32
    // bytes each per instruction: the load phase of a launch took 8.1 k cycles and the stores of the tail 4.3 k that way.)
    const int plr = 4 * (lr & 3) + (lr >> 2);
    double preg[16];
    d4 T[4];
    if (own) {
      const double* src = M + (r0 + lr) * ld;
      if (upd) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const d4 v = *reinterpret_cast<const d4*>(&src[j0 + 16 * q + 4 * lk]);
#pragma unroll
          for (int j = 0; j < 4; j++) preg[4 * q + j] = v[j];
        }
      }
#pragma unroll
      for (int jb = 0; jb < 4; jb++) T[jb] = *reinterpret_cast<const d4*>(&src[t0 + 16 * jb + 4 * lk]);
    }
#define MSFM_KC(s) (16 * ((s) >> 2) + 4 * lk + ((s) & 3))
    // tiles of the diagonal block this wave forms: (wave, 0) now, then (ta, 1) and (tb, tc):
    // wave 1: (1,1),(2,2)   wave 2: (2,1),(3,2)   wave 3: (3,1),(3,3)
    const int ta = wave, tb = wave == 1 ? 2 : 3, tc = wave == 3 ? 3 : 2;
    d4 D0, D1, D2;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      D0[i] = M[(size_t)(t0 + 16 * wave + lk + 4 * i) * ld + t0 + lr];
      D1[i] = M[(size_t)(t0 + 16 * ta + lk + 4 * i) * ld + t0 + 16 + lr];
      D2[i] = M[(size_t)(t0 + 16 * tb + lk + 4 * i) * ld + t0 + 16 * tc + lr];
    }
    if (upd) p0_store(Bs, tid, pv);
    __syncthreads();
    // ---- column 0 of the updated diagonal block: one 16x16 tile per wave ----
    if (upd) D0 = mm_nt_neg<16>(Bs, 16 * wave, Bs, 0, 0, lr, lk, D0);
    tile_st(Ls, wave, 0, lr, lk, D0);
    __syncthreads();
    d4 X[4];
    // ---- B0 (wave 0 factors sub-panel 0): tile (h,1) and block 0 of the own rows ----
    if (upd) D1 = mm_nt_neg<16>(Bs, 16 * ta, Bs, 16, 0, lr, lk, D1);
    tile_st(Ls, ta, 1, lr, lk, D1);
    if (own && upd) {
#pragma unroll
      for (int s = 0; s < 16; s++) T[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Bs[plr * LDT + MSFM_KC(s)], preg[s], T[0], 0, 0, 0);
    }
    __syncthreads();
    // ---- A1: (h,1) -= L_h0 L_10^T ----
    tile_st(Ls, ta, 1, lr, lk, mm_nt_neg<4>(Ls, 16 * ta, Ls, 16, 0, lr, lk, tile_ld(Ls, ta, 1, lr, lk)));
    __syncthreads();
    // ---- B1 ----
    if (upd) D2 = mm_nt_neg<16>(Bs, 16 * tb, Bs, 16 * tc, 0, lr, lk, D2);
    D2 = mm_nt_neg<4>(Ls, 16 * tb, Ls, 16 * tc, 0, lr, lk, D2);
    tile_st(Ls, tb, tc, lr, lk, D2);
    if (own) {
      d4 Y = {0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 4; s++) Y = __builtin_amdgcn_mfma_f64_16x16x4f64(dinv[plr * DV + 4 * lk + s], T[0][s], Y, 0, 0, 0);
      X[0] = Y;
      if (upd) {
#pragma unroll
        for (int s = 0; s < 16; s++) T[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Bs[(16 + plr) * LDT + MSFM_KC(s)], preg[s], T[1], 0, 0, 0);
      }
#pragma unroll
      for (int s = 0; s < 4; s++) T[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ls[(16 + plr) * LDT + 4 * lk + s], X[0][s], T[1], 0, 0, 0);
    }
    if (b == 0) publish_subpanel(Ls, dinv, pub_lo, pub_out, 0, tid - 64, 192);
    __syncthreads();
    // ---- A2: (2,2), (3,2) -= L_x1 L_21^T ----
    if (wave != 3) tile_st(Ls, tb, 2, lr, lk, mm_nt_neg<4>(Ls, 16 * tb, Ls, 32, 16, lr, lk, tile_ld(Ls, tb, 2, lr, lk)));
    __syncthreads();
    // ---- B2 ----
    if (wave == 3) tile_st(Ls, 3, 3, lr, lk, mm_nt_neg<4>(Ls, 48, Ls, 48, 16, lr, lk, tile_ld(Ls, 3, 3, lr, lk)));
    if (own) {
      d4 Y = {0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 4; s++) Y = __builtin_amdgcn_mfma_f64_16x16x4f64(dinv[(16 + plr) * DV + 4 * lk + s], T[1][s], Y, 0, 0, 0);
      X[1] = Y;
      if (upd) {
#pragma unroll
        for (int s = 0; s < 16; s++) T[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Bs[(32 + plr) * LDT + MSFM_KC(s)], preg[s], T[2], 0, 0, 0);
      }
#pragma unroll
      for (int i2 = 0; i2 < 2; i2++)
#pragma unroll
        for (int s = 0; s < 4; s++) T[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ls[(32 + plr) * LDT + 16 * i2 + 4 * lk + s], X[i2][s], T[2], 0, 0, 0);
    }
    if (b == 0) publish_subpanel(Ls, dinv, pub_lo, pub_out, 1, tid - 64, 192);
    __syncthreads();
    // ---- A3: (3,3) -= L_32 L_32^T ----
    if (wave == 3) tile_st(Ls, 3, 3, lr, lk, mm_nt_neg<4>(Ls, 48, Ls, 48, 32, lr, lk, tile_ld(Ls, 3, 3, lr, lk)));
    __syncthreads();
    // ---- B3 ----
    if (own) {
      d4 Y = {0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 4; s++) Y = __builtin_amdgcn_mfma_f64_16x16x4f64(dinv[(32 + plr) * DV + 4 * lk + s], T[2][s], Y, 0, 0, 0);
      X[2] = Y;
      if (upd) {
#pragma unroll
        for (int s = 0; s < 16; s++) T[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Bs[(48 + plr) * LDT + MSFM_KC(s)], preg[s], T[3], 0, 0, 0);
      }
#pragma unroll
      for (int i2 = 0; i2 < 3; i2++)
#pragma unroll
        for (int s = 0; s < 4; s++) T[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ls[(48 + plr) * LDT + 16 * i2 + 4 * lk + s], X[i2][s], T[3], 0, 0, 0);
    }
    if (b == 0) publish_subpanel(Ls, dinv, pub_lo, pub_out, 2, tid - 64, 192);
    __syncthreads();
    // ---- tail: X_3 = T_3 Dinv_3^T, store the own rows ----
    if (own) {
      d4 Y = {0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 4; s++) Y = __builtin_amdgcn_mfma_f64_16x16x4f64(dinv[(48 + plr) * DV + 4 * lk + s], T[3][s], Y, 0, 0, 0);
      X[3] = Y;
      double* dst = M + (r0 + lr) * ld + t0;
#pragma unroll
      for (int jb = 0; jb < 4; jb++) *reinterpret_cast<d4*>(&dst[16 * jb + 4 * lk]) = X[jb];
    }
#undef MSFM_KC
  }
  if (b == 0) {
    // the last sub-panel of the diagonal block's factor (the helper waves wrote the other three during the launch)
    publish_subpanel(Ls, dinv, Ldiag + (size_t)(t0 / NB) * NB * NB, Dinv + (size_t)(t0 / NB) * 1024, 3, tid, 256);
    if (!FULL) {
      // last, partial block: it also holds the rhs row (row n), whose entries are the tail of
      // w = L^-1 rhs that the back substitution reads from M.  No other workgroup reads this tile
      // in this launch (there are no rows below it).
      for (int e = tid; e < NB * NB; e += 256) {
        const int r = e >> 6, c = e & 63;
        if (c <= r) M[(size_t)(t0 + r) * ld + t0 + c] = Ls[r * LDT + c];
      }
    }
  }
  MSFM_PROBE(10);
}

// grid: for every job its column-0 workgroups, then the 64x64 tiles (I, J), jmin <= J <= I < nt, of the
// trailing update  C_IJ -= P_I P_J^T  (jmin = 1 when the job also factors: the column-0 workgroups own J = 0).
template <bool FULL>
__global__ __launch_bounds__(256) void k_panel_v2(double* __restrict__ M, int ld, int n, double* __restrict__ Dinv,
                                                      double* __restrict__ Ldiag, int* fail, PanelJobs jobs) {
  // small arrays and the diagonal block first: their uniform-address reads then fit the 16-bit DS offset
  __shared__ double sm[80 + 64 * DV + 2 * 64 * LDT];
  double* As = sm + 80 + 64 * DV;
  double* Bs = As + 64 * LDT;
  const int tid = threadIdx.x;
  int ji = 0;
  for (int k = 1; k < jobs.count; k++)
    if ((int)blockIdx.x >= jobs.job[k].wg0) ji = k;
  const PanelJob& jb = jobs.job[ji];
  const int bl = blockIdx.x - jb.wg0;
  MSFM_PROBE_ARM(jb.j0);
  MSFM_PROBE(0);
  if (bl < jb.ncw) {
    panel_col0<FULL>(M, ld, jb, bl, n, Dinv, Ldiag, fail, /*Bs=*/Bs, /*Ls=*/As, /*dinv=*/sm + 80, /*dvec=*/sm);
    return;
  }
  const int wave = tid >> 6, lane = tid & 63;
  const int wr = wave >> 1, wc = wave & 1;
  const int lr = lane & 15, lk = lane >> 4;
  const int j0 = jb.j0;
  const int jmin = jb.t0 >= 0 ? 1 : 0, nA64 = jb.na16 / 4;
  const int nbw = jb.nwg - jb.ncw;
  const int qrow = 32 * wr + lk, qcol = 32 * wc + lr;
  // A bulk workgroup walks tiles q, q + nbw, ...: the panel tiles and the C quadrant of the NEXT tile are
  // fetched into registers while the MFMAs of the current one run (the kernel runs one workgroup per CU,
  // so nothing else hides that latency).
  auto locate = [&](int q, int& ri, int& rj, double*& C, int& ldC) {
    // pairs (I, J) with jmin <= J <= I < nt, enumerated row by row (shifted by one when jmin == 0)
    int I, J;
    const int ntri = nA64 * (nA64 - 1) / 2;  // pairs 1 <= J <= I < nA64
    if (jb.defer_corner && q >= ntri) {
      // rows of range B against the columns of range A only (defer_corner jobs always factor: jmin == 1)
      const int w = nA64 - 1, e = q - ntri;
      I = nA64 + e / w;
      J = 1 + e % w;
    } else {
      I = (int)((sqrt(8.0 * q + 1.0) + 1.0) * 0.5);
      while (I * (I - 1) / 2 > q) I--;
      while ((I + 1) * I / 2 <= q) I++;
      J = q - I * (I - 1) / 2 + 1;
      if (!jmin) { I--; J--; }
    }
    ri = job_row64(jb, I);
    rj = job_row64(jb, J);
    // destination of the tile: M, or the job's private corner when both tiles lie in range B
    if (jb.corner && I >= nA64 && J >= nA64) {
      ldC = jb.ldc;
      C = jb.corner + (size_t)(64 * (I - nA64)) * ldC + 64 * (J - nA64);
    } else {
      ldC = ld;
      C = M + (size_t)ri * ld + rj;
    }
  };
  d2 va[8], vb[8];
  d4 c00, c01, c10, c11;
  auto fetch = [&](int ri, int rj, const double* C, int ldC) {
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int e = tid + 256 * it, r = e >> 5, c2 = (e & 31) * 2;
      va[it] = *reinterpret_cast<const d2*>(&M[(size_t)(ri + r) * ld + j0 + c2]);
      vb[it] = *reinterpret_cast<const d2*>(&M[(size_t)(rj + r) * ld + j0 + c2]);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const double* p0 = &C[(size_t)(qrow + 4 * i) * ldC + qcol];
      const double* p1 = &C[(size_t)(qrow + 16 + 4 * i) * ldC + qcol];
      c00[i] = p0[0]; c01[i] = p0[16]; c10[i] = p1[0]; c11[i] = p1[16];
    }
  };
  int q = bl - jb.ncw;
  int ri, rj, ldC;
  double* C;
  locate(q, ri, rj, C, ldC);
  fetch(ri, rj, C, ldC);
  for (;;) {
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int e = tid + 256 * it, r = e >> 5, c2 = (e & 31) * 2;
      As[r * LDT + c2] = -va[it].x;  // negated once here: the MFMAs then accumulate C - P_I P_J^T directly
      As[r * LDT + c2 + 1] = -va[it].y;
      Bs[r * LDT + c2] = vb[it].x;
      Bs[r * LDT + c2 + 1] = vb[it].y;
    }
    d4 acc00 = c00, acc01 = c01, acc10 = c10, acc11 = c11;
    double* Cw = C;
    const int ldw = ldC;
    __syncthreads();
    const int qn = q + nbw;
    const bool more = qn < jb.ntile;
    if (more) {
      locate(qn, ri, rj, C, ldC);
      fetch(ri, rj, C, ldC);
    }
    quad_abt(As, Bs, wr, wc, lr, lk, acc00, acc01, acc10, acc11);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      double* p0 = &Cw[(size_t)(qrow + 4 * i) * ldw + qcol];
      double* p1 = &Cw[(size_t)(qrow + 16 + 4 * i) * ldw + qcol];
      p0[0] = acc00[i]; p0[16] = acc01[i]; p1[0] = acc10[i]; p1[16] = acc11[i];
    }
    if (!more) break;
    q = qn;
    __syncthreads();  // every wave is done with As / Bs
  }
}

// =======================================================================================
// The panel chain of a tree level as ONE persistent launch (round 4; replaces the level's k_panel_v2 launches).
//
// The work is the launch chain's, decomposed the same way - row owners (the column-0 workgroups: every one of them factors the
// updated diagonal block redundantly and solves its own three 16-row tiles behind the sub-panels) and bulk workgroups (64 x 64
// tiles of the trailing update) - but nothing ends between two 64-column steps:
//  * a row owner owns ABSOLUTE row tiles for the whole chain.  Its rows of the new panel stay in registers and are the
//    `preg` operand of its next step; only the 64 x 64 block L[t+1, t] (the rows of the next diagonal block) has to travel,
//    from the one or two workgroups that own those rows to everybody else;
//  * that block travels through `hb`, a buffer of self-tagged values: every double is its own {data, tag} granule, published
//    with an agent-scope (sc1) 8-byte store and polled by the consumers against MSFM_Z_PENDING - one memory round trip, no
//    flag, no fence (the mechanism of k_backsolve_chain).  Two buffers alternate from solve to solve; a launch marks the
//    other one "pending" for its own blocks;
//  * everything else that crosses workgroups (the rows of a finished panel for the bulk tiles, the updated tiles for the row
//    owners' next column) is written with sc1 stores, read with sc1 loads and ordered by counters: rowflag[job][tile] = steps
//    finished by the owner of that 16-row tile, tileflag[I][J] = panels the bulk workgroups have applied to tile (I, J); both
//    carry the solve's epoch in their upper bits, so nothing is ever reset;
//  * bulk tiles are handed out by a ticket counter in (step, job, column-major) order: any resident bulk workgroup can take
//    any tile, so progress needs the row owners resident (they are the first blocks of the grid) and one bulk workgroup.
// Every poll is bounded and gives up for the whole launch once MSFM_FAIL_SYNC is set (the host returns MSFM_E_DEVICE).
// The arithmetic and its order are exactly those of the launch chain: the factor is bit-identical.
// =======================================================================================
// cycle stamps of one row-owner workgroup per step (scripts/chol_probe.hip; compiled out of the library)
#ifdef MSFM_CHAIN_STAMPS
__device__ long long g_chain_stamp[256][8];
__device__ int g_chain_stamp_wg = 0;
__device__ int g_chain_stamp_jobs = 0;        // stamp only launches with this many jobs (0: all)
__device__ volatile int g_chain_stamp_on = 0;
__device__ long long g_chain_task_t[8192][4];   // per ticket: taken, counters there, products done / stored, published (wall clock, 10 ns)
__device__ long long g_chain_row_t[256][2];     // per step of the stamped row owner: step start, pivots done (wall clock)
#define BSTAMP(tk, i) do { if (threadIdx.x == 0 && g_chain_stamp_on && (tk) >= 0 && (tk) < 8192) g_chain_task_t[tk][i] = (long long)wall_clock64(); } while (0)
#define RSTAMP(i) do { if (threadIdx.x == 0 && (int)blockIdx.x == g_chain_stamp_wg && l < 256 && g_chain_stamp_on) g_chain_row_t[l][i] = (long long)wall_clock64(); } while (0)
#define CSTAMP(i) do { if (threadIdx.x == 0 && (int)blockIdx.x == g_chain_stamp_wg && l < 256 && g_chain_stamp_on) g_chain_stamp[l][i] = (long long)__builtin_readcyclecounter(); } while (0)
#define CSTAMP_H(i) do { if (threadIdx.x == 64 && (int)blockIdx.x == g_chain_stamp_wg && l < 256 && g_chain_stamp_on) g_chain_stamp[l][i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define BSTAMP(tk, i) do {} while (0)
#define RSTAMP(i) do {} while (0)
#define CSTAMP(i) do {} while (0)
#define CSTAMP_H(i) do {} while (0)
#endif
struct ChainJob {
  int begin;            // first column of the node
  int P;                // factor steps = 64-column blocks of the node
  int nA64;             // 64-row blocks of range A (the node itself; the dense root chain: everything down to the rhs row)
  int nb16, nB64, nseg; // range B: 16-row tiles with data, 64-row blocks, segments
  int sb0[4], sn16[4];
  int nrt;              // 16-row tiles that carry data, A then B, counted from the node's first row
  int ncw, wg0;         // row-owner workgroups [wg0, wg0 + ncw)
  int flag0;            // first entry of the job in rowflag
};
struct ChainJobs {
  int count, n_row_wg, n_bulk_wg, n_steps;   // n_steps: entries of task_first / 8
  ChainJob job[8];
};
// A bulk task.  type 0: trailing tile - apply the panel of step l - 1 of job k to the tile at row tiles (ti, tj) of the job.
// type 1: corner piece - add the product of the job's panel of step l, rows of the B blocks I and J, to split `piece` of the
// level's corner tile (I, J) (what k_corner_syrk does after the chains; `seq` = panels the piece has received before).
// type 2: merge - M[corner tile (I, J)] += the sum of its pieces (k_merge_corners); l = panels of the tile, piece = its splits.
struct ChainTask { short type, k, l, ti, tj, I, J, piece, seq, urgent, pad0, pad1; };
struct ChainCtl {
  unsigned* rowflag;          // [jobs][tiles]
  unsigned* tileflag;         // [nblk][nblk] by absolute 64-blocks of M
  double* hb;                 // [nblk][64][64] the rows of diagonal block i in panel i - 1, self-tagged
  unsigned long long* hb_next;
  int* ticket;                // this launch's counter (zero when the launch starts)
  int* ticket_next;           // zeroed here for the next launch
  const ChainTask* tasks;     // the launch's tiles in ticket order
  int n_tasks;
  unsigned base;              // epoch << 12
  unsigned spin_limit;
  int nblk;                   // npad / 64
  // the level's corner update folded into the launch (corners == nullptr: done by launches behind it)
  double* corners;            // [8][ldc][ldc] the pieces
  unsigned* cornerflag;       // [tile (I, J) of the B square][8]: panels a piece has received
  int ldc, corner_b0;         // leading dimension of a piece; first row of the B square
  int* dbg;                   // [8] the first poll that gave up: site, workgroup, step, awaited count, seen value, index
};
__device__ __forceinline__ void chain_note(int* dbg, int site, int step, unsigned need, unsigned seen, int index) {
  if (atomicCAS(&dbg[0], 0, site) == 0) { dbg[1] = (int)blockIdx.x; dbg[2] = step; dbg[3] = (int)need; dbg[4] = (int)seen; dbg[5] = index; dbg[6] = (int)threadIdx.x; }
}
// Coherent (agent-scope, sc1) accesses of any width through buffer instructions: the compiler tracks them like any load
// (8-byte __hip_atomic accesses cost an instruction per double: the 44 loads and 32 stores of a helper lane per step took
// ~10 k cycles to issue).  Offsets are bytes from the start of the buffer (32 bits: the host refuses systems past 4 GB).
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
typedef unsigned int u2v __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t coh_buf;
#define MSFM_COH_SC1 16
__device__ __forceinline__ coh_buf coh_make(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)std::min<size_t>(bytes, 0xFFFFFFFFull), 0x00020000);
}
__device__ __forceinline__ double ld_coh(coh_buf r, size_t idx) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)(unsigned)(idx * 8), 0, MSFM_COH_SC1));
}
__device__ __forceinline__ d2 ld_coh2(coh_buf r, size_t idx) {
  return __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(unsigned)(idx * 8), 0, MSFM_COH_SC1));
}
__device__ __forceinline__ d4 ld_coh4(coh_buf r, size_t idx) {
  const d2 a = ld_coh2(r, idx), b = ld_coh2(r, idx + 2);
  d4 v = {a.x, a.y, b.x, b.y};
  return v;
}
__device__ __forceinline__ void st_coh(coh_buf r, size_t idx, double v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, v), r, (int)(unsigned)(idx * 8), 0, MSFM_COH_SC1);
}
__device__ __forceinline__ void st_coh2(coh_buf r, size_t idx, d2 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v), r, (int)(unsigned)(idx * 8), 0, MSFM_COH_SC1);
}
__device__ __forceinline__ bool chain_aborted(int* fail) {
  return (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & MSFM_FAIL_SYNC) != 0;
}
// wave-uniform wait for a counter of this solve to reach `need`
__device__ __forceinline__ void wait_count(const unsigned* f, unsigned base, unsigned need, unsigned limit, int* fail, int* dbg, int site, int step,
                                           int index) {
  unsigned spins = 0;
  for (;;) {
    const unsigned d = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base;
    if (d < 4096u && d >= need) return;
    if ((++spins & 1023u) == 0 && (spins > limit || chain_aborted(fail))) {
      if (spins > limit) chain_note(dbg, site, step, need, d, index);
      atomicOr(fail, MSFM_FAIL_SYNC);
      return;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}
__device__ __forceinline__ int chain_row16(const ChainJob& jb, int at) {
  if (at < 4 * jb.nA64) return jb.begin + 16 * at;
  int r = at - 4 * jb.nA64, g = 0;
  while (g + 1 < jb.nseg && r >= jb.sn16[g]) { r -= jb.sn16[g]; g++; }
  return jb.sb0[g] + 16 * r;
}
// the 64 x 64 block L[t, t-1] from the hand-off buffer into pv (the layout of p0_load), polled value by value
// (rows from `rows` on carry no data - the system's last block - and have no owner: they are zero, as in M)
__device__ __forceinline__ void hb_poll(coh_buf hb, size_t blk, int tid, int rows, d2 (&pv)[8], unsigned limit, int* fail, int* dbg, int step) {
  const double pend_d = __longlong_as_double((long long)MSFM_Z_PENDING);
  auto is_pend = [](double v) { return (unsigned long long)__double_as_longlong(v) == MSFM_Z_PENDING; };
#pragma unroll
  for (int it = 0; it < 8; it++) pv[it].x = pv[it].y = ((tid + 256 * it) >> 5) < rows ? pend_d : 0.0;
  unsigned spins = 0;
  for (;;) {
    bool pend = false;
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int e = tid + 256 * it, r = e >> 5, c2 = (e & 31) * 2;
      if (is_pend(pv[it].x) || is_pend(pv[it].y)) pv[it] = ld_coh2(hb, blk * NB * NB + r * NB + c2);
    }
#pragma unroll
    for (int it = 0; it < 8; it++) pend |= is_pend(pv[it].x) || is_pend(pv[it].y);
    if (!__builtin_amdgcn_ballot_w64(pend)) break;
    if ((++spins & 255u) == 0 && (spins > limit || chain_aborted(fail))) {
      if (spins > limit && pend) chain_note(dbg, 3, step, 0u, 0u, tid);
      atomicOr(fail, MSFM_FAIL_SYNC);
#pragma unroll
      for (int it = 0; it < 8; it++) { if (is_pend(pv[it].x)) pv[it].x = 0.0; if (is_pend(pv[it].y)) pv[it].y = 0.0; }
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}
__device__ __forceinline__ double hb_safe(double v) {   // (no arithmetic produces the mark; keep the protocol safe anyway)
  return (unsigned long long)__double_as_longlong(v) == MSFM_Z_PENDING ? __longlong_as_double(0x7FF8000000000000ll) : v;
}

// A row-owner workgroup runs two programs with the same barriers per step, each with its own loop over the steps (so that
// a wave's registers hold either the pivot chain's state or a helper's tiles and carried rows, never both):
// the pivot wave (wave 0) and three helper waves with one 16-row tile of the panel each.  The bodies are panel_col0's; what
// differs is where the operands come from and go to.
//
// One coherent (sc1) round trip costs ~2.5 k cycles on MI355X (measured: scripts/chol_probe), a step's pivot chain 19 k, so
// a step must not pay more than the one round trip it cannot avoid (L[t, t-1] from its owners):
//  * the next step's tiles of the diagonal block and of the own rows are fetched DURING the current step, at the start of a
//    helper phase, once their counters (asked for one phase earlier, tested without waiting) say that the bulk workgroups
//    are done with them; wave 1 also fetches tile (0, 0) for the pivot wave and hands it over through LDS;
//  * the rows of the new panel go out sixteen columns at a time, each block in the helper phase in which it becomes final
//    (only X_3 is left for the tail), and without waiting; the counter that releases them to the bulk workgroups is written
//    behind the next step's poll of the hand-off buffer (an explicit s_waitcnt there costs nothing by then), or on exit.
struct ChainWave {
  int b, at;        // workgroup of the job, the wave's row tile counted from the node's first row
  size_t r0;        // first row of that tile in M
};
__device__ __forceinline__ bool chain_alive(const ChainJob& jb, int b, int l) { return l < jb.P && (3 * b + 2 >= 4 * l || b == jb.ncw - 1); }
__device__ __forceinline__ bool chain_is_pub(const ChainJob& jb, int b, int l) { return b == min(l == 0 ? 0 : (4 * l) / 3, jb.ncw - 1); }
__device__ __forceinline__ bool count_ready(unsigned v, unsigned base, unsigned need) { const unsigned d = v - base; return d < 4096u && d >= need; }

// The nine workgroup barriers of a row owner's step, by NAME.  Wave 0 (chain_pivot_step) and waves 1-3 (chain_helper_step) run
// different programs between the same barriers; s_barrier only counts arrivals, so a one-sided edit would pair barrier k of one
// program with barrier k + 1 of the other and LDS tiles would be read before they are written - a wrong factor, not an error.
// Both programs therefore spell every barrier as CHAIN_BAR(name), and the build with -DMSFM_CHAIN_BARCHECK
// (metricsfm_amd/libmsfm_barcheck.so, tests/test_gpu_ba.py::test_chain_barriers_pair_up) makes every wave log (step, name) in
// front of the barrier and compare the four logs behind it: a mismatch raises MSFM_FAIL_SYNC (-> MSFM_E_DEVICE) with note 90.
enum { CB_S1 = 1, CB_S2, CB_P0, CB_A1, CB_P1, CB_A2, CB_P2, CB_A3, CB_P3 };
#ifdef MSFM_CHAIN_BARCHECK
#define CHAIN_BAR(name) do { \
    if ((threadIdx.x & 63) == 0) barlog[threadIdx.x >> 6] = (l << 8) | (name); \
    __syncthreads(); \
    { const int want_ = (l << 8) | (name); \
      if (barlog[0] != want_ || barlog[1] != want_ || barlog[2] != want_ || barlog[3] != want_) { \
        chain_note(ctl.dbg, 90, l, (unsigned)(name), (unsigned)barlog[0], barlog[1] ^ barlog[2] ^ barlog[3]); atomicOr(fail, MSFM_FAIL_SYNC); } } \
    __syncthreads(); \
  } while (0)
#else
#define CHAIN_BAR(name) __syncthreads()
#endif
template <bool FULL>
__device__ __forceinline__ void chain_pivot_step(const ChainJob& jb, const ChainCtl& ctl, coh_buf cH, int l, int n, int* fail, double* Bs, double* Ls,
                                                 double* dinv, double* dvec, const double* d00s, int* barlog) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int lr = lane & 15, lk = lane >> 4;
  const int t0 = jb.begin + NB * l;
  const bool upd = l > 0;
  const int ncol = min(NB, n - t0);
  const int tb = t0 / NB;
  CSTAMP(0);
  RSTAMP(0);
  if (upd) {
    d2 pv[8];
    hb_poll(cH, (size_t)tb, tid, 16 * min(4, jb.nrt - 4 * l), pv, ctl.spin_limit, fail, ctl.dbg, l);
    p0_store(Bs, tid, pv);
  }
  CSTAMP(2);
  CHAIN_BAR(CB_S1);   // L[t, t-1] and the block's tile (0, 0) are in LDS
  d4 D0;
#pragma unroll
  for (int i = 0; i < 4; i++) D0[i] = d00s[4 * lane + i];
  if (upd) D0 = mm_nt_neg<16>(Bs, 0, Bs, 0, 0, lr, lk, D0);
  tile_st(Ls, 0, 0, lr, lk, D0);
  CHAIN_BAR(CB_S2);
  CSTAMP(3);
  potrf16_v2<0, FULL>(Ls, dinv, dvec, ncol, lane, fail);
  CHAIN_BAR(CB_P0);
  CHAIN_BAR(CB_A1);
  if (FULL || 16 < ncol) potrf16_v2<1, FULL>(Ls, dinv, dvec, ncol, lane, fail);
  else potrf16_skip<1>(dinv, lane);
  CHAIN_BAR(CB_P1);
  CHAIN_BAR(CB_A2);
  if (FULL || 32 < ncol) potrf16_v2<2, FULL>(Ls, dinv, dvec, ncol, lane, fail);
  else potrf16_skip<2>(dinv, lane);
  CHAIN_BAR(CB_P2);
  CHAIN_BAR(CB_A3);
  if (FULL || 48 < ncol) potrf16_v2<3, FULL>(Ls, dinv, dvec, ncol, lane, fail);
  else potrf16_skip<3>(dinv, lane);
  CHAIN_BAR(CB_P3);
  CSTAMP(4);
  RSTAMP(1);
}

// what a helper wave carries from step to step
struct ChainCarry {
  double xr[16];          // its rows of the previous panel (the `preg` operand)
  d4 Tn[4], Dn0, Dn1, Dn2, Dn00;   // next step's operands, fetched ahead
  bool have_next;         // ... are in the registers above
  unsigned fd, fo;        // the two counters they depend on, as last read
  bool polled;
  int flag_step;          // > 0: rowflag[at] = base + flag_step is still to be written (the rows went out without waiting)
};
// Sixteen columns (block q) of a helper wave's rows of the new panel, the moment they are final: to M for the bulk tiles and,
// when they are rows of the next diagonal block, to the hand-off buffer.  Spread over the helper phases (X_0 .. X_2 are final
// one phase each before the step ends), only X_3 is left for the tail.
__device__ __forceinline__ void chain_emit(coh_buf cM, coh_buf cH, int ld, size_t mrow, bool to_hb, size_t hrow, int q, d4 x, int lk) {
  const d2 lo = {x[0], x[1]}, hi = {x[2], x[3]};
  if (to_hb) {
    const d2 slo = {hb_safe(x[0]), hb_safe(x[1])}, shi = {hb_safe(x[2]), hb_safe(x[3])};
    st_coh2(cH, hrow + 16 * q + 4 * lk, slo);
    st_coh2(cH, hrow + 16 * q + 4 * lk + 2, shi);
  }
  st_coh2(cM, mrow + 16 * q + 4 * lk, lo);
  st_coh2(cM, mrow + 16 * q + 4 * lk + 2, hi);
}
__device__ __forceinline__ void chain_release_rows(const ChainJob& jb, const ChainCtl& ctl, int at, int flag_step, int lane) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the rows have reached the coherence point before the counter moves
  if (lane == 0) __hip_atomic_store(&ctl.rowflag[jb.flag0 + at], ctl.base + (unsigned)flag_step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The operands of step `lt` of a helper wave (its rows of column lt, its tiles of the diagonal block; wave 1: tile (0, 0) for
// the pivot wave), fetched as soon as their counters allow, never waiting: a call either looks at the counters it asked for
// in the call before and, if they are there, issues the loads - or asks for the counters again.
__device__ __forceinline__ void chain_fetch_ahead(coh_buf cM, int ld, const ChainJob& jb, const ChainCtl& ctl, const ChainWave& W, int lt, ChainCarry& C) {
  if (C.have_next || !chain_alive(jb, W.b, lt)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
  const int t1 = jb.begin + NB * lt, tb1 = t1 / NB;
  const bool own_n = W.at >= 4 * (lt + 1) && W.at < jb.nrt;
  const unsigned need_n = lt >= 2 ? (unsigned)(lt - 1) : 0u;
  const int ta = wave, tb2 = wave == 1 ? 2 : 3, tc = wave == 3 ? 3 : 2;
  if (need_n == 0 || (C.polled && count_ready(C.fd, ctl.base, need_n) && (!own_n || count_ready(C.fo, ctl.base, need_n)))) {
    if (own_n) {
      const size_t src = (W.r0 + lr) * ld;
#pragma unroll
      for (int q = 0; q < 4; q++) C.Tn[q] = ld_coh4(cM, src + t1 + 16 * q + 4 * lk);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      C.Dn0[i] = ld_coh(cM, (size_t)(t1 + 16 * wave + lk + 4 * i) * ld + t1 + lr);
      C.Dn1[i] = ld_coh(cM, (size_t)(t1 + 16 * ta + lk + 4 * i) * ld + t1 + 16 + lr);
      C.Dn2[i] = ld_coh(cM, (size_t)(t1 + 16 * tb2 + lk + 4 * i) * ld + t1 + 16 * tc + lr);
      if (wave == 1) C.Dn00[i] = ld_coh(cM, (size_t)(t1 + lk + 4 * i) * ld + t1 + lr);
    }
    C.have_next = true;
  } else {
    C.fd = __hip_atomic_load(&ctl.tileflag[(size_t)tb1 * ctl.nblk + tb1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    C.fo = own_n ? __hip_atomic_load(&ctl.tileflag[(size_t)(W.r0 / NB) * ctl.nblk + tb1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    C.polled = true;
  }
}

template <bool FULL>
__device__ __forceinline__ void chain_helper_step(double* __restrict__ M, coh_buf cM, coh_buf cH, int ld, const ChainJob& jb, const ChainCtl& ctl,
                                                  const ChainWave& W, int l, int n,
                                                  double* __restrict__ Dinv, double* __restrict__ Ldiag, int* fail, double* Bs, double* Ls, double* dinv,
                                                  double* d00s, ChainCarry& C, int* barlog) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 15, lk = lane >> 4;
  const int t0 = jb.begin + NB * l, at = W.at;
  const size_t r0 = W.r0;
  const bool upd = l > 0;
  const int tb = t0 / NB;
  const bool is_pub = chain_is_pub(jb, W.b, l);
  const bool own = at >= 4 * (l + 1) && at < jb.nrt;
  double* const pub_lo = Ldiag + (size_t)tb * NB * NB;
  double* const pub_out = Dinv + (size_t)tb * 1024;
  const int plr = 4 * (lr & 3) + (lr >> 2);
  const int ta = wave, tb2 = wave == 1 ? 2 : 3, tc = wave == 3 ? 3 : 2;
  const size_t mrow = (r0 + lr) * ld + t0;                                        // the lane's row of the new panel in M
  const bool to_hb = own && at < 4 * (l + 2) && l + 1 < jb.P;                     // ... a row of the next diagonal block
  const size_t hrow = (size_t)(tb + 1) * NB * NB + (size_t)(16 * (at & 3) + lr) * NB;
  d4 T[4], D0, D1, D2, D00;
  // this step's operands: fetched during the step before; or now, under the poll below, when their counters (asked for at the
  // end of the step before) have arrived since; or, failing that, behind a wait
  chain_fetch_ahead(cM, ld, jb, ctl, W, l, C);
  d2 pv[8];
  if (upd) hb_poll(cH, (size_t)tb, tid, 16 * min(4, jb.nrt - 4 * l), pv, ctl.spin_limit, fail, ctl.dbg, l);
  if (C.flag_step > 0) { chain_release_rows(jb, ctl, at, C.flag_step, lane); C.flag_step = 0; }   // the rows of the step before are out
  if (upd) p0_store(Bs, tid, pv);
  if (!C.have_next) {
    const unsigned need = l >= 2 ? (unsigned)(l - 1) : 0u;
    if (need) wait_count(&ctl.tileflag[(size_t)tb * ctl.nblk + tb], ctl.base, need, ctl.spin_limit, fail, ctl.dbg, 1, l, tb * ctl.nblk + tb);
    if (own && need) wait_count(&ctl.tileflag[(size_t)(r0 / NB) * ctl.nblk + tb], ctl.base, need, ctl.spin_limit, fail, ctl.dbg, 2, l, (int)(r0 / NB) * ctl.nblk + tb);
    C.polled = true; C.fd = C.fo = ctl.base + 4095u;   // (the counters are there now)
    chain_fetch_ahead(cM, ld, jb, ctl, W, l, C);
  }
#pragma unroll
  for (int q = 0; q < 4; q++) T[q] = C.Tn[q];
  D0 = C.Dn0; D1 = C.Dn1; D2 = C.Dn2; D00 = C.Dn00;
  C.have_next = false;
  C.polled = false;
  auto fetch_ahead = [&]() { chain_fetch_ahead(cM, ld, jb, ctl, W, l + 1, C); };
  if (wave == 1) {
#pragma unroll
    for (int i = 0; i < 4; i++) d00s[4 * lane + i] = D00[i];
  }
  CSTAMP_H(5);
  CHAIN_BAR(CB_S1);   // S1
#define MSFM_KC(s) (16 * ((s) >> 2) + 4 * lk + ((s) & 3))
  // ---- column 0 of the updated diagonal block: one 16x16 tile per wave ----
  if (upd) D0 = mm_nt_neg<16>(Bs, 16 * wave, Bs, 0, 0, lr, lk, D0);
  tile_st(Ls, wave, 0, lr, lk, D0);
  CHAIN_BAR(CB_S2);   // S2
  d4 X[4];
  // ---- B0 ----
  fetch_ahead();
  if (upd) D1 = mm_nt_neg<16>(Bs, 16 * ta, Bs, 16, 0, lr, lk, D1);
  tile_st(Ls, ta, 1, lr, lk, D1);
  if (own && upd) {
#pragma unroll
    for (int s = 0; s < 16; s++) T[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Bs[plr * LDT + MSFM_KC(s)], C.xr[s], T[0], 0, 0, 0);
  }
  CHAIN_BAR(CB_P0);
  // ---- A1 ----
  tile_st(Ls, ta, 1, lr, lk, mm_nt_neg<4>(Ls, 16 * ta, Ls, 16, 0, lr, lk, tile_ld(Ls, ta, 1, lr, lk)));
  CHAIN_BAR(CB_A1);
  // ---- B1 ----
  fetch_ahead();
  if (upd) D2 = mm_nt_neg<16>(Bs, 16 * tb2, Bs, 16 * tc, 0, lr, lk, D2);
  D2 = mm_nt_neg<4>(Ls, 16 * tb2, Ls, 16 * tc, 0, lr, lk, D2);
  tile_st(Ls, tb2, tc, lr, lk, D2);
  if (own) {
    d4 Y = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; s++) Y = __builtin_amdgcn_mfma_f64_16x16x4f64(dinv[plr * DV + 4 * lk + s], T[0][s], Y, 0, 0, 0);
    X[0] = Y;
    chain_emit(cM, cH, ld, mrow, to_hb, hrow, 0, Y, lk);
    if (upd) {
#pragma unroll
      for (int s = 0; s < 16; s++) T[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Bs[(16 + plr) * LDT + MSFM_KC(s)], C.xr[s], T[1], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 4; s++) T[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ls[(16 + plr) * LDT + 4 * lk + s], X[0][s], T[1], 0, 0, 0);
  }
  if (is_pub) publish_subpanel(Ls, dinv, pub_lo, pub_out, 0, tid - 64, 192);
  CHAIN_BAR(CB_P1);
  // ---- A2 ----
  if (wave != 3) tile_st(Ls, tb2, 2, lr, lk, mm_nt_neg<4>(Ls, 16 * tb2, Ls, 32, 16, lr, lk, tile_ld(Ls, tb2, 2, lr, lk)));
  CHAIN_BAR(CB_A2);
  // ---- B2 ----
  fetch_ahead();
  if (wave == 3) tile_st(Ls, 3, 3, lr, lk, mm_nt_neg<4>(Ls, 48, Ls, 48, 16, lr, lk, tile_ld(Ls, 3, 3, lr, lk)));
  if (own) {
    d4 Y = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; s++) Y = __builtin_amdgcn_mfma_f64_16x16x4f64(dinv[(16 + plr) * DV + 4 * lk + s], T[1][s], Y, 0, 0, 0);
    X[1] = Y;
    chain_emit(cM, cH, ld, mrow, to_hb, hrow, 1, Y, lk);
    if (upd) {
#pragma unroll
      for (int s = 0; s < 16; s++) T[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Bs[(32 + plr) * LDT + MSFM_KC(s)], C.xr[s], T[2], 0, 0, 0);
    }
#pragma unroll
    for (int i2 = 0; i2 < 2; i2++)
#pragma unroll
      for (int s = 0; s < 4; s++) T[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ls[(32 + plr) * LDT + 16 * i2 + 4 * lk + s], X[i2][s], T[2], 0, 0, 0);
  }
  if (is_pub) publish_subpanel(Ls, dinv, pub_lo, pub_out, 1, tid - 64, 192);
  CHAIN_BAR(CB_P2);
  // ---- A3 ----
  if (wave == 3) tile_st(Ls, 3, 3, lr, lk, mm_nt_neg<4>(Ls, 48, Ls, 48, 32, lr, lk, tile_ld(Ls, 3, 3, lr, lk)));
  CHAIN_BAR(CB_A3);
  // ---- B3 ----
  fetch_ahead();
  if (own) {
    d4 Y = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; s++) Y = __builtin_amdgcn_mfma_f64_16x16x4f64(dinv[(32 + plr) * DV + 4 * lk + s], T[2][s], Y, 0, 0, 0);
    X[2] = Y;
    chain_emit(cM, cH, ld, mrow, to_hb, hrow, 2, Y, lk);
    if (upd) {
#pragma unroll
      for (int s = 0; s < 16; s++) T[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Bs[(48 + plr) * LDT + MSFM_KC(s)], C.xr[s], T[3], 0, 0, 0);
    }
#pragma unroll
    for (int i2 = 0; i2 < 3; i2++)
#pragma unroll
      for (int s = 0; s < 4; s++) T[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ls[(48 + plr) * LDT + 16 * i2 + 4 * lk + s], X[i2][s], T[3], 0, 0, 0);
  }
  if (is_pub) publish_subpanel(Ls, dinv, pub_lo, pub_out, 2, tid - 64, 192);
  CHAIN_BAR(CB_P3);   // S9
  // ---- tail: X_3 = T_3 Dinv_3^T; the rows go out: to the hand-off buffer first when they are the rows of the next diagonal
  //      block (everybody's next step waits for them), then to M for the bulk tiles ----
  if (own) {
    d4 Y = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; s++) Y = __builtin_amdgcn_mfma_f64_16x16x4f64(dinv[(48 + plr) * DV + 4 * lk + s], T[3][s], Y, 0, 0, 0);
    X[3] = Y;
    chain_emit(cM, cH, ld, mrow, to_hb, hrow, 3, Y, lk);   // the last sixteen columns: everybody's next step waits for them
    CSTAMP_H(6);
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int j = 0; j < 4; j++) C.xr[4 * q + j] = X[q][j];
    C.flag_step = l + 1;   // released behind the next poll (or on exit)
  }
  fetch_ahead();
  CSTAMP_H(7);
#undef MSFM_KC
  if (is_pub) {
    // (publish_subpanel of the last quarter by the helper waves alone: the pivot wave is already on its way to the next step)
    publish_subpanel(Ls, dinv, pub_lo, pub_out, 3, tid - 64, 192);
    if (!FULL) {
      for (int e = tid - 64; e < NB * NB; e += 192) {
        const int r = e >> 6, c = e & 63;
        if (c <= r) M[(size_t)(t0 + r) * ld + t0 + c] = Ls[r * LDT + c];
      }
    }
  }
}

// Bulk workgroup (and every row owner after its last step): 64 x 64 tiles of the trailing updates, taken by ticket from a
// list the host has put in PRIORITY order (chain_build): tile (I, J) of launch step l - "apply the panel of step l - 1" -
// has the key 0.4 l + 0.6 J, so the tiles of the column right behind the diagonal block (what the next step's row owners
// wait for) come before the far columns of the step before, and a far tile receives its panels a little later, in order.
// The key grows by less than 1 per step for one tile and every task's inputs have smaller keys: the list is a linear
// extension of the dependencies, whoever holds the lowest open ticket can always finish.
// The loop keeps the next tile's operands in flight under the products of the current one: the next ticket and its counters
// are asked for before the MFMAs and looked at after them; a tile that is not urgent publishes its counter one task later
// (when the next task's loads have come back, memory operations of a wave completing in issue order, its own stores are out).
__device__ __forceinline__ void chain_bulk(double* __restrict__ M, coh_buf cM, int ld, int* fail, const ChainJobs& jobs, const ChainCtl& ctl, double* As,
                                           double* Bs, int* task, int bi) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (bi >= 0) {
    // the other hand-off buffer becomes "pending" for the blocks of this launch; the next launch's ticket counter zero
    for (int k = 0; k < jobs.count; k++) {
      unsigned long long* dst = ctl.hb_next + (size_t)(jobs.job[k].begin / NB) * NB * NB;
      const size_t cnt = (size_t)jobs.job[k].P * NB * NB;
      for (size_t e = (size_t)bi * 256 + tid; e < cnt; e += (size_t)jobs.n_bulk_wg * 256) dst[e] = MSFM_Z_PENDING;
    }
    if (bi == 0 && tid == 0) *ctl.ticket_next = 0;
  }
  const int wr = wave >> 1, wc = wave & 1;
  const int lr = lane & 15, lk = lane >> 4;
  const int qrow = 32 * wr + lk, qcol = 32 * wc + lr;
  const int total = ctl.n_tasks;
  int* const tkt = task + 37;   // the tickets of the three slots (stamps only)
  const coh_buf cC = coh_make(ctl.corners ? ctl.corners : M, ctl.corners ? (size_t)8 * ctl.ldc * ctl.ldc * sizeof(double) : 8);
  // l < 0: none.  For the products (types 0, 1): rows ri / rj of the panel at column j0; the tile lives at coff in M (type 0) or in
  // the corner pieces (type 1, leading dimension cld; zero: its first panel, nothing to load); the counters it waits for are
  // the eight row tiles of the panel (>= rneed) and its own history (hist >= hneed); done = the value its history gets
  struct Tile { int l, type, ri, rj, j0, urgent, flag0, ti, tj, nrt, tk, rneed, cld, zero, hneed, dval, I, J, piece; size_t coff; unsigned* hist; };
  auto take = [&](int slot) {   // thread 0: next ticket -> task[slot * 12 ..]
    const int t = atomicAdd(ctl.ticket, 1);
    ChainTask q = {0, 0, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (t < total) q = ctl.tasks[t];
    int* o = task + 12 * slot;
    o[0] = q.l; o[1] = q.k; o[2] = q.type; o[3] = q.ti; o[4] = q.tj; o[5] = q.I; o[6] = q.J; o[7] = q.piece; o[8] = q.seq; o[9] = q.urgent;
    tkt[slot] = t;
    BSTAMP(t < total ? t : -1, 0);
  };
  auto decode = [&](int slot, Tile& T) {
    const int* o = task + 12 * slot;
    T.l = o[0];
    T.tk = tkt[slot];
    if (T.l < 0) return;
    const ChainJob& jb = jobs.job[o[1]];
    T.type = o[2]; T.ti = o[3]; T.tj = o[4]; T.I = o[5]; T.J = o[6]; T.piece = o[7];
    T.flag0 = jb.flag0; T.nrt = jb.nrt;
    if (T.type == 2) return;   // (merge: handled outside the pipeline)
    T.ri = chain_row16(jb, T.ti); T.rj = chain_row16(jb, T.tj);
    T.urgent = o[9];
    if (T.type == 0) {
      T.j0 = jb.begin + NB * (T.l - 1);
      T.rneed = T.l;
      T.coff = (size_t)T.ri * ld + T.rj; T.cld = ld; T.zero = 0;
      T.hist = &ctl.tileflag[(size_t)(T.ri / NB) * ctl.nblk + T.rj / NB];
      T.hneed = T.l - 1; T.dval = T.l;
    } else {
      T.j0 = jb.begin + NB * T.l;
      T.rneed = T.l + 1;
      T.coff = (size_t)T.piece * ctl.ldc * ctl.ldc + (size_t)(NB * T.I) * ctl.ldc + NB * T.J; T.cld = ctl.ldc; T.zero = o[8] == 0;
      T.hist = &ctl.cornerflag[(size_t)(T.I * (T.I + 1) / 2 + T.J) * 8 + T.piece];
      T.hneed = o[8]; T.dval = o[8] + 1;
    }
  };
  // the counters a tile waits for: the eight row tiles of its panel (lanes 0..7 of wave 0) and its own history (lane 8)
  auto flag_ptr = [&](const Tile& T, unsigned& need) -> const unsigned* {
    need = 0u;
    if (lane < 8) {
      const int tt = (lane < 4 ? T.ti : T.tj) + (lane & 3);
      if (tt < T.nrt) { need = (unsigned)T.rneed; return &ctl.rowflag[T.flag0 + tt]; }
    } else if (lane == 8 && T.hneed >= 1) {
      need = (unsigned)T.hneed;
      return T.hist;
    }
    return nullptr;
  };
  d2 va[8], vb[8];
  d4 c00, c01, c10, c11;
  auto fetch = [&](const Tile& T) {
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int e = tid + 256 * it, r = e >> 5, c2 = (e & 31) * 2;
      va[it] = ld_coh2(cM, (size_t)(T.ri + r) * ld + T.j0 + c2);
      vb[it] = ld_coh2(cM, (size_t)(T.rj + r) * ld + T.j0 + c2);
    }
    if (T.zero) {
      c00 = d4{0, 0, 0, 0}; c01 = c00; c10 = c00; c11 = c00;
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const size_t p0 = T.coff + (size_t)(qrow + 4 * i) * T.cld + qcol;
      const size_t p1 = T.coff + (size_t)(qrow + 16 + 4 * i) * T.cld + qcol;
      if (T.type == 0) { c00[i] = ld_coh(cM, p0); c01[i] = ld_coh(cM, p0 + 16); c10[i] = ld_coh(cM, p1); c11[i] = ld_coh(cM, p1 + 16); }
      else { c00[i] = ld_coh(cC, p0); c01[i] = ld_coh(cC, p0 + 16); c10[i] = ld_coh(cC, p1); c11[i] = ld_coh(cC, p1 + 16); }
    }
  };
  auto wait_flags = [&](const Tile& T) {   // wave 0, blocking
    unsigned need;
    const unsigned* f = flag_ptr(T, need);
    if (f) wait_count(f, ctl.base, need, ctl.spin_limit, fail, ctl.dbg, lane < 8 ? 4 : 5, T.l, lane);
  };
  // M[corner tile] += the sum of its pieces, once every piece has all its panels (k_merge_corners' sums, in its order)
  auto merge = [&](const Tile& T) {
    const int len = T.l, ns = T.piece;
    if (wave == 0 && lane < ns) {
      const int cnt = len > lane ? (len - lane - 1) / ns + 1 : 0;
      if (cnt > 0)
        wait_count(&ctl.cornerflag[(size_t)(T.I * (T.I + 1) / 2 + T.J) * 8 + lane], ctl.base, (unsigned)cnt, ctl.spin_limit, fail, ctl.dbg, 6, len, lane);
    }
    __syncthreads();
    const size_t pl = (size_t)ctl.ldc * ctl.ldc;
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int e = u * 256 + tid, r = NB * T.I + (e >> 6), c = NB * T.J + (e & 63);
      double v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = k < ns ? ld_coh(cC, (size_t)k * pl + (size_t)r * ctl.ldc + c) : 0.0;
      double sacc = 0.0;
#pragma unroll
      for (int k = 0; k < 8; k++) sacc += v[k];
      M[(size_t)(ctl.corner_b0 + r) * ld + ctl.corner_b0 + c] += sacc;
    }
    __syncthreads();
  };
  // the merges sit behind every product in the list: from the first one on a workgroup only merges
  auto merge_loop = [&](Tile T) {
    for (;;) {
      merge(T);
      if (tid == 0) take(0);
      __syncthreads();
      decode(0, T);
      if (T.l < 0) return;
    }
  };
  // Three tasks in flight: `cur` (operands in registers, then in LDS under the products), `nxt` (known; its counters asked
  // for a round earlier; fetched under cur's products when they are there) and `nn` (ticket taken, counters asked for).
  Tile cur, nxt, nn;
  unsigned* pending = nullptr;
  int pending_l = 0;   // (thread 0) the counter that is still to be written, and its value
  unsigned fneed = 0u, fval = 0u;    // (wave 0) the counter of nxt this lane looks at, as read a round ago
  const unsigned* fp = nullptr;
  auto ask = [&](const Tile& T) {     // wave 0: read the lane's counter of T without waiting for it
    fp = nullptr; fneed = 0u; fval = 0u;
    if (T.l >= 0) {
      fp = flag_ptr(T, fneed);
      if (fp) fval = __hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  if (tid == 0) take(0);
  __syncthreads();
  decode(0, cur);
  if (cur.l < 0) return;
  if (cur.type == 2) { merge_loop(cur); return; }
  if (tid == 0) take(1);
  __syncthreads();
  decode(1, nxt);
  const auto is_product = [](const Tile& T) { return T.l >= 0 && T.type != 2; };
  Tile mtask;          // the first merge task this workgroup drew (handled when the products before it are done)
  mtask.l = -1;
  if (nxt.l >= 0 && nxt.type == 2) { mtask = nxt; nxt.l = -1; }
  if (wave == 0) { wait_flags(cur); if (is_product(nxt)) ask(nxt); }
  __syncthreads();
  BSTAMP(cur.tk, 1);
  fetch(cur);
  for (;;) {
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int e = tid + 256 * it, r = e >> 5, c2 = (e & 31) * 2;
      As[r * LDT + c2] = -va[it].x;
      As[r * LDT + c2 + 1] = -va[it].y;
      Bs[r * LDT + c2] = vb[it].x;
      Bs[r * LDT + c2 + 1] = vb[it].y;
    }
    d4 a00 = c00, a01 = c01, a10 = c10, a11 = c11;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's loads are back; so are its stores of the tile before
    if (wave == 0) {
      const bool ok = nxt.l < 0 || !fp || count_ready(fval, ctl.base, fneed);
      const bool all = __builtin_amdgcn_ballot_w64(!ok) == 0ull;
      if (lane == 0) {
        task[36] = all ? 1 : 0;
        // a third ticket only when the second one can go ahead: a workgroup that is going to wait for its next tile must not
        // sit on another one meanwhile (-2: not taken yet)
        if (all && nxt.l >= 0) take(2); else task[24] = nxt.l >= 0 ? -2 : -1;
      }
    }
    __syncthreads();   // B1: the operands are in LDS; the stores of the tile before are out (every wave waited above)
    if (tid == 0 && pending) {
      __hip_atomic_store(pending, ctl.base + (unsigned)pending_l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pending = nullptr;
    }
    const bool ready = task[36] != 0 && nxt.l >= 0;
    if (ready) { fetch(nxt); BSTAMP(nxt.tk, 1); }   // in flight under the products below
    decode(2, nn);
    if (nn.l >= 0 && nn.type == 2) { mtask = nn; nn.l = -1; }   // a merge: no more products for this workgroup
    if (wave == 0 && nn.l >= 0) ask(nn);
    quad_abt(As, Bs, wr, wc, lr, lk, a00, a01, a10, a11);
    {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const size_t p0 = cur.coff + (size_t)(qrow + 4 * i) * cur.cld + qcol;
        const size_t p1 = cur.coff + (size_t)(qrow + 16 + 4 * i) * cur.cld + qcol;
        if (cur.type == 0) { st_coh(cM, p0, a00[i]); st_coh(cM, p0 + 16, a01[i]); st_coh(cM, p1, a10[i]); st_coh(cM, p1 + 16, a11[i]); }
        else { st_coh(cC, p0, a00[i]); st_coh(cC, p0 + 16, a01[i]); st_coh(cC, p1, a10[i]); st_coh(cC, p1 + 16, a11[i]); }
      }
    }
    const bool last = nxt.l < 0;
    if (cur.urgent || last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next step's row owners wait for this tile
    __syncthreads();   // B2: every wave's part of the tile is issued (urgent: out); everybody is done with As / Bs and task[]
    BSTAMP(cur.tk, 2);
    if (tid == 0) {
      if (cur.urgent || last) __hip_atomic_store(cur.hist, ctl.base + (unsigned)cur.dval, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else { pending = cur.hist; pending_l = cur.dval; }
    }
    if (last) break;
    if (!ready) {   // (uniform) the counters were not there a round ago: wait for them now
      // ... but never with a counter of our own unpublished: what nxt waits for may hang on it (the same tile a step later)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0 && pending) {
        __hip_atomic_store(pending, ctl.base + (unsigned)pending_l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pending = nullptr;
      }
      if (wave == 0) wait_flags(nxt);
      __syncthreads();
      BSTAMP(nxt.tk, 1);
      fetch(nxt);
    }
    cur = nxt;
    if (nn.l == -2) {   // the third ticket, now that the second one is on its way
      if (tid == 0) take(2);
      __syncthreads();
      decode(2, nn);
      if (nn.l >= 0 && nn.type == 2) { mtask = nn; nn.l = -1; }
      if (wave == 0 && nn.l >= 0) ask(nn);
    }
    nxt = nn;
  }
  if (mtask.l >= 0) { __syncthreads(); merge_loop(mtask); }
}

__global__ __launch_bounds__(256) void k_chain(double* __restrict__ M, int ld, int n, double* __restrict__ Dinv, double* __restrict__ Ldiag,
                                                int* fail, ChainJobs jobs, ChainCtl ctl) {
  __shared__ double sm[80 + 64 * DV + 2 * 64 * LDT];
  __shared__ double d00s[256];
  __shared__ int task[40];   // three ticket slots of 12 words, the 'next is ready' word, the three tickets
  __shared__ int barlog_s[4];   // (MSFM_CHAIN_BARCHECK: what every wave says it is waiting at)
  int* const barlog = barlog_s;
  double* As = sm + 80 + 64 * DV;
  double* Bs = As + 64 * LDT;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const coh_buf cM = coh_make(M, (size_t)ld * ld * sizeof(double));
  const coh_buf cH = coh_make(ctl.hb, (size_t)ctl.nblk * NB * NB * sizeof(double));
  if ((int)blockIdx.x < jobs.n_row_wg) {
    // ---------------- row owner ----------------
    int ji = 0;
    for (int k = 1; k < jobs.count; k++)
      if ((int)blockIdx.x >= jobs.job[k].wg0) ji = k;
    const ChainJob& jb = jobs.job[ji];
    const int b = blockIdx.x - jb.wg0;
#ifdef MSFM_CHAIN_STAMPS
    if (tid == 0 && (int)blockIdx.x == g_chain_stamp_wg) g_chain_stamp_on = g_chain_stamp_jobs == 0 || jobs.count == g_chain_stamp_jobs;
#endif
    if (wave == 0) {
      for (int l = 0; chain_alive(jb, b, l); l++) {
        if (n - (jb.begin + NB * l) >= NB) chain_pivot_step<true>(jb, ctl, cH, l, n, fail, Bs, As, sm + 80, sm, d00s, barlog);
        else chain_pivot_step<false>(jb, ctl, cH, l, n, fail, Bs, As, sm + 80, sm, d00s, barlog);
      }
    } else {
      ChainWave W;
      W.b = b;
      W.at = 4 + 3 * b + wave - 1;
      W.r0 = (size_t)chain_row16(jb, W.at < jb.nrt ? W.at : 0);
      ChainCarry C;
#pragma unroll
      for (int i = 0; i < 16; i++) C.xr[i] = 0.0;
      C.have_next = false; C.polled = false; C.fd = C.fo = 0u; C.flag_step = 0;
      for (int l = 0; chain_alive(jb, b, l); l++) {
        if (n - (jb.begin + NB * l) >= NB) chain_helper_step<true>(M, cM, cH, ld, jb, ctl, W, l, n, Dinv, Ldiag, fail, Bs, As, sm + 80, d00s, C, barlog);
        else chain_helper_step<false>(M, cM, cH, ld, jb, ctl, W, l, n, Dinv, Ldiag, fail, Bs, As, sm + 80, d00s, C, barlog);
      }
      if (C.flag_step > 0) chain_release_rows(jb, ctl, W.at, C.flag_step, lane);
    }
    // a row owner whose rows have all passed the diagonal takes bulk tiles for the rest of the launch
    __syncthreads();
    chain_bulk(M, cM, ld, fail, jobs, ctl, As, Bs, task, -1);
    return;
  }
  // ---------------- bulk: tiles of the trailing updates, by ticket ----------------
  chain_bulk(M, cM, ld, fail, jobs, ctl, As, Bs, task, (int)blockIdx.x - jobs.n_row_wg);
}

// The separator x separator part of the domain chains' trailing updates, all at once:  corner_r[I][J] = -sum_p X_I,p X_J,p^T
// over the 64-column panels p = p_begin + r, p_begin + r + nsplit, ... of the level's columns [64 p_begin, b0) (X = the rows
// of the factor from b0 on).
// grid = lower 64x64 tiles of the separator square x nsplit; the next panel's tiles are fetched during the MFMAs.
// Only panels of leaves / nodes under BOTH blocks' tree nodes contribute (everything else is structurally zero):
// blk_plo / blk_phi give, per 64-row block of the square, the panel range of the level that lies under its node.
// K-split per tile (round 3): a tile under the root receives every panel of the level, a tile under a deep separator only
// its own subtree's - with one split count for all tiles the root x root workgroups walked twice as many panels as the rest
// and set the launch's length (config 3, level 0: 11 panel products against 5-6).  Now tile (I, J) is cut into
// ceil(panels / target) pieces, at most MSFM_CORNER_SPLIT_MAX; the grid holds MSFM_CORNER_SPLIT_MAX workgroups per tile and
// the surplus ones leave at once (a thousand empty workgroups start and end within a microsecond).
#define MSFM_CORNER_SPLIT_MAX 8
struct CornerRanges { short plo[MSFM_CORNER_MAX_BLOCKS], phi[MSFM_CORNER_MAX_BLOCKS]; };
__host__ __device__ __forceinline__ int corner_splits(int panels, int target) {
  const int s = (panels + target - 1) / target;
  return s < 1 ? 1 : (s > MSFM_CORNER_SPLIT_MAX ? MSFM_CORNER_SPLIT_MAX : s);
}
__global__ __launch_bounds__(256) void k_corner_syrk(const double* __restrict__ M, int ld, int b0, int target, int smax,
                                                      double* __restrict__ corners, int ldc, CornerRanges R) {
  __shared__ double sm[2 * 64 * LDT];
  double* As = sm;
  double* Bs = sm + 64 * LDT;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wr = wave >> 1, wc = wave & 1, lr = lane & 15, lk = lane >> 4;
  const int tile = blockIdx.x / smax, r = blockIdx.x % smax;
  int I = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
  while (I * (I + 1) / 2 > tile) I--;
  while ((I + 1) * (I + 2) / 2 <= tile) I++;
  const int J = tile - I * (I + 1) / 2;
  const int ri = b0 + 64 * I, rj = b0 + 64 * J;
  const int p_begin = max((int)R.plo[I], (int)R.plo[J]), np = min((int)R.phi[I], (int)R.phi[J]);
  const int nsplit = corner_splits(np - p_begin, target);
  if (r >= nsplit) return;
  d4 acc00 = {0, 0, 0, 0}, acc01 = acc00, acc10 = acc00, acc11 = acc00;
  d2 va[8], vb[8];
  auto fetch = [&](int p) {
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int e = tid + 256 * it, row = e >> 5, c2 = (e & 31) * 2;
      va[it] = *reinterpret_cast<const d2*>(&M[(size_t)(ri + row) * ld + 64 * p + c2]);
      vb[it] = *reinterpret_cast<const d2*>(&M[(size_t)(rj + row) * ld + 64 * p + c2]);
    }
  };
  int p = p_begin + r;
  if (p < np) fetch(p);
  for (; p < np; p += nsplit) {
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int e = tid + 256 * it, row = e >> 5, c2 = (e & 31) * 2;
      As[row * LDT + c2] = -va[it].x;
      As[row * LDT + c2 + 1] = -va[it].y;
      Bs[row * LDT + c2] = vb[it].x;
      Bs[row * LDT + c2 + 1] = vb[it].y;
    }
    __syncthreads();
    if (p + nsplit < np) fetch(p + nsplit);
    quad_abt(As, Bs, wr, wc, lr, lk, acc00, acc01, acc10, acc11);
    __syncthreads();
  }
  double* C = corners + (size_t)r * ldc * ldc + (size_t)(64 * I) * ldc + 64 * J;
  const int qrow = 32 * wr + lk, qcol = 32 * wc + lr;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    double* c0 = &C[(size_t)(qrow + 4 * i) * ldc + qcol];
    double* c1 = &C[(size_t)(qrow + 16 + 4 * i) * ldc + qcol];
    c0[0] = acc00[i]; c0[16] = acc01[i]; c1[0] = acc10[i]; c1[16] = acc11[i];
  }
}

// M[b0.., b0..] += sum_k corner_k (lower 64x64 tiles of the separator square), after the domain chains
__global__ __launch_bounds__(256) void k_merge_corners(double* __restrict__ M, int ld, int b0, int nB64, const double* __restrict__ corners,
                                                        int ldc, int target, CornerRanges R) {
  // 4 workgroups per lower 64x64 tile, 4 elements per thread, all loads issued before the sums
  const int tile = blockIdx.x >> 2, part = blockIdx.x & 3;
  int I = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
  while (I * (I + 1) / 2 > tile) I--;
  while ((I + 1) * (I + 2) / 2 <= tile) I++;
  const int J = tile - I * (I + 1) / 2;
  (void)nB64;
  const int ncorner = corner_splits(min((int)R.phi[I], (int)R.phi[J]) - max((int)R.plo[I], (int)R.plo[J]), target);
  double v[4][8];
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int e = part * 1024 + u * 256 + threadIdx.x;
    const int r = 64 * I + (e >> 6), c = 64 * J + (e & 63);
#pragma unroll
    for (int k = 0; k < 8; k++) v[u][k] = k < ncorner ? corners[(size_t)k * ldc * ldc + (size_t)r * ldc + c] : 0.0;
  }
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int e = part * 1024 + u * 256 + threadIdx.x;
    const int r = 64 * I + (e >> 6), c = 64 * J + (e & 63);
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 8; k++) s += v[u][k];
    M[(size_t)(b0 + r) * ld + b0 + c] += s;
  }
}

// ---------------------------------------------------------------------------------------
// Full inverses of all 64x64 diagonal blocks of the factor (one workgroup per block, after the
// factorisation, off the critical path): from the 16x16 inverses by two doubling steps
//   inv([A 0; B C]) = [A^-1 0; -C^-1 B A^-1  C^-1].   Linv[blk] row-major 64x64.
// ---------------------------------------------------------------------------------------
// Workgroups past the last diagonal block copy the eliminated rhs row (row n of M) into w for the back substitution
// (one launch fewer on the dependent chain).
// C (16 x 16 tile at rows rc, columns cc of Cm) = sign * A[ra.., ka..ka+K) * B[kb..kb+K, cb..) with all three row-major in LDS
// (stride LDT), one wave, K a multiple of 4: the f64 MFMA takes A[m = lane & 15][k = lane >> 4] and B[k = lane >> 4][n = lane & 15].
template <int K>
__device__ __forceinline__ void tile_mm(double* Cm, int rc, int cc, const double* A, int ra, int ka, const double* B, int kb, int cb, bool negate,
                                        int lr, int lk) {
  d4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int ks = 0; ks < K / 4; ks++) {
    const double a = A[(ra + lr) * LDT + ka + 4 * ks + lk];
    const double b = B[(kb + 4 * ks + lk) * LDT + cb + lr];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(negate ? -a : a, b, acc, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) Cm[(rc + lk + 4 * i) * LDT + cc + lr] = acc[i];
}

// The full inverse of the 64 x 64 diagonal block `blk` of the factor in v (LDS, stride LDT), by all 256 threads of a
// workgroup; L and t are scratch of the same size.
__device__ __forceinline__ void trinv64_block(const double* __restrict__ Ldiag, const double* __restrict__ Dinv, int blk, int n, double* L, double* v,
                                              double* t) {
  const int j0 = blk * NB, tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
  const int ncol = min(NB, n - j0);
  for (int e = tid; e < NB * NB; e += 256) {
    const int r = e >> 6, c = e & 63;
    L[r * LDT + c] = (c <= r && r < ncol) ? Ldiag[(size_t)blk * NB * NB + r * NB + c] : 0.0;
    const bool diag16 = (r >> 4) == (c >> 4);
    v[r * LDT + c] = diag16 ? Dinv[(size_t)blk * 1024 + ((r >> 4) * 16 + (r & 15)) * 16 + (c & 15)] : 0.0;
  }
  __syncthreads();
  // s = 16: the two 32 x 32 diagonal blocks, off-diagonal tile  -C^-1 (B A^-1)  each (waves 0 and 1)
  if (wave < 2) {
    const int o = 32 * wave;
    tile_mm<16>(t, o + 16, o, L, o + 16, o, v, o, o, false, lr, lk);            // t = B A^-1
  }
  __syncthreads();
  if (wave < 2) {
    const int o = 32 * wave;
    tile_mm<16>(v, o + 16, o, v, o + 16, o + 16, t, o + 16, o, true, lr, lk);   // -C^-1 t
  }
  __syncthreads();
  // s = 32: the lower-left 32 x 32 block, one 16 x 16 tile per wave
  {
    const int tr = wave >> 1, tc = wave & 1;
    tile_mm<32>(t, 32 + 16 * tr, 16 * tc, L, 32 + 16 * tr, 0, v, 0, 16 * tc, false, lr, lk);
    __syncthreads();
    tile_mm<32>(v, 32 + 16 * tr, 16 * tc, v, 32 + 16 * tr, 32, t, 32, 16 * tc, true, lr, lk);
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void k_trinv64_full(const double* __restrict__ Ldiag, int n, const double* __restrict__ Dinv,
                                                       double* __restrict__ Linv, int nblk, const double* __restrict__ M, int ld,
                                                       double* __restrict__ w, int npad, unsigned long long* __restrict__ zfill) {
  __shared__ double L[NB * LDT];
  __shared__ double v[NB * LDT];
  __shared__ double t[NB * LDT];
  if ((int)blockIdx.x >= nblk) {
    const int i = ((int)blockIdx.x - nblk) * 256 + (int)threadIdx.x;
    if (i < npad) {
      w[i] = (i < n) ? M[(size_t)n * ld + i] : 0.0;
      if (zfill) zfill[i] = MSFM_Z_PENDING;   // "not solved yet" for k_backsolve_chain
    }
    return;
  }
  trinv64_block(Ldiag, Dinv, (int)blockIdx.x, n, L, v, t);
  double* out = Linv + (size_t)blockIdx.x * NB * NB;
  for (int e = threadIdx.x; e < NB * NB; e += 256) out[e] = v[(e >> 6) * LDT + (e & 63)];
}

// ---------------------------------------------------------------------------------------
// Back substitution step for block jb (from the last block down):
//   z_j = Linv_j^T w_j ;  w_i -= L_ji^T z_j for the blocks i in [ib, ib + ni) (all blocks before j in
//   the dense order; with camera domains only the blocks of j's own domain couple to it, and the
//   steps of different domains share a launch).
// Per job max(ni, 1) workgroups of 256 threads; every workgroup recomputes z_j itself (64x64
// mat-vec against the explicit inverse) so there is no in-launch dependency; the job's first
// workgroup stores z_j.  Nobody writes w_j in this launch.
// ---------------------------------------------------------------------------------------
struct BackJob { int jb, ib, ni, wg0; };
struct BackJobs { int count; BackJob job[16]; };
__global__ __launch_bounds__(256) void k_backsolve_step(const double* __restrict__ M, int ld, int n, BackJobs jobs,
                                                         const double* __restrict__ Linv, double* __restrict__ w,
                                                         double* __restrict__ z) {
  __shared__ double part[4][NB];
  __shared__ double zj[NB];
  int ji = 0;
  for (int k = 1; k < jobs.count; k++)
    if ((int)blockIdx.x >= jobs.job[k].wg0) ji = k;
  const int jb = jobs.job[ji].jb, bl = blockIdx.x - jobs.job[ji].wg0, ni = jobs.job[ji].ni;
  const int tid = threadIdx.x, j0 = jb * NB;
  const int col = tid & 63, kq = tid >> 6;
  const double* Li = Linv + (size_t)jb * NB * NB;
  // issue the loads of both phases up front: they do not depend on each other
  double lv[16], mv[16];
  const int i0 = (jobs.job[ji].ib + bl) * NB;
#pragma unroll
  for (int k = 0; k < 16; k++) lv[k] = Li[(16 * kq + k) * NB + col];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int r = j0 + 16 * kq + k;
    mv[k] = (ni > 0 && r < n) ? M[(size_t)r * ld + i0 + col] : 0.0;
  }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 16; k++) s += lv[k] * w[j0 + 16 * kq + k];
  part[kq][col] = s;
  __syncthreads();
  if (tid < NB) {
    const double zz = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    zj[tid] = zz;
    if (bl == 0) z[j0 + tid] = zz;
  }
  __syncthreads();
  if (ni == 0) return;
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < 16; k++) acc += mv[k] * zj[16 * kq + k];
  __syncthreads();
  part[kq][col] = acc;
  __syncthreads();
  if (tid < NB) w[i0 + tid] -= (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
}

// Two adjacent blocks (jb-1, jb) of the backward solve in one launch: z2 = L22^-T w2, w1' = w1 - L21^T z2,
// z1 = L11^-T w1', then w_i -= L_{jb,i}^T z2 and w_i -= L_{jb-1,i}^T z1 for the job's earlier blocks i - the operations
// of two consecutive k_backsolve_step launches in the same order and with the same partial-sum trees (bit-identical
// result), every workgroup recomputing z2, w1' and z1 for itself.  Halves the dependent launches of the back substitution.
__global__ __launch_bounds__(256) void k_backsolve_pair(const double* __restrict__ M, int ld, int n, BackJobs jobs,
                                                         const double* __restrict__ Linv, double* __restrict__ w,
                                                         double* __restrict__ z) {
  __shared__ double part[4][NB];
  __shared__ double z2[NB], z1[NB], w1p[NB];
  int ji = 0;
  for (int k = 1; k < jobs.count; k++)
    if ((int)blockIdx.x >= jobs.job[k].wg0) ji = k;
  const int jb = jobs.job[ji].jb, bl = blockIdx.x - jobs.job[ji].wg0, ni = jobs.job[ji].ni;
  const int tid = threadIdx.x, j2 = jb * NB, j1 = (jb - 1) * NB;
  const int col = tid & 63, kq = tid >> 6;
  const double* Li2 = Linv + (size_t)jb * NB * NB;
  const double* Li1 = Linv + (size_t)(jb - 1) * NB * NB;
  const int i0 = (jobs.job[ji].ib + bl) * NB;
  double lv2[16], lv1[16], m21[16], mv2[16], mv1[16];
#pragma unroll
  for (int k = 0; k < 16; k++) { lv2[k] = Li2[(16 * kq + k) * NB + col]; lv1[k] = Li1[(16 * kq + k) * NB + col]; }
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int r2 = j2 + 16 * kq + k, r1 = j1 + 16 * kq + k;
    m21[k] = r2 < n ? M[(size_t)r2 * ld + j1 + col] : 0.0;
    mv2[k] = (ni > 0 && r2 < n) ? M[(size_t)r2 * ld + i0 + col] : 0.0;
    mv1[k] = (ni > 0 && r1 < n) ? M[(size_t)r1 * ld + i0 + col] : 0.0;
  }
  // z2 = L22^-T w2
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 16; k++) s += lv2[k] * w[j2 + 16 * kq + k];
  part[kq][col] = s;
  __syncthreads();
  if (tid < NB) {
    const double zz = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    z2[tid] = zz;
    if (bl == 0) z[j2 + tid] = zz;
  }
  __syncthreads();
  // w1' = w1 - L21^T z2
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < 16; k++) acc += m21[k] * z2[16 * kq + k];
  __syncthreads();
  part[kq][col] = acc;
  __syncthreads();
  if (tid < NB) w1p[tid] = w[j1 + tid] - ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]));
  __syncthreads();
  // z1 = L11^-T w1'
  s = 0.0;
#pragma unroll
  for (int k = 0; k < 16; k++) s += lv1[k] * w1p[16 * kq + k];
  __syncthreads();
  part[kq][col] = s;
  __syncthreads();
  if (tid < NB) {
    const double zz = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    z1[tid] = zz;
    if (bl == 0) z[j1 + tid] = zz;
  }
  __syncthreads();
  if (ni == 0) return;
  // w_i -= L_{jb,i}^T z2, then w_i -= L_{jb-1,i}^T z1
  acc = 0.0;
#pragma unroll
  for (int k = 0; k < 16; k++) acc += mv2[k] * z2[16 * kq + k];
  __syncthreads();
  part[kq][col] = acc;
  __syncthreads();
  double wi = 0.0;
  if (tid < NB) wi = w[i0 + tid] - ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]));
  acc = 0.0;
#pragma unroll
  for (int k = 0; k < 16; k++) acc += mv1[k] * z1[16 * kq + k];
  __syncthreads();
  part[kq][col] = acc;
  __syncthreads();
  if (tid < NB) w[i0 + tid] = wi - ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]));
}

// ---------------------------------------------------------------------------------------
// The whole back substitution as ONE launch: one workgroup per 64-column block, workgroup b takes block i = nblk - 1 - b.
// It keeps w_i in LDS, walks the blocks j > i that couple to it in descending order (the later blocks of its own tree
// node, its ancestors' blocks, the root chain), takes z_j as soon as the workgroup of block j has published it, subtracts
// L_ji^T z_j and finally publishes z_i = Linv_i^T w_i.  The arithmetic and its order are those of k_backsolve_step /
// k_backsolve_pair (bit-identical result); what goes is the chain of twelve dependent launches (config 3: 85 -> 40 us).
//
// Hand-off: z is filled with MSFM_Z_PENDING (a NaN payload no arithmetic produces) by k_trinv64_full; a block's 64 values are
// published by ONE wave with 8-byte agent-scope (sc1) stores and consumed by agent-scope loads polled per value - every
// double is its own {data, tag} granule, so no flag, fence or ordering between the values is needed.  A workgroup only ever
// waits for blocks with a HIGHER index, i.e. workgroups with a lower blockIdx; the launch is used only while all of them fit
// on the chip together (host: nblk <= MSFM_BACKSOLVE_CHAIN_MAX), and every poll loop is bounded: on expiry the failure word
// gets MSFM_FAIL_SYNC (the host returns MSFM_E_DEVICE) and the value is taken as 0 so that every workgroup still drains.
// ---------------------------------------------------------------------------------------
#define MSFM_BACKSOLVE_CHAIN_MAX 224   // upper bound; the limit in force is what the device can hold at once (resident_workgroups)
struct BackTree {
  int n_levels;
  int root_blk;                       // first block of the root chain
  struct { int K; struct { int b0, b1, leaf_lo, leaf_hi; } node[8]; } level[3];   // block ranges of the nodes
};
__global__ __launch_bounds__(256) void k_backsolve_chain(const double* __restrict__ M, int ld, int n, int nblk, BackTree tree,
                                                          const double* __restrict__ Ldiag, const double* __restrict__ Dinv,
                                                          double* __restrict__ z, unsigned long long* __restrict__ z_next,
                                                          int* __restrict__ fail, unsigned spin_limit) {
  __shared__ double part[4][NB];
  __shared__ double zj[NB], wi[NB];
  __shared__ int rlo[5], rhi[5], nr;
  __shared__ double Lb[NB * LDT], Vb[NB * LDT], Tb[NB * LDT];   // the block's own 64 x 64 inverse is formed here (Vb), not by a launch before
  const int tid = threadIdx.x, col = tid & 63, kq = tid >> 6;
  const int i = nblk - 1 - (int)blockIdx.x, i0 = i * NB;
  if (tid == 0) {
    // ranges of coupled later blocks, in the order they are walked (descending block index): root chain, ancestors from
    // the shallowest to the deepest, then the rest of the own node
    int c = 0;
    if (tree.n_levels == 0 || i >= tree.root_blk) {
      rlo[c] = i + 1; rhi[c] = nblk; c++;
    } else {
      int lv = 0, k = 0;
      for (int l = 0; l < tree.n_levels; l++)
        for (int q = 0; q < tree.level[l].K; q++)
          if (i >= tree.level[l].node[q].b0 && i < tree.level[l].node[q].b1) { lv = l; k = q; }
      const int lo = tree.level[lv].node[k].leaf_lo, hi = tree.level[lv].node[k].leaf_hi;
      rlo[c] = tree.root_blk; rhi[c] = nblk; c++;
      for (int h = tree.n_levels - 1; h > lv; h--)
        for (int q = 0; q < tree.level[h].K; q++)
          if (tree.level[h].node[q].leaf_lo <= lo && tree.level[h].node[q].leaf_hi >= hi) { rlo[c] = tree.level[h].node[q].b0; rhi[c] = tree.level[h].node[q].b1; c++; }
      rlo[c] = i + 1; rhi[c] = tree.level[lv].node[k].b1; c++;
    }
    nr = c;
  }
  if (tid < NB) {
    wi[tid] = (i0 + tid < n) ? M[(size_t)n * ld + i0 + tid] : 0.0;   // the eliminated right-hand side: row n of the factor
    if (z_next) z_next[i0 + tid] = MSFM_Z_PENDING;                     // the other solution buffer is ready for the next solve
  }
  __syncthreads();
  auto fetch = [&](int j, double (&mv)[16]) {
    const int j0 = j * NB;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int r = j0 + 16 * kq + k;
      mv[k] = r < n ? M[(size_t)r * ld + i0 + col] : 0.0;
    }
  };
  // flat walk over the ranges with the next block's tile in flight while the current block's z is awaited
  int ri = 0, j = -1;
  auto advance = [&]() {   // next (range, block) in walking order, j = -1 when done
    while (ri < nr) {
      if (j < 0) j = rhi[ri] - 1; else j--;
      if (j >= rlo[ri]) return;
      ri++; j = -1;
    }
    j = -1;
  };
  advance();
  double mv[16], mvn[16];
  if (j >= 0) fetch(j, mv);
  // the own block's inverse while the first tile is in flight (and, for all but the last block, while z of the later
  // blocks is still being produced): what k_trinv64_full did in a launch of its own on the critical path
  trinv64_block(Ldiag, Dinv, i, n, Lb, Vb, Tb);
  while (j >= 0) {
    const int jc = j;
    advance();
    if (j >= 0) fetch(j, mvn);
    if (tid < NB) {
      // one wave, one value per lane
      const unsigned long long* src = reinterpret_cast<const unsigned long long*>(z) + (size_t)jc * NB + tid;
      unsigned long long v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      while (v == MSFM_Z_PENDING) {
        if (++spins > spin_limit) { atomicOr(fail, MSFM_FAIL_SYNC); v = 0ull; break; }
        __builtin_amdgcn_s_sleep(2);
        v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      zj[tid] = __longlong_as_double((long long)v);
    }
    __syncthreads();
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k++) acc += mv[k] * zj[16 * kq + k];
    part[kq][col] = acc;
    __syncthreads();
    if (tid < NB) wi[tid] -= (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; k++) mv[k] = mvn[k];
  }
  // z_i = Linv_i^T w_i
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 16; k++) s += Vb[(16 * kq + k) * LDT + col] * wi[16 * kq + k];
  part[kq][col] = s;
  __syncthreads();
  if (tid < NB) {
    const double zz = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    unsigned long long bits = (unsigned long long)__double_as_longlong(zz);
    if (bits == MSFM_Z_PENDING) bits = 0x7FF8000000000000ull;   // (cannot come out of arithmetic; keep the protocol safe anyway)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(z) + (size_t)i0 + tid, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Host driver.  M: npad x npad, row n = rhs.  On return z[0..n) solves S z = rhs.
// `fail` (device int) is OR-ed with 1 when S is not positive definite.
// work: npad*16 doubles (16x16 inverses) + npad*64 (full block inverses) + npad*64 (diagonal blocks of L).
// plan (optional): K uncoupled domains whose panel chains advance together, one job each per launch.
// One tile per bulk workgroup: the hardware dispatcher balances them over the CUs the column-0 workgroups leave free
// (4 tiles per workgroup with register prefetch of the next tile measured slower: coarser quantisation, same time per tile).
static int bulk_workgroups(int ntile) { return ntile; }

static PanelJob make_job(int j0, int t0, int a0, int nA64, int b0, int nrows_b /*data rows in B*/, double* corner, int ldc) {
  PanelJob jb;
  jb.j0 = j0; jb.t0 = t0; jb.a0 = a0; jb.na16 = 4 * nA64; jb.b0 = b0; jb.nb16 = cdiv(std::max(0, nrows_b), 16);
  jb.nseg = 0;
  for (int g = 0; g < 4; g++) { jb.sb0[g] = 0; jb.sn16[g] = 0; }
  jb.nrt = jb.na16 + jb.nb16;
  const int nt = nA64 + cdiv(jb.nb16, 4);
  if (t0 >= 0) {
    jb.ncw = std::max(1, cdiv(jb.nrt - 4, 3));  // column-0 workgroups: three 16-row tiles each
    jb.ntile = j0 >= 0 ? nt * (nt - 1) / 2 : 0;
  } else {
    jb.ncw = 0;
    jb.ntile = nt * (nt + 1) / 2;
  }
  jb.nwg = jb.ncw + bulk_workgroups(jb.ntile);
  jb.wg0 = 0; jb.corner = corner; jb.ldc = ldc; jb.defer_corner = 0;
  return jb;
}

// Workgroups of `kernel` (256 threads, static LDS only) that the device of `ctx` keeps resident at the same time: a kernel
// whose workgroups wait for each other inside one launch is used only up to this many (asked once per device and kernel).
// k_backsolve_chain's waits only ever go to workgroups with a LOWER blockIdx, so with every workgroup resident no dispatch
// order can starve them.  k_chain is different: its row owners (the first blocks of the grid) also wait for tiles that BULK
// workgroups - higher blockIdx - publish, so it needs every row owner AND at least one bulk workgroup resident; that is what
// the `fits` rule of chain_build leaves room for.  Every poll is bounded besides (MSFM_FAIL_SYNC -> MSFM_E_DEVICE).
template <class K>
static int resident_workgroups(msfm_ctx* ctx, K kernel, int threads) {
  int per_cu = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess) return 0;
  return std::max(0, per_cu) * std::max(0, cus);
}

__global__ __launch_bounds__(256) void k_fill_pending(int n, unsigned long long* __restrict__ z) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) z[i] = MSFM_Z_PENDING;
}
int msfm_chol_fill_pending(msfm_ctx* ctx, double* z, int npad) {
  hipLaunchKernelGGL(k_fill_pending, dim3(cdiv(npad, 256)), dim3(256), 0, ctx->stream, npad, reinterpret_cast<unsigned long long*>(z));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return msfm_set_error(ctx, MSFM_E_DEVICE, "fill: %s", hipGetErrorString(e));
  return MSFM_OK;
}

// ---- host side of the persistent chain ----
struct ChainLaunch {
  ChainJobs jobs;
  int task_off = 0, n_tasks = 0;   // the launch's tiles in ws->tasks
  int steps = 0;        // panel steps on its critical path (for the timers)
  bool usable = false;
  bool corner_folded = false;   // the level's corner update runs as tasks of the launch
  int corner_b0 = 0;
};
// The deferred corner update of a tree level: per 64-row block of the B square (rows from b0 on) the panels of the level that lie
// under the block's node, and into how many pieces a tile's panels are cut (k_corner_syrk / the corner tasks of k_chain).
struct CornerPlan { CornerRanges R; int nB64 = 0, target = 1, smax = 1, sb = 0; };
static void corner_plan(const msfm_chol_plan* plan, int lv, int nrows, CornerPlan& C) {
  const int n_levels = plan->n_levels;
  const msfm_chol_level& L = plan->level[lv];
  const int K = L.K, sb = L.b0;
  C.sb = sb;
  C.nB64 = cdiv(nrows - sb, 64);
  const int nB64 = std::min(C.nB64, MSFM_CORNER_MAX_BLOCKS);
  CornerRanges& R = C.R;
  for (int I = 0; I < nB64; I++) {
    const int row = sb + 64 * I;
    int lo = 0, hi = 0x7fffffff;   // leaf interval of the block's node; the root covers everything
    for (int h = lv + 1; h < n_levels; h++)
      for (int q = 0; q < plan->level[h].K; q++)
        if (row >= plan->level[h].node[q].begin && row < plan->level[h].node[q].end) { lo = plan->level[h].node[q].leaf_lo; hi = plan->level[h].node[q].leaf_hi; }
    int plo = 0, phi = 0;
    bool any = false;
    for (int q = 0; q < K; q++)
      if (L.node[q].leaf_lo >= lo && L.node[q].leaf_hi <= hi) { if (!any) plo = L.node[q].begin / NB; phi = L.node[q].end / NB; any = true; }
    R.plo[I] = (short)plo; R.phi[I] = (short)phi;
  }
  // panels per piece: the smallest count with which the pieces cover the chip about once
  int target = 1, smax = 1;
  for (;; target++) {
    long wgs = 0;
    smax = 1;
    for (int I = 0; I < nB64; I++)
      for (int J = 0; J <= I; J++) {
        const int sp = corner_splits(std::min((int)R.phi[I], (int)R.phi[J]) - std::max((int)R.plo[I], (int)R.plo[J]), target);
        wgs += sp;
        smax = std::max(smax, sp);
      }
    if (wgs <= 560 || target >= 64) break;
  }
  static const int target_env = getenv("MSFM_CORNER_TARGET") ? atoi(getenv("MSFM_CORNER_TARGET")) : 0;
  if (target_env > 0) {
    target = target_env; smax = 1;
    for (int I = 0; I < nB64; I++)
      for (int J = 0; J <= I; J++)
        smax = std::max(smax, corner_splits(std::min((int)R.phi[I], (int)R.phi[J]) - std::max((int)R.plo[I], (int)R.plo[J]), target));
  }
  C.target = target; C.smax = smax;
}

struct msfm_chol_ws {
  msfm_ctx* ctx = nullptr;
  int npad = 0;
  DevBuf<unsigned> flags;     // rowflag [8][npad / 16], then tileflag [(npad / 64)^2]
  DevBuf<double> hb;          // two hand-off buffers of npad x 64
  DevBuf<int> tickets;        // [8]
  DevBuf<ChainTask> tasks;
  unsigned epoch = 0, n_launch = 0, flag_epoch = 0;   // solves (hand-off buffer parity), launches (ticket slot), launches (counter tag)
  bool dirty = true;          // the hand-off buffers and tickets must be (re)initialised before the next use
  // launches of the plan they were built for (levels in order, then the root chain)
  std::vector<ChainLaunch> launch;
  unsigned long long sig = 0;
  int capacity = 0;           // resident workgroups of k_chain on the context's device
};

int msfm_chol_ws_create(msfm_ctx* ctx, int npad, msfm_chol_ws** out) {
  if (!ctx || !out || npad < NB || npad % NB) return MSFM_E_INVAL;
  std::unique_ptr<msfm_chol_ws> w(new msfm_chol_ws());
  w->ctx = ctx; w->npad = npad;
  const size_t nt16 = npad / 16, nb = npad / NB;
  HIP_TRY(ctx, w->flags.alloc(8 * nt16 + nb * nb + 8 * (nb * (nb + 1) / 2)));   // ... then cornerflag [tiles of the B square][8]
  HIP_TRY(ctx, w->hb.alloc(2 * (size_t)npad * NB));
  HIP_TRY(ctx, w->tickets.alloc(16));   // eight ticket counters, then the eight words of the give-up note
  HIP_TRY(ctx, hipMemsetAsync(w->tickets.p, 0, sizeof(int) * 16, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(w->flags.p, 0, sizeof(unsigned) * w->flags.n, ctx->stream));
  w->capacity = resident_workgroups(ctx, k_chain, 256);
  *out = w.release();
  return MSFM_OK;
}
void msfm_chol_ws_destroy(msfm_chol_ws* w) { delete w; }

// The launches of a plan: one per tree level (its K node chains side by side), then the root chain.
static int chain_build(msfm_chol_ws* ws, int n, const msfm_chol_plan* plan) {
  msfm_ctx* ctx = ws->ctx;
  const int nrows = n + 1, n_levels = plan ? plan->n_levels : 0;
  unsigned long long sig = 1469598103934665603ull;
  auto mix = [&](long v) { sig = (sig ^ (unsigned long long)v) * 1099511628211ull; };
  mix(n); mix(n_levels);
  for (int lv = 0; lv < n_levels; lv++) {
    mix(plan->level[lv].K); mix(plan->level[lv].b0);
    for (int k = 0; k < plan->level[lv].K; k++) { mix(plan->level[lv].node[k].begin); mix(plan->level[lv].node[k].end); mix(plan->level[lv].node[k].leaf_lo); mix(plan->level[lv].node[k].leaf_hi); }
  }
  if (sig == ws->sig && !ws->launch.empty()) return MSFM_OK;
  ws->launch.clear();
  std::vector<ChainTask> table;
  const int root_begin = n_levels ? plan->level[n_levels - 1].b0 : 0;
  const int nt16 = ws->npad / 16;
  for (int lv = 0; lv <= n_levels; lv++) {
    ChainLaunch L;
    ChainJobs& J = L.jobs;
    memset(&J, 0, sizeof J);
    int wg = 0, maxp = 0;
    if (lv < n_levels) {
      const msfm_chol_level& PL = plan->level[lv];
      for (int k = 0; k < PL.K; k++) {
        const msfm_chol_node& nd = PL.node[k];
        const int P = (nd.end - nd.begin) / NB;
        if (P <= 0) continue;
        ChainJob& jb = J.job[J.count];
        jb.begin = nd.begin; jb.P = P; jb.nA64 = P;
        jb.nseg = 0;
        int rows16 = 0;
        for (int h = lv + 1; h < n_levels; h++)
          for (int q = 0; q < plan->level[h].K; q++) {
            const msfm_chol_node& a = plan->level[h].node[q];
            if (a.leaf_lo <= nd.leaf_lo && a.leaf_hi >= nd.leaf_hi) { jb.sb0[jb.nseg] = a.begin; jb.sn16[jb.nseg] = (a.end - a.begin) / 16; rows16 += jb.sn16[jb.nseg]; jb.nseg++; }
          }
        jb.sb0[jb.nseg] = root_begin; jb.sn16[jb.nseg] = cdiv(nrows - root_begin, 16); rows16 += jb.sn16[jb.nseg]; jb.nseg++;
        jb.nb16 = rows16; jb.nB64 = cdiv(rows16, 4);
        jb.nrt = 4 * P + rows16;
        jb.ncw = std::max(1, cdiv(jb.nrt - 4, 3));
        jb.wg0 = wg; wg += jb.ncw;
        jb.flag0 = J.count * nt16;
        if (jb.nrt > nt16) return msfm_set_error(ctx, MSFM_E_INVAL, "cholesky: chain job larger than the flag table");
        maxp = std::max(maxp, P);
        J.count++;
      }
    } else {
      const int t_first = root_begin;
      if (t_first < n) {
        ChainJob& jb = J.job[0];
        jb.begin = t_first; jb.P = cdiv(n - t_first, NB); jb.nA64 = cdiv(nrows - t_first, NB);
        jb.nseg = 0; jb.nb16 = 0; jb.nB64 = 0;
        jb.nrt = cdiv(nrows - t_first, 16);
        jb.ncw = std::max(1, cdiv(jb.nrt - 4, 3));
        jb.wg0 = 0; wg = jb.ncw; jb.flag0 = 0;
        maxp = jb.P;
        J.count = 1;
      }
    }
    J.n_row_wg = wg;
    J.n_steps = std::max(0, maxp - 1);
    L.steps = maxp;
    L.task_off = (int)table.size();
    int max_step_tasks = 0;
    {
      // every tile of every launch step, in priority order: key = l + 0.6 (J - 1) (see chain_bulk)
      struct Keyed { int key; ChainTask t; };
      std::vector<Keyed> lt;
      static const int slope = getenv("MSFM_CHAIN_SLOPE") ? std::max(0, std::min(9, atoi(getenv("MSFM_CHAIN_SLOPE")))) : 6;   // tenths; 0: step by step
      for (int l = 1; l <= J.n_steps; l++) {
        int step_tasks = 0;
        for (int k = 0; k < J.count; k++) {
          if (l >= J.job[k].P) continue;
          const int nA = J.job[k].nA64 - l, nB = J.job[k].nB64;
          for (int Jc = 1; Jc < nA; Jc++)
            for (int I = Jc; I < nA + nB; I++) {
              lt.push_back(Keyed{10 * l + slope * (Jc - 1), ChainTask{0, (short)k, (short)l, (short)(4 * (I + l)), (short)(4 * (Jc + l)), (short)I, (short)Jc, 0, 0,
                                                                           (short)(Jc == 1 ? 1 : 0), 0, 0}});
              step_tasks++;
            }
        }
        max_step_tasks = std::max(max_step_tasks, step_tasks);
      }
      // the level's corner update as tasks of the same launch (instead of k_corner_syrk / k_merge_corners behind it):
      // for every tile of the B square and every piece of its panel list one task per panel, in the panel order of k_corner_syrk
      // (the pieces' sums are then bit for bit those of the launches); a piece's tasks get non-decreasing keys, so the list stays
      // a linear extension of the dependencies; the merges come last.
      // Measured at config 3's tree (scripts/chol_probe 3137 . plan): 0.535 ms folded against 0.456 ms with the two launches - the
      // 3 140 per-panel corner tasks (each reads and writes its 32 KB piece: k_corner_syrk keeps it in registers over a piece's
      // panels) more than double the bulk work and sit in front of the next steps' urgent tiles.  So the launches stay the
      // default; MSFM_CORNER_FOLD=1 takes this path (bit-identical factor, checked by the probe).
      static const bool corner_launches = getenv("MSFM_CORNER_FOLD") == nullptr;
      std::vector<Keyed> merges;
      if (lv < n_levels && !corner_launches && plan->corners && J.count > 0) {
        CornerPlan CP;
        corner_plan(plan, lv, nrows, CP);
        if (CP.nB64 <= MSFM_CORNER_MAX_BLOCKS && CP.nB64 <= 180) {
          L.corner_folded = true;
          L.corner_b0 = CP.sb;
          auto job_of_panel = [&](int p, int& k, int& l) {
            for (k = 0; k < J.count; k++)
              if (64 * p >= J.job[k].begin && 64 * p < J.job[k].begin + 64 * J.job[k].P) { l = p - J.job[k].begin / 64; return true; }
            return false;
          };
          auto row_tile = [&](const ChainJob& jb, int row) {   // the job's tile index of a row of range B (-1: not among its rows)
            int t = 4 * jb.nA64;
            for (int g = 0; g < jb.nseg; g++) {
              if (row >= jb.sb0[g] && row < jb.sb0[g] + 16 * jb.sn16[g]) return t + (row - jb.sb0[g]) / 16;
              t += jb.sn16[g];
            }
            return -1;
          };
          for (int I = 0; I < CP.nB64 && L.corner_folded; I++)
            for (int Jc = 0; Jc <= I && L.corner_folded; Jc++) {
              const int p_begin = std::max((int)CP.R.plo[I], (int)CP.R.plo[Jc]), np = std::min((int)CP.R.phi[I], (int)CP.R.phi[Jc]);
              const int len = np - p_begin;
              if (len <= 0) continue;
              const int ns = corner_splits(len, CP.target);
              for (int r = 0; r < ns; r++) {
                int key = 0, seq = 0;
                for (int pp = p_begin + r; pp < np; pp += ns, seq++) {
                  int k = 0, l = 0;
                  if (!job_of_panel(pp, k, l)) { L.corner_folded = false; break; }
                  const int ti = row_tile(J.job[k], CP.sb + 64 * I), tj = row_tile(J.job[k], CP.sb + 64 * Jc);
                  if (ti < 0 || tj < 0) { L.corner_folded = false; break; }
                  key = std::max(key, 10 * (l + 1) + 14);
                  lt.push_back(Keyed{key, ChainTask{1, (short)k, (short)l, (short)ti, (short)tj, (short)I, (short)Jc, (short)r, (short)seq, 0, 0, 0}});
                }
              }
              merges.push_back(Keyed{0x7fffffff, ChainTask{2, 0, (short)len, 0, 0, (short)I, (short)Jc, (short)ns, 0, 0, 0, 0}});
            }
          if (!L.corner_folded) {   // (cannot happen with the trees choose_dissection builds; keep the launches then)
            lt.erase(std::remove_if(lt.begin(), lt.end(), [](const Keyed& q) { return q.t.type != 0; }), lt.end());
            merges.clear();
          }
        }
      }
      std::stable_sort(lt.begin(), lt.end(), [](const Keyed& a, const Keyed& b) { return a.key < b.key; });
      for (const Keyed& q : merges) lt.push_back(q);
      for (const Keyed& q : lt) table.push_back(q.t);
      L.n_tasks = (int)lt.size();
    }
    // bulk workgroups: what the device holds beside the row owners, at most one per tile of the busiest step
    const int room = ws->capacity - wg;
    J.n_bulk_wg = std::max(1, std::min(room, max_step_tasks));
    // usable: every row owner resident with room to spare (other processes may share the device), a bulk workgroup for
    // every three tiles of the busiest step at least
    static const bool force = getenv("MSFM_CHAIN_FORCE") != nullptr;   // (probes: take the chain whatever the tile count)
    // (a device shared with other contexts that launch the same kernel at the same time - ranks are synchronised by the
    //  reduction in front of the factorisation - must hold the row owners of all of them, or none makes progress)
    const int share = std::max(1, ctx->device_share);
    // One context per device: the row owners may take just under half of the resident workgroups.  Two independent PROCESSES
    // that share a GPU without MSFM_DEVICE_SHARE set each see share == 1; with exactly half each, their row owners together
    // could fill the device and leave no bulk workgroup of either resident - both would spin to the poll limit.  Two short of
    // half, one bulk workgroup of one of the two launches always finds a slot, that launch drains (whoever holds the lowest
    // open ticket can finish), and the other follows.  (A quarter, as the round-4 review suggested, would send systems of more
    // than ~170 blocks - config 5 in full - back to the launch chain for a case the bounded polls already turn into an error.)
    const bool fits = share == 1 ? wg <= ws->capacity / 2 - 2 : (long)wg * share <= 3L * ws->capacity / 4;
    L.usable = J.count > 0 && fits && (max_step_tasks == 0 || 3 * room >= max_step_tasks || force) && maxp < 4000;
    ws->launch.push_back(L);
  }
  HIP_TRY(ctx, ws->tasks.alloc(std::max<size_t>(1, table.size())));
  HIP_TRY(ctx, hipMemcpyAsync(ws->tasks.p, table.data(), sizeof(ChainTask) * table.size(), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // (the host table goes away on return; built once per plan)
  ws->sig = sig;
  ws->dirty = true;   // a new plan: hand-off blocks of the old layout must not be read as published values
  return MSFM_OK;
}

static unsigned chain_spin_limit() {
  static const unsigned v = [] {
    const char* e = getenv("MSFM_SYNC_TIMEOUT_S");
    const double s = e ? atof(e) : 120.0;
    return (unsigned)std::min(4.0e9, std::max(1.0e4, (s > 0 ? s : 120.0) * 3.0e6));
  }();
  return v;
}

static int chain_launch(msfm_chol_ws* ws, const ChainLaunch& L, double* M, int npad, int n, double* Dinv, double* Ldiag, int* fail, double* corners,
                        int ldc) {
  msfm_ctx* ctx = ws->ctx;
  ChainCtl c;
  const size_t nt16 = npad / 16, nb = npad / NB;
  c.rowflag = ws->flags.p;
  c.tileflag = ws->flags.p + 8 * nt16;
  const unsigned par = ws->epoch & 1u;
  c.hb = ws->hb.p + (size_t)par * npad * NB;
  c.hb_next = reinterpret_cast<unsigned long long*>(ws->hb.p + (size_t)(par ^ 1u) * npad * NB);
  const unsigned slot = ws->n_launch++ & 7u;
  c.ticket = ws->tickets.p + slot;
  c.ticket_next = ws->tickets.p + ((slot + 1) & 7u);
  c.tasks = ws->tasks.p + L.task_off;
  c.n_tasks = L.n_tasks;
  c.base = (++ws->flag_epoch & 0xFFFFFu) << 12;   // per LAUNCH: the launches of one solve reuse the rowflag slots of their jobs
  c.spin_limit = chain_spin_limit();
  c.nblk = (int)nb;
  c.corners = L.corner_folded ? corners : nullptr;
  c.cornerflag = ws->flags.p + 8 * nt16 + nb * nb;
  c.ldc = ldc; c.corner_b0 = L.corner_b0;
  c.dbg = ws->tickets.p + 8;
#ifdef MSFM_CHAIN_BARCHECK
  if (getenv("MSFM_CHAIN_TRACE")) fprintf(stderr, "k_chain: n %d, %d row owners, %d bulk workgroups, %d steps\n", n, L.jobs.n_row_wg, L.jobs.n_bulk_wg, L.steps);
#endif
  hipLaunchKernelGGL(k_chain, dim3(L.jobs.n_row_wg + L.jobs.n_bulk_wg), dim3(256), 0, ctx->stream, M, npad, n, Dinv, Ldiag, fail, L.jobs, c);
  return MSFM_OK;
}

// z_next (optional): a second solution buffer of npad doubles.  With it the caller promises that `z` already holds
// MSFM_Z_PENDING everywhere (msfm_chol_fill_pending once, afterwards the previous call's z_next) and gets z_next back in that
// state - the two buffers alternate from solve to solve and no fill launch sits on the critical path.  Without it the
// function fills z itself first.
int msfm_chol_factor_solve(msfm_ctx* ctx, double* M, int npad, int n, double* work, double* w, double* z, int* fail,
                           const msfm_chol_plan* plan, double* z_next, msfm_chol_ws* ws) {
  if (!M || !work || !w || !z || !fail || npad % NB != 0 || n < 1 || n + 1 > npad)
    return msfm_set_error(ctx, MSFM_E_INVAL, "cholesky: bad workspace (null buffer or size)");
  hipStream_t s = ctx->stream;
  const int nrows = n + 1;  // rows that carry data (S plus the rhs row)
  double* Dinv = work;
  double* Linv = work + (size_t)npad * 16;
  double* Ldiag = work + (size_t)npad * 80;
  int t_first = 0;
  const int n_levels = plan ? plan->n_levels : 0;
  // the persistent chain (one launch per tree level) when the caller keeps a workspace for it; MSFM_CHOL_LAUNCHES=1: the
  // round-3 chain of one launch per 64-column panel, for comparison
  const bool chol_launches_env = getenv("MSFM_CHOL_LAUNCHES") != nullptr;   // (read per call: the tests switch it between solves)
  bool use_chain = ws && !chol_launches_env && ws->npad == npad && ws->capacity > 0 && (size_t)npad * npad * sizeof(double) < 0xFFFFFFFFull;   // (32-bit buffer offsets)
  if (use_chain) {
    MSFM_TRY(chain_build(ws, n, plan));
    if ((int)ws->launch.size() != n_levels + 1) use_chain = false;
  }
  if (use_chain) {
    ws->epoch++;
    if (ws->dirty) {
      MSFM_TRY(msfm_chol_fill_pending(ctx, ws->hb.p, 2 * npad * NB));
      HIP_TRY(ctx, hipMemsetAsync(ws->tickets.p, 0, sizeof(int) * 8, s));
      ws->n_launch = 0;
      ws->dirty = false;
    }
  }
  if (n_levels > 0) {
    if (n_levels > 3 || !plan->corners) return msfm_set_error(ctx, MSFM_E_INVAL, "cholesky: bad plan");
    // every level is checked before the first launch: a refusal must not leave half a factorisation behind
    for (int lv = 0; lv < n_levels; lv++) {
      const msfm_chol_level& L = plan->level[lv];
      if (L.K < 1 || L.K > 8 || L.b0 % NB || L.begin % NB || plan->ldc < 64 * cdiv(nrows - L.b0, 64))
        return msfm_set_error(ctx, MSFM_E_INVAL, "cholesky: bad plan level");
      if (cdiv(nrows - L.b0, 64) > MSFM_CORNER_MAX_BLOCKS)
        return msfm_set_error(ctx, MSFM_E_INVAL, "cholesky: separator part too large for the corner update (%d blocks)", cdiv(nrows - L.b0, 64));
    }
    for (int lv = 0; lv < n_levels; lv++) {
      // ---- the K chains of this level, step by step: step l of node k factors its block l (after applying its panel
      //      l-1); the step after a node's last block only applies that last panel to the rows from b0 on ----
      const msfm_chol_level& L = plan->level[lv];
      const int K = L.K, sb = L.b0, ldc = plan->ldc;
      if (K < 1 || K > 8 || sb % NB || L.begin % NB || ldc < 64 * cdiv(nrows - sb, 64)) return msfm_set_error(ctx, MSFM_E_INVAL, "cholesky: bad plan level");
      const int root_begin = plan->level[n_levels - 1].b0;
      // rows a node's panels reach below the node itself: its ancestors (one node per higher level), then the root
      // chain with the rhs row - everything else from b0 on is structurally zero in its columns
      auto ancestors = [&](const msfm_chol_node& nd, PanelJob& jb) {
        jb.nseg = 0;
        int rows = 0;
        for (int h = lv + 1; h < n_levels; h++)
          for (int q = 0; q < plan->level[h].K; q++) {
            const msfm_chol_node& a = plan->level[h].node[q];
            if (a.leaf_lo <= nd.leaf_lo && a.leaf_hi >= nd.leaf_hi) { jb.sb0[jb.nseg] = a.begin; jb.sn16[jb.nseg] = (a.end - a.begin) / 16; jb.nseg++; rows += a.end - a.begin; }
          }
        jb.sb0[jb.nseg] = root_begin; jb.sn16[jb.nseg] = cdiv(nrows - root_begin, 16); jb.nseg++;
        rows += nrows - root_begin;
        return rows;
      };
      int maxp = 0;
      for (int k = 0; k < K; k++) maxp = std::max(maxp, (L.node[k].end - L.node[k].begin) / NB);
      KTimer chain_timer(ctx, "chol_panel_mfma");   // the level's chain of panel launches as a whole
      chain_timer.count = 0;
      const bool lv_chain = use_chain && ws->launch[lv].usable;
      if (lv_chain) {
        MSFM_TRY(chain_launch(ws, ws->launch[lv], M, npad, n, Dinv, Ldiag, fail, plan->corners, ldc));
        chain_timer.count = ws->launch[lv].steps;   // (counted in panel steps, so that a step's time compares with the launch chain's)
      }
      for (int l = 0; l < maxp && !lv_chain; l++) {
        PanelJobs jobs;
        jobs.count = 0;
        int wg = 0;
        for (int k = 0; k < K; k++) {
          const int P = (L.node[k].end - L.node[k].begin) / NB;
          if (l >= P) continue;
          const int t0 = L.node[k].begin + NB * l;
          const int j0 = l > 0 ? L.node[k].begin + NB * (l - 1) : -1;
          PanelJob seg;
          const int brows = ancestors(L.node[k], seg);
          PanelJob jb = make_job(j0, t0, t0, P - l, sb, brows, nullptr, 0);
          jb.nseg = seg.nseg;
          for (int g = 0; g < 4; g++) { jb.sb0[g] = seg.sb0[g]; jb.sn16[g] = seg.sn16[g]; }
          // the tiles among the rows from b0 on are formed once, after the chains (k_corner_syrk): leave them out here
          const int nA64 = P - l, nB64 = cdiv(jb.nb16, 4);
          jb.defer_corner = 1;
          jb.ntile = j0 >= 0 ? nA64 * (nA64 - 1) / 2 + nB64 * (nA64 - 1) : 0;
          jb.nwg = jb.ncw + bulk_workgroups(jb.ntile);
          jb.wg0 = wg;
          wg += jb.nwg;
          jobs.job[jobs.count++] = jb;
        }
        if (!jobs.count) break;
        hipLaunchKernelGGL(k_panel_v2<true>, dim3(wg), dim3(256), 0, s, M, npad, n, Dinv, Ldiag, fail, jobs);
        chain_timer.count++;
      }
      chain_timer.stop();
      if (maxp > 0 && !(lv_chain && ws->launch[lv].corner_folded)) {
        KTimer t(ctx, "chol_corner_syrk");
        t.count = 2;
        const int nB64 = cdiv(nrows - sb, 64), ntile = nB64 * (nB64 + 1) / 2;
        if (nB64 > MSFM_CORNER_MAX_BLOCKS)   // (choose_dissection never picks such a tree: checked before anything is launched)
          return msfm_set_error(ctx, MSFM_E_INVAL, "cholesky: separator part too large for the corner update (%d blocks)", nB64);
        CornerPlan CP;
        corner_plan(plan, lv, nrows, CP);
        const CornerRanges& R = CP.R;
        const int target = CP.target, smax = CP.smax;
        hipLaunchKernelGGL(k_corner_syrk, dim3(ntile * smax), dim3(256), 0, s, M, npad, sb, target, smax, plan->corners, ldc, R);
        hipLaunchKernelGGL(k_merge_corners, dim3(4 * ntile), dim3(256), 0, s, M, npad, sb, nB64, plan->corners, ldc, target, R);
      }
      t_first = sb;
    }
  }
  KTimer root_timer(ctx, "chol_panel_mfma");
  root_timer.count = 0;
  const bool root_chain = use_chain && ws->launch[n_levels].usable;
  if (root_chain) {
    MSFM_TRY(chain_launch(ws, ws->launch[n_levels], M, npad, n, Dinv, Ldiag, fail, nullptr, 0));
    root_timer.count = ws->launch[n_levels].steps;
  }
  for (int t0 = t_first; t0 < n && !root_chain; t0 += NB) {
    // the separator (or the whole matrix): one job per launch, its first block has nothing left to apply
    const int j0 = t0 > t_first ? t0 - NB : -1;
    PanelJobs jobs;
    jobs.count = 1;
    jobs.job[0] = make_job(j0, t0, t0, cdiv(nrows - t0, 64), 0, 0, nullptr, 0);
    jobs.job[0].nrt = cdiv(nrows - t0, 16);  // 16-row tiles from t0 down that carry data
    jobs.job[0].ncw = std::max(1, cdiv(jobs.job[0].nrt - 4, 3));
    jobs.job[0].nwg = jobs.job[0].ncw + bulk_workgroups(jobs.job[0].ntile);
    // trailing update with panel j0 + potrf / trsm of the panel at t0
    if (n - t0 >= NB) hipLaunchKernelGGL(k_panel_v2<true>, dim3(jobs.job[0].nwg), dim3(256), 0, s, M, npad, n, Dinv, Ldiag, fail, jobs);
    else hipLaunchKernelGGL(k_panel_v2<false>, dim3(jobs.job[0].nwg), dim3(256), 0, s, M, npad, n, Dinv, Ldiag, fail, jobs);
    root_timer.count++;
  }
  root_timer.stop();
  {
    KTimer t(ctx, "chol_backsolve");
    const int nblk = cdiv(n, NB);
    static const bool launches_env = getenv("MSFM_BACKSOLVE_LAUNCHES") != nullptr;   // the round-2 chain of launches, for comparison
    // per device: min(MSFM_BACKSOLVE_CHAIN_MAX, resident workgroups of k_backsolve_chain), 0 = not asked yet (atomic: the
    // per-rank host threads of msfm_multi come through here at the same time; both would store the same value)
    static std::atomic<int> chain_max[64];
    int cmax = chain_max[ctx->device & 63].load(std::memory_order_relaxed);
    if (cmax == 0) {
      cmax = std::max(1, std::min(MSFM_BACKSOLVE_CHAIN_MAX, resident_workgroups(ctx, k_backsolve_chain, 256)));
      chain_max[ctx->device & 63].store(cmax, std::memory_order_relaxed);
    }
    const bool chain = !launches_env && nblk <= cmax;
    if (!chain) hipLaunchKernelGGL(k_trinv64_full, dim3(nblk + cdiv(npad, 256)), dim3(256), 0, s, Ldiag, n, Dinv, Linv, nblk, M, npad, w, npad,
                                   (unsigned long long*)nullptr);
    if (chain) {
      if (!z_next) MSFM_TRY(msfm_chol_fill_pending(ctx, z, npad));
      BackTree bt;
      bt.n_levels = n_levels;
      bt.root_blk = t_first / NB;
      for (int lv = 0; lv < 3; lv++) {
        bt.level[lv].K = lv < n_levels ? plan->level[lv].K : 0;
        for (int k = 0; k < 8; k++) {
          const bool live = lv < n_levels && k < plan->level[lv].K;
          const msfm_chol_node nd = live ? plan->level[lv].node[k] : msfm_chol_node{0, 0, 0, 0};
          bt.level[lv].node[k].b0 = nd.begin / NB; bt.level[lv].node[k].b1 = nd.end / NB;
          bt.level[lv].node[k].leaf_lo = nd.leaf_lo; bt.level[lv].node[k].leaf_hi = nd.leaf_hi;
        }
      }
      // a poll is an L2 round trip plus s_sleep: ~0.3 us; the bound is MSFM_SYNC_TIMEOUT_S (default 120 s, as for the host's own spin)
      static const unsigned spin_limit = [] {
        const char* e = getenv("MSFM_SYNC_TIMEOUT_S");
        const double v = e ? atof(e) : 120.0;
        return (unsigned)std::min(4.0e9, std::max(1.0e4, (v > 0 ? v : 120.0) * 3.0e6));
      }();
      hipLaunchKernelGGL(k_backsolve_chain, dim3(nblk), dim3(256), 0, s, M, npad, n, nblk, bt, Ldiag, Dinv, z,
                         reinterpret_cast<unsigned long long*>(z_next), fail, spin_limit);
    } else if (z_next) {
      MSFM_TRY(msfm_chol_fill_pending(ctx, z_next, npad));   // keep the caller's alternation intact on the launch-chain path
    }
    const int first_dense = t_first / NB;
    for (int jb = nblk - 1; !chain && jb >= first_dense;) {  // root chain (or everything): couples to every block before it
      BackJobs bj;
      bj.count = 1;
      if (jb - 1 >= first_dense) {   // blocks jb and jb-1 together
        bj.job[0] = BackJob{jb, 0, jb - 1, 0};
        hipLaunchKernelGGL(k_backsolve_pair, dim3(jb - 1 > 0 ? jb - 1 : 1), dim3(256), 0, s, M, npad, n, bj, Linv, w, z);
        jb -= 2;
      } else {
        bj.job[0] = BackJob{jb, 0, jb, 0};
        hipLaunchKernelGGL(k_backsolve_step, dim3(jb > 0 ? jb : 1), dim3(256), 0, s, M, npad, n, bj, Linv, w, z);
        jb -= 1;
      }
    }
    for (int lv = n_levels - 1; !chain && lv >= 0; lv--) {
      // Blocks P_k - 1 - 2l and P_k - 2 - 2l of every node of the level in one launch (a node's last odd block alone).  A
      // block couples to the earlier blocks of its own node and to its descendants in the lower levels (contiguous in
      // every level: tree order) - one job per range, all of them recomputing z for themselves; siblings never touch the
      // same rows.
      const msfm_chol_level& L = plan->level[lv];
      int maxp = 0;
      for (int k = 0; k < L.K; k++) maxp = std::max(maxp, (L.node[k].end - L.node[k].begin) / NB);
      for (int l = 0; 2 * l < maxp; l++) {
        BackJobs pj, sj;
        pj.count = sj.count = 0;
        int pwg = 0, swg = 0;
        for (int k = 0; k < L.K; k++) {
          const int ib = L.node[k].begin / NB, P = (L.node[k].end - L.node[k].begin) / NB;
          if (2 * l >= P) continue;
          const int jb = ib + P - 1 - 2 * l;
          const bool pair = jb - 1 >= ib;
          BackJobs& J = pair ? pj : sj;
          int& wg = pair ? pwg : swg;
          const int own = pair ? jb - 1 - ib : jb - ib;   // earlier blocks of the node itself
          J.job[J.count++] = BackJob{jb, ib, own, wg};
          wg += std::max(1, own);
          for (int lo = lv - 1; lo >= 0; lo--) {   // descendants, level by level
            const msfm_chol_level& D = plan->level[lo];
            int d0 = -1, d1 = -1;
            for (int q = 0; q < D.K; q++)
              if (D.node[q].leaf_lo >= L.node[k].leaf_lo && D.node[q].leaf_hi <= L.node[k].leaf_hi) { if (d0 < 0) d0 = q; d1 = q; }
            if (d0 < 0) continue;
            const int rb = D.node[d0].begin / NB, rn = (D.node[d1].end - D.node[d0].begin) / NB;
            if (rn <= 0) continue;
            if (J.count >= 16) return msfm_set_error(ctx, MSFM_E_INVAL, "cholesky: too many back-substitution ranges in one launch");
            J.job[J.count++] = BackJob{jb, rb, rn, wg};
            wg += rn;
          }
        }
        if (pj.count) hipLaunchKernelGGL(k_backsolve_pair, dim3(pwg), dim3(256), 0, s, M, npad, n, pj, Linv, w, z);
        if (sj.count) hipLaunchKernelGGL(k_backsolve_step, dim3(swg), dim3(256), 0, s, M, npad, n, sj, Linv, w, z);
      }
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    if (ws) ws->dirty = true;
    return msfm_set_error(ctx, MSFM_E_DEVICE, "cholesky launch: %s", hipGetErrorString(e));
  }
  return MSFM_OK;
}
