// 16 x 16 tile factorisation of the dense FP64 Cholesky (chol.hip), one wave, two-dimensional lane layout.
//
// Part of the replacement of Eigen's LLT inside Ceres' DENSE_SCHUR solver (ceres::Solve, SfM/src/optimizer.cc:133).
//
// Round 2 factored a 16-column sub-panel with lane = row and the 16 columns of a row in registers (potrf16_v2, kept below
// for the last, partial block): every pivot costs the wave ~33 instructions (the dot-product terms of the next column, three
// v_readlane pairs, seven LDS operations, the reciprocal chain), measured 265-360 cycles per pivot although the dependent
// chain itself is ~40 cycles - the wave is issue-bound (scripts/lat_probe.hip: 4-8 cycles per FP64 instruction, 25 per
// readlane pair, 76 for an LDS round trip).  Here the wave owns ONLY the 16 x 16 diagonal tile plus 16 identity rows (which
// turn into the inverse of the factor under the same column operations) and spreads them over lanes in two dimensions:
//
//   lane = (i, h):  i = lane & 31 = row (0..15 tile rows, 16..31 identity rows),  h = lane >> 5 = column parity
//   register a[r], r = 0..7  =  entry (i, 2 r + h)
//
// Right-looking: pivot c publishes column c (32 values, unscaled) to LDS, every lane reads back its own row's entry and
// the eight entries that belong to its columns, and one FMA per register that still lies right of the pivot applies the
// rank-1 update - about 16 instructions per pivot.  The column that becomes the next pivot is updated and published first;
// the pivot itself travels by v_readlane so that its reciprocal (v_rcp_f64 + one cubic step) runs beside the LDS round
// trip.  The rows of the 64 x 64 block below the tile are no longer carried by this wave: the helper waves form them with
// MFMA triangular solves against the inverse (panel_col0 in chol.hip).
#pragma once
#include <hip/hip_runtime.h>

#ifndef NB
#define NB 64
#endif
#ifndef LDT
#define LDT 66  // LDS row stride in doubles: 132 dwords = 4 mod 64 -> conflict-free ds_read_b64 fragments
#endif
#ifndef DV
#define DV 17   // row stride of the 16x16 inverse blocks in LDS
#endif

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
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

// Factor the 16 x 16 tile (JB, JB) of the 64 x 64 block in Ls (row stride LDT) in place - lower triangle, entries above the
// diagonal are neither read nor meaningful afterwards - and write the inverse of its factor to dinv:
// dinv[(16 JB + r) * DV + l] = (L^-1)[r][l].  dvec: 80 doubles of scratch.  ONE wave, all 64 lanes.
// The published columns live in the never-used upper tiles of the block: column c of the pass at Ls[c * LDT + 16 .. 48).
template <int JB>
__device__ __forceinline__ void potrf16_t(double* Ls, double* dinv, double* dvec, int lane, int* fail) {
  constexpr int c0 = 16 * JB;
  const int i = lane & 31, h = lane >> 5;
  // position of row i inside a published column: tile rows split by parity (even rows 0..7, odd rows 8..15) so that the
  // eight entries of a lane's columns 2 r + h are contiguous; identity rows behind them
  const int pos = i < 16 ? ((i & 1) * 8 + (i >> 1)) : i;
  double* const colb = Ls + 16;       // column c at colb[c * LDT + pos]
  double a[8];
  if (i < 16) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const int j = 2 * r + h;
      const double v = Ls[(c0 + i) * LDT + c0 + j];
      a[r] = j <= i ? v : 0.0;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 8; r++) a[r] = (2 * r + h == i - 16) ? 1.0 : 0.0;
  }
  // Where a lane publishes its entry of an even / odd column: its row's slot when it holds that column, a dump slot (columns
  // 48..63 of the same LDS row, also never used) when the other half does - the stores then need no branch.
  const int dump = 32 + (i & 15);
  const int wpos_even = h == 0 ? pos : dump, wpos_odd = h == 1 ? pos : dump;
  // pivot 0: publish column 0
  colb[wpos_even] = a[0];
  double d = readlane_f64(a[0], 0);
  // Software pipeline: the reads of pivot c + 1 (its own row's entry `my`, the entries `w` of its columns) are issued right
  // behind the store that publishes column c + 1, into a second register set, and the rest of pivot c's rank-1 update runs
  // while they are in flight.  sched_barrier pins that order (the compiler would otherwise reuse the registers of w and
  // push the loads behind the update).
  double my = colb[pos], w[8];
  {
    const d2* wp = reinterpret_cast<const d2*>(&colb[8 * h]);
#pragma unroll
    for (int r2 = 0; r2 < 4; r2++) { const d2 v = wp[r2]; w[2 * r2] = v.x; w[2 * r2 + 1] = v.y; }
  }
#pragma unroll
  for (int c = 0; c < 16; c++) {
    const int hc = c & 1, rc = c >> 1;
    const double inv = rcp3(d);
    dvec[c] = d;   // every lane, same value
    const double ns = -(my * inv);
    double myn = 0.0, wn[8];
    // The column that becomes the next pivot first.  Entry (i, c + 1) lives in register rc of the odd half when c is even
    // (the even half's register rc is column c itself: final, published, never read again - updating it too is harmless)
    // and in register rc + 1 of the even half when c is odd.
    if (c < 15) {
      const int rn = rc + hc;
      a[rn] = fma(ns, w[rn], a[rn]);
      colb[(c + 1) * LDT + (hc == 0 ? wpos_odd : wpos_even)] = a[rn];
      d = readlane_f64(a[rn], hc == 0 ? 32 + c + 1 : c + 1);
      __builtin_amdgcn_sched_barrier(0);
      myn = colb[(c + 1) * LDT + pos];
      const d2* wp = reinterpret_cast<const d2*>(&colb[(c + 1) * LDT + 8 * h]);
      const int rcn = (c + 1) >> 1;
#pragma unroll
      for (int r2 = 0; r2 < 4; r2++) {
        if (2 * r2 + 1 >= rcn) {   // (registers left of the pivot are never touched again)
          const d2 v = wp[r2];
          wn[2 * r2] = v.x; wn[2 * r2 + 1] = v.y;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // the rest of the rank-1 update
#pragma unroll
    for (int r = 0; r < 8; r++) {
      if (r > rc + hc) a[r] = fma(ns, w[r], a[r]);
    }
    __builtin_amdgcn_sched_barrier(0);
    my = myn;
#pragma unroll
    for (int r = 0; r < 8; r++) w[r] = wn[r];
  }
  // 1 / sqrt(d) of the 16 pivots in one vector operation, then back as uniform values
  {
    const double dl = dvec[lane & 15];
    if (!(dl > 0.0)) atomicOr(fail, 1);  // Eigen LLT: info() != Success
    dvec[64 + (lane & 15)] = rsq3(dl);
  }
  // L = U D^-1/2 for the tile rows, (L^-1)^T = (identity rows) D^-1/2; lane (i, h) scales the columns 2 r + h of its row
  {
    const d2* rv = reinterpret_cast<const d2*>(&dvec[64]);
    double out[8];
#pragma unroll
    for (int r2 = 0; r2 < 4; r2++) {
      const d2 rs01 = rv[2 * r2], rs23 = rv[2 * r2 + 1];   // rs[4 r2 .. 4 r2 + 3]
      const int ca = 4 * r2 + h, cb = 4 * r2 + 2 + h;       // this lane's columns 2 r + h for r = 2 r2, 2 r2 + 1
      out[2 * r2] = colb[ca * LDT + pos] * (h ? rs01.y : rs01.x);
      out[2 * r2 + 1] = colb[cb * LDT + pos] * (h ? rs23.y : rs23.x);
    }
    // (every read of the published columns above must be complete before a tile row below the first is overwritten: the
    // columns of pass JB lie in rows 0..15 of the block, columns 16..47 - outside every lower tile - so nothing overlaps)
    if (i < 16) {
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const int j = 2 * r + h;
        if (j <= i) Ls[(c0 + i) * LDT + c0 + j] = out[r];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 8; r++) dinv[(c0 + 2 * r + h) * DV + (i - 16)] = out[r];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same tile factorisation on the matrix pipe.  The tile T (kept SYMMETRIC, both triangles) and J = the transpose of
// the identity rows live in two 16 x 16 accumulator tiles (entry (row lk + 4 i, column lr) in register i of lane
// lr + 16 lk).  Row c of either tile is then register c >> 2 of the sixteen lanes of group lk = c & 3 - which is exactly
// where v_mfma_f64_16x16x4 expects k-slice c & 3 of its A operand (lane (m, k) <-> A[m][k]) and of its B operand (lane
// (n, k) <-> B[k][n]).  With u = row c of T (= column c, by symmetry) masked to that lane group, one MFMA is the whole
// rank-1 update of a pivot:
//     T <- T + u (x) (-u / d)            A = u,        B = -u / d
//     J <- J + (-u / d) (x) J[c][.]      A = -u / d,   B = row c of J
// No LDS, no cross-lane traffic except the two v_readlane pairs that fetch the NEXT pivot's d = T[c+1][c+1] - T[c][c+1]^2 / d
// from the tile before the update lands, so that its reciprocal is ready when the accumulator is.  The matrix pipe is the
// bound: two 64-cycle MFMAs per pivot.  Rows of J that have been passed (and columns of T left of the pivot) pick up
// rounding-level residue; they are extracted at their own pivot, before that happens.
// At the end U[c] = row c of T at its pivot and R[c] = row c of J: L[m][c] = U[c][m] / sqrt(d_c) (m >= c) and
// (L^-1)[c][m] = R[c][m] / sqrt(d_c).
template <int JB>
__device__ __forceinline__ void potrf16_m(double* Ls, double* dinv, double* dvec, int lane, int* fail) {
  constexpr int c0 = 16 * JB;
  const int lr = lane & 15, lk = lane >> 4;
  d4 T, J;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int row = lk + 4 * i;
    // only the lower triangle of the tile in LDS is meaningful: mirror it
    T[i] = row >= lr ? Ls[(c0 + row) * LDT + c0 + lr] : Ls[(c0 + lr) * LDT + c0 + row];
    J[i] = row == lr ? 1.0 : 0.0;
  }
  double U[16], R[16];
  double d = readlane_f64(T[0], 0);   // T[0][0]: register 0 of lane (lr 0, lk 0)
  double ninv = -rcp3(d);
#pragma unroll
  for (int c = 0; c < 16; c++) {
    const int b = c >> 2, g = c & 3;
    const bool mine = lk == g;
    const double uT = mine ? T[b] : 0.0;
    const double uJ = mine ? J[b] : 0.0;
    U[c] = uT; R[c] = uJ;
    dvec[c] = d;   // every lane, same value
    const double Bv = uT * ninv;
    double dn = 1.0;
    if (c < 15) {
      // next pivot from the tile as it stands: t = T[c+1][c+1], x = T[c][c+1]
      const double t = readlane_f64(T[(c + 1) >> 2], 16 * ((c + 1) & 3) + c + 1);
      const double x = readlane_f64(T[b], 16 * g + c + 1);
      dn = fma(x, x * ninv, t);
    }
    T = __builtin_amdgcn_mfma_f64_16x16x4f64(uT, Bv, T, 0, 0, 0);
#ifndef NO_J
    J = __builtin_amdgcn_mfma_f64_16x16x4f64(Bv, uJ, J, 0, 0, 0);
#endif
    if (c < 15) { d = dn; ninv = -rcp3(dn); }
  }
  // 1 / sqrt(d) of the 16 pivots in one vector operation, then back as uniform values
  {
    const double dl = dvec[lane & 15];
    if (!(dl > 0.0)) atomicOr(fail, 1);  // Eigen LLT: info() != Success
    dvec[64 + (lane & 15)] = rsq3(dl);
  }
  {
    // U[c] / R[c] are non-zero in the lanes of group c & 3 only, so a lane's four rows c = lk + 4 q fall out of plain sums
    // (three of the four terms are zero); lane lr is the row index m of L's column / the column index of the inverse's row
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const double uq = (U[4 * q] + U[4 * q + 1]) + (U[4 * q + 2] + U[4 * q + 3]);
      const double rq = (R[4 * q] + R[4 * q + 1]) + (R[4 * q + 2] + R[4 * q + 3]);
      const int c = lk + 4 * q;
      const double rs = dvec[64 + c];
      Ls[(c0 + lr) * LDT + c0 + c] = uq * rs;     // (entries above the diagonal: never read by anyone)
      dinv[(c0 + c) * DV + lr] = rq * rs;
    }
  }
}
