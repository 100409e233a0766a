// Bundle adjustment on MI355X: Levenberg–Marquardt with a dense Schur complement, replacing
// `ceres::Solve(options_, &problem_, &summary_)` at SfM/src/optimizer.cc:133 (problem assembly
// :59-129, options :42-48) and SfM/src/slam_gps.cc:841 (GPS residuals :818-830).
//
// Per linearisation (k_linearize, one thread per observation): residual + closed-form Jacobian
// of the reference's projection model, Huber(1) corrector, Jacobi column scaling; written
// point-major SoA (for the per-point kernels) and camera-major AoS (for the per-camera sums).
// Per linear solve:
//   k_point      one thread per point: V = Jp^T Jp + D^2, 3x3 Cholesky, T = (Jc^T Jp) L^-T per
//                observation, T.u for the right-hand side
//   k_ftf        one wave per chunk of a camera's observations: Jc^T Jc, Jm^T Jc, Jm^T Jm, J^T r
//   k_pairs      one wave per chunk of a (block row, block col) pair list: sum T_a T_b^T — the
//                Schur complement contributions W V^-1 W^T, reduced in a fixed order (no atomics)
//   k_asm_*      assemble S and rhs into the padded dense matrix; chol.hip factors and solves
//   k_backsub    one thread per point: back substitution, candidate point, model cost change
// The LM control flow (step acceptance, radius update, stopping rules) follows Ceres 1.13's
// TrustRegionMinimizer and runs on the host; one small scalar read-back per phase.
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>

#include <rocprim/rocprim.hpp>

#include "ba_device.h"
#include "common.h"


#define PSTRIDE 80   // doubles per FTF partial
#define CHUNK 1024   // entries per reduction chunk (16 per lane)
// k_ftf's chunks are shorter: a camera has a few thousand rows, and one wave per 1024 of them leaves part of the chip without
// a wave (C3: 1500 waves for 1024 SIMDs x 2); 512 rows per wave = 8 passes of 64 lanes (measured 128 .. 1024: 0.117, 0.107,
// 0.102, 0.100 ms for the three per-camera kernels together - flat from 512 on)
#define FTF_CHUNK 512
// Small problems (the window of one new camera: a few hundred thousand rows) would leave most of the chip without a wave at
// those lengths: chunks shrink, in steps of 64, until a list makes about 1024 waves (more, shorter chunks cost the
// assembly, which adds a block's chunk partials in order, what they save here: measured at config 3).
static int chunk_for(long total, int max_chunk) {
  const long c = 64 * ((total / 1024 + 63) / 64);
  return (int)std::max<long>(64, std::min<long>(max_chunk, c));
}
static int ftf_chunk(long rows) {
  const char* e = getenv("MSFM_FTF_CHUNK");   // (experiments)
  const int v = e ? atoi(e) : 0;
  return v >= 64 ? v : chunk_for(rows, FTF_CHUNK);
}
// FTF partial layout
#define F_JCJC 0     // 36
#define F_JMJC 36    // 18 (3x6)
#define F_JMJM 54    // 9
#define F_JCR 63     // 6
#define F_JMR 69     // 3
#define F_TU 72      // 6

// scalar slots (device `scal` array)
// Scalars of one LM iteration.  [0, S_GMAX) are sums over ranks (S_FAIL: non-zero on any rank = failure), S_GMAX a
// maximum.  S_XCOST is the cost at x (kept across the speculative solve), S_COST the cost at the candidate.  With more
// than one rank the kernels write the rank's partials to `sloc` and ONE sum per iteration brings them into `scal`:
// the maximum travels as one slot per rank (S_RANK0 + r), of which only rank r's is non-zero.
enum { S_COST = 0, S_MCC, S_DX2, S_X2, S_FAIL, S_XCOST, S_GMAX, S_RANK0 = 8, S_N = 64 };

struct BaPtrs {
  int A, AE, ncb, nmb, npb, NCR;
  const double *cam, *model, *pt;  // parameters being evaluated
  const double* rot;               // msfm_rot_prepare of `cam` ([Nc][4])
  const int *o_cam, *o_model, *o_pt, *o_cb, *o_mb, *o_pb, *o_cpos, *o_pm;
  const double *o_x, *o_y, *o_w;
  double *lin_r, *lin_Jc, *lin_Jm;   // the rows [AE, A) only
  const double *scale_c, *scale_m, *scale_p;
  double huber;
};

// --------------------------------------------------------------------------------------
// The linearisation of one observation row: corrected (Huber) and column-scaled residual and Jacobian blocks, as the
// eliminator and the back substitution use them.  Returns the row's cost 1/2 rho(s).  k_point, k_backsub and k_linearize
// all evaluate THIS function (26 doubles per row are cheaper to recompute from 52 bytes of row data and cached
// parameters than to write once and read twice: 0.75 GB per LM iteration at config 3).
// Rows of frozen blocks come out as zeros (their scales are 0).
// --------------------------------------------------------------------------------------
// The arithmetic of obs_linearize on values already in registers (k_ftf evaluates it camera by camera: pose, rotation cache,
// intrinsics and their column scales are the same for a whole chunk of rows).
__device__ __forceinline__ double linearize_core(const double (&pose)[6], const double (&rc)[4], const double (&cm)[3], const double (&X)[3], double ox,
                                                 double oy, double ow, const double (&sc)[6], const double (&sm)[3], const double (&sp)[3], double huber,
                                                 double& r0, double& r1, double (&jc)[12], double (&jm)[6], double (&jp)[6]) {
  double r[2], J[24];
  msfm_reproj(pose, rc, cm, X, ox, oy, ow, r, J);
  double rho0, rho1;
  msfm_huber(huber, r[0] * r[0] + r[1] * r[1], rho0, rho1);
  const double sq = sqrt(rho1);
  r0 = sq * r[0]; r1 = sq * r[1];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    const double s = sq * sc[j];
    jc[j] = s * J[j];
    jc[6 + j] = s * J[12 + j];
  }
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const double s = sq * sm[j];
    jm[j] = s * J[6 + j];
    jm[3 + j] = s * J[12 + 6 + j];
    const double spj = sq * sp[j];
    jp[j] = spj * J[9 + j];
    jp[3 + j] = spj * J[12 + 9 + j];
  }
  return 0.5 * rho0;
}
__device__ __forceinline__ double obs_linearize(const BaPtrs& P, int i, double& r0, double& r1, double (&jc)[12], double (&jm)[6], double (&jp)[6]) {
  const int c = P.o_cam[i], m = P.o_model[i], p = P.o_pt[i];
  const int cb = P.o_cb[i], mb = P.o_mb[i], pb = P.o_pb[i];
  double pose[6], rc[4], cm[3], X[3];
#pragma unroll
  for (int j = 0; j < 6; j++) pose[j] = P.cam[6 * (size_t)c + j];
#pragma unroll
  for (int j = 0; j < 4; j++) rc[j] = P.rot[4 * (size_t)c + j];
#pragma unroll
  for (int j = 0; j < 3; j++) { cm[j] = P.model[3 * (size_t)m + j]; X[j] = P.pt[3 * (size_t)p + j]; }
  const double ox = P.o_x[i], oy = P.o_y[i], ow = P.o_w[i];
  double r[2], J[24];
  msfm_reproj(pose, rc, cm, X, ox, oy, ow, r, J);
  // the column scales are fetched only now: twelve registers less while the projection is evaluated
  __builtin_amdgcn_sched_barrier(0);
  double sc[6], sm[3], sp[3];
#pragma unroll
  for (int j = 0; j < 6; j++) sc[j] = cb >= 0 ? P.scale_c[6 * cb + j] : 0.0;
#pragma unroll
  for (int j = 0; j < 3; j++) { sm[j] = mb >= 0 ? P.scale_m[3 * mb + j] : 0.0; sp[j] = pb >= 0 ? P.scale_p[3 * (size_t)pb + j] : 0.0; }
  double rho0, rho1;
  msfm_huber(P.huber, r[0] * r[0] + r[1] * r[1], rho0, rho1);
  const double sq = sqrt(rho1);
  r0 = sq * r[0]; r1 = sq * r[1];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    const double s = sq * sc[j];
    jc[j] = s * J[j];
    jc[6 + j] = s * J[12 + j];
  }
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const double s = sq * sm[j];
    jm[j] = s * J[6 + j];
    jm[3 + j] = s * J[12 + 6 + j];
    const double spj = sq * sp[j];
    jp[j] = spj * J[9 + j];
    jp[3 + j] = spj * J[12 + 9 + j];
  }
  return 0.5 * rho0;
}
// --------------------------------------------------------------------------------------
// k_linearize: thread per active observation, rows [i0, A).  <false>: the cost only (trial points).  <true>: the stored
// linearisation of the rows that belong to no eliminated point (i0 = AE; their point is frozen) - the rows of the
// eliminated points are linearised inside k_point.
// --------------------------------------------------------------------------------------
template <bool WRITE_JAC>
__device__ __forceinline__ void linearize_block(const BaPtrs& P, int i0, int blk, double* __restrict__ cost_partial, double* sh) {
  const int i = i0 + blk * 256 + threadIdx.x;
  double cost = 0.0;
  if (i < P.A) {
    if (WRITE_JAC) {
      double r0, r1, jc[12], jm[6], jp[6];
      cost = obs_linearize(P, i, r0, r1, jc, jm, jp);
      // stored rows exist for [AE, A) only: component-major over the A - AE rows (read by k_mcc_rest)
      const size_t nt = (size_t)(P.A - P.AE), it = (size_t)(i - P.AE);
      P.lin_r[it] = r0;
      P.lin_r[nt + it] = r1;
#pragma unroll
      for (int j = 0; j < 12; j++) P.lin_Jc[(size_t)j * nt + it] = jc[j];
#pragma unroll
      for (int j = 0; j < 6; j++) P.lin_Jm[(size_t)j * nt + it] = jm[j];
    } else {
      const int c = P.o_cam[i], m = P.o_model[i], p = P.o_pt[i];
      double pose[6], rc[4], cm[3], X[3];
#pragma unroll
      for (int j = 0; j < 6; j++) pose[j] = P.cam[6 * (size_t)c + j];
#pragma unroll
      for (int j = 0; j < 4; j++) rc[j] = P.rot[4 * (size_t)c + j];
#pragma unroll
      for (int j = 0; j < 3; j++) { cm[j] = P.model[3 * (size_t)m + j]; X[j] = P.pt[3 * (size_t)p + j]; }
      double r[2];
      msfm_reproj(pose, rc, cm, X, P.o_x[i], P.o_y[i], P.o_w[i], r, nullptr);
      double rho0, rho1;
      msfm_huber(P.huber, r[0] * r[0] + r[1] * r[1], rho0, rho1);
      cost = 0.5 * rho0;
    }
  }
  const double t = block_sum256(cost, sh);
  if (threadIdx.x == 0) cost_partial[blk] = t;
}
template <bool WRITE_JAC>
__global__ __launch_bounds__(256) void k_linearize(BaPtrs P, int i0, double* __restrict__ cost_partial, const double* __restrict__ spec = nullptr) {
  __shared__ double sh[4];
  if (spec && spec[0] == 0.0) return;   // (see PointPtrs::spec)
  linearize_block<WRITE_JAC>(P, i0, blockIdx.x, cost_partial, sh);
}

// rot[c] = msfm_rot_prepare(cam[c]) for all cameras (run start; the candidates' are formed inside k_backsub)
__global__ __launch_bounds__(256) void k_rot_cache(int Nc, const double* __restrict__ cam, double* __restrict__ rot) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < Nc) msfm_rot_prepare(cam + 6 * (size_t)c, rot + 4 * (size_t)c);
}

// GPS residual per camera block (gps_error_pose_absolute.h:31-44; d|x|/dx = x<0 ? -1 : 1).
// Writes corrected, column-scaled r and J (diagonal) and the cost into cost_partial[slot].
template <bool WRITE_JAC>
__device__ __forceinline__ void gps_block(int blk, int ncb, const int* __restrict__ cb_cam, const double* __restrict__ cam,
                                          const double* __restrict__ gps, double w, double huber,
                                          const double* __restrict__ scale_c, double* __restrict__ g_r,
                                          double* __restrict__ g_J, double* __restrict__ cost_partial, double* sh) {
  const int cb = blk * 256 + threadIdx.x;
  double cost = 0.0;
  if (cb < ncb) {
    const int c = cb_cam[cb];
    const double wz[3] = {w, w, w / 5.0};
    double r[3], J[3], s = 0.0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const double d = cam[6 * (size_t)c + 3 + k] - gps[3 * (size_t)cb + k];
      r[k] = wz[k] * fabs(d);
      J[k] = wz[k] * (d < 0.0 ? -1.0 : 1.0);
      s += r[k] * r[k];
    }
    double rho0, rho1;
    msfm_huber(huber, s, rho0, rho1);
    cost = 0.5 * rho0;
    if (WRITE_JAC) {
      const double sq = sqrt(rho1);
#pragma unroll
      for (int k = 0; k < 3; k++) {
        g_r[3 * (size_t)cb + k] = sq * r[k];
        g_J[3 * (size_t)cb + k] = sq * J[k] * scale_c[6 * cb + 3 + k];
      }
    }
  }
  const double t = block_sum256(cost, sh);
  if (threadIdx.x == 0) cost_partial[blk] = t;
}
template <bool WRITE_JAC>
__global__ __launch_bounds__(256) void k_gps(int ncb, const int* __restrict__ cb_cam, const double* __restrict__ cam,
                                              const double* __restrict__ gps, double w, double huber,
                                              const double* __restrict__ scale_c, double* __restrict__ g_r,
                                              double* __restrict__ g_J, double* __restrict__ cost_partial, const double* __restrict__ spec = nullptr) {
  __shared__ double sh[4];
  if (spec && spec[0] == 0.0) return;   // (see PointPtrs::spec)
  gps_block<WRITE_JAC>(blockIdx.x, ncb, cb_cam, cam, gps, w, huber, scale_c, g_r, g_J, cost_partial, sh);
}

// Sum (or max) `n` partials in a fixed order into scal[slot]: one workgroup of 1024 threads, each
// thread a fixed strided subset (independent loads in flight), then a fixed wave / block tree.
// Up to four independent reductions of per-workgroup partials in one launch (block b = job b), each in the fixed order
// of the single-job form; block 0 can also publish the failure flag.  One launch instead of one per scalar: every tiny
// dependent launch costs ~4.5 us on the LM iteration's critical path.
struct ReduceJobs {
  int count;
  struct { const double* p; int n; int slot; int is_max; } job[4];
  const int* fail;
  int fail_slot;
};
__global__ __launch_bounds__(1024) void k_reduce(ReduceJobs J, double* __restrict__ scal) {
  __shared__ double sh[16];
  const double* __restrict__ partial = J.job[blockIdx.x].p;
  const int n = J.job[blockIdx.x].n, slot = J.job[blockIdx.x].slot, is_max = J.job[blockIdx.x].is_max;
  double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
  int i = threadIdx.x;
  for (; i + 3072 < n; i += 4096) {
    const double a = partial[i], b = partial[i + 1024], c = partial[i + 2048], d = partial[i + 3072];
    if (is_max) { v0 = fmax(v0, a); v1 = fmax(v1, b); v2 = fmax(v2, c); v3 = fmax(v3, d); }
    else { v0 += a; v1 += b; v2 += c; v3 += d; }
  }
  for (; i < n; i += 1024) v0 = is_max ? fmax(v0, partial[i]) : v0 + partial[i];
  double v = is_max ? fmax(fmax(v0, v1), fmax(v2, v3)) : (v0 + v1) + (v2 + v3);
  v = is_max ? wave_max(v) : wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = sh[0];
    for (int w = 1; w < 16; w++) t = is_max ? fmax(t, sh[w]) : t + sh[w];
    scal[slot] = t;
    if (blockIdx.x == 0 && J.fail) scal[J.fail_slot] = (double)*J.fail;
  }
}

// --------------------------------------------------------------------------------------
// k_point: the eliminated points.  mode 0: full; mode 1: raw squared column norms only.
// --------------------------------------------------------------------------------------
// One pass of a folding workgroup (FoldTables below): `words` 32-bit words at fold_stream + off - the headers of n_slots slots
// (first entry relative to the entry area | count << 16; padded to a multiple of four words), then the slots' entries
// (record_i | record_j << 16) - which one wave instruction per 256 words copies straight into LDS.  slot0: the first slot's
// index in workgroup-major numbering (for its rank); n_diag: how many of the pass's slots (its first ones) are camera-diagonal.
struct FoldPass { int off, words, slot0, n_slots, n_diag, pad0, pad1, pad2; };
struct PointPtrs {
  BaPtrs B;            // row data, parameters and scales the rows are linearised with
  int npb, NCR;
  const int *pt_first, *pm_first, *pm_mb;
  double *diag_p;
  double *ptL, *ptg, *T, *Tu, *Tm, *Tmu;
  double radius, dmin, dmax;
  int reuse_diag, mode;
  int store_rows;      // first pass at this linearisation point: also the cost
  double* cost_partial;
  int* fail;
  // FoldTables (fold_wg nullptr: every product goes through the pair lists)
  const int *fold_wg, *fold_ovf_off, *fold_wg_pass_first, *fold_slot_rank;
  const FoldPass* fold_pass;
  const unsigned* fold_stream;
  double* fold_partial;
  double* fold_mc_partial;   // intrinsics x camera products of the same workgroups (one intrinsics block only; nullptr: gather path)
  // A launch enqueued BEFORE the host has seen the step it follows (msfm_ba_run): spec[0] != 0 if the device-side decision
  // (k_publish_scalars) accepted that step - else the launch ends at once - and spec[1] = the new trust-region radius.
  const double* spec;
  int keep_T;   // MSFM_KEEP_T=1: store every T record as rounds 1-3 did (comparison)
  int tu_direct;   // (MSFM_TU_DIRECT=0: T.u through the lane exchange as well)
};

// (8-lane groups; after the first two steps the four lanes of a quad hold the same value, so adding lane 7 - i is adding lane i ^ 4)
#define GROUP_SUM(x) { x += dpp_f64<MSFM_DPP_XOR1>(x); x += dpp_f64<MSFM_DPP_XOR2>(x); x += dpp_f64<MSFM_DPP_HALF_MIRROR>(x); }

// 8 lanes per point, lane = observation (rounds of 8 for longer tracks).  Every lane linearises its own row
// (obs_linearize: the row data is read coalesced, the parameters come from cache), the per-point sums are 3-step
// reductions inside the 8-lane group, and every lane then finishes its own observation's T = (Jc^T Jp) L^-T.
// 256 threads = 32 points.  Tracks of up to 8 views keep their rows in registers; longer ones linearise them again.
#define FOLD_OVF 51   // second-round records (rows 8..15 of a point) a workgroup can park beside the 256 first-round ones
#define FOLD_NREC (256 + FOLD_OVF)   // records of 2 x 10 doubles (two 16-byte aligned halves of 9) in the 48 KB row park
#define FOLD_WORDS 1024      // words (slot headers + entries) of one pass, staged in LDS
#define FOLD_PASS_SLOTS 128  // slots of one pass at most: four threads per slot, two rounds
#ifdef MSFM_FOLD_STAMPS
// developer build only (make CXXFLAGS+=-DMSFM_FOLD_STAMPS): clock stamps of wave 0 of every folding workgroup
__device__ long long g_fold_stamps[8192][8];
#define FSTAMP(i) do { if (fold && tid == 0 && blockIdx.x < 8192) g_fold_stamps[blockIdx.x][i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define FSTAMP(i) do {} while (0)
#endif
__device__ __forceinline__ void k_point_body(const PointPtrs& P, double* __restrict__ gmax_partial, double* sh, double* park, unsigned* ent_s) {
  const int tid = threadIdx.x, sub = tid & 7;
  const int pb = blockIdx.x * 32 + (tid >> 3);
  const bool act = pb < P.npb;
  // FoldTables: the slot headers and entries of the workgroup's first pass start on their way into LDS now (one
  // global_load_lds_dwordx4 per wave and 256 words; nothing waits for them before the fold phase at the end)
  const bool fold = P.fold_wg != nullptr && P.mode != 1 && P.fold_wg[blockIdx.x] != 0;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  auto stage = [&](int off, int words) {
    if (256 * wv < words)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(P.fold_stream + (size_t)off + 256 * wv + 4 * (tid & 63)),
                                       (__attribute__((address_space(3))) void*)(ent_s + 256 * wv), 16, 0, 0);
  };
  int fp0 = 0, fp1 = 0;
  if (fold) {
    fp0 = __builtin_amdgcn_readfirstlane(P.fold_wg_pass_first[blockIdx.x]);
    fp1 = __builtin_amdgcn_readfirstlane(P.fold_wg_pass_first[blockIdx.x + 1]);
    stage(__builtin_amdgcn_readfirstlane(P.fold_pass[fp0].off), __builtin_amdgcn_readfirstlane(P.fold_pass[fp0].words));
  }
  FSTAMP(0);
  int f = 0, l = 0;
  if (act) { f = P.pt_first[pb]; l = P.pt_first[pb + 1]; }
  const bool single = (l - f) <= 8;
  double V00 = 0, V10 = 0, V11 = 0, V20 = 0, V21 = 0, V22 = 0, g0 = 0, g1 = 0, g2 = 0;
  // The lane's row of the (only) round waits in LDS (component-major, one column per thread: no bank conflicts) while
  // the point's 3x3 system is reduced and factored - 48 registers less across that phase.
  double cost = 0.0;
  for (int base = f; base < l; base += 8) {
    const int i = base + sub;
    if (i < l) {
      double r0, r1, jcs[12], jms[6], jps[6];
      const double ci = obs_linearize(P.B, i, r0, r1, jcs, jms, jps);
#pragma unroll
      for (int k = 0; k < 12; k++) park[k * 256 + tid] = jcs[k];
#pragma unroll
      for (int k = 0; k < 6; k++) { park[(12 + k) * 256 + tid] = jms[k]; park[(18 + k) * 256 + tid] = jps[k]; }
      if (P.store_rows) cost += ci;
      const double a0 = jps[0], a1 = jps[1], a2 = jps[2], b0 = jps[3], b1 = jps[4], b2 = jps[5];
      V00 += a0 * a0 + b0 * b0; V10 += a1 * a0 + b1 * b0; V11 += a1 * a1 + b1 * b1;
      V20 += a2 * a0 + b2 * b0; V21 += a2 * a1 + b2 * b1; V22 += a2 * a2 + b2 * b2;
      g0 += a0 * r0 + b0 * r1; g1 += a1 * r0 + b1 * r1; g2 += a2 * r0 + b2 * r1;
    }
  }
  FSTAMP(1);
  GROUP_SUM(V00) GROUP_SUM(V10) GROUP_SUM(V11) GROUP_SUM(V20) GROUP_SUM(V21) GROUP_SUM(V22)
  GROUP_SUM(g0) GROUP_SUM(g1) GROUP_SUM(g2)

  double gmax = 0.0;
  double l00 = 1, l10 = 0, l11 = 1, l20 = 0, l21 = 0, l22 = 1, i00 = 1, i11 = 1, i22 = 1, u0 = 0, u1 = 0, u2 = 0;
  if (act) {
    double* dg = P.diag_p + 3 * (size_t)pb;
    if (P.mode == 1) {
      if (sub == 0) { dg[0] = V00; dg[1] = V11; dg[2] = V22; }
    } else {
      double d0, d1, d2;
      if (!P.reuse_diag) {
        d0 = fmin(fmax(V00, P.dmin), P.dmax); d1 = fmin(fmax(V11, P.dmin), P.dmax); d2 = fmin(fmax(V22, P.dmin), P.dmax);
        if (sub == 0) { dg[0] = d0; dg[1] = d1; dg[2] = d2; }
      } else {
        d0 = dg[0]; d1 = dg[1]; d2 = dg[2];
      }
      // lm_diagonal = sqrt(diagonal / radius); the eliminator adds its square
      const double radius = P.spec ? P.spec[1] : P.radius;
      const double q0 = sqrt(d0 / radius), q1 = sqrt(d1 / radius), q2 = sqrt(d2 / radius);
      V00 += q0 * q0; V11 += q1 * q1; V22 += q2 * q2;
      // 3x3 Cholesky (Eigen LLT on the e-block in Ceres' InvertPSDMatrix)
      bool ok = V00 > 0.0;
      l00 = sqrt(V00);
      l10 = V10 / l00; l20 = V20 / l00;
      const double e1 = V11 - l10 * l10;
      ok = ok && e1 > 0.0;
      l11 = sqrt(e1);
      l21 = (V21 - l20 * l10) / l11;
      const double e2 = V22 - l20 * l20 - l21 * l21;
      ok = ok && e2 > 0.0;
      l22 = sqrt(e2);
      i00 = 1.0 / l00; i11 = 1.0 / l11; i22 = 1.0 / l22;
      u0 = g0 * i00;
      u1 = (g1 - l10 * u0) * i11;
      u2 = (g2 - l20 * u0 - l21 * u1) * i22;
      if (sub == 0) {
        if (!ok) atomicOr(P.fail, 2);
        double* L = P.ptL + 6 * (size_t)pb;
        // (the diagonal as reciprocals: the back substitution multiplies - six binary64 divisions, ~170 instructions that one lane in
        //  eight needs and the whole wave waits for, were a ninth of k_backsub's instruction stream)
        L[0] = i00; L[1] = l10; L[2] = i11; L[3] = l20; L[4] = l21; L[5] = i22;
        // (g itself is not stored: the back substitution forms it again from the rows)
        const double* sp = P.B.scale_p + 3 * (size_t)pb;
        gmax = fmax(fabs(g0 / sp[0]), fmax(fabs(g1 / sp[1]), fabs(g2 / sp[2])));
      }
    }
  }
  FSTAMP(2);
  if (P.mode != 1) {
    // camera entries: T = (Jc^T Jp) L^-T (one 144-byte record per observation, camera-major), T.u
    // Stores: neighbouring lanes hold the observations of ONE point - eight different cameras, eight far-apart addresses,
    // one 8-byte write request per lane and component (21.6 M at C3).  The lanes of a wave therefore trade records first,
    // (point q, observation s) -> lane 8 s + q: now neighbouring lanes hold the same observation slot of consecutive
    // points, which sit at consecutive positions of one camera wherever the points share their cameras, and the stores
    // of a quad merge (k_point 0.229 -> 0.223 ms at C3; a record-major T was also measured: the same here, but k_pairs
    // 0.28 -> 0.49 ms).
    // Round 3 (FoldTables): in a folding workgroup every record also stays in LDS (record-major, where the parked rows were,
    // once every lane is done with those) and the camera x camera products are formed from there below.
    const int wl = tid & 63, tsrc = ((wl & 7) << 3) | (wl >> 3);
    const bool need_T = !fold || P.fold_mc_partial == nullptr || P.keep_T;   // (uniform over the workgroup)
    auto round = [&](int rd, double (&Tk)[18], int& cpk) {
      const int i = f + rd + sub;
      int cp = -1;
      if (i < l) cp = P.B.o_cpos[i];
      double T[18], tu[6];
#pragma unroll
      for (int k = 0; k < 18; k++) T[k] = 0.0;
#pragma unroll
      for (int k = 0; k < 6; k++) tu[k] = 0.0;
      if (cp >= 0) {
        double jcs[12], jms[6], jps[6];
        if (!single) { double r0, r1; obs_linearize(P.B, i, r0, r1, jcs, jms, jps); }
        else {
#pragma unroll
          for (int k = 0; k < 12; k++) jcs[k] = park[k * 256 + tid];
#pragma unroll
          for (int k = 0; k < 6; k++) jps[k] = park[(18 + k) * 256 + tid];
        }
        const double a0 = jps[0], a1 = jps[1], a2 = jps[2], b0 = jps[3], b1 = jps[4], b2 = jps[5];
#pragma unroll
        for (int a = 0; a < 6; a++) {
          const double ja = jcs[a], jb = jcs[6 + a];
          const double w0 = ja * a0 + jb * b0, w1 = ja * a1 + jb * b1, w2 = ja * a2 + jb * b2;
          const double t0 = w0 * i00;
          const double t1 = (w1 - l10 * t0) * i11;
          const double t2 = (w2 - l20 * t0 - l21 * t1) * i22;
          T[a * 3 + 0] = t0; T[a * 3 + 1] = t1; T[a * 3 + 2] = t2;
          tu[a] = t0 * u0 + t1 * u1 + t2 * u2;
        }
      }
      cpk = cp;
#pragma unroll
      for (int k = 0; k < 18; k++) Tk[k] = T[k];
      if (!need_T && P.tu_direct) {
        // (only T.u leaves the workgroup: 48 contiguous bytes per row, stored from the lane that formed them - no exchange)
        if (cp >= 0) {
          double2* Tu2 = reinterpret_cast<double2*>(P.Tu + 6 * (size_t)cp);
#pragma unroll
          for (int k = 0; k < 3; k++) Tu2[k] = make_double2(tu[2 * k], tu[2 * k + 1]);
        }
        return;
      }
      const int cpt = __shfl(cp, tsrc, 64);
      // The T records in memory are read by the pair kernels only - by the entries that did NOT fold.  A workgroup folds all
      // of its entries or none (both records of an entry belong to one point), so a folding workgroup whose intrinsics x camera
      // products fold too has no reader for its records: it keeps them in LDS for its own products and does not store them
      // (173 MB per iteration at config 3, and the 36 lane exchanges per round that line the stores up).
      if (need_T) {
#pragma unroll
        for (int k = 0; k < 18; k++) T[k] = __shfl(T[k], tsrc, 64);
      }
#pragma unroll
      for (int k = 0; k < 6; k++) tu[k] = __shfl(tu[k], tsrc, 64);
      if (cpt >= 0) {
        double2* Tu2 = reinterpret_cast<double2*>(P.Tu + 6 * (size_t)cpt);
#pragma unroll
        for (int k = 0; k < 3; k++) Tu2[k] = make_double2(tu[2 * k], tu[2 * k + 1]);
        // component-major (T[k][position], positions camera-major): records of consecutive points of a camera are
        // neighbours in every component plane, so these stores and the pair kernel's loads coalesce over the runs of
        // points that share their cameras
        if (need_T) {
#pragma unroll
          for (int k = 0; k < 18; k++) P.T[(size_t)k * P.NCR + cpt] = T[k];
        }
      }
    };
    // records in LDS: two halves (rows 0..2 and 3..5 of the 6 x 3 record) of 10 doubles each, record-major, so that a slot
    // thread fetches its nine values with four 16-byte reads and one 8-byte read
    double* const H0 = park;
    double* const H1 = park + FOLD_NREC * 10;
    auto rec_st = [&](int rec, const double (&Tk)[18]) {
#pragma unroll
      for (int k = 0; k < 9; k++) { H0[rec * 10 + k] = Tk[k]; H1[rec * 10 + k] = Tk[9 + k]; }
    };
    {
      double Tk[18];
      int cpk;
      FSTAMP(3);
      round(0, Tk, cpk);
      FSTAMP(4);
      // (point, intrinsics-block) entries: Tm = (sum Jm^T Jp) L^-T.  After the first round and after the park has turned into
      // the record store, so that the lane that forms Tm can put it beside the records (its first entry's: with one intrinsics
      // block that is the only one; a point without an entry leaves zeros): the lane's Jm and Jp rows cross the barrier in
      // registers, and what the section asks from memory (three dependent loads) is on its way during the barrier.
      int pe0 = 0, pe1 = 0, my_mb = -2;
      if (act) {
        pe0 = P.pm_first[pb]; pe1 = P.pm_first[pb + 1];
        if (f + sub < l) my_mb = P.B.o_mb[f + sub];   // (the lane's row of the first round)
      }
      double pjm[6], pjp[6];
      if (single) {
#pragma unroll
        for (int k = 0; k < 6; k++) { pjm[k] = park[(12 + k) * 256 + tid]; pjp[k] = park[(18 + k) * 256 + tid]; }
      }
      const int mb_first = pe0 < pe1 ? P.pm_mb[pe0] : -1;
      if (fold) {
        __syncthreads();   // every lane has consumed its parked row data: the park becomes the record store
        if (cpk >= 0) rec_st(tid, Tk);
      }
      auto tm_finish = [&](int e, double (&W)[9]) {
#pragma unroll
        for (int k = 0; k < 9; k++) GROUP_SUM(W[k])
        if (sub == 0) {
          double* Tm = P.Tm + 9 * (size_t)e;
          double* Tmu = P.Tmu + 3 * (size_t)e;
          const bool pads = fold && P.fold_mc_partial && e == pe0;
#pragma unroll
          for (int a = 0; a < 3; a++) {
            const double t0 = W[a * 3] * i00;
            const double t1 = (W[a * 3 + 1] - l10 * t0) * i11;
            const double t2 = (W[a * 3 + 2] - l20 * t0 - l21 * t1) * i22;
            Tm[a * 3 + 0] = t0; Tm[a * 3 + 1] = t1; Tm[a * 3 + 2] = t2;
            Tmu[a] = t0 * u0 + t1 * u1 + t2 * u2;
            if (pads) {   // into the tenth doubles of the point's first five records, for the intrinsics x camera products below
              const double tv[3] = {t0, t1, t2};
#pragma unroll
              for (int c = 0; c < 3; c++) { const int k = a * 3 + c; ((k & 1) ? H1 : H0)[(tid + (k >> 1)) * 10 + 9] = tv[c]; }
            }
          }
        }
      };
      auto tm_row = [&](double (&W)[9], const double (&jms)[6], const double (&jps)[6]) {
        const double a0 = jps[0], a1 = jps[1], a2 = jps[2], b0 = jps[3], b1 = jps[4], b2 = jps[5];
#pragma unroll
        for (int a = 0; a < 3; a++) {
          const double ja = jms[a], jb = jms[3 + a];
          W[a * 3 + 0] += ja * a0 + jb * b0;
          W[a * 3 + 1] += ja * a1 + jb * b1;
          W[a * 3 + 2] += ja * a2 + jb * b2;
        }
      };
      // (two separate paths, so that the rows held in registers are not alive beside the linearisation of the long tracks)
      if (single) {
        for (int e = pe0; e < pe1; e++) {
          const int mb = e == pe0 ? mb_first : P.pm_mb[e];
          double W[9];
#pragma unroll
          for (int k = 0; k < 9; k++) W[k] = 0.0;
          if (f + sub < l && my_mb == mb) tm_row(W, pjm, pjp);
          tm_finish(e, W);
        }
      } else {
        for (int e = pe0; e < pe1; e++) {
          const int mb = e == pe0 ? mb_first : P.pm_mb[e];
          double W[9];
#pragma unroll
          for (int k = 0; k < 9; k++) W[k] = 0.0;
          for (int base = f; base < l; base += 8) {
            const int i = base + sub;
            if (i < l && P.B.o_mb[i] == mb) {
              double r0, r1, jcs[12], jms[6], jps[6];
              obs_linearize(P.B, i, r0, r1, jcs, jms, jps);
              tm_row(W, jms, jps);
            }
          }
          tm_finish(e, W);
        }
      }
      if (fold && P.fold_mc_partial && sub == 0 && act && pe0 == pe1) {
#pragma unroll
        for (int k = 0; k < 9; k++) ((k & 1) ? H1 : H0)[(tid + (k >> 1)) * 10 + 9] = 0.0;
      }
      FSTAMP(5);
      for (int rd = 8; __any(f + rd < l); rd += 8) {
        const int ovf = (fold && act) ? P.fold_ovf_off[pb] : 0;   // (asked for before the round's own loads)
        round(rd, Tk, cpk);
        if (fold && cpk >= 0) {
          const int rec = 256 + ovf + (rd - 8) + sub;
          rec_st(rec, Tk);
          H0[rec * 10 + 9] = (double)(tid >> 3);   // (a second-round record names its point)
        }
      }
    }
    FSTAMP(6);
    if (fold) {
      const int q = tid & 3, ra = q >> 1, rb = q & 1, sl4 = tid >> 2;   // rows 3 ra .. of record i times rows 3 rb .. of record j
      for (int p = fp0; p < fp1; p++) {
        const int slot0 = __builtin_amdgcn_readfirstlane(P.fold_pass[p].slot0), n_slots = __builtin_amdgcn_readfirstlane(P.fold_pass[p].n_slots);
        const int n_diag = P.fold_mc_partial ? __builtin_amdgcn_readfirstlane(P.fold_pass[p].n_diag) : 0;
        if (p > fp0) {
          __syncthreads();   // (the previous pass is done with the staged words)
          stage(__builtin_amdgcn_readfirstlane(P.fold_pass[p].off), __builtin_amdgcn_readfirstlane(P.fold_pass[p].words));
        }
        // the LDS-DMA copy of stage() counts in vmcnt, and s_barrier itself does not wait for it: every wave drains its own
        // copy explicitly before the barrier (this toolchain happens to emit the wait; nothing guarantees that it always will)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();     // records stored (first pass); the pass's words have arrived in LDS
        // where the results go: asked for now, needed after the loops
        int rank0 = 0, rank1 = 0;
        if (sl4 < n_slots) rank0 = P.fold_slot_rank[slot0 + sl4];
        if (sl4 + 64 < n_slots) rank1 = P.fold_slot_rank[slot0 + sl4 + 64];
        const int hdr = (n_slots + 3) & ~3;
        for (int sl = sl4, rnk = rank0; sl < n_slots; sl += 64, rnk = rank1) {
          // the slot's four threads form nine entries of the 6 x 6 product each, over the slot's entries in point order
          // (measured alternatives, config 3: the next entry's records kept in flight behind the multiplies - no change; every
          // fourth entry per thread with the whole 6 x 6 product and a sum across the quad - k_point 0.41 -> 0.52 ms.  The
          // phase is bound by the LDS array: the records a lane group gathers fall on the 16 four-bank groups at random)
          const unsigned h = ent_s[sl];
          const int e0 = hdr + (int)(h & 0xffffu), e1 = e0 + (int)(h >> 16);
          // intrinsics x camera: Tm_p T_rec^T summed over the records of one camera in this workgroup - exactly the entries of
          // the camera's diagonal slot that pair a record with itself (a camera that sees a point twice also has the two cross
          // entries there).  The two threads that hold rows 0..2 of record i take nine entries of the 3 x 6 product each: the
          // rows of record j they have loaded anyway, times the point's Tm from the spare doubles of its records.
          // (rows 0 and 1 of the 3 x 6 product with the threads that hold rows 0..2 of record i, row 2 with the other two)
          const bool mcd = sl < n_diag;
          const int ma0 = ra ? 2 : 0;
          double acc[9], accm[6];
#pragma unroll
          for (int k = 0; k < 9; k++) acc[k] = 0.0;
#pragma unroll
          for (int k = 0; k < 6; k++) accm[k] = 0.0;
          for (int e = e0; e < e1; e++) {
            const unsigned pr = ent_s[e];
            const double* oi = (ra ? H1 : H0) + (int)(pr & 0xffffu) * 10;
            const double* oj = (rb ? H1 : H0) + (int)(pr >> 16) * 10;
            double ti[9], tj[9];
#pragma unroll
            for (int k2 = 0; k2 < 4; k2++) {
              const double2 vi = *reinterpret_cast<const double2*>(oi + 2 * k2);
              const double2 vj = *reinterpret_cast<const double2*>(oj + 2 * k2);
              ti[2 * k2] = vi.x; ti[2 * k2 + 1] = vi.y; tj[2 * k2] = vj.x; tj[2 * k2 + 1] = vj.y;
            }
            ti[8] = oi[8]; tj[8] = oj[8];
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
              for (int c = 0; c < 3; c++)
                acc[a * 3 + c] += ti[a * 3] * tj[c * 3] + ti[a * 3 + 1] * tj[c * 3 + 1] + ti[a * 3 + 2] * tj[c * 3 + 2];
            if (mcd && (pr & 0xffffu) == (pr >> 16)) {
              const int rec = (int)(pr & 0xffffu);
              int r0 = rec & ~7;
              if (rec >= 256) r0 = 8 * (int)H0[rec * 10 + 9];   // (second-round record: its point is written beside it)
              double tm[6];
#pragma unroll
              for (int k6 = 0; k6 < 6; k6++) { const int k = 3 * ma0 + k6; tm[k6] = ((k & 1) ? H1 : H0)[(r0 + (k >> 1)) * 10 + 9]; }   // (row 3 is never used)
#pragma unroll
              for (int a = 0; a < 2; a++)
#pragma unroll
                for (int c = 0; c < 3; c++)
                  accm[a * 3 + c] += tm[a * 3] * tj[c * 3] + tm[a * 3 + 1] * tj[c * 3 + 1] + tm[a * 3 + 2] * tj[c * 3 + 2];
            }
          }
          if (mcd) {
            double* om = P.fold_mc_partial + (size_t)rnk * 18;
#pragma unroll
            for (int a = 0; a < 2; a++)
              if (ma0 + a < 3) {
#pragma unroll
                for (int c = 0; c < 3; c++) om[(ma0 + a) * 6 + 3 * rb + c] = accm[a * 3 + c];
              }
          }
          double* out = P.fold_partial + (size_t)rnk * 36;
#pragma unroll
          for (int a = 0; a < 3; a++)
#pragma unroll
            for (int c = 0; c < 3; c++) out[(3 * ra + a) * 6 + 3 * rb + c] = acc[a * 3 + c];
        }
      }
      FSTAMP(7);
    }
  }
  if (P.store_rows) {
    const double tc = block_sum256(cost, sh);
    if (threadIdx.x == 0) P.cost_partial[blockIdx.x] = tc;
  }
  const double t = block_max256(gmax, sh);
  if (threadIdx.x == 0) gmax_partial[blockIdx.x] = t;
}
// (three waves per SIMD: 168 registers with three dwords of scratch, against 175 and two waves: 0.337 -> 0.309 ms at C3;
// the 48 KB of parked rows allow exactly three workgroups per CU)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_point(PointPtrs P, double* __restrict__ gmax_partial) {
  __shared__ double sh[4];
  __shared__ double park[24 * 256];
  __shared__ __attribute__((aligned(16))) unsigned ent_s[FOLD_WORDS];
  if (P.spec && P.spec[0] == 0.0) return;   // enqueued ahead of a step that was not accepted
  k_point_body(P, gmax_partial, sh, park, ent_s);
}

// The static inputs of a row in camera-major order (k_ftf): built once per problem from the resident row arrays.
__global__ __launch_bounds__(256) void k_cam_rows(int A, const int* __restrict__ o_cpos, const int* __restrict__ o_pt, const double* __restrict__ o_x,
                                                   const double* __restrict__ o_y, const double* __restrict__ o_w, int* __restrict__ cm_pt,
                                                   double* __restrict__ cm_x, double* __restrict__ cm_y, double* __restrict__ cm_w, int* __restrict__ cm_row) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= A) return;
  const int cp = o_cpos[i];
  if (cp < 0) return;
  cm_pt[cp] = o_pt[i]; cm_x[cp] = o_x[i]; cm_y[cp] = o_y[i]; cm_w[cp] = o_w[i]; cm_row[cp] = i;
}
// cm_X / cm_Xc at the start of a run: the coordinates of every camera-major row's point (the rows of frozen points keep them)
__global__ __launch_bounds__(256) void k_cam_points(int ncr, const int* __restrict__ cm_pt, const double* __restrict__ pt, double* __restrict__ X0,
                                                     double* __restrict__ X1) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= ncr) return;
  const size_t p = (size_t)cm_pt[e];
#pragma unroll
  for (int j = 0; j < 3; j++) { const double v = pt[3 * p + j]; X0[3 * (size_t)e + j] = v; X1[3 * (size_t)e + j] = v; }
}
__global__ __launch_bounds__(256) void k_chunk_cam(int nchunk, const int* __restrict__ ch_start, const int* __restrict__ cm_row, const int* __restrict__ o_cam,
                                                    const int* __restrict__ o_model, const int* __restrict__ o_cb, const int* __restrict__ o_mb,
                                                    int4* __restrict__ chunk_cam) {
  const int ch = blockIdx.x * 256 + threadIdx.x;
  if (ch >= nchunk) return;
  const int i = cm_row[ch_start[ch]];
  chunk_cam[ch] = make_int4(o_cam[i], o_model[i], o_cb[i], o_mb[i]);
}

// --------------------------------------------------------------------------------------
// k_ftf: one wave per chunk of a camera's (camera-major) observation rows.
// partial[chunk][PSTRIDE]: Jc^T Jc | Jm^T Jc | Jm^T Jm | Jc^T r | Jm^T r | sum T.u
// --------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wave_reduce_store(double (&acc)[N], double* out, int lane) {
  double v;
  int k;
  wave_reduce_scatter<N>(acc, lane, v, k);
  if (k >= 0) out[k] = v;   // one store instruction: every value ends in its own lane
}

// Round 4: the rows are LINEARISED AGAIN here instead of being read back from a 160-byte camera-major copy that k_point
// wrote (192 MB written + 258 MB read per LM iteration at config 3): per row the camera-major statics (point index and
// x, y, weight: 28 bytes, coalesced) and the point (24 bytes, gathered); pose, rotation cache, intrinsics and their column
// scales are those of the chunk's camera and live in scalar registers.  Same function, same inputs, same order of the sums
// as before: the per-camera sums are what the stored rows gave.
struct CamRows {
  const int* pt;             // [NCR] point of the row at camera-major position e
  const double* X;           // [NCR][3] the point's coordinates at the linearisation point, in camera-major order (cm_X: written by
                             // k_backsub for the candidate, swapped with the parameters; the 24-byte gather through `pt` moved
                             // 100 MB of cache lines for 29 MB of coordinates at config 3)
  const double *x, *y, *w;   // [NCR]
  const int4* chunk_cam;     // [chunks] camera, intrinsics, camera block, intrinsics block (-1: frozen) of the chunk's rows
};
__device__ __forceinline__ void ftf_body(int blk, int nchunk, const int* __restrict__ ch_start, const int* __restrict__ ch_end, const BaPtrs& P, const CamRows& R,
                                         const double* __restrict__ Tu, const int* __restrict__ cpos_pb, double* __restrict__ partial) {
  const int chunk = blk * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (chunk >= nchunk) return;
  // Jc^T Jc and Jm^T Jm are symmetric: 21 + 6 of their 36 + 9 entries are accumulated (the mirrored ones are the same
  // products in the other order, so the stored 78 values are what the full loops gave).
  // acc: Jc^T Jc lower triangle (21, row-wise) | Jm^T Jc (18) | Jm^T Jm lower (6) | Jc^T r (6) | Jm^T r (3) | sum T.u (6)
  constexpr int O_JC = 0, O_JMC = 21, O_JM = 39, O_CR = 45, O_MR = 51, O_TU = 54;
  double acc[60];
#pragma unroll
  for (int k = 0; k < 60; k++) acc[k] = 0.0;
  // the chunk's camera: uniform over the wave
  const int4 cc = R.chunk_cam[chunk];
  const int c = __builtin_amdgcn_readfirstlane(cc.x), m = __builtin_amdgcn_readfirstlane(cc.y);
  const int cb = __builtin_amdgcn_readfirstlane(cc.z), mb = __builtin_amdgcn_readfirstlane(cc.w);
  double pose[6], rc[4], cm[3], sc[6], sm[3];
  const double sp[3] = {0.0, 0.0, 0.0};   // (the point columns are not needed here)
#pragma unroll
  for (int j = 0; j < 6; j++) { pose[j] = P.cam[6 * (size_t)c + j]; sc[j] = cb >= 0 ? P.scale_c[6 * cb + j] : 0.0; }
#pragma unroll
  for (int j = 0; j < 4; j++) rc[j] = P.rot[4 * (size_t)c + j];
#pragma unroll
  for (int j = 0; j < 3; j++) { cm[j] = P.model[3 * (size_t)m + j]; sm[j] = mb >= 0 ? P.scale_m[3 * mb + j] : 0.0; }
  const int e1 = ch_end[chunk];
  int e = ch_start[chunk] + lane;
  // one row ahead: statics and point (camera-major copies, all coalesced) of row e + 64
  double xn = 0, yn = 0, wn = 0, Xn[3] = {0, 0, 0};
  if (e < e1) {
    xn = R.x[e]; yn = R.y[e]; wn = R.w[e];
#pragma unroll
    for (int j = 0; j < 3; j++) Xn[j] = R.X[3 * (size_t)e + j];
  }
  for (; e < e1; e += 64) {
    const double X[3] = {Xn[0], Xn[1], Xn[2]};
    const double ox = xn, oy = yn, ow = wn;
    const bool has_tu = cpos_pb[e] >= 0;
    double2 tu2[3];
    if (has_tu) {
      const double2* tp = reinterpret_cast<const double2*>(Tu + 6 * (size_t)e);
#pragma unroll
      for (int a = 0; a < 3; a++) tu2[a] = tp[a];
    }
    if (e + 64 < e1) {
      xn = R.x[e + 64]; yn = R.y[e + 64]; wn = R.w[e + 64];
#pragma unroll
      for (int j = 0; j < 3; j++) Xn[j] = R.X[3 * (size_t)(e + 64) + j];
    }
    double r0, r1, jcr[12], jmr[6], jpr[6];
    linearize_core(pose, rc, cm, X, ox, oy, ow, sc, sm, sp, P.huber, r0, r1, jcr, jmr, jpr);
    const double* jc = jcr;
    const double* jm = jmr;
#pragma unroll
    for (int a = 0; a < 6; a++) {
#pragma unroll
      for (int b = 0; b <= a; b++) acc[O_JC + a * (a + 1) / 2 + b] += jc[a] * jc[b] + jc[6 + a] * jc[6 + b];
      acc[O_CR + a] += jc[a] * r0 + jc[6 + a] * r1;
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
      for (int b = 0; b < 6; b++) acc[O_JMC + a * 6 + b] += jm[a] * jc[b] + jm[3 + a] * jc[6 + b];
#pragma unroll
      for (int b = 0; b <= a; b++) acc[O_JM + a * (a + 1) / 2 + b] += jm[a] * jm[b] + jm[3 + a] * jm[3 + b];
      acc[O_MR + a] += jm[a] * r0 + jm[3 + a] * r1;
    }
    if (has_tu) {
#pragma unroll
      for (int a = 0; a < 3; a++) { acc[O_TU + 2 * a] += tu2[a].x; acc[O_TU + 2 * a + 1] += tu2[a].y; }
    }
  }
  // every sum ends in one lane (wave_reduce_scatter); that lane stores it, and its mirror image for the two symmetric blocks
  double* out = partial + (size_t)chunk * PSTRIDE;
  double v;
  int k;
  wave_reduce_scatter<60>(acc, lane, v, k);
  if (k >= 0) {
    int o1, o2;
    if (k < O_JMC) {
      int a = 0;
      while ((a + 1) * (a + 2) / 2 <= k) a++;
      const int b_ = k - a * (a + 1) / 2;
      o1 = F_JCJC + a * 6 + b_; o2 = F_JCJC + b_ * 6 + a;
    } else if (k < O_JM) {
      o1 = o2 = F_JMJC + (k - O_JMC);
    } else if (k < O_CR) {
      const int t = k - O_JM;
      int a = 0;
      while ((a + 1) * (a + 2) / 2 <= t) a++;
      const int b_ = t - a * (a + 1) / 2;
      o1 = F_JMJM + a * 3 + b_; o2 = F_JMJM + b_ * 3 + a;
    } else if (k < O_MR) {
      o1 = o2 = F_JCR + (k - O_CR);
    } else if (k < O_TU) {
      o1 = o2 = F_JMR + (k - O_MR);
    } else {
      o1 = o2 = F_TU + (k - O_TU);
    }
    out[o1] = v;
    if (o2 != o1) out[o2] = v;
  }
}
__global__ __launch_bounds__(256) void k_ftf(int nchunk, const int* __restrict__ ch_start, const int* __restrict__ ch_end, BaPtrs P, CamRows R,
                                              const double* __restrict__ Tu, const int* __restrict__ cpos_pb, double* __restrict__ partial) {
  ftf_body(blockIdx.x, nchunk, ch_start, ch_end, P, R, Tu, cpos_pb, partial);
}

// Sum the FTF partials of each camera block (fixed order) -> camftf[cb][PSTRIDE]; the GPS rows
// are folded in by the lead rank only (the buffer is summed over ranks afterwards).
__global__ __launch_bounds__(128) void k_camftf(int ncb, const int* __restrict__ cam_chunk_first,
                                                 const double* __restrict__ partial, double* __restrict__ camftf,
                                                 const double* __restrict__ g_r, const double* __restrict__ g_J, int add_gps, int post,
                                                 double* __restrict__ diag_c, const double* __restrict__ scale_c, int reuse_diag, int mode,
                                                 double dmin, double dmax, double* __restrict__ gmax_c) {
  __shared__ double sums[PSTRIDE];
  const int cb = blockIdx.x, t = threadIdx.x;
  if (t < PSTRIDE) {
    double s = 0.0;
    int ch = cam_chunk_first[cb];
    const int ch1 = cam_chunk_first[cb + 1];
    for (; ch + 4 <= ch1; ch += 4) {   // four loads in flight, summed in chunk order as the single loop would
      const double* q = partial + (size_t)ch * PSTRIDE + t;
      const double p0 = q[0], p1 = q[PSTRIDE], p2 = q[2 * PSTRIDE], p3 = q[3 * PSTRIDE];
      s += p0; s += p1; s += p2; s += p3;
    }
    for (; ch < ch1; ch++) s += partial[(size_t)ch * PSTRIDE + t];
    if (add_gps) {
      if (t >= F_JCJC && t < F_JCJC + 36) {
        const int a = (t - F_JCJC) / 6, b = (t - F_JCJC) % 6;
        if (a == b && a >= 3) { const double j = g_J[3 * (size_t)cb + a - 3]; s += j * j; }
      } else if (t >= F_JCR + 3 && t < F_JCR + 6) {
        const int k = t - F_JCR - 3;
        s += g_J[3 * (size_t)cb + k] * g_r[3 * (size_t)cb + k];
      }
    }
    camftf[(size_t)cb * PSTRIDE + t] = s;
    sums[t] = s;
  }
  if (!post) return;   // with several ranks the per-camera sums are all-reduced first, k_cam_post follows the exchange
  __syncthreads();
  if (t < 6) {
    const int i = 6 * cb + t;
    const double d = sums[F_JCJC + t * 6 + t];
    if (mode == 1) diag_c[i] = d;
    else if (!reuse_diag) diag_c[i] = fmin(fmax(d, dmin), dmax);
    gmax_c[i] = fabs(sums[F_JCR + t] / scale_c[i]);
  }
}

// From the (globally summed) camftf: LM diagonal / raw column norms of the camera columns and
// the gradient entries |g / scale|.
__global__ void k_cam_post(int ncb, const double* __restrict__ camftf, double* __restrict__ diag_c,
                           const double* __restrict__ scale_c, int reuse_diag, int mode, double dmin, double dmax,
                           double* __restrict__ gmax_c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 6 * ncb) return;
  const int cb = i / 6, a = i % 6;
  const double* f = camftf + (size_t)cb * PSTRIDE;
  const double s = f[F_JCJC + a * 6 + a];
  if (mode == 1) diag_c[i] = s;
  else if (!reuse_diag) diag_c[i] = fmin(fmax(s, dmin), dmax);
  gmax_c[i] = fabs(f[F_JCR + a] / scale_c[i]);
}

// Per intrinsics block: sum camftf(Jm^T Jm | Jm^T r) over its cameras (one wave, lanes strided,
// fixed butterfly) -> modelsum[mb][12]; LM diagonal / raw norms; gradient.
__global__ __launch_bounds__(64) void k_modelsum(const int* __restrict__ mcam_first, const int* __restrict__ mcam,
                                                  const double* __restrict__ camftf, double* __restrict__ modelsum,
                                                  double* __restrict__ diag_m, const double* __restrict__ scale_m, int reuse_diag,
                                                  int mode, double dmin, double dmax, double* __restrict__ gmax_m) {
  const int mb = blockIdx.x, lane = threadIdx.x;
  double acc[12];
#pragma unroll
  for (int k = 0; k < 12; k++) acc[k] = 0.0;
  // (a lane's cameras q, q + 64, ... in that order; the records of FOUR of them are asked for before the first is added - the
  //  launch is one wave whose time is its dependent load rounds: 8 rounds of 12 loads at config 3, 12.7 us, before)
  const int q1 = mcam_first[mb + 1];
  for (int q = mcam_first[mb] + lane; q < q1; q += 256) {
    double v[4][12];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int qq = q + 64 * j;
      const double* f = camftf + (size_t)mcam[min(qq, q1 - 1)] * PSTRIDE;
#pragma unroll
      for (int k = 0; k < 9; k++) v[j][k] = f[F_JMJM + k];
#pragma unroll
      for (int k = 0; k < 3; k++) v[j][9 + k] = f[F_JMR + k];
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (q + 64 * j < q1) {
#pragma unroll
        for (int k = 0; k < 12; k++) acc[k] += v[j][k];
      }
  }
#pragma unroll
  for (int k = 0; k < 12; k++) acc[k] = wave_sum(acc[k]);
  if (lane == 0) {
    for (int k = 0; k < 12; k++) modelsum[12 * (size_t)mb + k] = acc[k];
    for (int a = 0; a < 3; a++) {
      const double s = acc[a * 3 + a];
      if (mode == 1) diag_m[3 * mb + a] = s;
      else if (!reuse_diag) diag_m[3 * mb + a] = fmin(fmax(s, dmin), dmax);
      gmax_m[3 * mb + a] = fabs(acc[9 + a] / scale_m[3 * mb + a]);
    }
  }
}

// jacobian_scaling = 1 / (1 + sqrt(squared column norm))  (Ceres, iteration 0)
__global__ void k_make_scale(int n, const double* __restrict__ norm2, double* __restrict__ scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) scale[i] = 1.0 / (1.0 + sqrt(norm2[i]));
}

// --------------------------------------------------------------------------------------
// k_pairs: one wave per chunk of a block-pair list: sum_e A[pa[e]] (DA x 3) * B[pb[e]]^T (3 x DB).
// --------------------------------------------------------------------------------------
// SPARSE: the list of a problem whose products were formed inside k_point (FoldTables) - only a few entries of a visited chunk
// are still live (0.4 % at config 3).  All index words of the chunk are asked for at once and the sixty-four-entry steps
// without a live entry are skipped as a wave, instead of sixteen dependent load -> test rounds (the three residue launches
// were 20 us each for a few thousand products).
template <int DA, int DB, bool WITH_U, bool SPARSE = false>
__device__ __forceinline__ void pairs_body(int blk, int nblk, int nchunk, const int* __restrict__ ch_start, const int* __restrict__ ch_end,
                                           const int* __restrict__ pa, const int* __restrict__ pb,
                                           const double* __restrict__ TA, const double* __restrict__ TB,
                                           const double* __restrict__ UA, size_t plane, double* __restrict__ partial,
                                           const int* __restrict__ live) {
  constexpr int NOUT = DA * DB + (WITH_U ? DA : 0);
  // XCD-aware chunk order: workgroups b, b+8, ... share an XCD (and its L2), so hand each XCD a
  // contiguous range of chunks — chunks are sorted by (row camera, col camera), a contiguous range
  // keeps re-reading the same cameras' T segments (speed only; any mapping is correct)
  const int nwg = nblk, xcd = blk & 7, loc = blk >> 3;
  const int qn = nwg >> 3, rm = nwg & 7;
  const int swz = (xcd < rm ? xcd * (qn + 1) : rm * (qn + 1) + (xcd - rm) * qn) + loc;
  const int cidx = swz * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (cidx >= nchunk) return;
  const int chunk = live ? live[cidx] : cidx;   // (FoldTables: only the chunks that still hold a live entry are visited)
  double acc[NOUT];
#pragma unroll
  for (int k = 0; k < NOUT; k++) acc[k] = 0.0;
  const int e0 = ch_start[chunk] + lane, e1 = ch_end[chunk];
  constexpr int NPRE = SPARSE ? CHUNK / 64 : 1;
  int pre_a[NPRE];
  if constexpr (SPARSE) {
#pragma unroll
    for (int k = 0; k < NPRE; k++) pre_a[k] = e0 + 64 * k < e1 ? pa[e0 + 64 * k] : -1;
  }
#pragma unroll 1
  for (int e = e0, step = 0; e < e1 || (SPARSE && step < NPRE); e += 64, step++) {
    int ia;
    if constexpr (SPARSE) {
      if (step >= NPRE) break;
      ia = -1;
#pragma unroll
      for (int k = 0; k < NPRE; k++) if (k == step) ia = pre_a[k];
      if (__builtin_amdgcn_ballot_w64(ia >= 0) == 0ull) continue;   // nothing live in these sixty-four entries
    } else {
      ia = pa[e];
    }
    if (ia < 0) continue;   // formed inside k_point (FoldTables)
    const int ib = pb[e];
    double ta[DA * 3], tb[DB * 3];
    // 6-row records (camera blocks) live component-major, T[k][position] with `plane` positions per component;
    // 3-row records (intrinsics entries) are small contiguous records
    if constexpr (DA == 6) {
#pragma unroll
      for (int k = 0; k < 18; k++) ta[k] = TA[(size_t)k * plane + ia];
    } else {
      const double* pa_ = TA + (size_t)ia * (DA * 3);
#pragma unroll
      for (int k = 0; k < DA * 3; k++) ta[k] = pa_[k];
    }
    if constexpr (DB == 6) {
#pragma unroll
      for (int k = 0; k < 18; k++) tb[k] = TB[(size_t)k * plane + ib];
    } else {
      const double* pb_ = TB + (size_t)ib * (DB * 3);
#pragma unroll
      for (int k = 0; k < DB * 3; k++) tb[k] = pb_[k];
    }
#pragma unroll
    for (int a = 0; a < DA; a++)
#pragma unroll
      for (int b = 0; b < DB; b++)
        acc[a * DB + b] += ta[a * 3] * tb[b * 3] + ta[a * 3 + 1] * tb[b * 3 + 1] + ta[a * 3 + 2] * tb[b * 3 + 2];
    if (WITH_U && ia == ib) {
#pragma unroll
      for (int a = 0; a < DA; a++) acc[DA * DB + a] += UA[(size_t)ia * DA + a];
    }
  }
  wave_reduce_store<NOUT>(acc, partial + (size_t)chunk * NOUT, lane);
}
template <int DA, int DB, bool WITH_U, bool SPARSE = false>
__global__ __launch_bounds__(256) void k_pairs(int nchunk, const int* __restrict__ ch_start, const int* __restrict__ ch_end,
                                                const int* __restrict__ pa, const int* __restrict__ pb,
                                                const double* __restrict__ TA, const double* __restrict__ TB,
                                                const double* __restrict__ UA, size_t plane, double* __restrict__ partial,
                                                const int* __restrict__ live = nullptr) {
  pairs_body<DA, DB, WITH_U, SPARSE>(blockIdx.x, gridDim.x, nchunk, ch_start, ch_end, pa, pb, TA, TB, UA, plane, partial, live);
}

// --------------------------------------------------------------------------------------
// Assembly of the padded dense system M (row-major npad x npad, lower triangle; row n = rhs).
// --------------------------------------------------------------------------------------
// Zero fill of the reduced system before the assembly: only the 64 x 64 tiles of the lower triangle that the factorisation
// reads or writes.  With an elimination tree those are, for a column block, the rows of its own node, of the nodes above it
// and of the root chain (everything else is structurally zero and never touched: at config 3 one tile in five of the square).
// lev[b] / lo[b] / hi[b]: level and leaf interval of the node that owns 64-block b (root: level 127, every leaf).
struct ZeroMap { unsigned char lev[256]; short lo[256], hi[256]; int nb; };
__device__ __forceinline__ void zero_tile_body(int tile, double* __restrict__ M, int ld, const ZeroMap& Z) {
  int I = (int)((sqrt(8.0 * tile + 1.0) - 1.0) * 0.5);
  while (I * (I + 1) / 2 > tile) I--;
  while ((I + 1) * (I + 2) / 2 <= tile) I++;
  const int J = tile - I * (I + 1) / 2;
  const bool same = Z.lev[I] == Z.lev[J] && Z.lo[I] == Z.lo[J] && Z.hi[I] == Z.hi[J];
  const bool above = Z.lev[I] > Z.lev[J] && Z.lo[I] <= Z.lo[J] && Z.hi[I] >= Z.hi[J];
  if (!same && !above) return;
  double2* base = reinterpret_cast<double2*>(M + (size_t)(64 * I) * ld + 64 * J);
  for (int e = threadIdx.x; e < 64 * 32; e += 256) base[(size_t)(e >> 5) * (ld / 2) + (e & 31)] = make_double2(0.0, 0.0);
}
__global__ __launch_bounds__(256) void k_zero_system(double* __restrict__ M, int ld, ZeroMap Z) { zero_tile_body(blockIdx.x, M, ld, Z); }

// With the Schur products formed inside k_point (FoldTables) what is left between k_point and the assembly is the per-camera
// sums (k_ftf's chunks), a residue of the three pair lists (a few thousand live entries; the intrinsics x intrinsics list in
// full: one 3 x 3 product per point) and the zero fill of the reduced system - rounds 3-4 ran the residue and the fill as
// four launches on a second stream beside k_ftf: two events on the main stream (13 + 6 us of idle time around them in the
// kernel trace) and three 19 us launches for next to no work.  Here they are ONE launch on the main stream: the residue and
// the zero tiles are its first workgroups and finish in the shadow of the camera chunks.  Same arithmetic per chunk, so the
// partials - and the solve - are bit-identical to the separate launches (MSFM_FUSED_SUMS=0).
struct SumsArgs {
  int n_zero, n_mc_wg, n_mm_wg, n_cc_wg, n_ftf_wg;   // workgroups of each part, in this order
  // zero fill
  double* M; int ld; ZeroMap Z;
  // pair lists: intrinsics x camera (live chunks), intrinsics x intrinsics (all), camera x camera (live chunks)
  int mc_n; const int *mc_start, *mc_end, *mc_pa, *mc_pb, *mc_live; double* mc_partial;
  int mm_n; const int *mm_start, *mm_end, *mm_pa, *mm_pb; double* mm_partial;
  int cc_n; const int *cc_start, *cc_end, *cc_pa, *cc_pb, *cc_live; double* cc_partial;
  const double *T, *Tm, *Tmu; size_t plane;
  // per-camera sums
  int f_n; const int *f_start, *f_end; BaPtrs P; CamRows R; const double* Tu; const int* cpos_pb; double* f_partial;
};
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_sums(SumsArgs a) {
  int b = blockIdx.x;
  if (b < a.n_zero) { zero_tile_body(b, a.M, a.ld, a.Z); return; }
  b -= a.n_zero;
  if (b < a.n_mc_wg) {   // (a list without live chunks is a full one: small problems, several intrinsics blocks)
    if (a.mc_live) pairs_body<3, 6, false, true>(b, a.n_mc_wg, a.mc_n, a.mc_start, a.mc_end, a.mc_pa, a.mc_pb, a.Tm, a.T, nullptr, a.plane, a.mc_partial, a.mc_live);
    else pairs_body<3, 6, false, false>(b, a.n_mc_wg, a.mc_n, a.mc_start, a.mc_end, a.mc_pa, a.mc_pb, a.Tm, a.T, nullptr, a.plane, a.mc_partial, nullptr);
    return;
  }
  b -= a.n_mc_wg;
  if (b < a.n_mm_wg) { pairs_body<3, 3, true, false>(b, a.n_mm_wg, a.mm_n, a.mm_start, a.mm_end, a.mm_pa, a.mm_pb, a.Tm, a.Tm, a.Tmu, (size_t)0, a.mm_partial, nullptr); return; }
  b -= a.n_mm_wg;
  if (b < a.n_cc_wg) {
    if (a.cc_live) pairs_body<6, 6, false, true>(b, a.n_cc_wg, a.cc_n, a.cc_start, a.cc_end, a.cc_pa, a.cc_pb, a.T, a.T, nullptr, a.plane, a.cc_partial, a.cc_live);
    else pairs_body<6, 6, false, false>(b, a.n_cc_wg, a.cc_n, a.cc_start, a.cc_end, a.cc_pa, a.cc_pb, a.T, a.T, nullptr, a.plane, a.cc_partial, nullptr);
    return;
  }
  b -= a.n_cc_wg;
  ftf_body(b, a.f_n, a.f_start, a.f_end, a.P, a.R, a.Tu, a.cpos_pb, a.f_partial);
}

// camera-camera blocks: one block per 64-thread workgroup; 18 threads hold two neighbouring entries of a row each (16-byte
// loads of the partials, one 16-byte store into M), and THREE such groups share the block's list of fold partials (group g
// sums partials g, g + 3, ...; the three sums meet in group 0 at the end).  The launch is bound by the dependent load rounds
// of its longest block, not by bytes: a camera's diagonal block has a partial from every workgroup of k_point that sees the
// camera (138 on average at config 3).  History at config 3: one block per wave, four loads per round 0.055 ms (three blocks
// per wave; 0.052 with the zero chunk partials skipped), sixteen loads per round 0.042, the list split over three groups 0.0xx.
__device__ __forceinline__ void asm_cc(int b, int sub, int t2, const int* __restrict__ blk_row, const int* __restrict__ blk_col,
                                       const int* __restrict__ blk_chunk_first, const uint8_t* __restrict__ blk_live, const double* __restrict__ partial,
                                       const int* __restrict__ blk_fold_range, const double* __restrict__ fold_partial,
                                       const double* __restrict__ camftf, const double* __restrict__ diag_c,
                                       double radius, const int* __restrict__ cb_off, double* __restrict__ M, int ld, int lead) {
  const int rb = blk_row[b], cbk = blk_col[b];
  const int t = 2 * t2;
  double s0 = 0.0, s1 = 0.0;
  // (blk_chunk_first nullptr: every entry was folded into k_point, the gather kernel did not run; blk_live: which blocks still
  //  have an entry on the gather path - the chunk partials of the others are zero and their two dependent loads are skipped)
  if (sub == 0 && blk_chunk_first && (!blk_live || blk_live[b]))
    for (int ch = blk_chunk_first[b]; ch < blk_chunk_first[b + 1]; ch++) {
      const double2 v = *reinterpret_cast<const double2*>(partial + (size_t)ch * 36 + t);
      s0 += v.x; s1 += v.y;
    }
  if (blk_fold_range) {   // the products formed inside k_point: one partial per (workgroup, block), in workgroup order
    // four running sums per group (its partials 0, 1, 2, 3 mod 4), sixteen loads in flight per round, combined in a fixed order
    double2 q0 = make_double2(0.0, 0.0), q1 = q0, q2 = q0, q3 = q0;
    const int f0 = blk_fold_range[2 * b], f1 = sub < 3 ? blk_fold_range[2 * b + 1] : 0;
    const double2* fp = reinterpret_cast<const double2*>(fold_partial + t);
    int sl = f0 + sub;
    for (; sl + 45 < f1; sl += 48) {
      double2 v[16];
#pragma unroll
      for (int j = 0; j < 16; j++) v[j] = fp[(size_t)(sl + 3 * j) * 18];
#pragma unroll
      for (int j = 0; j < 16; j += 4) {
        q0.x += v[j].x; q0.y += v[j].y; q1.x += v[j + 1].x; q1.y += v[j + 1].y;
        q2.x += v[j + 2].x; q2.y += v[j + 2].y; q3.x += v[j + 3].x; q3.y += v[j + 3].y;
      }
    }
    for (; sl + 9 < f1; sl += 12) {
      const double2 v0 = fp[(size_t)sl * 18], v1 = fp[(size_t)(sl + 3) * 18], v2 = fp[(size_t)(sl + 6) * 18], v3 = fp[(size_t)(sl + 9) * 18];
      q0.x += v0.x; q0.y += v0.y; q1.x += v1.x; q1.y += v1.y; q2.x += v2.x; q2.y += v2.y; q3.x += v3.x; q3.y += v3.y;
    }
    for (; sl < f1; sl += 3) { const double2 v = fp[(size_t)sl * 18]; q0.x += v.x; q0.y += v.y; }
    const double g0 = (q0.x + q1.x) + (q2.x + q3.x), g1 = (q0.y + q1.y) + (q2.y + q3.y);
    // groups 1 and 2 hand their sums to group 0 (every lane of the wave takes part in the exchange)
    const int lane = (int)threadIdx.x;
    const double a0 = __shfl(g0, lane + 18), a1 = __shfl(g1, lane + 18);
    const double b0 = __shfl(g0, lane + 36), b1 = __shfl(g1, lane + 36);
    s0 += (g0 + a0) + b0;
    s1 += (g1 + a1) + b1;
  }
  if (sub != 0) return;
  double v0 = -s0, v1 = -s1;
  const int a = t / 6, c = t % 6;
  if (rb == cbk && lead) {
    const double2 f = *reinterpret_cast<const double2*>(camftf + (size_t)rb * PSTRIDE + F_JCJC + t);
    v0 += f.x; v1 += f.y;
    if (a == c) { const double q = sqrt(diag_c[6 * rb + a] / radius); v0 += q * q; }
    if (a == c + 1) { const double q = sqrt(diag_c[6 * rb + a] / radius); v1 += q * q; }
  }
  *reinterpret_cast<double2*>(&M[(size_t)(cb_off[rb] + a) * ld + cb_off[cbk] + c]) = make_double2(v0, v1);
}

// intrinsics-camera blocks (rows 6*ncb + 3*mb.., cols 6*cb..): 18 entries; three groups of 18 threads share the list of
// fold partials as in asm_cc.
__device__ __forceinline__ void asm_mc(int b, const int* __restrict__ blk_row, const int* __restrict__ blk_col,
                                       const int* __restrict__ blk_chunk_first, const double* __restrict__ partial,
                                       const int* __restrict__ fold_range, const double* __restrict__ fold_partial,
                                       const double* __restrict__ camftf, const int* __restrict__ cb_mb,
                                       const int* __restrict__ cb_off, int mo, double* __restrict__ M, int ld, int lead) {
  const int lane = threadIdx.x, sub = lane / 18, t = lane - 18 * sub;
  const int mb = blk_row[b], cb = blk_col[b];
  double s = 0.0;
  if (sub == 0 && blk_chunk_first)
    for (int ch = blk_chunk_first[b]; ch < blk_chunk_first[b + 1]; ch++) s += partial[(size_t)ch * 18 + t];
  if (fold_range) {   // the products formed inside k_point: one partial per (workgroup, camera), in workgroup order
    double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
    const int f0 = fold_range[2 * b], f1 = sub < 3 ? fold_range[2 * b + 1] : 0;
    int sl = f0 + sub;
    for (; sl + 45 < f1; sl += 48) {
      double v[16];
#pragma unroll
      for (int j = 0; j < 16; j++) v[j] = fold_partial[(size_t)(sl + 3 * j) * 18 + t];
#pragma unroll
      for (int j = 0; j < 16; j += 4) { q0 += v[j]; q1 += v[j + 1]; q2 += v[j + 2]; q3 += v[j + 3]; }
    }
    for (; sl + 9 < f1; sl += 12) {
      const double v0 = fold_partial[(size_t)sl * 18 + t], v1 = fold_partial[(size_t)(sl + 3) * 18 + t];
      const double v2 = fold_partial[(size_t)(sl + 6) * 18 + t], v3 = fold_partial[(size_t)(sl + 9) * 18 + t];
      q0 += v0; q1 += v1; q2 += v2; q3 += v3;
    }
    for (; sl < f1; sl += 3) q0 += fold_partial[(size_t)sl * 18 + t];
    const double g = (q0 + q1) + (q2 + q3);
    const double ga = __shfl(g, lane + 18), gb = __shfl(g, lane + 36);
    s += (g + ga) + gb;
  }
  if (sub != 0) return;
  double v = -s;
  if (lead && cb_mb[cb] == mb) v += camftf[(size_t)cb * PSTRIDE + F_JMJC + t];
  const int a = t / 6, c = t % 6;
  M[(size_t)(mo + 3 * mb + a) * ld + cb_off[cb] + c] = v;
}

// intrinsics-intrinsics blocks: one wave per block, lanes strided over chunks.
// partial layout per chunk: 9 (Tm Tm'^T) + 3 (sum Tm.u, self pairs only).
__device__ __forceinline__ void asm_mm(int b, const int* __restrict__ blk_row, const int* __restrict__ blk_col,
                                       const int* __restrict__ blk_chunk_first, const double* __restrict__ partial,
                                       const double* __restrict__ modelsum, const double* __restrict__ diag_m, double radius,
                                       int mo, int n, double* __restrict__ M, int ld, int lead) {
  const int lane = threadIdx.x;
  const int rb = blk_row[b], cbk = blk_col[b];
  double acc[12];
#pragma unroll
  for (int k = 0; k < 12; k++) acc[k] = 0.0;
  for (int ch = blk_chunk_first[b] + lane; ch < blk_chunk_first[b + 1]; ch += 64)
#pragma unroll
    for (int k = 0; k < 12; k++) acc[k] += partial[(size_t)ch * 12 + k];
#pragma unroll
  for (int k = 0; k < 12; k++) acc[k] = wave_sum(acc[k]);
  if (lane < 9) {
    const int a = lane / 3, c = lane % 3;
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 9; k++) if (k == lane) v = -acc[k];
    if (rb == cbk && lead) {
      v += modelsum[12 * (size_t)rb + lane];
      if (a == c) { const double q = sqrt(diag_m[3 * rb + a] / radius); v += q * q; }
    }
    M[(size_t)(mo + 3 * rb + a) * ld + mo + 3 * cbk + c] = v;
  } else if (lane < 12 && rb == cbk) {
    const int a = lane - 9;
    double u = 0.0;
#pragma unroll
    for (int k = 9; k < 12; k++) if (k == lane) u = acc[k];
    M[(size_t)n * ld + mo + 3 * rb + a] = (lead ? modelsum[12 * (size_t)rb + 9 + a] : 0.0) - u;  // rhs row
  }
}

// rhs of the camera columns + intrinsics blocks that have no (point, intrinsics) entries.
// camftf is already global (summed over ranks), so only the lead rank contributes it.
__device__ __forceinline__ void asm_rhs_cam(int i, int ncb, const double* __restrict__ camftf, const int* __restrict__ cb_off,
                                            double* __restrict__ M, int ld, int n, int lead) {
  if (i >= 6 * ncb) return;
  const int cb = i / 6, a = i % 6;
  const double* f = camftf + (size_t)cb * PSTRIDE;
  M[(size_t)n * ld + cb_off[cb] + a] = lead ? f[F_JCR + a] - f[F_TU + a] : 0.0;
}

// The four assembly passes in one launch of 64-thread workgroups: [0, n_cc) camera-camera blocks, then intrinsics-camera
// blocks, intrinsics-intrinsics blocks, and the camera part of the rhs row (64 entries per workgroup).
struct AsmArgs {
  int n_cc, n_mc, n_mm, n_rhs, n_padcol;   // n_padcol > 0 only on a single rank (with several, the padding follows the exchange)
  const int* padcol;
  const int* cc_fold_first;
  const uint8_t* cc_live;
  const double* cc_fold_partial;
  const int* mc_fold_range;
  const double* mc_fold_partial;
  const int *cc_row, *cc_col, *cc_first, *mc_row, *mc_col, *mc_first, *mm_row, *mm_col, *mm_first, *cb_mb, *cb_off;
  const double *cc_partial, *mc_partial, *mm_partial, *camftf, *diag_c, *modelsum, *diag_m;
  double radius;
  int ncb, mo, n, ld, lead;
  double* M;
};
__global__ __launch_bounds__(64) void k_asm_all(AsmArgs a) {
  int b = blockIdx.x;
  if (b < a.n_cc) {
    const int sub = (int)threadIdx.x / 18;
    asm_cc(b, sub, (int)threadIdx.x - 18 * sub, a.cc_row, a.cc_col, a.cc_first, a.cc_live, a.cc_partial, a.cc_fold_first, a.cc_fold_partial, a.camftf, a.diag_c, a.radius, a.cb_off, a.M, a.ld, a.lead);
    return;
  }
  b -= a.n_cc;
  if (b < a.n_mc) { asm_mc(b, a.mc_row, a.mc_col, a.mc_first, a.mc_partial, a.mc_fold_range, a.mc_fold_partial, a.camftf, a.cb_mb, a.cb_off, a.mo, a.M, a.ld, a.lead); return; }
  b -= a.n_mc;
  if (b < a.n_mm) { asm_mm(b, a.mm_row, a.mm_col, a.mm_first, a.mm_partial, a.modelsum, a.diag_m, a.radius, a.mo, a.n, a.M, a.ld, a.lead); return; }
  b -= a.n_mm;
  if (b < a.n_rhs) { asm_rhs_cam(b * 64 + (int)threadIdx.x, a.ncb, a.camftf, a.cb_off, a.M, a.ld, a.n, a.lead); return; }
  b -= a.n_rhs;
  const int i = b * 64 + (int)threadIdx.x;   // identity on the padding columns (k_pad_diag)
  if (i < a.n_padcol) a.M[(size_t)a.padcol[i] * a.ld + a.padcol[i]] = 1.0;
}

// identity on the padding columns that align the camera domains to 64 (their solution component is 0)
__global__ void k_pad_diag(int npadcol, const int* __restrict__ padcol, double* __restrict__ M, int ld) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npadcol) M[(size_t)padcol[i] * ld + padcol[i]] = 1.0;
}

// candidate buffers start as copies of x so that inactive blocks carry over: the three copies in one launch (run start)
__global__ __launch_bounds__(256) void k_copy3(size_t n0, const double* __restrict__ a0, double* __restrict__ b0, size_t n1,
                                               const double* __restrict__ a1, double* __restrict__ b1, size_t n2,
                                               const double* __restrict__ a2, double* __restrict__ b2) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n0) b0[i] = a0[i];
  if (i < n1) b1[i] = a1[i];
  if (i < n2) b2[i] = a2[i];
}


__global__ void k_scale_scal(double* __restrict__ scal, int slot, double f) { scal[slot] *= f; }

// --------------------------------------------------------------------------------------
// After the reduced solve: camera / intrinsics update, point back substitution, model cost.
// z = solution of S z = rhs (scaled space); step = -z.
// --------------------------------------------------------------------------------------
// The camera and intrinsics part of the step in one launch: the solution leaves the solver's elimination order (zsys) for
// the block order of the BA kernels (z, read by k_backsub / k_mcc_rest), with the finiteness check on the way, and the
// candidates x - z * scale go to the candidate buffers.  (Inactive blocks of the candidate buffers were filled once, at the
// start of the run: nothing writes them.)  One launch instead of k_gather_z + k_copy3 + 2 x k_update_blocks.
__global__ __launch_bounds__(256) void k_update_params(int ncb, int nmb, const int* __restrict__ cb_cam, const int* __restrict__ mb_model,
                                                        const int* __restrict__ cb_off, int mo, const double* __restrict__ zsys,
                                                        const double* __restrict__ scale_c, const double* __restrict__ scale_m,
                                                        const double* __restrict__ cam, double* __restrict__ cam_c,
                                                        const double* __restrict__ model, double* __restrict__ model_c, double* __restrict__ z,
                                                        int* __restrict__ fail, double* __restrict__ dx2_partial, double* __restrict__ x2_partial,
                                                        double count_weight) {
  __shared__ double sh[4];
  const int i = blockIdx.x * 256 + threadIdx.x;
  double dx2 = 0.0, x2 = 0.0;
  if (i < 6 * ncb + 3 * nmb) {
    double zv, sc, xv;
    double* dst;
    if (i < 6 * ncb) {
      const int b = i / 6, a = i % 6;
      zv = zsys[cb_off[b] + a];
      sc = scale_c[i];
      const size_t k = (size_t)cb_cam[b] * 6 + a;
      xv = cam[k];
      dst = cam_c + k;
    } else {
      const int j = i - 6 * ncb, b = j / 3, a = j % 3;
      zv = zsys[mo + j];
      sc = scale_m[j];
      const size_t k = (size_t)mb_model[b] * 3 + a;
      xv = model[k];
      dst = model_c + k;
    }
    z[i] = zv;
    if (!isfinite(zv)) atomicOr(fail, 4);
    const double cand = xv + (-zv) * sc;
    *dst = cand;
    const double d = cand - xv;
    dx2 = d * d * count_weight;  // replicated blocks are counted by the lead rank only
    x2 = xv * xv * count_weight;
  }
  const double t1 = block_sum256(dx2, sh);
  const double t2 = block_sum256(x2, sh);
  if (threadIdx.x == 0) { dx2_partial[blockIdx.x] = t1; x2_partial[blockIdx.x] = t2; }
}

struct BackPtrs {
  BaPtrs B;   // the rows are linearised again (obs_linearize) at the point the reduced system was built at
  int npb, ncb;
  const int *pt_first, *pb_pt;
  const double *ptL, *z;
  double* pt_c;
  double* cm_Xc;   // the candidate's coordinates again, at the camera-major positions of the point's rows (CamRows::X)
  int Nc;                  // the first Nc threads of the launch also prepare the candidate cameras' rotations
  const double* cam_c;     // (final before this launch: k_update_params)
  double* rot_c;
};

// 8 lanes per point, lane = observation, single pass: besides y = sum Jp^T (r + q) with
// q = -(Jc z_c + Jm z_m), the group accumulates the moments that give this point's model cost change
//   -(J s)^T (r + J s / 2) = -[ sp.g + sum q.r + 1/2 sp^T V sp + sp.(sum Jp^T q) + 1/2 sum q.q ]
// so the step sp (known only after the group reduction) never needs a second sweep.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_backsub(BackPtrs P, double* __restrict__ mcc_partial, double* __restrict__ dx2_partial,
                                                  double* __restrict__ x2_partial) {
  __shared__ double sh[4];
  const int tid = threadIdx.x, sub = tid & 7;
  const int pb = blockIdx.x * 32 + (tid >> 3);
  const bool act = pb < P.npb;
  for (int c = blockIdx.x * 256 + tid; c < P.Nc; c += gridDim.x * 256) msfm_rot_prepare(P.cam_c + 6 * (size_t)c, P.rot_c + 4 * (size_t)c);
  int f = 0, l = 0;
  if (act) { f = P.pt_first[pb]; l = P.pt_first[pb + 1]; }
  double V00 = 0, V10 = 0, V11 = 0, V20 = 0, V21 = 0, V22 = 0, g0 = 0, g1 = 0, g2 = 0, h0 = 0, h1 = 0, h2 = 0, qr = 0, qq = 0;
  for (int base = f; base < l; base += 8) {
    const int i = base + sub;
    if (i < l) {
      // rows of frozen blocks come out as zeros, so no branches
      double jp[6], r0, r1, q0, q1;
#ifdef MSFM_BACKSUB_DIRECTIONAL
      // Round 5, measured and NOT the default: q = -(J_c z_c + J_m z_m) as the directional derivative of the row's residual along
      // the (unscaled) camera / intrinsics step instead of eighteen Jacobian columns contracted with it (msfm_reproj_dir): the
      // same corrected, scaled quantities to rounding (every parity test green), 164 binary64 operations per row less (710 -> 546
      // in the kernel's stream) - and the kernel SLOWER, 101 -> 117 us at config 3 (1 077 -> 1 056 iterations/s): the step and
      // the scales now stand in front of the projection instead of behind it, one more level in the chain of dependent gathers
      // of a kernel that waits for memory at three waves per SIMD.  A frozen block has a zero scale, hence a zero step.
      {
        const BaPtrs& B = P.B;
        const int c = B.o_cam[i], m = B.o_model[i], p = B.o_pt[i];
        const int cb = B.o_cb[i], mb = B.o_mb[i], pbk = B.o_pb[i];
        double pose[6], rc[4], cm[3], X[3];
#pragma unroll
        for (int j = 0; j < 6; j++) pose[j] = B.cam[6 * (size_t)c + j];
#pragma unroll
        for (int j = 0; j < 4; j++) rc[j] = B.rot[4 * (size_t)c + j];
#pragma unroll
        for (int j = 0; j < 3; j++) { cm[j] = B.model[3 * (size_t)m + j]; X[j] = B.pt[3 * (size_t)p + j]; }
        const double* zcp = P.z + 6 * (cb >= 0 ? cb : 0);
        const double* zmp = P.z + 6 * P.ncb + 3 * (mb >= 0 ? mb : 0);
        double zw[3], zt[3], zmv[3], sp[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
          zw[j] = cb >= 0 ? B.scale_c[6 * cb + j] * zcp[j] : 0.0;
          zt[j] = cb >= 0 ? B.scale_c[6 * cb + 3 + j] * zcp[3 + j] : 0.0;
          zmv[j] = mb >= 0 ? B.scale_m[3 * mb + j] * zmp[j] : 0.0;
          sp[j] = pbk >= 0 ? B.scale_p[3 * (size_t)pbk + j] : 0.0;
        }
        double r[2], Jp[6], d[2];
        msfm_reproj_dir(pose, rc, cm, X, B.o_x[i], B.o_y[i], B.o_w[i], zw, zt, zmv, r, Jp, d);
        double rho0, rho1;
        msfm_huber(B.huber, r[0] * r[0] + r[1] * r[1], rho0, rho1);
        const double sq = sqrt(rho1);
        r0 = sq * r[0]; r1 = sq * r[1];
        q0 = -(sq * d[0]); q1 = -(sq * d[1]);
#pragma unroll
        for (int j = 0; j < 3; j++) { const double spj = sq * sp[j]; jp[j] = spj * Jp[j]; jp[3 + j] = spj * Jp[3 + j]; }
      }
#else
      {
        const int cb = P.B.o_cb[i], mb = P.B.o_mb[i];
        double jc[12], jm[6], zc[6], zm[3];
        obs_linearize(P.B, i, r0, r1, jc, jm, jp);
        __builtin_amdgcn_sched_barrier(0);   // the step is fetched after the row is formed: nine registers less before
        const double* zcp = P.z + 6 * (cb >= 0 ? cb : 0);
        const double* zmp = P.z + 6 * P.ncb + 3 * (mb >= 0 ? mb : 0);
#pragma unroll
        for (int j = 0; j < 6; j++) zc[j] = zcp[j];
#pragma unroll
        for (int j = 0; j < 3; j++) zm[j] = zmp[j];
        q0 = 0.0; q1 = 0.0;
#pragma unroll
        for (int j = 0; j < 6; j++) { q0 -= jc[j] * zc[j]; q1 -= jc[6 + j] * zc[j]; }
#pragma unroll
        for (int j = 0; j < 3; j++) { q0 -= jm[j] * zm[j]; q1 -= jm[3 + j] * zm[j]; }
      }
#endif
      const double a0 = jp[0], a1 = jp[1], a2 = jp[2], b0 = jp[3], b1 = jp[4], b2 = jp[5];
      V00 += a0 * a0 + b0 * b0; V10 += a1 * a0 + b1 * b0; V11 += a1 * a1 + b1 * b1;
      V20 += a2 * a0 + b2 * b0; V21 += a2 * a1 + b2 * b1; V22 += a2 * a2 + b2 * b2;
      g0 += a0 * r0 + b0 * r1; g1 += a1 * r0 + b1 * r1; g2 += a2 * r0 + b2 * r1;
      h0 += a0 * q0 + b0 * q1; h1 += a1 * q0 + b1 * q1; h2 += a2 * q0 + b2 * q1;
      qr += q0 * r0 + q1 * r1;
      qq += q0 * q0 + q1 * q1;
    }
  }
  GROUP_SUM(V00) GROUP_SUM(V10) GROUP_SUM(V11) GROUP_SUM(V20) GROUP_SUM(V21) GROUP_SUM(V22)
  GROUP_SUM(g0) GROUP_SUM(g1) GROUP_SUM(g2) GROUP_SUM(h0) GROUP_SUM(h1) GROUP_SUM(h2) GROUP_SUM(qr) GROUP_SUM(qq)
  double mcc = 0.0, dx2 = 0.0, x2 = 0.0, cand0 = 0.0, cand1 = 0.0, cand2 = 0.0;
  if (act && sub == 0) {
    const double* L = P.ptL + 6 * (size_t)pb;
    const double i00 = L[0], l10 = L[1], i11 = L[2], l20 = L[3], l21 = L[4], i22 = L[5];   // (1 / l00, 1 / l11, 1 / l22: k_point)
    // (L L^T) y' = y,  y = g + h
    double q0 = (g0 + h0) * i00;
    double q1 = ((g1 + h1) - l10 * q0) * i11;
    double q2 = ((g2 + h2) - l20 * q0 - l21 * q1) * i22;
    q2 = q2 * i22;
    q1 = (q1 - l21 * q2) * i11;
    q0 = (q0 - l10 * q1 - l20 * q2) * i00;
    const double s0 = -q0, s1 = -q1, s2 = -q2;  // step (scaled space)
    const double* sc = P.B.scale_p + 3 * (size_t)pb;
    const size_t p = P.pb_pt[pb];
    const double x0 = P.B.pt[3 * p], x1 = P.B.pt[3 * p + 1], x2v = P.B.pt[3 * p + 2];
    const double c0 = x0 + s0 * sc[0], c1 = x1 + s1 * sc[1], c2 = x2v + s2 * sc[2];
    P.pt_c[3 * p] = c0; P.pt_c[3 * p + 1] = c1; P.pt_c[3 * p + 2] = c2;
    cand0 = c0; cand1 = c1; cand2 = c2;
    dx2 = (c0 - x0) * (c0 - x0) + (c1 - x1) * (c1 - x1) + (c2 - x2v) * (c2 - x2v);
    x2 = x0 * x0 + x1 * x1 + x2v * x2v;
    const double sVs = s0 * (V00 * s0 + 2.0 * (V10 * s1 + V20 * s2)) + s1 * (V11 * s1 + 2.0 * V21 * s2) + s2 * V22 * s2;
    mcc = -((s0 * g0 + s1 * g1 + s2 * g2) + qr + 0.5 * sVs + (s0 * h0 + s1 * h1 + s2 * h2) + 0.5 * qq);
  }
  // the candidate to every row of the point in camera-major order (what k_ftf reads if the step is accepted): lane = row again
  // (measured: trading the records between the lanes first, as k_point does for its T stores, made this SLOWER - 104 against 98 us)
  cand0 = __shfl(cand0, 0, 8); cand1 = __shfl(cand1, 0, 8); cand2 = __shfl(cand2, 0, 8);
  for (int base = f; base < l; base += 8) {
    const int i = base + sub;
    if (i < l) {
      const int cp = P.B.o_cpos[i];
      if (cp >= 0) { double* xr = P.cm_Xc + 3 * (size_t)cp; xr[0] = cand0; xr[1] = cand1; xr[2] = cand2; }
    }
  }
  const double t0 = block_sum256(mcc, sh);
  const double t1 = block_sum256(dx2, sh);
  const double t2 = block_sum256(x2, sh);
  if (threadIdx.x == 0) { mcc_partial[blockIdx.x] = t0; dx2_partial[blockIdx.x] = t1; x2_partial[blockIdx.x] = t2; }
}

// model cost change of rows without an eliminated point (obs [AE, A)) and of the GPS rows.
__device__ __forceinline__ void mcc_rest_block(int blk, int A, int AE, int ncb, const int* __restrict__ o_cb, const int* __restrict__ o_mb,
                                               const double* __restrict__ lin_r, const double* __restrict__ lin_Jc,
                                               const double* __restrict__ lin_Jm, const double* __restrict__ z, int has_gps,
                                               const double* __restrict__ g_r, const double* __restrict__ g_J,
                                               double* __restrict__ mcc_partial, double* sh) {
  const int t = blk * 256 + threadIdx.x;
  const int nrest = A - AE;
  double mcc = 0.0;
  if (t < nrest) {
    const int i = AE + t;
    const size_t As = (size_t)nrest, it = (size_t)t;   // the stored rows are indexed from AE
    double m0 = 0, m1 = 0;
    const int cb = o_cb[i], mb = o_mb[i];
    if (cb >= 0)
      for (int j = 0; j < 6; j++) { const double sj = -z[6 * cb + j]; m0 += lin_Jc[(size_t)j * As + it] * sj; m1 += lin_Jc[(size_t)(6 + j) * As + it] * sj; }
    if (mb >= 0)
      for (int j = 0; j < 3; j++) { const double sj = -z[6 * ncb + 3 * mb + j]; m0 += lin_Jm[(size_t)j * As + it] * sj; m1 += lin_Jm[(size_t)(3 + j) * As + it] * sj; }
    mcc -= m0 * (lin_r[it] + m0 / 2.0) + m1 * (lin_r[As + it] + m1 / 2.0);
  } else if (has_gps && t < nrest + ncb) {
    const int cb = t - nrest;
    for (int k = 0; k < 3; k++) {
      const double m = g_J[3 * (size_t)cb + k] * (-z[6 * cb + 3 + k]);
      mcc -= m * (g_r[3 * (size_t)cb + k] + m / 2.0);
    }
  }
  const double s = block_sum256(mcc, sh);
  if (threadIdx.x == 0) mcc_partial[blk] = s;
}
__global__ __launch_bounds__(256) void k_mcc_rest(int A, int AE, int ncb, const int* __restrict__ o_cb, const int* __restrict__ o_mb,
                                                   const double* __restrict__ lin_r, const double* __restrict__ lin_Jc,
                                                   const double* __restrict__ lin_Jm, const double* __restrict__ z, int has_gps,
                                                   const double* __restrict__ g_r, const double* __restrict__ g_J,
                                                   double* __restrict__ mcc_partial) {
  __shared__ double sh[4];
  mcc_rest_block(blockIdx.x, A, AE, ncb, o_cb, o_mb, lin_r, lin_Jc, lin_Jm, z, has_gps, g_r, g_J, mcc_partial, sh);
}

// What follows the back substitution in ONE launch: the model cost change of the rows without an eliminated point and of the GPS
// rows (k_mcc_rest), the cost at the candidate (k_linearize<false> over all rows) and of its GPS rows (k_gps<false>) - three
// launches of ~5 us each for the window of a new camera, whose sums then share one k_reduce.  The same workgroups with the same
// partial slots as the separate launches (the cost partials in a buffer of their own: both sets are alive at once now).
struct TailArgs {
  int n_mcc, n_cost, n_gps;   // workgroups
  // model cost change of the remaining rows
  int A, AE, ncb; const int *o_cb, *o_mb; const double *lin_r, *lin_Jc, *lin_Jm, *z; int has_gps; const double *g_r, *g_J; double* mcc_partial;
  // cost at the candidate
  BaPtrs P; double* cost_partial;
  const int* cb_cam; const double *cam_c, *gps; double gps_weight, huber; const double* scale_c; double *g_r_w, *g_J_w;
};
__global__ __launch_bounds__(256) void k_tail(TailArgs a) {
  __shared__ double sh[4];
  int b = blockIdx.x;
  if (b < a.n_mcc) { mcc_rest_block(b, a.A, a.AE, a.ncb, a.o_cb, a.o_mb, a.lin_r, a.lin_Jc, a.lin_Jm, a.z, a.has_gps, a.g_r, a.g_J, a.mcc_partial, sh); return; }
  b -= a.n_mcc;
  if (b < a.n_cost) { linearize_block<false>(a.P, 0, b, a.cost_partial, sh); return; }
  b -= a.n_cost;
  gps_block<false>(b, a.ncb, a.cb_cam, a.cam_c, a.gps, a.gps_weight, a.huber, a.scale_c, a.g_r_w, a.g_J_w, a.cost_partial + a.n_cost, sh);
}

__global__ void k_zero_int(int* p) { *p = 0; }
// The iteration's scalars go to the host through mapped pinned memory: sixteen doubles, then (after a system-scope fence) the
// sequence number the host thread is spinning on - no copy engine, no event.  The failure bits were folded into
// scal[S_FAIL] by the last k_reduce, so the flag word is cleared here for the next iteration (one launch less at its start).
// The trust-region decision of the step whose scalars are being handed over (Ceres TrustRegionMinimizer, the order of
// msfm_ba_run's loop): taken HERE, on the device, so that the next linearisation (k_point and the rows of frozen points /
// GPS) can be enqueued before the host has seen anything; the host mirrors the code and takes the radius from here.
enum { LM_NONE = 0, LM_INVALID, LM_PARAM_TOL, LM_FUNC_TOL, LM_ACCEPT, LM_REJECT };
enum { H_SEQ = 16, H_CODE = 17, H_RADIUS = 18 };   // slots of the pinned block behind the sixteen scalars
struct LmDecide {
  int on;           // a step was enqueued with this reduced system
  int fresh;        // the system was built at a new linearisation point: the cost at x is scal[S_XCOST], else x_cost below
  double x_cost, radius;
  double min_relative_decrease, parameter_tolerance, function_tolerance, max_radius, min_radius;
};
// t^3 rounded once (the host's std::pow(t, 3) is correctly rounded in all but astronomically rare cases): the square and the
// product as unevaluated sums, added at the end
__device__ __forceinline__ double cube_rn(double t) {
  const double h = t * t, l = fma(t, t, -h);
  const double p = h * t, e = fma(h, t, -p);
  return p + fma(l, t, e);
}
__global__ __launch_bounds__(64) void k_publish_scalars(const double* __restrict__ scal, double* h_scal, unsigned long long seq, int* fail, LmDecide D,
                                                        double* __restrict__ spec) {
  const int lane = threadIdx.x;
  if (lane < 16) __hip_atomic_store(&h_scal[lane], scal[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (lane == 0) {
    int code = LM_NONE;
    double rnew = D.radius;
    if (D.on) {
      const double mcc = scal[S_MCC], cand = scal[S_COST], dx2 = scal[S_DX2], x2 = scal[S_X2];
      const double xc = D.fresh ? scal[S_XCOST] : D.x_cost;
      const bool solved = (int)scal[S_FAIL] == 0;
      if (!(solved && mcc > 0.0)) code = LM_INVALID;
      else if (sqrt(dx2) <= D.parameter_tolerance * (sqrt(x2) + D.parameter_tolerance)) code = LM_PARAM_TOL;
      else if (fabs(xc - cand) <= D.function_tolerance * xc) code = LM_FUNC_TOL;
      else {
        const double rho = (xc - cand) / mcc;
        if (rho > D.min_relative_decrease) {
          code = LM_ACCEPT;
          rnew = D.radius / fmax(1.0 / 3.0, 1.0 - cube_rn(2.0 * rho - 1.0));
          rnew = fmin(D.max_radius, rnew);
        } else code = LM_REJECT;
      }
    }
    // what follows an accepted step unless the radius has run out (the gradient test needs the new linearisation itself)
    spec[0] = (code == LM_ACCEPT && rnew > D.min_radius) ? 1.0 : 0.0;
    spec[1] = rnew;
    __hip_atomic_store(&h_scal[H_CODE], (double)code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&h_scal[H_RADIUS], rnew, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_system();
  if (lane == 0) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(h_scal + H_SEQ), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (fail) *fail = 0;
  }
}
__global__ void k_fill(int n, double v, double* __restrict__ p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}

// Multi-rank exchange of the camera-camera part of the reduced system: only the 6x6 blocks that
// exist on some rank travel (C3: 9335 blocks = 2.7 MB instead of the 72 MB dense square).
template <bool PACK>
__global__ __launch_bounds__(256) void k_pack_blocks(int nblk, const int* __restrict__ u_row, const int* __restrict__ u_col,
                                                      const int* __restrict__ cb_off, double* __restrict__ M, int ld,
                                                      double* __restrict__ pack) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= nblk * 36) return;
  const int b = e / 36, t = e - 36 * b;
  const size_t at = (size_t)(cb_off[u_row[b]] + t / 6) * ld + cb_off[u_col[b]] + t % 6;
  if (PACK) pack[e] = M[at];
  else M[at] = pack[e];
}

// =======================================================================================
// Host side
// =======================================================================================
// Camera x camera Schur products formed where the T records are produced (round 3).  A k_point workgroup holds the records of
// its 32 points in LDS; the entries (record i, record j) of those points are sorted by block at create time, every distinct
// block of a workgroup is a SLOT, four threads sum a slot's 6 x 6 products in point order and write ONE partial per slot
// (config 3: 268 k slots = 77 MB instead of 4.8 M entry pairs = 1.3 GB of gathered T records).  The assembly adds a
// block's slot partials (ranked by block) to its chunk partials.  Workgroups whose records do not fit (a point with more
// than 16 rows, more than FOLD_OVF second-round records) keep the gather path: their entries stay live in the pair list,
// the others are marked (pa < 0) and skipped by k_pairs.
struct FoldTables {
  bool on = false;
  bool all = false;   // every entry of the camera x camera list is folded: k_pairs<6,6> and its chunk partials are not needed at all
  int n_live = 0;     // chunks of the camera x camera list that still hold a live entry (the others' partials stay zero)
  DevBuf<int> live_chunk;
  int n_wg = 0, n_slots = 0, n_entries = 0, n_pass = 0;
  DevBuf<int> wg_fold, ovf_off, wg_pass_first, slot_rank, blk_range;
  DevBuf<uint8_t> blk_live;   // camera x camera blocks that still have an entry on the gather path
  DevBuf<FoldPass> pass;
  DevBuf<unsigned> stream;   // per pass: slot headers, then entries (FoldPass); FOLD_WORDS words of padding behind the last pass
  DevBuf<double> partial;
  // intrinsics x camera list (one intrinsics block): the records of a camera's diagonal slot are exactly its observations in
  // the workgroup, so the same tables serve; the 3 x 6 partials are indexed by the diagonal slots' ranks
  bool mc_on = false, mc_all = false;
  int n_diag = 0, mc_n_live = 0;
  long long mc_entries_folded = 0;
  DevBuf<int> mc_range, mc_live_chunk;
  DevBuf<double> mc_partial;
};

struct PairJobs {
  int n_pairs = 0, n_chunks = 0, n_blocks = 0;
  DevBuf<int> pa, pb, ch_start, ch_end, blk_row, blk_col, blk_chunk_first;
  DevBuf<double> partial;
  std::vector<int> h_row, h_col;  // block list on the host (union structure across ranks)
};

struct msfm_ba {
  msfm_ctx* ctx = nullptr;
  int Nc = 0, Nm = 0, Np = 0;
  int ncb = 0, nmb = 0, npb = 0, nred = 0, npad = 0;
  // solver layout of the reduced system: camera block cb at column cb_off[cb], intrinsics at mo + 3 mb, order nsys
  int nsys = 0, mo = 0, n_padcol = 0;
  msfm_chol_plan plan;
  msfm_chol_ws* chol_ws = nullptr;   // hand-off state of the persistent panel chain (chol.hip)
  DevBuf<int> cb_off, padcol;
  DevBuf<double> zsys, corners;
  int zflip = 0;   // which half of zsys the next solve writes (the other half is being marked "pending" meanwhile)
  int A = 0, AE = 0, NCR = 0, NPM = 0;
  bool has_gps = false;
  double gps_weight = 0;
  int n_residuals = 0;
  std::vector<int> h_cb_cam, h_mb_model, h_pb_pt;
  DevBuf<double> cam, model, pt, cam_c, model_c, pt_c;
  DevBuf<int> cb_cam, mb_model, pb_pt, cb_mb;
  DevBuf<int> o_cam, o_model, o_pt, o_cb, o_mb, o_pb, o_cpos, o_pm;
  DevBuf<double> o_x, o_y, o_w;
  DevBuf<double> lin_r, lin_Jc, lin_Jm, T, Tu, Tm, Tmu, rot, rot_c;
  DevBuf<int> cm_pt;            // camera-major statics of the rows (k_ftf linearises them again)
  DevBuf<double> cm_xyw;        // [3][NCR]
  DevBuf<double> cm_X, cm_Xc;   // [NCR][3] point coordinates in camera-major order at x / at the candidate (CamRows::X)
  DevBuf<int4> chunk_cam;
  DevBuf<int> cpos_pb;
  DevBuf<double> scale_c, scale_m, scale_p, diag_c, diag_m, diag_p;
  DevBuf<int> pt_first, pm_first, pm_mb;
  DevBuf<double> ptL, ptg;
  // ftf jobs
  int n_fchunks = 0;
  DevBuf<int> f_start, f_end, cam_chunk_first;
  DevBuf<double> f_partial, camftf, modelsum;
  DevBuf<int> mcam_first, mcam;
  PairJobs cc, mc, mm;
  FoldTables fold;
  DevBuf<double> M, Linv, w, z;
  DevBuf<double> gps, g_r, g_J;
  DevBuf<double> partial, partial2, partial3, partial4, gmax_buf, scal, sloc;
  double* swrite = nullptr;  // where the kernels put scalars: scal (one rank) or sloc (partials, summed by reduce_scalars)
  DevBuf<int> fail;
  double* h_scal = nullptr;  // pinned, mapped: [0, 16) scalars, [16] sequence number, [17] decision code, [18] radius (k_publish_scalars)
  DevBuf<double> spec;       // {go, radius} of the device-side decision, read by the launches enqueued ahead of the host (PointPtrs::spec)
  bool spec_on = true;       // MSFM_SPEC=0: the next linearisation is enqueued only after the host has seen the step
  double* h_scal_dev = nullptr;
  unsigned long long scal_seq = 0;
  int* h_fail = nullptr;
  int nblk_obs = 0, nblk_pt = 0;
  bool lin_pending = false;   // run_evaluate(jac) was asked for: the next k_point linearises, writes the camera rows and the cost
  double lin_huber = 1.0;
  double setup_ms = 0;
  int world_at_create = 1;
  // multi-rank: camera-camera blocks present on ANY rank, packed for the per-iteration sum
  int n_ublk = 0;
  DevBuf<int> u_row, u_col;
  DevBuf<double> pack;
};

// --------------------------------------------------------------------------------------
// Elimination order of the camera blocks.  Two cameras are coupled in S when they observe a common
// eliminated point.  If the camera graph falls apart into K domains once a separator is removed, the
// K domains can be factored concurrently (chol.hip, PanelJobs) and only the separator remains a
// single chain.  Nested bisection by BFS level sets from a pseudo-peripheral node (George-Liu):
// deterministic, O(edges) per cut.  Output: label per graph node, 0..K-1 = domain, -1 = separator.
// --------------------------------------------------------------------------------------
struct CamGraph {
  int n = 0;
  std::vector<std::vector<int>> adj;
};

static void graph_levels(const CamGraph& G, const std::vector<char>& in, int start, std::vector<int>& lev) {
  std::fill(lev.begin(), lev.end(), -1);
  std::vector<int> q{start}, nq;
  lev[start] = 0;
  while (!q.empty()) {
    nq.clear();
    for (int u : q)
      for (int v : G.adj[u])
        if (in[v] && lev[v] < 0) { lev[v] = lev[u] + 1; nq.push_back(v); }
    q.swap(nq);
  }
}

// Smallest vertex cover of the bipartite graph between the two one-hop boundaries L and R of a cut (Koenig: from a
// maximum matching, found by augmenting paths; Z = what alternating paths reach from the unmatched nodes of L; the cover
// is (L \ Z) + (R & Z)).  Removing the cover separates the sides, and it is never larger than the smaller boundary.
static void boundary_cover(const CamGraph& G, const std::vector<int>& L, const std::vector<int>& R, const std::vector<int>& r_index /*node -> position in R or -1*/,
                           std::vector<int>& cover) {
  const int nl = (int)L.size(), nr = (int)R.size();
  std::vector<std::vector<int>> e(nl);
  for (int l = 0; l < nl; l++)
    for (int v : G.adj[L[l]]) if (r_index[v] >= 0) e[l].push_back(r_index[v]);
  std::vector<int> ml(nl, -1), mr(nr, -1), seen(nr, -1), par(nr, -1);
  for (int l = 0; l < nl; l++)   // greedy start
    for (int r : e[l]) if (mr[r] < 0) { mr[r] = l; ml[l] = r; break; }
  std::vector<std::pair<int, size_t>> st;
  int stamp = 0;   // what a failed search has visited stays dead until the matching changes
  for (int l0 = 0; l0 < nl; l0++) {
    if (ml[l0] >= 0) continue;
    // iterative depth-first search for an augmenting path from l0; par[r] = the L node r was reached from
    st.clear();
    st.push_back({l0, 0});
    int end = -1;
    while (!st.empty() && end < 0) {
      auto& top = st.back();
      const int l = top.first;
      if (top.second == e[l].size()) { st.pop_back(); continue; }
      const int r = e[l][top.second++];
      if (seen[r] == stamp) continue;
      seen[r] = stamp;
      par[r] = l;
      if (mr[r] < 0) end = r; else st.push_back({mr[r], 0});
    }
    if (end >= 0) stamp++;
    while (end >= 0) { const int l = par[end], prev = ml[l]; ml[l] = end; mr[end] = l; end = prev; }
  }
  std::vector<char> zl(nl, 0), zr(nr, 0);
  std::vector<int> q;
  for (int l = 0; l < nl; l++) if (ml[l] < 0) { zl[l] = 1; q.push_back(l); }
  while (!q.empty()) {
    const int l = q.back();
    q.pop_back();
    for (int r : e[l]) {
      if (zr[r]) continue;
      zr[r] = 1;
      const int l2 = mr[r];
      if (l2 >= 0 && !zl[l2]) { zl[l2] = 1; q.push_back(l2); }
    }
  }
  cover.clear();
  for (int l = 0; l < nl; l++) if (!zl[l]) cover.push_back(L[l]);
  for (int r = 0; r < nr; r++) if (zr[r]) cover.push_back(R[r]);
}

static long steps64(double cams, int tail = 0) { return cdiv(6 * (long)std::ceil(cams) + tail, 64); }
// Estimated chain (64-column steps, 3 per level for its corner update) of a node of k cameras that will be cut `d` more
// times, if its separators shrink like those of a planar graph (the parent's, ns of total, scaled by sqrt(k / total)).
static double chain_estimate(double k, int d, double ns, double total) {
  if (d <= 0) return (double)steps64(k);
  const double s = ns * std::sqrt(std::max(k, 1.0) / std::max(total, 1.0));
  return chain_estimate(std::max((k - s) / 2, 1.0), d - 1, s, k) + (double)steps64(s) + 3;
}

// The cut order of a node set: nodes sorted by the difference of their hop distances to two far-apart nodes s and t,
// smoothed twice over the neighbours (hop distances alone take a handful of values on camera graphs with long-range
// overlap: the order inside a value would be arbitrary).  (Level sets of the distance to ONE node give L-shaped cuts on
// grid-like camera graphs; the difference gives the straight bisector.)  pos = place in the order, lo / hi = smallest and
// largest place among a node's neighbours in the set: a prefix [0, m) touches the rest through the nodes with hi >= m,
// the rest touches the prefix through those with lo < m, so a cut position costs O(nodes + boundary edges), not O(edges).
struct CutPrep {
  std::vector<char> in;
  int kind = 0;   // 0: cannot be split, 1: disconnected (a = the reached component, b = the rest, no separator), 2: ordered
  std::vector<char> a, b;
  std::vector<int> order, pos, lo, hi;
  struct Cand { int na, nb, nsep; std::vector<signed char> lab; };   // a cut position, evaluated (0: a, 1: b, 2: separator)
  std::vector<Cand> cands;
  bool cands_done = false;
};
static void cut_prepare(const CamGraph& G, const std::vector<char>& in, CutPrep& P) {
  const int n = G.n;
  P.in = in;
  P.kind = 0;
  int start = -1, total = 0;
  for (int i = 0; i < n; i++) if (in[i]) { if (start < 0) start = i; total++; }
  if (total < 3) return;
  std::vector<int> ls(n), lt(n);
  for (int it = 0; it < 4; it++) {  // pseudo-peripheral node: walk to the farthest node a few times
    graph_levels(G, in, start, ls);
    int far = start;
    for (int i = 0; i < n; i++) if (in[i] && ls[i] > ls[far]) far = i;
    if (far == start) break;
    start = far;
    if (it == 3) graph_levels(G, in, start, ls);
  }
  int t = start, unreached = 0;
  for (int i = 0; i < n; i++) if (in[i]) { if (ls[i] < 0) unreached++; else if (ls[i] > ls[t]) t = i; }
  if (unreached) {
    P.a.assign(n, 0); P.b.assign(n, 0);
    for (int i = 0; i < n; i++) if (in[i]) (ls[i] >= 0 ? P.a : P.b)[i] = 1;
    P.kind = 1;
    return;
  }
  if (t == start) return;
  graph_levels(G, in, t, lt);
  std::vector<int>& order = P.order;
  order.clear();
  for (int i = 0; i < n; i++) if (in[i]) order.push_back(i);
  std::vector<double> key(n, 0.0), key2(n, 0.0);
  for (int i : order) key[i] = (double)(ls[i] - lt[i]);
  for (int pass = 0; pass < 2; pass++) {
    for (int i : order) {
      double sum = 0;
      int deg = 0;
      for (int v : G.adj[i]) if (in[v]) { sum += key[v]; deg++; }
      key2[i] = deg ? 0.5 * key[i] + 0.5 * sum / deg : key[i];
    }
    key.swap(key2);
  }
  std::sort(order.begin(), order.end(), [&](int x, int y) {
    if (key[x] != key[y]) return key[x] < key[y];
    if (ls[x] != ls[y]) return ls[x] < ls[y];
    return x < y;
  });
  P.pos.assign(n, -1);
  for (int k = 0; k < total; k++) P.pos[order[k]] = k;
  P.lo.assign(n, total);
  P.hi.assign(n, -1);
  for (int i : order)
    for (int v : G.adj[i]) {
      const int pv = P.pos[v];
      if (pv < 0) continue;
      P.lo[i] = std::min(P.lo[i], pv);
      P.hi[i] = std::max(P.hi[i], pv);
    }
  P.kind = 2;
}

// Splits the node set of `P` into a, b, sep (no edge between a and b): a prefix of the cut order is side a, the rest side
// b, and the separator is the smallest vertex cover of the edges between the two one-hop boundaries, thinned.  Among nine
// cut positions between 34 % and 66 % (evaluated once per node set, whatever depth asks) the one with the shortest
// estimated panel chain wins: this node's own separator (`tail_cols` more columns behind it for the root) + the larger
// side, cut `depth_left - 1` more times; ties go to the smaller max(|a|, |b|) + |sep|.  false when the set cannot be split.
static void cut_candidates(const CamGraph& G, CutPrep& P) {
  const int n = G.n;
  const std::vector<int>& order = P.order;
  const int total = (int)order.size();
  std::vector<signed char> lab(n, -1);
  std::vector<int> La, Lb, r_index(n, -1), cover;
  int last_m = -1;
  for (int pct = 34; pct <= 66; pct += 4) {
    const int m = std::max(1, std::min(total - 1, (int)((long)total * pct / 100)));
    if (m == last_m) continue;
    last_m = m;
    La.clear(); Lb.clear();   // one-hop boundaries
    for (int k = 0; k < total; k++) {
      const int i = order[k];
      lab[i] = k < m ? 0 : 1;
      if (k < m ? P.hi[i] >= m : P.lo[i] < m) (k < m ? La : Lb).push_back(i);
    }
    for (size_t k = 0; k < Lb.size(); k++) r_index[Lb[k]] = (int)k;
    boundary_cover(G, La, Lb, r_index, cover);
    for (int v : Lb) r_index[v] = -1;
    for (int i : cover) lab[i] = 2;
    // thin the separator (in cut order): a separator node without a neighbour on one side belongs to the other side
    std::sort(cover.begin(), cover.end(), [&](int x, int y) { return P.pos[x] < P.pos[y]; });
    for (int i : cover) {
      bool ta = false, tb = false;
      for (int v : G.adj[i]) { ta = ta || lab[v] == 0; tb = tb || lab[v] == 1; }
      if (!tb) lab[i] = 0;
      else if (!ta) lab[i] = 1;
    }
    int cnt[3] = {0, 0, 0};
    for (int i : order) cnt[lab[i]]++;
    if (cnt[0] == 0 || cnt[1] == 0) continue;
    P.cands.push_back(CutPrep::Cand{cnt[0], cnt[1], cnt[2], lab});
  }
  P.cands_done = true;
}
static bool graph_bisect(const CamGraph& G, CutPrep& P, int depth_left, int tail_cols, std::vector<char>& a, std::vector<char>& b,
                         std::vector<char>& sep) {
  const int n = G.n;
  if (P.kind == 0) return false;
  if (P.kind == 1) { a = P.a; b = P.b; sep.assign(n, 0); return true; }
  if (!P.cands_done) cut_candidates(G, P);
  const int total = (int)P.order.size();
  double best = -1;
  long best_tie = 0;
  const CutPrep::Cand* pick = nullptr;
  for (const CutPrep::Cand& c : P.cands) {
    const double cost = (double)steps64(c.nsep, tail_cols) + chain_estimate(std::max(c.na, c.nb), depth_left - 1, c.nsep, total);
    const long tie = (long)std::max(c.na, c.nb) + c.nsep;
    if (best < 0 || cost < best || (cost == best && tie < best_tie)) { best = cost; best_tie = tie; pick = &c; }
  }
  if (!pick) return false;
  a.assign(n, 0); b.assign(n, 0); sep.assign(n, 0);
  for (int i : P.order) (pick->lab[i] == 0 ? a : pick->lab[i] == 1 ? b : sep)[i] = 1;
  return true;
}

// Nested dissection of the camera graph to a given depth.  Leaves and separators are collected in tree order (the "a"
// side of a cut before the "b" side), the separators per depth; every node remembers the interval of leaf indices under it.
struct NdNode { std::vector<int> cams; int leaf_lo, leaf_hi; };
struct NdTree {
  std::vector<NdNode> leaves;
  std::vector<std::vector<NdNode>> seps;   // seps[d]: separators cut at depth d (seps[0][0] = root separator)
};
typedef std::vector<CutPrep> CutCache;   // the cut orders of the node sets met so far (the depths tried share most of them)
static size_t cut_cached(const CamGraph& G, const std::vector<char>& in, CutCache& cache) {
  for (size_t k = 0; k < cache.size(); k++) if (cache[k].in == in) return k;
  cache.emplace_back();
  cut_prepare(G, in, cache.back());
  return cache.size() - 1;
}
static void nd_split(const CamGraph& G, const std::vector<char>& in, int depth_left, int d, int tail_cols, CutCache& cache, NdTree& T) {
  std::vector<char> a, b, s;
  auto members = [&](const std::vector<char>& m) { std::vector<int> v; for (int i = 0; i < G.n; i++) if (m[i]) v.push_back(i); return v; };
  bool cut = false;
  if (depth_left > 0) {
    const size_t k = cut_cached(G, in, cache);
    cut = graph_bisect(G, cache[k], depth_left, d == 0 ? tail_cols : 0, a, b, s);
  }
  if (!cut) {
    const int id = (int)T.leaves.size();
    T.leaves.push_back(NdNode{members(in), id, id});
    return;
  }
  if ((int)T.seps.size() <= d) T.seps.resize(d + 1);
  const size_t me = T.seps[d].size();
  T.seps[d].push_back(NdNode{members(s), (int)T.leaves.size(), -1});
  nd_split(G, a, depth_left - 1, d + 1, tail_cols, cache, T);
  nd_split(G, b, depth_left - 1, d + 1, tail_cols, cache, T);
  T.seps[d][me].leaf_hi = (int)T.leaves.size() - 1;
}

// Chooses the dissection depth (1..3) with the shortest estimated chain of 64-column panel launches: per level the
// longest node chain plus three launch-equivalents for the level's deferred corner update (one SYRK over all its panels +
// the merge: measured at config 3, depth 3 has one launch fewer than depth 2 and is 5 % slower), then the root separator
// with the intrinsics.  Returns false ("keep the dense order") unless a depth is at least 20 % shorter than the dense
// chain (or one is forced).
static bool choose_dissection(const CamGraph& G, int tail_cols, int force_depth, NdTree& best_tree) {
  const int n = G.n;
  const long dense = cdiv(6L * n + tail_cols, 64);
  long best = dense;
  bool found = false;
  CutCache cache;
  cache.reserve(16);
  long leaf_steps = dense;   // longest leaf chain of the depth before
  for (int depth = 1; depth <= 3; depth++) {
    if (force_depth >= 0 && depth != force_depth) continue;
    // one more cut replaces a leaf chain of p steps by ~p / 2 + its separator's + the 3 of the level: below a dozen steps that
    // cannot pay, and the cuts need not be evaluated (a third of the ordering's host time at config 3)
    if (force_depth < 0 && depth > 1 && leaf_steps < 12) break;
    NdTree T;
    std::vector<char> all(n, 1);
    nd_split(G, all, depth, 0, tail_cols, cache, T);
    if (T.leaves.size() < 2 || T.leaves.size() > 8 || T.seps.empty()) continue;
    bool ok = true;
    for (size_t d = 1; d < T.seps.size(); d++) ok = ok && T.seps[d].size() <= 8;
    // everything behind the leaves (padded separators, root, intrinsics, rhs row) is the square of the deferred corner
    // update, whose per-block panel ranges travel as a kernel argument of MSFM_CORNER_MAX_BLOCKS entries (chol.hip)
    long behind = 6 * (long)T.seps[0][0].cams.size() + tail_cols;
    for (size_t d = 1; d < T.seps.size(); d++)
      for (auto& q : T.seps[d]) behind += 64 * cdiv(6 * (long)q.cams.size(), 64);
    if (cdiv(behind, 64) > MSFM_CORNER_MAX_BLOCKS) ok = false;
    if (!ok) continue;
    long chain = 0, maxp = 0;
    for (auto& l : T.leaves) maxp = std::max<long>(maxp, cdiv(6 * (long)l.cams.size(), 64));
    leaf_steps = maxp;
    chain += maxp + 3;
    for (size_t d = T.seps.size() - 1; d >= 1; d--) {
      long mp = 0;
      for (auto& q : T.seps[d]) mp = std::max<long>(mp, cdiv(6 * (long)q.cams.size(), 64));
      if (mp) chain += mp + 3;
    }
    chain += cdiv(6 * (long)T.seps[0][0].cams.size() + tail_cols, 64);
    if (force_depth >= 0 || (chain < best && chain * 10 <= dense * 8)) {
      best = chain;
      best_tree = T;
      found = true;
    }
  }
  return found;
}

MSFM_API int msfm_camera_graph_dissection(int n_cams, const uint8_t* adjacency, int tail_cols, int force_depth, int32_t* label,
                                          int* n_leaves, int* chain_steps) {
  if (n_cams < 1 || !adjacency || !label || !n_leaves || !chain_steps || force_depth > 3) return MSFM_E_INVAL;
  CamGraph G;
  G.n = n_cams;
  G.adj.resize(n_cams);
  for (int a = 0; a < n_cams; a++)
    for (int b = 0; b < n_cams; b++) if (a != b && adjacency[(size_t)a * n_cams + b]) G.adj[a].push_back(b);
  for (int c = 0; c < n_cams; c++) label[c] = 0;
  *n_leaves = 0;
  *chain_steps = (int)cdiv(6L * n_cams + tail_cols, 64);
  NdTree T;
  if (!choose_dissection(G, tail_cols, force_depth, T)) return MSFM_OK;
  *n_leaves = (int)T.leaves.size();
  long chain = 0, mx = 0;
  for (size_t i = 0; i < T.leaves.size(); i++) {
    for (int c : T.leaves[i].cams) label[c] = (int)i;
    mx = std::max<long>(mx, cdiv(6 * (long)T.leaves[i].cams.size(), 64));
  }
  chain += mx;
  for (size_t d = 0; d < T.seps.size(); d++) {
    mx = 0;
    for (auto& q : T.seps[d]) {
      for (int c : q.cams) label[c] = -(int)(d + 1);
      mx = std::max<long>(mx, cdiv(6 * (long)q.cams.size() + (d == 0 ? tail_cols : 0), 64));
    }
    chain += mx;
  }
  *chain_steps = (int)chain;
  return MSFM_OK;
}

static bool is_mut(const uint8_t* m, int i) { return m == nullptr || m[i] != 0; }

// Numbers the camera blocks in elimination order and lays the reduced system out: leaves, then the separators from the
// deepest cut to the shallowest (every node padded with identity columns to a multiple of 64), the root separator and the
// intrinsics last.  adjb: ng x ng 0/1 adjacency of the graph nodes (cameras sharing an eliminated point), empty = no
// dissection.  Fills cam_slot, ba->h_cb_cam, ncb, mo, nsys, n_padcol, plan (levels) and the two column lists.
static void order_camera_blocks(msfm_ctx* ctx, msfm_ba* ba, const std::vector<int>& gcam, const std::vector<uint8_t>& adjb, int force,
                                std::vector<int>& cam_slot, std::vector<int>& cb_off_h, std::vector<int>& padcol_h) {
  const int ng = (int)gcam.size();
  NdTree T;
  bool dissect = false;
  if (!adjb.empty()) {
    CamGraph G;
    G.n = ng;
    G.adj.resize(ng);
    for (int a = 0; a < ng; a++)
      for (int b = 0; b < ng; b++) if (adjb[(size_t)a * ng + b]) G.adj[a].push_back(b);
    dissect = choose_dissection(G, 3 * ba->nmb + 1, force, T);
  }
  int col = 0;
  auto place = [&](int g) { cam_slot[gcam[g]] = ba->ncb++; ba->h_cb_cam.push_back(gcam[g]); cb_off_h.push_back(col); col += 6; };
  ba->plan = msfm_chol_plan();
  if (!dissect) {
    for (int g = 0; g < ng; g++) place(g);
  } else {
    // levels: 0 = leaves, then the separators of depth D-1, ..., 1; depth 0 is the root chain
    std::vector<std::vector<NdNode>*> levels;
    levels.push_back(&T.leaves);
    for (int d = (int)T.seps.size() - 1; d >= 1; d--) if (!T.seps[d].empty()) levels.push_back(&T.seps[d]);
    int nl = 0;
    for (auto* lv : levels) {
      msfm_chol_level& L = ba->plan.level[nl];
      L.K = 0;
      L.begin = col;
      for (auto& node : *lv) {
        if (node.cams.empty()) continue;   // a cut without separator nodes (disconnected parts)
        const int begin = col;
        for (int g : node.cams) place(g);
        while (col % 64) padcol_h.push_back(col++);
        L.node[L.K++] = msfm_chol_node{begin, col, node.leaf_lo, node.leaf_hi};
      }
      L.b0 = col;
      if (L.K > 0) nl++;
    }
    ba->plan.n_levels = nl;
    for (int g : T.seps[0][0].cams) place(g);
    if (getenv("MSFM_VERBOSE") && ctx->rank == 0) {
      fprintf(stderr, "msfm: camera graph %d nodes ->", ng);
      for (int l = 0; l < nl; l++) {
        fprintf(stderr, " level %d:", l);
        for (int k = 0; k < ba->plan.level[l].K; k++) fprintf(stderr, " %d", (ba->plan.level[l].node[k].end - ba->plan.level[l].node[k].begin) / 6);
        fprintf(stderr, " |");
      }
      fprintf(stderr, " root separator %zu cameras\n", T.seps[0][0].cams.size());
    }
  }
  ba->mo = col;
  ba->nsys = col + 3 * ba->nmb;
  ba->n_padcol = (int)padcol_h.size();
}


// Build chunk / block lists from pair entries already sorted by block key.
static int finish_jobs(msfm_ba* ba, PairJobs& J, const std::vector<int>& pa, const std::vector<int>& pb,
                       const std::vector<long>& key_first /*per present block: first entry*/, const std::vector<int>& brow,
                       const std::vector<int>& bcol, int nout) {
  hipStream_t s = ba->ctx->stream;
  J.n_pairs = (int)pa.size();
  J.n_blocks = (int)brow.size();
  J.h_row = brow;
  J.h_col = bcol;
  std::vector<int> cs, ce, bcf(J.n_blocks + 1, 0);
  const int chunk = chunk_for((long)pa.size(), CHUNK);
  for (int b = 0; b < J.n_blocks; b++) {
    bcf[b] = (int)cs.size();
    for (long e = key_first[b]; e < key_first[b + 1]; e += chunk) {
      cs.push_back((int)e);
      ce.push_back((int)std::min<long>(e + chunk, key_first[b + 1]));
    }
  }
  bcf[J.n_blocks] = (int)cs.size();
  J.n_chunks = (int)cs.size();
  HIP_TRY(ba->ctx, J.pa.from(pa, s));
  HIP_TRY(ba->ctx, J.pb.from(pb, s));
  HIP_TRY(ba->ctx, J.ch_start.from(cs, s));
  HIP_TRY(ba->ctx, J.ch_end.from(ce, s));
  HIP_TRY(ba->ctx, J.blk_row.from(brow, s));
  HIP_TRY(ba->ctx, J.blk_col.from(bcol, s));
  HIP_TRY(ba->ctx, J.blk_chunk_first.from(bcf, s));
  HIP_TRY(ba->ctx, J.partial.alloc((size_t)std::max(1, J.n_chunks) * nout));
  HIP_TRY(ba->ctx, hipStreamSynchronize(s));  // host vectors go out of scope
  return MSFM_OK;
}

MSFM_API void msfm_ba_options_default(msfm_ba_options* o) {
  if (!o) return;
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

MSFM_API void msfm_ba_destroy(msfm_ba* ba) {
  if (!ba) return;
  msfm_ctx* ctx = ba->ctx;
  (void)hipSetDevice(ctx->device);   // the caller's thread may have another device current (Python __del__ after set_device)
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);   // the pair kernels read T / Tm and write the partials freed below
  msfm_chol_ws_destroy(ba->chol_ws);
  ba->chol_ws = nullptr;
#ifdef MSFM_FOLD_STAMPS
  if (ba->fold.on) {
    static long long h[8192][8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fold_stamps), sizeof(h)) == hipSuccess) {
      double d[8] = {0}; int n = 0;
      for (int w = 0; w < std::min(8192, ba->fold.n_wg); w++) {
        if (h[w][7] <= h[w][0]) continue;
        for (int k = 1; k < 8; k++) d[k] += (double)(h[w][k] - h[w][k - 1]);
        n++;
      }
      fprintf(stderr, "msfm: k_point stamps (mean over %d workgroups, cycles of wave 0): linearise %.0f, reduce+factor %.0f, Tm %.0f, round 0 %.0f, records %.0f, later rounds %.0f, fold passes %.0f\n",
              n, d[1] / n, d[2] / n, d[3] / n, d[4] / n, d[5] / n, d[6] / n, d[7] / n);
    }
  }
#endif
  if (ba->h_scal) (void)hipHostFree(ba->h_scal);
  if (ba->h_fail) (void)hipHostFree(ba->h_fail);
  delete ba;
  msfm_ctx_child_released(ctx);
}

// Host work arrays of msfm_ba_create that scale with the observation count; owned by the context and reused.
struct BaScratch {
  std::vector<int> o_cam, o_model, o_pt, o_cb, o_mb, o_pb, o_cpos, o_pm, run_first, pt_first, cpos_pb, cpos_cb, pa, pb;
  std::vector<double> o_x, o_y, o_w;
  std::vector<std::pair<uint64_t, int>> keyed;
};
static BaScratch& ba_scratch(msfm_ctx* ctx) {
  if (!ctx->ba_scratch) {
    ctx->ba_scratch = new BaScratch();
    ctx->ba_scratch_free = [](void* p) { delete static_cast<BaScratch*>(p); };
  }
  return *static_cast<BaScratch*>(ctx->ba_scratch);
}

// =======================================================================================
// Index structures of a problem built ON THE DEVICE (default).  The reference hands over host arrays at every bundle
// adjustment of its incremental loop and the structure changes each time (sfm_incremental.cc:917-1014), so this set-up
// sits on the hot path of msfm_ba_solve: only the caller's own arrays cross PCIe (24 bytes per observation); block
// usage, the point order, point-major rows, camera-major positions, the (point, intrinsics) entries and the three
// block-pair lists are produced by scans, stable radix sorts (rocPRIM) and small kernels that write exactly what the
// host code below writes (create_structures_host, kept for MSFM_CREATE_HOST=1 and compared bit for bit in the tests).
// The host keeps what is O(cameras): slot numbering and the camera-graph bisection.
// =======================================================================================
namespace devsetup {

// Exclusive scan of a SHORT array by one workgroup in one launch (round 5): the set-up of a window-sized problem is ~190 launches
// of a few microseconds each, fifteen of them scans of a few thousand integers for which rocPRIM's look-back scan is two launches
// and a temporary.  Integer sums: the result is the same whatever the grouping.
template <class T>
__global__ __launch_bounds__(1024) void k_scan_small(const T* __restrict__ in, T* __restrict__ out, int n) {
  __shared__ T part[1024];
  const int t = threadIdx.x, per = (n + 1023) / 1024, lo = min(n, t * per), hi = min(n, lo + per);
  T sum = T(0);
  for (int i = lo; i < hi; i++) sum += in[i];
  part[t] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const T v = t >= off ? part[t - off] : T(0);
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  T run = part[t] - sum;
  for (int i = lo; i < hi; i++) { const T v = in[i]; out[i] = run; run += v; }
}
#define MSFM_SCAN_SMALL_MAX 65536
template <class T>
static hipError_t excl_scan(const T* in, T* out, size_t n, hipStream_t s, DevBuf<char>& tmp) {
  if (n == 0) return hipSuccess;
  if (n <= MSFM_SCAN_SMALL_MAX && in != out) {
    hipLaunchKernelGGL((k_scan_small<T>), dim3(1), dim3(1024), 0, s, in, out, (int)n);
    return hipGetLastError();
  }
  size_t bytes = 0;
  hipError_t e = rocprim::exclusive_scan(nullptr, bytes, in, out, T(0), n, rocprim::plus<T>(), s);
  if (e != hipSuccess) return e;
  if (tmp.n < bytes) { e = tmp.alloc(bytes); if (e != hipSuccess) return e; }
  return rocprim::exclusive_scan(tmp.p, bytes, in, out, T(0), n, rocprim::plus<T>(), s);
}
template <class K>
static hipError_t sort_pairs(const K* kin, K* kout, const int* vin, int* vout, size_t n, int bits, hipStream_t s, DevBuf<char>& tmp) {
  if (n == 0) return hipSuccess;
  size_t bytes = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0, bits, s);
  if (e != hipSuccess) return e;
  if (tmp.n < bytes) { e = tmp.alloc(bytes); if (e != hipSuccess) return e; }
  return rocprim::radix_sort_pairs(tmp.p, bytes, kin, kout, vin, vout, n, 0, bits, s);   // stable
}
static int bits_for(long n) { int b = 1; while ((1L << b) < n) b++; return b; }

__device__ __forceinline__ bool dmut(const uint8_t* m, int i) { return m == nullptr || m[i] != 0; }

// validation + block usage + observations per point.  err[0] = first out-of-range observation, err[1] = first order break
__global__ __launch_bounds__(256) void k_scan_obs(int No, int Nc, int Np, const int* __restrict__ obs_cam, const int* __restrict__ obs_pt,
                                                   const int* __restrict__ model_of_cam, const uint8_t* __restrict__ cam_mut,
                                                   const uint8_t* __restrict__ model_mut, const uint8_t* __restrict__ pt_mut,
                                                   uint8_t* __restrict__ cu, uint8_t* __restrict__ mu, uint8_t* __restrict__ pu,
                                                   int* __restrict__ cnt, int* __restrict__ err) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= No) return;
  const int c = obs_cam[o], p = obs_pt[o];
  if (c < 0 || c >= Nc || p < 0 || p >= Np) { atomicMin(&err[0], o); return; }
  if (o > 0 && p < obs_pt[o - 1]) atomicMin(&err[1], o);
  atomicAdd(&cnt[p], 1);
  const bool cm = dmut(cam_mut, c), pm = dmut(pt_mut, p);
  if (!cm && !pm) return;
  if (pm) pu[p] = 1;
  if (cm) {
    cu[c] = 1;
    const int m = model_of_cam[c];
    if (dmut(model_mut, m)) mu[m] = 1;
  }
}

// cameras that share an eliminated point (0/1 matrix over the graph nodes)
__global__ __launch_bounds__(256) void k_adjacency(int Np, const int* __restrict__ run_first, const int* __restrict__ obs_cam,
                                                    const uint8_t* __restrict__ pt_mut, const int* __restrict__ gnode, int ng,
                                                    uint8_t* __restrict__ adjb) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= Np || !dmut(pt_mut, p)) return;
  const int f = run_first[p], l = run_first[p + 1];
  for (int e1 = f; e1 < l; e1++) {
    const int a = gnode[obs_cam[e1]];
    if (a < 0) continue;
    for (int e2 = f; e2 < l; e2++) {
      const int b = gnode[obs_cam[e2]];
      if (b >= 0 && b != a) adjb[(size_t)a * ng + b] = 1;
    }
  }
}

// sort key of an eliminated point: its smallest camera blocks (4 x 16 bits, or 3 x 21 bits past 65535 blocks)
__global__ __launch_bounds__(256) void k_point_keys(int Np, const uint8_t* __restrict__ pu, const int* __restrict__ pu_pos,
                                                     const int* __restrict__ run_first, const int* __restrict__ obs_cam,
                                                     const int* __restrict__ cam_slot, int wide, unsigned long long* __restrict__ keys,
                                                     int* __restrict__ vals) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= Np || !pu[p]) return;
  const int nk = wide ? 3 : 4, bits = wide ? 21 : 16;
  const int none = (1 << bits) - 1;
  int best[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
  for (int e = run_first[p]; e < run_first[p + 1]; e++) {
    int v = cam_slot[obs_cam[e]];
    if (v < 0) continue;
#pragma unroll
    for (int j = 0; j < 4; j++) {   // sorted insert (duplicates kept, as the host's sorted list keeps them)
      const int lo = min(best[j], v);
      v = max(best[j], v);
      best[j] = lo;
    }
  }
  unsigned long long k = 0;
  for (int q = 0; q < nk; q++) k = (k << bits) | (unsigned long long)(best[q] == 0x7fffffff ? none : best[q]);
  keys[pu_pos[p]] = k;
  vals[pu_pos[p]] = p;
}

__global__ __launch_bounds__(256) void k_compact_used(int Np, const uint8_t* __restrict__ pu, const int* __restrict__ pu_pos, int* __restrict__ vals) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p < Np && pu[p]) vals[pu_pos[p]] = p;
}
__global__ __launch_bounds__(256) void k_u8_to_int(int n, const uint8_t* __restrict__ a, int* __restrict__ b) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) b[i] = a[i] ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_point_lengths(int npb, const int* __restrict__ pb_pt, const int* __restrict__ cnt, int* __restrict__ len,
                                                        int* __restrict__ pt_slot) {
  const int pb = blockIdx.x * 256 + threadIdx.x;
  if (pb >= npb) return;
  const int p = pb_pt[pb];
  len[pb] = cnt[p];
  pt_slot[p] = pb;
}

struct RowOut { int *o_cam, *o_model, *o_pt, *o_cb, *o_mb, *o_pb; double *o_x, *o_y, *o_w; };

// point-major rows of the eliminated points: row i of block pb is observation run_first[p] + (i - pt_first[pb])
__global__ __launch_bounds__(256) void k_fill_rows(int AE, int npb, const int* __restrict__ pt_first, const int* __restrict__ pb_pt,
                                                    const int* __restrict__ run_first, const int* __restrict__ obs_cam,
                                                    const double* __restrict__ obs_xy, const double* __restrict__ ptw,
                                                    const int* __restrict__ model_of_cam, const uint8_t* __restrict__ cam_mut,
                                                    const uint8_t* __restrict__ model_mut, const int* __restrict__ cam_slot,
                                                    const int* __restrict__ model_slot, RowOut R, int* __restrict__ cam_hist, int ncb) {
  // rows per camera block: counted in LDS first (a workgroup's 256 consecutive rows meet a few dozen cameras; 1.2 M atomic
  // adds on 500 counters made this kernel 0.6 ms at config 3), one global add per camera the workgroup has seen
  constexpr int HB = 4096;
  __shared__ int hist[HB];
  const bool lds_hist = ncb <= HB;
  if (lds_hist) {
    for (int k = threadIdx.x; k < ncb; k += 256) hist[k] = 0;
    __syncthreads();
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < AE) {
  int lo = 0, hi = npb - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (pt_first[mid] <= i) lo = mid; else hi = mid - 1;
  }
  const int pb = lo, p = pb_pt[pb], o = run_first[p] + (i - pt_first[pb]);
  const int c = obs_cam[o], m = model_of_cam[c];
  const bool cm = dmut(cam_mut, c);
  const int cb = cm ? cam_slot[c] : -1;
  R.o_cam[i] = c; R.o_model[i] = m; R.o_pt[i] = p;
  R.o_cb[i] = cb;
  R.o_mb[i] = (cm && dmut(model_mut, m)) ? model_slot[m] : -1;
  R.o_pb[i] = pb;
  R.o_x[i] = obs_xy[2 * (size_t)o]; R.o_y[i] = obs_xy[2 * (size_t)o + 1];
  R.o_w[i] = ptw ? ptw[p] : 1.0;
  if (cb >= 0) { if (lds_hist) atomicAdd(&hist[cb], 1); else atomicAdd(&cam_hist[cb], 1); }
  }
  if (lds_hist) {
    __syncthreads();
    for (int k = threadIdx.x; k < ncb; k += 256) { const int v = hist[k]; if (v) atomicAdd(&cam_hist[k], v); }
  }
}

// observations of frozen points by free cameras, in input order (only with a point mask)
__global__ __launch_bounds__(256) void k_flag_frozen(int No, const int* __restrict__ obs_cam, const int* __restrict__ obs_pt,
                                                      const uint8_t* __restrict__ cam_mut, const uint8_t* __restrict__ pt_mut, int* __restrict__ flag) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o < No) flag[o] = (!dmut(pt_mut, obs_pt[o]) && dmut(cam_mut, obs_cam[o])) ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_fill_frozen(int No, int AE, const int* __restrict__ flag, const int* __restrict__ pos,
                                                      const int* __restrict__ obs_cam, const int* __restrict__ obs_pt,
                                                      const double* __restrict__ obs_xy, const double* __restrict__ ptw,
                                                      const int* __restrict__ model_of_cam, const uint8_t* __restrict__ model_mut,
                                                      const int* __restrict__ cam_slot, const int* __restrict__ model_slot, RowOut R,
                                                      int* __restrict__ cam_hist) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= No || !flag[o]) return;
  const int i = AE + pos[o];
  const int c = obs_cam[o], p = obs_pt[o], m = model_of_cam[c];
  R.o_cam[i] = c; R.o_model[i] = m; R.o_pt[i] = p;
  R.o_cb[i] = cam_slot[c];
  R.o_mb[i] = dmut(model_mut, m) ? model_slot[m] : -1;
  R.o_pb[i] = -1;
  R.o_x[i] = obs_xy[2 * (size_t)o]; R.o_y[i] = obs_xy[2 * (size_t)o + 1];
  R.o_w[i] = ptw ? ptw[p] : 1.0;
  if (cam_slot[c] >= 0) atomicAdd(&cam_hist[cam_slot[c]], 1);
}

__global__ __launch_bounds__(256) void k_row_keys(int A, int ncb, const int* __restrict__ o_cb, int* __restrict__ keys, int* __restrict__ vals) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= A) return;
  keys[i] = o_cb[i] >= 0 ? o_cb[i] : ncb;   // rows without a camera block sort behind every camera
  vals[i] = i;
}
__global__ __launch_bounds__(256) void k_assign_positions(int NCR, const int* __restrict__ sorted_key, const int* __restrict__ sorted_row,
                                                           const int* __restrict__ o_pb, int* __restrict__ o_cpos, int* __restrict__ cpos_pb) {
  const int pos = blockIdx.x * 256 + threadIdx.x;
  if (pos >= NCR) return;
  (void)sorted_key;
  const int i = sorted_row[pos];
  o_cpos[i] = pos;
  cpos_pb[pos] = o_pb[i];
}
__global__ __launch_bounds__(256) void k_fill_int(int n, int v, int* __restrict__ p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}

// the sorted distinct intrinsics blocks of a point's rows: pass 0 counts, pass 1 writes pm_mb and o_pm
__global__ __launch_bounds__(256) void k_pm_entries(int npb, const int* __restrict__ pt_first, const int* __restrict__ o_mb, int pass,
                                                     int* __restrict__ pm_count, const int* __restrict__ pm_first, int* __restrict__ pm_mb,
                                                     int* __restrict__ o_pm, int* __restrict__ err) {
  const int pb = blockIdx.x * 256 + threadIdx.x;
  if (pb >= npb) return;
  int tmp[64];
  int nt = 0;
  for (int i = pt_first[pb]; i < pt_first[pb + 1]; i++) {
    const int mb = o_mb[i];
    if (mb < 0) continue;
    bool seen = false;
    for (int k = 0; k < nt; k++) seen |= tmp[k] == mb;
    if (!seen) {
      if (nt == 64) { atomicMin(err, pb); nt = -1; break; }
      tmp[nt++] = mb;
    }
  }
  if (pass == 0) { pm_count[pb] = nt < 0 ? 0 : nt; return; }
  if (nt < 0) return;
  for (int a = 1; a < nt; a++) {   // insertion sort
    const int v = tmp[a];
    int b = a - 1;
    while (b >= 0 && tmp[b] > v) { tmp[b + 1] = tmp[b]; b--; }
    tmp[b + 1] = v;
  }
  const int base = pm_first[pb];
  for (int k = 0; k < nt; k++) pm_mb[base + k] = tmp[k];
  for (int i = pt_first[pb]; i < pt_first[pb + 1]; i++)
    if (o_mb[i] >= 0)
      for (int k = 0; k < nt; k++) if (tmp[k] == o_mb[i]) o_pm[i] = base + k;
}

// ---- block-pair lists ------------------------------------------------------------------------
// kind 0: camera x camera (row block >= column block), 1: intrinsics x camera, 2: intrinsics x intrinsics (row >= column);
// a point's entries in the order of the host loops (i outer, j inner), points in block order.
template <int KIND, bool EMIT>
__global__ __launch_bounds__(256) void k_pairs_of_points(int npb, const int* __restrict__ pt_first, const int* __restrict__ o_cb,
                                                          const int* __restrict__ o_cpos, const int* __restrict__ pm_first,
                                                          const int* __restrict__ pm_mb, long ncol, int* __restrict__ count,
                                                          const int* __restrict__ offset, int* __restrict__ key, int* __restrict__ pa,
                                                          int* __restrict__ pbv) {
  const int pb = blockIdx.x * 256 + threadIdx.x;
  if (pb >= npb) return;
  const int f = pt_first[pb], l = pt_first[pb + 1];
  int n = 0;
  const int base = EMIT ? offset[pb] : 0;
  auto emit = [&](int r, int c, int a, int b) {
    if (EMIT) {
      const int k = (int)((long)r * ncol + c);
      key[base + n] = k; pa[base + n] = a; pbv[base + n] = b;
    }
    n++;
  };
  if (KIND == 0) {
    for (int i = f; i < l; i++) {
      if (o_cpos[i] < 0) continue;
      for (int j = f; j < l; j++) {
        if (o_cpos[j] < 0) continue;
        if (o_cb[i] >= o_cb[j]) emit(o_cb[i], o_cb[j], o_cpos[i], o_cpos[j]);
      }
    }
  } else if (KIND == 1) {
    for (int e = pm_first[pb]; e < pm_first[pb + 1]; e++)
      for (int j = f; j < l; j++) if (o_cpos[j] >= 0) emit(pm_mb[e], o_cb[j], e, o_cpos[j]);
  } else {
    for (int e = pm_first[pb]; e < pm_first[pb + 1]; e++)
      for (int g = pm_first[pb]; g <= e; g++) emit(pm_mb[e], pm_mb[g], e, g);
  }
  if (!EMIT) count[pb] = n;
}
// first entry of every key in the sorted key array (lower bound), and from that the entries per key: no atomics, so a
// list whose entries all share one key (one intrinsics block: 200k entries) costs the same as any other
__global__ __launch_bounds__(256) void k_key_first(long nkey, int total, const int* __restrict__ key_sorted, int* __restrict__ key_first) {
  const long k = (long)blockIdx.x * 256 + threadIdx.x;
  if (k > nkey) return;
  int lo = 0, hi = total;   // first position with key_sorted[pos] >= k
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (key_sorted[mid] < (int)k) lo = mid + 1; else hi = mid;
  }
  key_first[k] = lo;
}
__global__ __launch_bounds__(256) void k_key_hist(long nkey, const int* __restrict__ key_first, int* __restrict__ key_hist) {
  const long k = (long)blockIdx.x * 256 + threadIdx.x;
  if (k <= nkey) key_hist[k] = k < nkey ? key_first[k + 1] - key_first[k] : 0;
}
__global__ __launch_bounds__(256) void k_iota(int n, int* __restrict__ p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = i;
}
__global__ __launch_bounds__(256) void k_gather2(int n, const int* __restrict__ perm, const int* __restrict__ a, const int* __restrict__ b,
                                                  int* __restrict__ a2, int* __restrict__ b2) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) { a2[i] = a[perm[i]]; b2[i] = b[perm[i]]; }
}
// a block exists if it has entries or must be assembled from the F^T F terms alone
__global__ __launch_bounds__(256) void k_block_flags(long nkey, long ncol, int kind, int chunk, const int* __restrict__ key_hist,
                                                      const int* __restrict__ cb_mb, int* __restrict__ flag, int* __restrict__ nchunk) {
  const long k = (long)blockIdx.x * 256 + threadIdx.x;
  if (k >= nkey) return;
  const long r = k / ncol, c = k % ncol;
  const bool force = kind == 1 ? (cb_mb[c] == r) : (r == c);
  const int n = key_hist[k];
  flag[k] = (n > 0 || force) ? 1 : 0;
  nchunk[k] = (n + chunk - 1) / chunk;
}
__global__ __launch_bounds__(256) void k_block_lists(long nkey, long ncol, const int* __restrict__ flag, const int* __restrict__ blk_of_key,
                                                      const int* __restrict__ key_first, const int* __restrict__ key_hist,
                                                      const int* __restrict__ chunk_first_of_key, int chunk, int* __restrict__ blk_row,
                                                      int* __restrict__ blk_col, int* __restrict__ blk_chunk_first, int* __restrict__ ch_start,
                                                      int* __restrict__ ch_end) {
  const long k = (long)blockIdx.x * 256 + threadIdx.x;
  if (k >= nkey || !flag[k]) return;
  const int b = blk_of_key[k];
  blk_row[b] = (int)(k / ncol);
  blk_col[b] = (int)(k % ncol);
  const int cf = chunk_first_of_key[k];
  blk_chunk_first[b] = cf;
  const int e0 = key_first[k], n = key_hist[k];
  for (int q = 0, e = e0; e < e0 + n; e += chunk, q++) { ch_start[cf + q] = e; ch_end[cf + q] = min(e + chunk, e0 + n); }
}

}  // namespace devsetup

#define DTRY(expr) HIP_TRY(ctx, (expr))
// One block-pair list in three phases, so that the three lists of a problem share their host synchronisations (each phase ends
// in a read-back: the number of entries, then the numbers of blocks and chunks): the lists of a small problem - the window of a
// new camera - are a few dozen short launches each, and nine stream synchronisations were a third of its set-up.
template <int KIND>
struct PairBuild {
  msfm_ctx* ctx; msfm_ba* ba; PairJobs& J; int nout; const int *d_pt_first, *d_pm_first, *d_pm_mb; DevBuf<char>& tmp; bool want_host_blocks;
  long nrow = 0, ncol = 0, nkey = 0;
  DevBuf<int> count, offset, key, pa, pbv, key_hist, key_sorted, perm, perm_sorted, key_first, flag, nchunk, blk_of_key, chunk_first;
  int total = 0, nbc[2] = {0, 0}, chunk = 0;
  bool empty = false;
  PairBuild(msfm_ctx* c, msfm_ba* b, PairJobs& j, int no, const int* pf, const int* pmf, const int* pmm, DevBuf<char>& t, bool w)
      : ctx(c), ba(b), J(j), nout(no), d_pt_first(pf), d_pm_first(pmf), d_pm_mb(pmm), tmp(t), want_host_blocks(w) {}
  // entries per point, their offsets; asks for the total
  int phase1() {
    using namespace devsetup;
    hipStream_t s = ctx->stream;
    const int npb = ba->npb, ncb = ba->ncb, nmb = ba->nmb;
    nrow = KIND == 0 ? ncb : nmb; ncol = KIND == 2 ? nmb : ncb;
    nkey = nrow * ncol;
    J.n_pairs = J.n_chunks = J.n_blocks = 0;
    J.h_row.clear(); J.h_col.clear();
    if (nkey == 0) {
      empty = true;
      DTRY(J.pa.alloc(1)); DTRY(J.pb.alloc(1)); DTRY(J.ch_start.alloc(1)); DTRY(J.ch_end.alloc(1)); DTRY(J.blk_row.alloc(1)); DTRY(J.blk_col.alloc(1));
      DTRY(J.blk_chunk_first.alloc(1)); DTRY(hipMemsetAsync(J.blk_chunk_first.p, 0, sizeof(int), s)); DTRY(J.partial.alloc(nout));
      return MSFM_OK;
    }
    if (nkey > 0x7fffffffL) return msfm_set_error(ctx, MSFM_E_NOMEM, "block key space too large");
    const int nb_p = cdiv(std::max(1, npb), 256);
    DTRY(count.alloc((size_t)npb + 1)); DTRY(offset.alloc((size_t)npb + 1));
    DTRY(hipMemsetAsync(count.p, 0, sizeof(int) * ((size_t)npb + 1), s));
    if (npb) hipLaunchKernelGGL((k_pairs_of_points<KIND, false>), dim3(nb_p), dim3(256), 0, s, npb, d_pt_first, ba->o_cb.p, ba->o_cpos.p, d_pm_first, d_pm_mb,
                                ncol, count.p, (const int*)nullptr, (int*)nullptr, (int*)nullptr, (int*)nullptr);
    DTRY(excl_scan(count.p, offset.p, (size_t)npb + 1, s, tmp));
    DTRY(hipMemcpyAsync(&total, offset.p + npb, sizeof(int), hipMemcpyDeviceToHost, s));
    return MSFM_OK;
  }
  // (after a synchronisation) entries emitted, sorted by block key, blocks and chunks counted; asks for the two counts
  int phase2() {
    using namespace devsetup;
    if (empty) return MSFM_OK;
    hipStream_t s = ctx->stream;
    const int npb = ba->npb;
    const int nb_p = cdiv(std::max(1, npb), 256);
    if (total < 0) return msfm_set_error(ctx, MSFM_E_NOMEM, "pair list too long");
    const size_t nt = (size_t)std::max(1, total);
    DTRY(key.alloc(nt)); DTRY(pa.alloc(nt)); DTRY(pbv.alloc(nt)); DTRY(key_hist.alloc((size_t)nkey + 1)); DTRY(key_first.alloc((size_t)nkey + 1));
    if (npb && total) hipLaunchKernelGGL((k_pairs_of_points<KIND, true>), dim3(nb_p), dim3(256), 0, s, npb, d_pt_first, ba->o_cb.p, ba->o_cpos.p, d_pm_first,
                                         d_pm_mb, ncol, (int*)nullptr, offset.p, key.p, pa.p, pbv.p);
    // entries sorted by block key, point order kept inside a block (stable)
    DTRY(J.pa.alloc(nt)); DTRY(J.pb.alloc(nt)); DTRY(key_sorted.alloc(nt));
    if (total) {
      DTRY(perm.alloc(nt)); DTRY(perm_sorted.alloc(nt));
      hipLaunchKernelGGL(k_iota, dim3(cdiv(total, 256)), dim3(256), 0, s, total, perm.p);
      DTRY(sort_pairs(key.p, key_sorted.p, perm.p, perm_sorted.p, (size_t)total, bits_for(nkey), s, tmp));
      hipLaunchKernelGGL(k_gather2, dim3(cdiv(total, 256)), dim3(256), 0, s, total, perm_sorted.p, pa.p, pbv.p, J.pa.p, J.pb.p);
    }
    hipLaunchKernelGGL(k_key_first, dim3(cdiv(nkey + 1, 256)), dim3(256), 0, s, nkey, total, key_sorted.p, key_first.p);
    hipLaunchKernelGGL(k_key_hist, dim3(cdiv(nkey + 1, 256)), dim3(256), 0, s, nkey, key_first.p, key_hist.p);
    // blocks and chunks
    DTRY(flag.alloc((size_t)nkey + 1)); DTRY(nchunk.alloc((size_t)nkey + 1)); DTRY(blk_of_key.alloc((size_t)nkey + 1)); DTRY(chunk_first.alloc((size_t)nkey + 1));
    DTRY(hipMemsetAsync(flag.p + nkey, 0, sizeof(int), s)); DTRY(hipMemsetAsync(nchunk.p + nkey, 0, sizeof(int), s));
    chunk = chunk_for((long)total, CHUNK);   // (the host build, finish_jobs, takes the same length)
    hipLaunchKernelGGL(k_block_flags, dim3(cdiv(nkey, 256)), dim3(256), 0, s, nkey, ncol, KIND, chunk, key_hist.p, ba->cb_mb.p, flag.p, nchunk.p);
    DTRY(excl_scan(flag.p, blk_of_key.p, (size_t)nkey + 1, s, tmp));
    DTRY(excl_scan(nchunk.p, chunk_first.p, (size_t)nkey + 1, s, tmp));
    DTRY(hipMemcpyAsync(&nbc[0], blk_of_key.p + nkey, sizeof(int), hipMemcpyDeviceToHost, s));
    DTRY(hipMemcpyAsync(&nbc[1], chunk_first.p + nkey, sizeof(int), hipMemcpyDeviceToHost, s));
    return MSFM_OK;
  }
  // (after a synchronisation) the block and chunk lists; the caller synchronises once more before the temporaries go
  int phase3() {
    using namespace devsetup;
    if (empty) return MSFM_OK;
    hipStream_t s = ctx->stream;
    J.n_pairs = total; J.n_blocks = nbc[0]; J.n_chunks = nbc[1];
    DTRY(J.blk_row.alloc((size_t)std::max(1, J.n_blocks))); DTRY(J.blk_col.alloc((size_t)std::max(1, J.n_blocks)));
    DTRY(J.blk_chunk_first.alloc((size_t)J.n_blocks + 1));
    DTRY(J.ch_start.alloc((size_t)std::max(1, J.n_chunks))); DTRY(J.ch_end.alloc((size_t)std::max(1, J.n_chunks)));
    hipLaunchKernelGGL(k_block_lists, dim3(cdiv(nkey, 256)), dim3(256), 0, s, nkey, ncol, flag.p, blk_of_key.p, key_first.p, key_hist.p, chunk_first.p,
                       chunk, J.blk_row.p, J.blk_col.p, J.blk_chunk_first.p, J.ch_start.p, J.ch_end.p);
    DTRY(hipMemcpyAsync(J.blk_chunk_first.p + J.n_blocks, &J.n_chunks, sizeof(int), hipMemcpyHostToDevice, s));
    DTRY(J.partial.alloc((size_t)std::max(1, J.n_chunks) * nout));
    if (want_host_blocks && J.n_blocks) {
      J.h_row.resize(J.n_blocks); J.h_col.resize(J.n_blocks);
      DTRY(hipMemcpyAsync(J.h_row.data(), J.blk_row.p, sizeof(int) * J.n_blocks, hipMemcpyDeviceToHost, s));
      DTRY(hipMemcpyAsync(J.h_col.data(), J.blk_col.p, sizeof(int) * J.n_blocks, hipMemcpyDeviceToHost, s));
    }
    DTRY(hipGetLastError());
    return MSFM_OK;
  }
};

// ---- fold tables (FoldTables above), built from the resident index structures whichever way those were made ----
namespace devsetup {
__global__ __launch_bounds__(256) void k_fold_wg(int npb, int n_wg, const int* __restrict__ pt_first, int* __restrict__ ovf_off, int* __restrict__ wg_fold) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= n_wg) return;
  int ovf = 0, ok = 1;
  for (int pb = 32 * w; pb < min(npb, 32 * w + 32); pb++) {
    const int k = pt_first[pb + 1] - pt_first[pb];
    ovf_off[pb] = ovf;
    if (k > 16) ok = 0;
    ovf += max(0, k - 8);
  }
  wg_fold[w] = ok && ovf <= FOLD_OVF;
}
// block key of a slot: the camera-diagonal blocks first (a workgroup's diagonal slots are then its first ones, and their
// ranks are 0 .. number of diagonal slots - 1: the index of the intrinsics x camera partials)
__host__ __device__ inline unsigned fold_block_key(int row, int col, int ncb) { return row == col ? (unsigned)row : (unsigned)ncb + (unsigned)row * (unsigned)ncb + (unsigned)col; }
template <bool EMIT>
__global__ __launch_bounds__(256) void k_fold_entries(int npb, int ncb, const int* __restrict__ pt_first, const int* __restrict__ o_cb,
                                                       const int* __restrict__ o_cpos, const int* __restrict__ wg_fold, const int* __restrict__ ovf_off,
                                                       int* __restrict__ count, const int* __restrict__ offset, unsigned long long* __restrict__ key,
                                                       unsigned* __restrict__ val) {
  const int pb = blockIdx.x * 256 + threadIdx.x;
  if (pb >= npb) return;
  int n = 0;
  if (wg_fold[pb >> 5]) {
    const int f = pt_first[pb], l = pt_first[pb + 1], w = pb >> 5, base = EMIT ? offset[pb] : 0;
    auto rec = [&](int i) { const int r = i - f; return r < 8 ? ((pb & 31) << 3) + r : 256 + ovf_off[pb] + (r - 8); };
    for (int i = f; i < l; i++) {
      if (o_cpos[i] < 0) continue;
      for (int j = f; j < l; j++) {
        if (o_cpos[j] < 0 || o_cb[i] < o_cb[j]) continue;
        if (EMIT) {
          key[base + n] = ((unsigned long long)w << 32) | fold_block_key(o_cb[i], o_cb[j], ncb);
          val[base + n] = (unsigned)rec(i) | ((unsigned)rec(j) << 16);
        }
        n++;
      }
    }
  }
  if (!EMIT) count[pb] = n;
}
__global__ __launch_bounds__(256) void k_fold_heads(int E, const unsigned long long* __restrict__ key, int* __restrict__ head) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < E) head[e] = (e == 0 || key[e] != key[e - 1]) ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_fold_slots(int E, const unsigned long long* __restrict__ key, const int* __restrict__ head, const int* __restrict__ slot_of,
                                                     int* __restrict__ slot_ent_first, unsigned long long* __restrict__ slot_key2, int* __restrict__ slot_id) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E || !head[e]) return;
  const int sl = slot_of[e];
  slot_ent_first[sl] = e;
  slot_key2[sl] = (key[e] << 32) | (key[e] >> 32);   // block key major, workgroup minor: the order of the partials
  slot_id[sl] = sl;
}
// first slot of every workgroup: slots are sorted by (workgroup, block)
__global__ __launch_bounds__(256) void k_fold_wg_first(int n_wg, int n_slots, const unsigned long long* __restrict__ slot_key2, int* __restrict__ wg_slot_first) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w > n_wg) return;
  int lo = 0, hi = n_slots;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((int)(slot_key2[mid] & 0xffffffffull) < w) lo = mid + 1; else hi = mid;
  }
  wg_slot_first[w] = lo;
}
__global__ __launch_bounds__(256) void k_fold_rank(int n_slots, const int* __restrict__ sorted_id, int* __restrict__ rank) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r < n_slots) rank[sorted_id[r]] = r;
}
// the partials of every camera x camera block: ranks [range[2 b], range[2 b + 1])
__device__ inline int fold_lower_bound(const unsigned long long* __restrict__ sorted_key2, int n, unsigned long long k) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (sorted_key2[mid] < k) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__global__ __launch_bounds__(256) void k_fold_blk_range(int n_blocks, int n_slots, int ncb, const int* __restrict__ blk_row, const int* __restrict__ blk_col,
                                                         const unsigned long long* __restrict__ sorted_key2, int* __restrict__ range) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= n_blocks) return;
  const unsigned long long k = fold_block_key(blk_row[b], blk_col[b], ncb);
  range[2 * b] = fold_lower_bound(sorted_key2, n_slots, k << 32);
  range[2 * b + 1] = fold_lower_bound(sorted_key2, n_slots, (k + 1) << 32);
}
// Order of a workgroup's slots inside its passes: the camera-diagonal ones first (k_point counts on that), then by falling
// entry count - the sixteen slots of a wave then take about equally long (a wave lasts as long as its longest slot: in block
// order the waves' lanes were idle for more than half of the phase).
__global__ __launch_bounds__(256) void k_fold_perm_key(int NS, int ncb, const unsigned long long* __restrict__ slot_key2, const int* __restrict__ slot_ent_first,
                                                        const int* __restrict__ wg_slot_first, unsigned long long* __restrict__ key, int* __restrict__ id) {
  const int sl = blockIdx.x * 256 + threadIdx.x;
  if (sl >= NS) return;
  const unsigned w = (unsigned)(slot_key2[sl] & 0xffffffffull);
  const unsigned dg = (unsigned)(slot_key2[sl] >> 32) < (unsigned)ncb ? 0u : 1u;
  const unsigned cnt = (unsigned)min(0x7fff, slot_ent_first[sl + 1] - slot_ent_first[sl]);
  const unsigned loc = (unsigned)(sl - wg_slot_first[w]) & 0xffffu;
  key[sl] = ((unsigned long long)w << 32) | (dg << 31) | ((0x7fffu - cnt) << 16) | loc;
  id[sl] = sl;
}
__global__ __launch_bounds__(256) void k_fold_gather_rank(int NS, const int* __restrict__ perm, const int* __restrict__ rank, int* __restrict__ rank_pos) {
  const int pos = blockIdx.x * 256 + threadIdx.x;
  if (pos < NS) rank_pos[pos] = rank[perm[pos]];
}
// passes of every workgroup: as many slots as fit FOLD_WORDS words and FOLD_PASS_SLOTS slots.  COUNT: passes and words per
// workgroup; EMIT: the pass descriptors, the slot headers (written into the stream here) and where every slot's entries go.
template <bool EMIT>
__global__ __launch_bounds__(256) void k_fold_passes(int n_wg, int ncb, const int* __restrict__ wg_slot_first, const int* __restrict__ slot_ent_first,
                                                      const unsigned long long* __restrict__ slot_key2, const int* __restrict__ perm, int* __restrict__ wg_npass, int* __restrict__ wg_words,
                                                      const int* __restrict__ pass_first, const int* __restrict__ words_first, FoldPass* __restrict__ pass,
                                                      unsigned* __restrict__ stream, int* __restrict__ slot_ent_pos, int* __restrict__ too_large) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= n_wg) return;
  const int s0 = wg_slot_first[w], s1 = wg_slot_first[w + 1];
  int np = 0, words = 0;
  int a = s0;
  while (a < s1) {
    int b = a, ent = 0;
    while (b < s1 && b - a < FOLD_PASS_SLOTS) {
      const int c = slot_ent_first[perm[b] + 1] - slot_ent_first[perm[b]];
      if (b > a && ((b + 1 - a + 3) & ~3) + ent + c > FOLD_WORDS) break;   // (one slot always fits: at most one entry per pair of records)
      ent += c; b++;
    }
    const int hdr = (b - a + 3) & ~3, pw = (hdr + ent + 3) & ~3;
    if (EMIT) {
      const int off = words_first[w] + words;
      int nd = 0, rel = 0;
      for (int pos = a; pos < b; pos++) {
        const int sl = perm[pos];
        const int c = slot_ent_first[sl + 1] - slot_ent_first[sl];
        if ((unsigned)(slot_key2[sl] >> 32) < (unsigned)ncb) nd++;
        stream[off + (pos - a)] = (unsigned)rel | ((unsigned)c << 16);
        slot_ent_pos[sl] = off + hdr + rel;
        rel += c;
      }
      FoldPass fp;
      fp.off = off; fp.words = pw; fp.slot0 = a; fp.n_slots = b - a; fp.n_diag = nd; fp.pad0 = fp.pad1 = fp.pad2 = 0;
      pass[pass_first[w] + np] = fp;
    }
    // one slot alone does not fit the staging area (thousands of repeated observations of one camera pair inside 32 points):
    // the host reads the flag beside the counts and keeps the gather path (a flag of its own: the counts are summed by a
    // 32-bit scan, a sentinel inside them could wrap)
    if (pw > FOLD_WORDS && too_large) atomicOr(too_large, 1);
    np++; words += pw;
    a = b;
  }
  if (!EMIT) { wg_npass[w] = np; wg_words[w] = words; }
}
__global__ __launch_bounds__(256) void k_fold_stream_entries(int E, const int* __restrict__ head, const int* __restrict__ slot_of, const int* __restrict__ slot_ent_first,
                                                              const int* __restrict__ slot_ent_pos, const unsigned* __restrict__ ent, unsigned* __restrict__ stream) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const int sl = slot_of[e] + head[e] - 1;   // (slot_of: exclusive count of slot heads)
  stream[slot_ent_pos[sl] + (e - slot_ent_first[sl])] = ent[e];
}
__global__ __launch_bounds__(256) void k_fold_mark_cpos(int npb, const int* __restrict__ pt_first, const int* __restrict__ o_cpos, const int* __restrict__ wg_fold,
                                                         uint8_t* __restrict__ folded) {
  const int pb = blockIdx.x * 256 + threadIdx.x;
  if (pb >= npb || !wg_fold[pb >> 5]) return;
  for (int i = pt_first[pb]; i < pt_first[pb + 1]; i++) if (o_cpos[i] >= 0) folded[o_cpos[i]] = 1;
}
__global__ __launch_bounds__(256) void k_fold_live_flags(int nch, const int* __restrict__ ch_start, const int* __restrict__ ch_end, const int* __restrict__ pa,
                                                          int* __restrict__ flag) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;   // one wave per chunk
  if (c >= nch) return;
  int any = 0;
  for (int e = ch_start[c] + lane; e < ch_end[c]; e += 64) any |= pa[e] >= 0;
  any = __any(any);
  if (lane == 0) flag[c] = any;
}
__global__ __launch_bounds__(256) void k_fold_blk_live(int n_blocks, const int* __restrict__ blk_chunk_first, const int* __restrict__ flag, uint8_t* __restrict__ live) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= n_blocks) return;
  int any = 0;
  for (int c = blk_chunk_first[b]; c < blk_chunk_first[b + 1]; c++) any |= flag[c];
  live[b] = (uint8_t)any;
}
__global__ __launch_bounds__(256) void k_fold_live_list(int nch, const int* __restrict__ flag, const int* __restrict__ pos, int* __restrict__ live) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < nch && flag[c]) live[pos[c]] = c;
}
// the partials of every intrinsics x camera block: the ranks of the camera's diagonal slots
__global__ __launch_bounds__(256) void k_fold_mc_range(int n_blocks, int n_slots, const int* __restrict__ blk_col, const unsigned long long* __restrict__ sorted_key2,
                                                        int* __restrict__ range) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= n_blocks) return;
  const unsigned long long k = (unsigned)blk_col[b];
  range[2 * b] = fold_lower_bound(sorted_key2, n_slots, k << 32);
  range[2 * b + 1] = fold_lower_bound(sorted_key2, n_slots, (k + 1) << 32);
}
// (grid-stride, one atomic add per workgroup: one per wave on a single counter was 19 k serialised atomics, 0.2 ms, at config 3)
__global__ __launch_bounds__(256) void k_fold_count_marked(int n, const int* __restrict__ pa, int* __restrict__ count) {
  __shared__ int part[4];
  int c = 0;
  for (int e = blockIdx.x * 256 + threadIdx.x; e - (int)threadIdx.x < n; e += gridDim.x * 256) c += __popcll(__ballot(e < n && pa[e] < 0));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) { const int t = part[0] + part[1] + part[2] + part[3]; if (t) atomicAdd(count, t); }
}
__global__ __launch_bounds__(256) void k_fold_mark_mc(int n, const uint8_t* __restrict__ folded, int* __restrict__ pa, const int* __restrict__ pb) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < n && pa[e] >= 0 && folded[pb[e]]) pa[e] = ~pa[e];
}
__global__ __launch_bounds__(256) void k_fold_mark_pairs(int n, const uint8_t* __restrict__ folded, int* __restrict__ pa) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < n && pa[e] >= 0 && folded[pa[e]]) pa[e] = ~pa[e];   // (both records of an entry belong to one point)
}
}  // namespace devsetup

static int build_fold_device(msfm_ctx* ctx, msfm_ba* ba) {
  using namespace devsetup;
  FoldTables& F = ba->fold;
  F.on = false;
  static const bool off = getenv("MSFM_NO_FOLD") != nullptr;
  const int npb = ba->npb, ncb = ba->ncb;
  // (small problems keep the gather path: their pair list fits the caches anyway and the tables would only lengthen the set-up;
  //  MSFM_FOLD_MIN overrides the threshold, e.g. 0 to fold everything in a test)
  static const long fold_min = getenv("MSFM_FOLD_MIN") ? atol(getenv("MSFM_FOLD_MIN")) : 262144;
  if (off || npb == 0 || ncb == 0 || ba->cc.n_pairs == 0 || ba->cc.n_pairs < fold_min || (long)ncb * (ncb + 1) > 0x7fffffffL) return MSFM_OK;
  hipStream_t s = ctx->stream;
  DevBuf<char> tmp;
  const int n_wg = cdiv(npb, 32);
  F.n_wg = n_wg;
  static const bool laps = getenv("MSFM_FOLD_LAPS") != nullptr;   // developer switch: where the set-up time goes
  auto t_last = std::chrono::steady_clock::now();
  auto flap = [&](const char* what) {
    if (!laps) return;
    (void)hipStreamSynchronize(s);
    const auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "msfm: fold set-up %-28s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
    t_last = t;
  };
  DTRY(F.wg_fold.alloc(n_wg)); DTRY(F.ovf_off.alloc(npb));
  hipLaunchKernelGGL(k_fold_wg, dim3(cdiv(n_wg, 256)), dim3(256), 0, s, npb, n_wg, ba->pt_first.p, F.ovf_off.p, F.wg_fold.p);
  DevBuf<int> count, offset;
  DTRY(count.alloc((size_t)npb + 1)); DTRY(offset.alloc((size_t)npb + 1));
  DTRY(hipMemsetAsync(count.p, 0, sizeof(int) * ((size_t)npb + 1), s));
  hipLaunchKernelGGL((k_fold_entries<false>), dim3(cdiv(npb, 256)), dim3(256), 0, s, npb, ncb, ba->pt_first.p, ba->o_cb.p, ba->o_cpos.p, F.wg_fold.p,
                     F.ovf_off.p, count.p, (const int*)nullptr, (unsigned long long*)nullptr, (unsigned*)nullptr);
  DTRY(excl_scan(count.p, offset.p, (size_t)npb + 1, s, tmp));
  int E = 0;
  DTRY(hipMemcpyAsync(&E, offset.p + npb, sizeof(int), hipMemcpyDeviceToHost, s));
  DTRY(hipStreamSynchronize(s));
  if (E <= 0) return MSFM_OK;
  flap("count entries");
  DevBuf<unsigned long long> key, key_s, slot_key2, slot_key2_s;
  DevBuf<unsigned> val;
  DevBuf<int> head, slot_of, slot_id, slot_id_s;
  DevBuf<unsigned> ent_sorted;
  DTRY(key.alloc(E)); DTRY(key_s.alloc(E)); DTRY(val.alloc(E)); DTRY(ent_sorted.alloc(E));
  hipLaunchKernelGGL((k_fold_entries<true>), dim3(cdiv(npb, 256)), dim3(256), 0, s, npb, ncb, ba->pt_first.p, ba->o_cb.p, ba->o_cpos.p, F.wg_fold.p,
                     F.ovf_off.p, (int*)nullptr, offset.p, key.p, val.p);
  {
    size_t bytes = 0;
    const int bits = 32 + bits_for(std::max(2, n_wg));
    DTRY(rocprim::radix_sort_pairs(nullptr, bytes, key.p, key_s.p, val.p, ent_sorted.p, (size_t)E, 0, bits, s));
    if (tmp.n < bytes) DTRY(tmp.alloc(bytes));
    flap("emit entries");
    DTRY(rocprim::radix_sort_pairs(tmp.p, bytes, key.p, key_s.p, val.p, ent_sorted.p, (size_t)E, 0, bits, s));   // stable: point order inside a slot
  }
  flap("sort entries");
  DTRY(head.alloc((size_t)E + 1)); DTRY(slot_of.alloc((size_t)E + 1));
  DTRY(hipMemsetAsync(head.p + E, 0, sizeof(int), s));
  hipLaunchKernelGGL(k_fold_heads, dim3(cdiv(E, 256)), dim3(256), 0, s, E, key_s.p, head.p);
  DTRY(excl_scan(head.p, slot_of.p, (size_t)E + 1, s, tmp));
  int NS = 0;
  DTRY(hipMemcpyAsync(&NS, slot_of.p + E, sizeof(int), hipMemcpyDeviceToHost, s));
  DTRY(hipStreamSynchronize(s));
  DevBuf<int> slot_ent_first;
  DTRY(slot_ent_first.alloc((size_t)NS + 1)); DTRY(slot_key2.alloc(NS)); DTRY(slot_key2_s.alloc(NS)); DTRY(slot_id.alloc(NS)); DTRY(slot_id_s.alloc(NS));
  hipLaunchKernelGGL(k_fold_slots, dim3(cdiv(E, 256)), dim3(256), 0, s, E, key_s.p, head.p, slot_of.p, slot_ent_first.p, slot_key2.p, slot_id.p);
  DTRY(hipMemcpyAsync(slot_ent_first.p + NS, &E, sizeof(int), hipMemcpyHostToDevice, s));
  DevBuf<int> wg_slot_first;
  DTRY(wg_slot_first.alloc((size_t)n_wg + 1));
  hipLaunchKernelGGL(k_fold_wg_first, dim3(cdiv(n_wg + 1, 256)), dim3(256), 0, s, n_wg, NS, slot_key2.p, wg_slot_first.p);
  flap("slots");
  // the passes and the stream they read
  DevBuf<int> perm;
  {
    DevBuf<unsigned long long> pk, pk_s;
    DevBuf<int> pid;
    DTRY(pk.alloc(NS)); DTRY(pk_s.alloc(NS)); DTRY(pid.alloc(NS)); DTRY(perm.alloc(NS));
    hipLaunchKernelGGL(k_fold_perm_key, dim3(cdiv(NS, 256)), dim3(256), 0, s, NS, ncb, slot_key2.p, slot_ent_first.p, wg_slot_first.p, pk.p, pid.p);
    size_t bytes = 0;
    const int bits = 32 + bits_for(std::max(2, n_wg));
    DTRY(rocprim::radix_sort_pairs(nullptr, bytes, pk.p, pk_s.p, pid.p, perm.p, (size_t)NS, 0, bits, s));
    if (tmp.n < bytes) DTRY(tmp.alloc(bytes));
    DTRY(rocprim::radix_sort_pairs(tmp.p, bytes, pk.p, pk_s.p, pid.p, perm.p, (size_t)NS, 0, bits, s));
    DTRY(hipStreamSynchronize(s));
  }
  flap("slot order");
  {
    DevBuf<int> wg_npass, wg_words, words_first, slot_ent_pos;
    DTRY(wg_npass.alloc((size_t)n_wg + 1)); DTRY(wg_words.alloc((size_t)n_wg + 1)); DTRY(words_first.alloc((size_t)n_wg + 1));
    DTRY(F.wg_pass_first.alloc((size_t)n_wg + 1)); DTRY(slot_ent_pos.alloc(NS));
    DevBuf<int> too_large;
    DTRY(too_large.alloc(1)); DTRY(hipMemsetAsync(too_large.p, 0, sizeof(int), s));
    DTRY(hipMemsetAsync(wg_npass.p + n_wg, 0, sizeof(int), s)); DTRY(hipMemsetAsync(wg_words.p + n_wg, 0, sizeof(int), s));
    hipLaunchKernelGGL((k_fold_passes<false>), dim3(cdiv(n_wg, 256)), dim3(256), 0, s, n_wg, ncb, wg_slot_first.p, slot_ent_first.p, slot_key2.p, perm.p, wg_npass.p, wg_words.p,
                       (const int*)nullptr, (const int*)nullptr, (FoldPass*)nullptr, (unsigned*)nullptr, (int*)nullptr, too_large.p);
    DTRY(excl_scan(wg_npass.p, F.wg_pass_first.p, (size_t)n_wg + 1, s, tmp));
    DTRY(excl_scan(wg_words.p, words_first.p, (size_t)n_wg + 1, s, tmp));
    int npass = 0, nwords = 0, h_too_large = 0;
    DTRY(hipMemcpyAsync(&h_too_large, too_large.p, sizeof(int), hipMemcpyDeviceToHost, s));
    DTRY(hipMemcpyAsync(&npass, F.wg_pass_first.p + n_wg, sizeof(int), hipMemcpyDeviceToHost, s));
    DTRY(hipMemcpyAsync(&nwords, words_first.p + n_wg, sizeof(int), hipMemcpyDeviceToHost, s));
    DTRY(hipStreamSynchronize(s));
    if (h_too_large || npass < 0 || nwords < 0) return MSFM_OK;   // (k_fold_passes: a slot too large to stage; nothing has been marked yet)
    DTRY(F.pass.alloc((size_t)std::max(1, npass))); DTRY(F.stream.alloc((size_t)nwords + FOLD_WORDS));
    DTRY(hipMemsetAsync(F.stream.p, 0, sizeof(unsigned) * ((size_t)nwords + FOLD_WORDS), s));
    hipLaunchKernelGGL((k_fold_passes<true>), dim3(cdiv(n_wg, 256)), dim3(256), 0, s, n_wg, ncb, wg_slot_first.p, slot_ent_first.p, slot_key2.p, perm.p, (int*)nullptr, (int*)nullptr,
                       F.wg_pass_first.p, words_first.p, F.pass.p, F.stream.p, slot_ent_pos.p, (int*)nullptr);
    hipLaunchKernelGGL(k_fold_stream_entries, dim3(cdiv(E, 256)), dim3(256), 0, s, E, head.p, slot_of.p, slot_ent_first.p, slot_ent_pos.p, ent_sorted.p, F.stream.p);
    F.n_pass = npass;
    DTRY(hipStreamSynchronize(s));
  }
  {
    size_t bytes = 0;
    DTRY(rocprim::radix_sort_pairs(nullptr, bytes, slot_key2.p, slot_key2_s.p, slot_id.p, slot_id_s.p, (size_t)NS, 0, 64, s));
    if (tmp.n < bytes) DTRY(tmp.alloc(bytes));
    DTRY(rocprim::radix_sort_pairs(tmp.p, bytes, slot_key2.p, slot_key2_s.p, slot_id.p, slot_id_s.p, (size_t)NS, 0, 64, s));
  }
  {
    DevBuf<int> rank;   // by slot; k_point asks by position inside the workgroup's passes
    DTRY(rank.alloc(NS)); DTRY(F.slot_rank.alloc(NS));
    hipLaunchKernelGGL(k_fold_rank, dim3(cdiv(NS, 256)), dim3(256), 0, s, NS, slot_id_s.p, rank.p);
    hipLaunchKernelGGL(k_fold_gather_rank, dim3(cdiv(NS, 256)), dim3(256), 0, s, NS, perm.p, rank.p, F.slot_rank.p);
    DTRY(hipStreamSynchronize(s));
  }
  DTRY(F.blk_range.alloc(2 * (size_t)std::max(1, ba->cc.n_blocks)));
  hipLaunchKernelGGL(k_fold_blk_range, dim3(cdiv(std::max(1, ba->cc.n_blocks), 256)), dim3(256), 0, s, ba->cc.n_blocks, NS, ncb, ba->cc.blk_row.p, ba->cc.blk_col.p,
                     slot_key2_s.p, F.blk_range.p);
  flap("passes, stream, ranks, ranges");
  // the same entries leave the gather path
  DevBuf<uint8_t> folded;
  DTRY(folded.alloc((size_t)std::max(1, ba->NCR)));
  DTRY(hipMemsetAsync(folded.p, 0, (size_t)std::max(1, ba->NCR), s));
  hipLaunchKernelGGL(k_fold_mark_cpos, dim3(cdiv(npb, 256)), dim3(256), 0, s, npb, ba->pt_first.p, ba->o_cpos.p, F.wg_fold.p, folded.p);
  hipLaunchKernelGGL(k_fold_mark_pairs, dim3(cdiv(ba->cc.n_pairs, 256)), dim3(256), 0, s, ba->cc.n_pairs, folded.p, ba->cc.pa.p);
  DTRY(F.partial.alloc((size_t)NS * 36));
  // chunks that keep a live entry: the gather kernel visits only those, the partials of the others are zero for good
  auto live_chunks = [&](PairJobs& J, int width, int& n_live, DevBuf<int>& live, DevBuf<uint8_t>* blk_live) -> int {
    DevBuf<int> lf, lpos;
    const int nch = J.n_chunks;
    DTRY(lf.alloc((size_t)nch + 1)); DTRY(lpos.alloc((size_t)nch + 1));
    DTRY(hipMemsetAsync(lf.p + nch, 0, sizeof(int), s));
    hipLaunchKernelGGL(k_fold_live_flags, dim3(cdiv(std::max(1, nch), 4)), dim3(256), 0, s, nch, J.ch_start.p, J.ch_end.p, J.pa.p, lf.p);
    DTRY(excl_scan(lf.p, lpos.p, (size_t)nch + 1, s, tmp));
    DTRY(hipMemcpyAsync(&n_live, lpos.p + nch, sizeof(int), hipMemcpyDeviceToHost, s));
    DTRY(hipStreamSynchronize(s));
    DTRY(live.alloc((size_t)std::max(1, n_live)));
    hipLaunchKernelGGL(k_fold_live_list, dim3(cdiv(std::max(1, nch), 256)), dim3(256), 0, s, nch, lf.p, lpos.p, live.p);
    if (blk_live) {
      DTRY(blk_live->alloc((size_t)std::max(1, J.n_blocks)));
      hipLaunchKernelGGL(k_fold_blk_live, dim3(cdiv(std::max(1, J.n_blocks), 256)), dim3(256), 0, s, J.n_blocks, J.blk_chunk_first.p, lf.p, blk_live->p);
    }
    DTRY(hipMemsetAsync(J.partial.p, 0, sizeof(double) * width * (size_t)std::max(1, nch), s));
    DTRY(hipStreamSynchronize(s));
    return MSFM_OK;
  };
  MSFM_TRY(live_chunks(ba->cc, 36, F.n_live, F.live_chunk, &F.blk_live));
  flap("mark + live chunks");
  // intrinsics x camera list: with ONE intrinsics block every point has at most one (point, intrinsics) entry and the
  // products Tm_p T^T of a camera are summed over exactly the records of its diagonal slots
  F.mc_on = false;
  static const bool mc_off = getenv("MSFM_NO_FOLD_MC") != nullptr;
  if (!mc_off && ba->nmb == 1 && ba->mc.n_pairs > 0) {
    DTRY(F.mc_range.alloc(2 * (size_t)ba->mc.n_blocks));
    hipLaunchKernelGGL(k_fold_mc_range, dim3(cdiv(ba->mc.n_blocks, 256)), dim3(256), 0, s, ba->mc.n_blocks, NS, ba->mc.blk_col.p, slot_key2_s.p, F.mc_range.p);
    // (diagonal keys are the ncb smallest: their slots have the ranks 0 .. n_diag - 1)
    {
      DevBuf<int> one;
      DTRY(one.alloc(2));
      int col = ncb;   // k_fold_mc_range on a single pseudo block with column ncb gives lower_bound(ncb << 32) in range[0]
      DevBuf<int> colbuf;
      DTRY(colbuf.alloc(1));
      DTRY(hipMemcpyAsync(colbuf.p, &col, sizeof(int), hipMemcpyHostToDevice, s));
      hipLaunchKernelGGL(k_fold_mc_range, dim3(1), dim3(256), 0, s, 1, NS, colbuf.p, slot_key2_s.p, one.p);
      int h[2] = {0, 0};
      DTRY(hipMemcpyAsync(h, one.p, sizeof(h), hipMemcpyDeviceToHost, s));
      DTRY(hipStreamSynchronize(s));
      F.n_diag = h[0];
    }
    if (F.n_diag > 0) {
      DTRY(F.mc_partial.alloc((size_t)F.n_diag * 18));
      DTRY(hipMemsetAsync(F.mc_partial.p, 0, sizeof(double) * 18 * (size_t)F.n_diag, s));
      hipLaunchKernelGGL(k_fold_mark_mc, dim3(cdiv(ba->mc.n_pairs, 256)), dim3(256), 0, s, ba->mc.n_pairs, folded.p, ba->mc.pa.p, ba->mc.pb.p);
      MSFM_TRY(live_chunks(ba->mc, 18, F.mc_n_live, F.mc_live_chunk, nullptr));
      {
        // (how many entries left the gather list: reported by msfm_ba_get_layout)
        DevBuf<int> cnt;
        DTRY(cnt.alloc(1));
        DTRY(hipMemsetAsync(cnt.p, 0, sizeof(int), s));
        hipLaunchKernelGGL(k_fold_count_marked, dim3(std::min(1024, cdiv(ba->mc.n_pairs, 256))), dim3(256), 0, s, ba->mc.n_pairs, ba->mc.pa.p, cnt.p);
        int h = 0;
        DTRY(hipMemcpyAsync(&h, cnt.p, sizeof(int), hipMemcpyDeviceToHost, s));
        DTRY(hipStreamSynchronize(s));
        F.mc_entries_folded = h;
      }
      F.mc_on = true;
      F.mc_all = F.mc_n_live == 0;
    }
  }
  flap("intrinsics x camera");
  DTRY(hipGetLastError());
  DTRY(hipStreamSynchronize(s));   // the temporaries above go back to the pool
  F.n_slots = NS; F.n_entries = E;
  F.on = true;
  F.all = E == ba->cc.n_pairs;
  if (getenv("MSFM_VERBOSE") && ctx->rank == 0)
    fprintf(stderr, "msfm: fold tables: %d workgroups, %d passes, %d slots (%.1f MB of partials), %d of %d entries folded; intrinsics x camera: %s, %d diagonal slots, %d of %d chunks live\n",
            n_wg, F.n_pass, NS, NS * 288e-6, E, ba->cc.n_pairs, F.mc_on ? "folded" : "gathered", F.n_diag, F.mc_n_live, ba->mc.n_chunks);
  return MSFM_OK;
}

// Everything msfm_ba_create needs between the caller's arrays and the allocation of the work buffers, on the device.
static int create_structures_device(msfm_ctx* ctx, const msfm_ba_problem* P, msfm_ba* ba, const std::function<void(const char*)>& lap,
                                    bool bulk_on_device) {
  using namespace devsetup;
  hipStream_t s = ctx->stream;
  // bulk arrays that are already resident (msfm_chain_ba_create) are copied device to device
  auto bulk = [&](void* dst, const void* src, size_t bytes) {
    return hipMemcpyAsync(dst, src, bytes, bulk_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s);
  };
  const int Nc = ba->Nc = P->n_cams, Nm = ba->Nm = P->n_models, Np = ba->Np = P->n_points, No = P->n_obs;
  DevBuf<char> tmp;
  // ---- the caller's arrays (the only bulk PCIe traffic of the set-up) ----
  DevBuf<int> d_obs_cam, d_obs_pt, d_model_of_cam;
  DevBuf<double> d_obs_xy, d_ptw;
  DevBuf<uint8_t> d_cam_mut, d_model_mut, d_pt_mut;
  DTRY(d_obs_cam.alloc((size_t)std::max(1, No))); DTRY(d_obs_pt.alloc((size_t)std::max(1, No))); DTRY(d_obs_xy.alloc(2 * (size_t)std::max(1, No)));
  DTRY(d_model_of_cam.alloc(Nc));
  if (No) {
    DTRY(bulk(d_obs_cam.p, P->obs_cam, sizeof(int) * (size_t)No)); DTRY(bulk(d_obs_pt.p, P->obs_pt, sizeof(int) * (size_t)No));
    DTRY(bulk(d_obs_xy.p, P->obs_xy, sizeof(double) * 2 * (size_t)No));
  }
  DTRY(d_model_of_cam.upload(P->cam_model_of_cam, Nc, s));
  if (P->pt_weight && Np) { DTRY(d_ptw.alloc(Np)); DTRY(bulk(d_ptw.p, P->pt_weight, sizeof(double) * (size_t)Np)); }
  if (P->cam_mutable) { DTRY(d_cam_mut.alloc(Nc)); DTRY(d_cam_mut.upload(P->cam_mutable, Nc, s)); }
  if (P->model_mutable) { DTRY(d_model_mut.alloc(Nm)); DTRY(d_model_mut.upload(P->model_mutable, Nm, s)); }
  if (P->pt_mutable && Np) { DTRY(d_pt_mut.alloc(Np)); DTRY(bulk(d_pt_mut.p, P->pt_mutable, (size_t)Np)); }
  // ---- validation, block usage, observations per point ----
  DevBuf<uint8_t> d_cu, d_mu, d_pu;
  DevBuf<int> d_cnt, d_run_first, d_err;
  DTRY(d_cu.alloc(Nc)); DTRY(d_mu.alloc(Nm)); DTRY(d_pu.alloc((size_t)std::max(1, Np)));
  DTRY(d_cnt.alloc((size_t)Np + 1)); DTRY(d_run_first.alloc((size_t)Np + 1)); DTRY(d_err.alloc(4));
  DTRY(hipMemsetAsync(d_cu.p, 0, Nc, s)); DTRY(hipMemsetAsync(d_mu.p, 0, Nm, s)); DTRY(hipMemsetAsync(d_pu.p, 0, (size_t)std::max(1, Np), s));
  DTRY(hipMemsetAsync(d_cnt.p, 0, sizeof(int) * ((size_t)Np + 1), s));
  DTRY(hipMemsetAsync(d_err.p, 0x7f, sizeof(int) * 4, s));
  if (No) hipLaunchKernelGGL(k_scan_obs, dim3(cdiv(No, 256)), dim3(256), 0, s, No, Nc, Np, d_obs_cam.p, d_obs_pt.p, d_model_of_cam.p, d_cam_mut.p,
                             d_model_mut.p, d_pt_mut.p, d_cu.p, d_mu.p, d_pu.p, d_cnt.p, d_err.p);
  DTRY(excl_scan(d_cnt.p, d_run_first.p, (size_t)Np + 1, s, tmp));
  std::vector<uint8_t> cu(Nc), mu(Nm);
  int err[4];
  DTRY(hipMemcpyAsync(cu.data(), d_cu.p, Nc, hipMemcpyDeviceToHost, s));
  DTRY(hipMemcpyAsync(mu.data(), d_mu.p, Nm, hipMemcpyDeviceToHost, s));
  DTRY(hipMemcpyAsync(err, d_err.p, sizeof err, hipMemcpyDeviceToHost, s));
  DTRY(hipStreamSynchronize(s));
  if (err[0] < 0x7f7f7f7f && (err[1] == 0x7f7f7f7f || err[0] <= err[1])) return msfm_set_error(ctx, MSFM_E_INVAL, "observation %d: index out of range", err[0]);
  if (err[1] < 0x7f7f7f7f) return msfm_set_error(ctx, MSFM_E_INVAL, "obs_pt must be non-decreasing (gather order, optimizer.cc:62)");
  lap("uploaded + scanned");
  // ---- slots of the camera / intrinsics blocks (host: O(cameras)) ----
  if (P->gps_xyz) for (int c = 0; c < Nc; c++) if (is_mut(P->cam_mutable, c)) cu[c] = 1;
  if (ctx->world > 1)
    for (int c = 0; c < Nc; c++) if (is_mut(P->cam_mutable, c)) { cu[c] = 1; if (is_mut(P->model_mutable, P->cam_model_of_cam[c])) mu[P->cam_model_of_cam[c]] = 1; }
  std::vector<int> cam_slot(Nc, -1), model_slot(Nm, -1);
  for (int m = 0; m < Nm; m++) if (mu[m]) { model_slot[m] = ba->nmb++; ba->h_mb_model.push_back(m); }
  std::vector<int> gnode(Nc, -1), gcam;
  for (int c = 0; c < Nc; c++) if (cu[c]) { gnode[c] = (int)gcam.size(); gcam.push_back(c); }
  const int ng = (int)gcam.size();
  const char* env = getenv("MSFM_CHOL_DOMAINS");
  const int force = env ? atoi(env) : -1;
  std::vector<uint8_t> adjb;
  if (ng >= 128 && ng <= 4096 && force != 0) {
    DevBuf<int> d_gnode;
    DevBuf<uint8_t> d_adj;
    DTRY(d_gnode.from(gnode, s));
    DTRY(d_adj.alloc((size_t)ng * ng));
    DTRY(hipMemsetAsync(d_adj.p, 0, (size_t)ng * ng, s));
    if (Np) hipLaunchKernelGGL(k_adjacency, dim3(cdiv(Np, 256)), dim3(256), 0, s, Np, d_run_first.p, d_obs_cam.p, d_pt_mut.p, d_gnode.p, ng, d_adj.p);
    adjb.resize((size_t)ng * ng);
    DTRY(hipMemcpyAsync(adjb.data(), d_adj.p, adjb.size(), hipMemcpyDeviceToHost, s));
    DTRY(hipStreamSynchronize(s));
    if (ctx->world > 1) {   // the graph must be the same on every rank: max-reduce over the ranks' shards
      std::vector<double> adjm(adjb.size());
      for (size_t k = 0; k < adjm.size(); k++) adjm[k] = adjb[k] ? 1.0 : 0.0;
      DevBuf<double> dadj;
      DTRY(dadj.from(adjm, s));
      DTRY(hipStreamSynchronize(s));
      const int rc = ctx->allreduce(ctx->allreduce_user, dadj.p, adjm.size(), MSFM_REDUCE_MAX, (void*)s);
      if (rc != 0) return msfm_set_error(ctx, MSFM_E_DEVICE, "all-reduce hook failed: %d", rc);
      DTRY(hipMemcpyAsync(adjm.data(), dadj.p, sizeof(double) * adjm.size(), hipMemcpyDeviceToHost, s));
      MSFM_TRY(msfm_stream_wait_bounded(ctx, s, "the all-reduce of the camera graph"));   // (bounded: a peer that failed never joins)
      for (size_t k = 0; k < adjm.size(); k++) adjb[k] = adjm[k] != 0.0;
    }
  }
  std::vector<int> cb_off_h, padcol_h;
  order_camera_blocks(ctx, ba, gcam, adjb, force, cam_slot, cb_off_h, padcol_h);
  const int ncb = ba->ncb, nmb = ba->nmb;
  ba->nred = 6 * ncb + 3 * nmb;
  ba->npad = 64 * cdiv(ba->nsys + 1, 64);
  if (ba->plan.n_levels > 0) {
    ba->plan.ldc = 64 * cdiv(ba->nsys + 1 - ba->plan.level[0].b0, 64);
    DTRY(ba->corners.alloc((size_t)8 * ba->plan.ldc * ba->plan.ldc));   // up to eight K-splits of the corner update (MSFM_CORNER_SPLIT_MAX)
    ba->plan.corners = ba->corners.p;
  }
  ba->gps_weight = P->gps_weight;
  DTRY(ba->cb_off.from(cb_off_h.empty() ? std::vector<int>(1, 0) : cb_off_h, s));
  DTRY(ba->padcol.from(padcol_h.empty() ? std::vector<int>(1, 0) : padcol_h, s));
  DevBuf<int> d_cam_slot, d_model_slot;
  DTRY(d_cam_slot.from(cam_slot, s)); DTRY(d_model_slot.from(model_slot, s));
  // cameras of each intrinsics block (host, O(cameras))
  std::vector<int> cb_mb(std::max(1, ncb), -1), mcam_first(nmb + 1, 0), mcam;
  for (int cb = 0; cb < ncb; cb++) {
    const int m = P->cam_model_of_cam[ba->h_cb_cam[cb]];
    cb_mb[cb] = is_mut(P->model_mutable, m) ? model_slot[m] : -1;
  }
  for (int mb = 0; mb < nmb; mb++) {
    mcam_first[mb] = (int)mcam.size();
    for (int cb = 0; cb < ncb; cb++) if (cb_mb[cb] == mb) mcam.push_back(cb);
  }
  mcam_first[nmb] = (int)mcam.size();
  DTRY(ba->cb_cam.from(ba->h_cb_cam.empty() ? std::vector<int>(1, 0) : ba->h_cb_cam, s));
  DTRY(ba->mb_model.from(ba->h_mb_model.empty() ? std::vector<int>(1, 0) : ba->h_mb_model, s));
  DTRY(ba->cb_mb.from(cb_mb, s)); DTRY(ba->mcam_first.from(mcam_first, s));
  if (!mcam.empty()) DTRY(ba->mcam.from(mcam, s));
  std::vector<double> gps_cb;
  ba->has_gps = P->gps_xyz != nullptr;
  if (ba->has_gps) {
    gps_cb.resize(3 * (size_t)std::max(1, ncb));
    for (int cb = 0; cb < ncb; cb++) for (int k = 0; k < 3; k++) gps_cb[3 * (size_t)cb + k] = P->gps_xyz[3 * (size_t)ba->h_cb_cam[cb] + k];
    DTRY(ba->gps.from(gps_cb, s));
  }
  lap("camera order");
  // ---- order of the eliminated points: by their smallest camera blocks, ties by the caller's index (stable sort) ----
  DevBuf<int> d_pu_int, d_pu_pos, d_vals, d_vals_sorted, d_pt_slot, d_len;
  DevBuf<unsigned long long> d_keys, d_keys_sorted;
  DTRY(d_pu_int.alloc((size_t)Np + 1)); DTRY(d_pu_pos.alloc((size_t)Np + 1));
  DTRY(hipMemsetAsync(d_pu_int.p + Np, 0, sizeof(int), s));
  if (Np) hipLaunchKernelGGL(k_u8_to_int, dim3(cdiv(Np, 256)), dim3(256), 0, s, Np, d_pu.p, d_pu_int.p);
  DTRY(excl_scan(d_pu_int.p, d_pu_pos.p, (size_t)Np + 1, s, tmp));
  int npb = 0;
  DTRY(hipMemcpyAsync(&npb, d_pu_pos.p + Np, sizeof(int), hipMemcpyDeviceToHost, s));
  DTRY(hipStreamSynchronize(s));
  ba->npb = npb;
  DTRY(ba->pb_pt.alloc((size_t)std::max(1, npb)));
  const bool order_points = !getenv("MSFM_POINT_ORDER") || atoi(getenv("MSFM_POINT_ORDER")) != 0;
  if (npb) {
    if (order_points) {
      DTRY(d_keys.alloc(npb)); DTRY(d_keys_sorted.alloc(npb)); DTRY(d_vals.alloc(npb));
      hipLaunchKernelGGL(k_point_keys, dim3(cdiv(Np, 256)), dim3(256), 0, s, Np, d_pu.p, d_pu_pos.p, d_run_first.p, d_obs_cam.p, d_cam_slot.p,
                         ncb >= 0xFFFF ? 1 : 0, d_keys.p, d_vals.p);
      DTRY(sort_pairs(d_keys.p, d_keys_sorted.p, d_vals.p, ba->pb_pt.p, (size_t)npb, 64, s, tmp));
    } else {
      hipLaunchKernelGGL(k_compact_used, dim3(cdiv(Np, 256)), dim3(256), 0, s, Np, d_pu.p, d_pu_pos.p, ba->pb_pt.p);
    }
  }
  DTRY(d_pt_slot.alloc((size_t)std::max(1, Np))); DTRY(d_len.alloc((size_t)npb + 1)); DTRY(ba->pt_first.alloc((size_t)npb + 1));
  DTRY(hipMemsetAsync(d_len.p + npb, 0, sizeof(int), s));
  if (npb) hipLaunchKernelGGL(k_point_lengths, dim3(cdiv(npb, 256)), dim3(256), 0, s, npb, ba->pb_pt.p, d_cnt.p, d_len.p, d_pt_slot.p);
  DTRY(excl_scan(d_len.p, ba->pt_first.p, (size_t)npb + 1, s, tmp));
  // frozen points seen by free cameras: flags and their ranks
  DevBuf<int> d_fflag, d_fpos;
  int n_frozen_rows = 0;
  if (P->pt_mutable && No) {
    DTRY(d_fflag.alloc((size_t)No + 1)); DTRY(d_fpos.alloc((size_t)No + 1));
    DTRY(hipMemsetAsync(d_fflag.p + No, 0, sizeof(int), s));
    hipLaunchKernelGGL(k_flag_frozen, dim3(cdiv(No, 256)), dim3(256), 0, s, No, d_obs_cam.p, d_obs_pt.p, d_cam_mut.p, d_pt_mut.p, d_fflag.p);
    DTRY(excl_scan(d_fflag.p, d_fpos.p, (size_t)No + 1, s, tmp));
    DTRY(hipMemcpyAsync(&n_frozen_rows, d_fpos.p + No, sizeof(int), hipMemcpyDeviceToHost, s));
  }
  int AE = 0;
  DTRY(hipMemcpyAsync(&AE, ba->pt_first.p + npb, sizeof(int), hipMemcpyDeviceToHost, s));
  DTRY(hipStreamSynchronize(s));
  ba->AE = AE;
  const int A = ba->A = AE + n_frozen_rows;
  ba->n_residuals = 2 * A + (ba->has_gps ? 3 * ncb : 0);
  lap("point order");
  // ---- rows ----
  const size_t As = (size_t)std::max(1, A);
  DTRY(ba->o_cam.alloc(As)); DTRY(ba->o_model.alloc(As)); DTRY(ba->o_pt.alloc(As)); DTRY(ba->o_cb.alloc(As)); DTRY(ba->o_mb.alloc(As));
  DTRY(ba->o_pb.alloc(As)); DTRY(ba->o_cpos.alloc(As)); DTRY(ba->o_pm.alloc(As)); DTRY(ba->o_x.alloc(As)); DTRY(ba->o_y.alloc(As)); DTRY(ba->o_w.alloc(As));
  DevBuf<int> d_cam_hist, d_cam_first;
  DTRY(d_cam_hist.alloc((size_t)ncb + 1)); DTRY(d_cam_first.alloc((size_t)ncb + 1));
  DTRY(hipMemsetAsync(d_cam_hist.p, 0, sizeof(int) * ((size_t)ncb + 1), s));
  RowOut R{ba->o_cam.p, ba->o_model.p, ba->o_pt.p, ba->o_cb.p, ba->o_mb.p, ba->o_pb.p, ba->o_x.p, ba->o_y.p, ba->o_w.p};
  if (AE) hipLaunchKernelGGL(k_fill_rows, dim3(cdiv(AE, 256)), dim3(256), 0, s, AE, npb, ba->pt_first.p, ba->pb_pt.p, d_run_first.p, d_obs_cam.p, d_obs_xy.p,
                             d_ptw.p, d_model_of_cam.p, d_cam_mut.p, d_model_mut.p, d_cam_slot.p, d_model_slot.p, R, d_cam_hist.p, ba->ncb);
  if (n_frozen_rows) hipLaunchKernelGGL(k_fill_frozen, dim3(cdiv(No, 256)), dim3(256), 0, s, No, AE, d_fflag.p, d_fpos.p, d_obs_cam.p, d_obs_pt.p, d_obs_xy.p,
                                        d_ptw.p, d_model_of_cam.p, d_model_mut.p, d_cam_slot.p, d_model_slot.p, R, d_cam_hist.p);
  // ---- camera-major positions: stable sort of the rows by camera block ----
  DTRY(excl_scan(d_cam_hist.p, d_cam_first.p, (size_t)ncb + 1, s, tmp));
  std::vector<int> cam_first(ncb + 1, 0);
  DTRY(hipMemcpyAsync(cam_first.data(), d_cam_first.p, sizeof(int) * ((size_t)ncb + 1), hipMemcpyDeviceToHost, s));
  DTRY(hipStreamSynchronize(s));
  const int NCR = ba->NCR = cam_first[ncb];
  DTRY(ba->cpos_pb.alloc((size_t)std::max(1, NCR)));
  hipLaunchKernelGGL(k_fill_int, dim3(cdiv(As, 256)), dim3(256), 0, s, (int)As, -1, ba->o_cpos.p);
  hipLaunchKernelGGL(k_fill_int, dim3(cdiv(As, 256)), dim3(256), 0, s, (int)As, -1, ba->o_pm.p);
  hipLaunchKernelGGL(k_fill_int, dim3(cdiv(std::max(1, NCR), 256)), dim3(256), 0, s, std::max(1, NCR), -1, ba->cpos_pb.p);
  if (A && NCR) {
    DevBuf<int> rk, rv, rks, rvs;
    DTRY(rk.alloc(A)); DTRY(rv.alloc(A)); DTRY(rks.alloc(A)); DTRY(rvs.alloc(A));
    hipLaunchKernelGGL(k_row_keys, dim3(cdiv(A, 256)), dim3(256), 0, s, A, ncb, ba->o_cb.p, rk.p, rv.p);
    DTRY(sort_pairs(rk.p, rks.p, rv.p, rvs.p, (size_t)A, bits_for((long)ncb + 1), s, tmp));
    hipLaunchKernelGGL(k_assign_positions, dim3(cdiv(NCR, 256)), dim3(256), 0, s, NCR, rks.p, rvs.p, ba->o_pb.p, ba->o_cpos.p, ba->cpos_pb.p);
    DTRY(hipStreamSynchronize(s));   // the sort buffers go out of scope
  }
  lap("rows + camera positions");
  // ---- (point, intrinsics block) entries ----
  DevBuf<int> d_pm_count;
  DTRY(d_pm_count.alloc((size_t)npb + 1)); DTRY(ba->pm_first.alloc((size_t)npb + 1));
  DTRY(hipMemsetAsync(d_pm_count.p, 0, sizeof(int) * ((size_t)npb + 1), s));
  DTRY(hipMemsetAsync(d_err.p, 0x7f, sizeof(int) * 4, s));
  if (npb) hipLaunchKernelGGL(k_pm_entries, dim3(cdiv(npb, 256)), dim3(256), 0, s, npb, ba->pt_first.p, ba->o_mb.p, 0, d_pm_count.p, (const int*)nullptr,
                              (int*)nullptr, (int*)nullptr, d_err.p);
  DTRY(excl_scan(d_pm_count.p, ba->pm_first.p, (size_t)npb + 1, s, tmp));
  int NPM = 0;
  DTRY(hipMemcpyAsync(&NPM, ba->pm_first.p + npb, sizeof(int), hipMemcpyDeviceToHost, s));
  DTRY(hipMemcpyAsync(err, d_err.p, sizeof err, hipMemcpyDeviceToHost, s));
  DTRY(hipStreamSynchronize(s));
  if (err[0] < 0x7f7f7f7f) return msfm_set_error(ctx, MSFM_E_INVAL, "a point touches more than 64 intrinsics blocks");
  ba->NPM = NPM;
  DTRY(ba->pm_mb.alloc((size_t)std::max(1, NPM)));
  if (npb && NPM) hipLaunchKernelGGL(k_pm_entries, dim3(cdiv(npb, 256)), dim3(256), 0, s, npb, ba->pt_first.p, ba->o_mb.p, 1, d_pm_count.p, ba->pm_first.p,
                                     ba->pm_mb.p, ba->o_pm.p, d_err.p);
  // ---- FTF chunks (host: O(cameras + rows / 1024)) ----
  std::vector<int> f_start, f_end, cam_chunk_first(ncb + 1, 0);
  const int fchunk = ftf_chunk(cam_first[ncb]);
  for (int c = 0; c < ncb; c++) {
    cam_chunk_first[c] = (int)f_start.size();
    for (int e = cam_first[c]; e < cam_first[c + 1]; e += fchunk) { f_start.push_back(e); f_end.push_back(std::min(e + fchunk, cam_first[c + 1])); }
  }
  cam_chunk_first[ncb] = (int)f_start.size();
  ba->n_fchunks = (int)f_start.size();
  DTRY(ba->f_start.from(f_start.empty() ? std::vector<int>(1, 0) : f_start, s));
  DTRY(ba->f_end.from(f_end.empty() ? std::vector<int>(1, 0) : f_end, s));
  DTRY(ba->cam_chunk_first.from(cam_chunk_first, s));
  DTRY(hipStreamSynchronize(s));
  lap("entries + chunks");
  // ---- block-pair lists ----
  const bool want_blocks = ctx->world > 1;
  {
    PairBuild<0> b0(ctx, ba, ba->cc, 36, ba->pt_first.p, ba->pm_first.p, ba->pm_mb.p, tmp, want_blocks);
    PairBuild<1> b1(ctx, ba, ba->mc, 18, ba->pt_first.p, ba->pm_first.p, ba->pm_mb.p, tmp, false);
    PairBuild<2> b2(ctx, ba, ba->mm, 12, ba->pt_first.p, ba->pm_first.p, ba->pm_mb.p, tmp, false);
    MSFM_TRY(b0.phase1()); MSFM_TRY(b1.phase1()); MSFM_TRY(b2.phase1());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    MSFM_TRY(b0.phase2()); MSFM_TRY(b1.phase2()); MSFM_TRY(b2.phase2());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    MSFM_TRY(b0.phase3()); MSFM_TRY(b1.phase3()); MSFM_TRY(b2.phase3());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // the host copies of the counts and the temporaries go out of scope
  }
  lap("pairs cc");
  lap("pairs mc mm");
  return MSFM_OK;
}
#undef DTRY

// The same structures built by host threads (MSFM_CREATE_HOST=1): the first implementation, kept as the reference the
// device build is compared with bit for bit (tests/test_gpu_ba.py).
static int create_structures_host(msfm_ctx* ctx, const msfm_ba_problem* P, msfm_ba* ba, const std::function<void(const char*)>& lap) {
  hipStream_t s = ctx->stream;
  {
    const int nt = host_threads();
    std::vector<long> bad_range(nt, -1), bad_order(nt, -1);
    par_ranges((size_t)P->n_obs, nt, [&](int t, size_t o0, size_t o1) {
      for (size_t o = o0; o < o1; o++) {
        if (bad_range[t] < 0 && (P->obs_cam[o] < 0 || P->obs_cam[o] >= P->n_cams || P->obs_pt[o] < 0 || P->obs_pt[o] >= P->n_points)) bad_range[t] = (long)o;
        if (bad_order[t] < 0 && o > 0 && P->obs_pt[o] < P->obs_pt[o - 1]) bad_order[t] = (long)o;
      }
    });
    for (int t = 0; t < nt; t++) {   // the first offender in input order, range errors first (as the sequential check reported them)
      if (bad_range[t] >= 0 && (bad_order[t] < 0 || bad_range[t] <= bad_order[t]))
        return msfm_set_error(ctx, MSFM_E_INVAL, "observation %d: index out of range", (int)bad_range[t]);
      if (bad_order[t] >= 0) return msfm_set_error(ctx, MSFM_E_INVAL, "obs_pt must be non-decreasing (gather order, optimizer.cc:62)");
    }
  }
  const int Nc = ba->Nc = P->n_cams, Nm = ba->Nm = P->n_models, Np = ba->Np = P->n_points, No = P->n_obs;
  // ---- which parameter blocks exist (a block exists iff some residual uses it) ----
  BaScratch& H = ba_scratch(ctx);
  std::vector<int> cam_slot(Nc, -1), model_slot(Nm, -1), pt_slot(Np, -1);
  std::vector<int>& run_first = H.run_first;
  {
    std::vector<char> cu(Nc, 0), mu(Nm, 0), pu(Np, 0);
    par_ranges((size_t)No, host_threads(), [&](int, size_t o0, size_t o1) {   // all stores write 1: relaxed byte stores
      for (size_t o = o0; o < o1; o++) {
        const int c = P->obs_cam[o], p = P->obs_pt[o], m = P->cam_model_of_cam[c];
        const bool cm = is_mut(P->cam_mutable, c), pm = is_mut(P->pt_mutable, p);
        if (!cm && !pm) continue;
        if (pm && !pu[p]) __atomic_store_n(&pu[p], (char)1, __ATOMIC_RELAXED);
        if (cm) {
          if (!cu[c]) __atomic_store_n(&cu[c], (char)1, __ATOMIC_RELAXED);
          if (is_mut(P->model_mutable, m) && !mu[m]) __atomic_store_n(&mu[m], (char)1, __ATOMIC_RELAXED);
        }
      }
    });
    // input runs per point (obs_pt is non-decreasing)
    run_first.assign(Np + 1, 0);
    for (int o = 0; o < No; o++) run_first[P->obs_pt[o] + 1]++;
    for (int p = 0; p < Np; p++) run_first[p + 1] += run_first[p];
    if (P->gps_xyz) for (int c = 0; c < Nc; c++) if (is_mut(P->cam_mutable, c)) cu[c] = 1;
    if (ctx->world > 1) {
      // points are sharded over ranks: the block structure of the reduced system must be the same
      // everywhere, so every mutable camera / intrinsics block gets a slot whether or not this
      // rank's shard observes it
      for (int c = 0; c < Nc; c++) if (is_mut(P->cam_mutable, c)) { cu[c] = 1; if (is_mut(P->model_mutable, P->cam_model_of_cam[c])) mu[P->cam_model_of_cam[c]] = 1; }
    }
    for (int m = 0; m < Nm; m++) if (mu[m]) { model_slot[m] = ba->nmb++; ba->h_mb_model.push_back(m); }
    for (int p = 0; p < Np; p++) if (pu[p]) ba->h_pb_pt.push_back(p);  // slots are assigned below, once the camera order is known
    // ---- elimination order of the camera blocks (see partition_cameras) ----
    std::vector<int> gnode(Nc, -1), gcam;
    for (int c = 0; c < Nc; c++) if (cu[c]) { gnode[c] = (int)gcam.size(); gcam.push_back(c); }
    const int ng = (int)gcam.size();
    const char* env = getenv("MSFM_CHOL_DOMAINS");  // "0": dense order, "1".."3": force that bisection depth
    const int force = env ? atoi(env) : -1;
    std::vector<uint8_t> adjb8;
    if (ng >= 128 && ng <= 4096 && force != 0) {
      // adjacency as a 0/1 matrix: cameras sharing an eliminated point; max-reduced over the ranks' shards
      std::vector<char> adjb((size_t)ng * ng, 0);
      par_ranges((size_t)Np, host_threads(), [&](int, size_t p0, size_t p1) {
        std::vector<int> run;
        for (size_t p = p0; p < p1; p++) {
          if (!is_mut(P->pt_mutable, (int)p)) continue;
          run.clear();
          for (int e = run_first[p]; e < run_first[p + 1]; e++)
            if (gnode[P->obs_cam[e]] >= 0) run.push_back(gnode[P->obs_cam[e]]);
          for (int a : run) for (int b : run) if (a != b && !adjb[(size_t)a * ng + b]) __atomic_store_n(&adjb[(size_t)a * ng + b], (char)1, __ATOMIC_RELAXED);
        }
      });
      if (ctx->world > 1) {
        std::vector<double> adjm((size_t)ng * ng);
        for (size_t k = 0; k < adjm.size(); k++) adjm[k] = adjb[k] ? 1.0 : 0.0;
        DevBuf<double> dadj;
        HIP_TRY(ctx, dadj.from(adjm, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));
        const int rc = ctx->allreduce(ctx->allreduce_user, dadj.p, adjm.size(), MSFM_REDUCE_MAX, (void*)s);
        if (rc != 0) return msfm_set_error(ctx, MSFM_E_DEVICE, "all-reduce hook failed: %d", rc);
        HIP_TRY(ctx, hipMemcpyAsync(adjm.data(), dadj.p, sizeof(double) * adjm.size(), hipMemcpyDeviceToHost, s));
        MSFM_TRY(msfm_stream_wait_bounded(ctx, s, "the all-reduce of the camera graph"));
        for (size_t k = 0; k < adjm.size(); k++) adjb[k] = adjm[k] != 0.0;
      }
      adjb8.assign(adjb.begin(), adjb.end());
    }
    std::vector<int> cb_off_h, padcol_h;
    order_camera_blocks(ctx, ba, gcam, adjb8, force, cam_slot, cb_off_h, padcol_h);
    // ---- order of the eliminated points: by the (sorted) list of camera blocks that see them ----
    // Points seen by the same cameras become neighbours, so the records of a camera pair's common points are
    // runs in both cameras' segments (the pair kernel's gathers and k_point's scattered stores turn near-sequential).
    // Internal only: h_pb_pt maps block -> caller's point index.
    if (!getenv("MSFM_POINT_ORDER") || atoi(getenv("MSFM_POINT_ORDER")) != 0) {
      // key: the four smallest camera blocks, 16 bits each (21 bits x 3 when there are more than 65535 blocks)
      const bool wide = ba->ncb >= 0xFFFF;
      std::vector<std::pair<uint64_t, int>>& keyed = H.keyed;
      keyed.resize(ba->h_pb_pt.size());
      const int nk = wide ? 3 : 4, bits = wide ? 21 : 16;
      par_ranges(keyed.size(), host_threads(), [&](int, size_t i0, size_t i1) {
        std::vector<int> cams;
        for (size_t i = i0; i < i1; i++) {
          const int p = ba->h_pb_pt[i];   // still in ascending point order here
          cams.clear();
          for (int e = run_first[p]; e < run_first[p + 1]; e++)
            if (cam_slot[P->obs_cam[e]] >= 0) cams.push_back(cam_slot[P->obs_cam[e]]);
          std::sort(cams.begin(), cams.end());
          uint64_t k = 0;
          for (int q = 0; q < nk; q++) k = (k << bits) | (uint64_t)(q < (int)cams.size() ? cams[q] : ((1 << bits) - 1));
          keyed[i] = {k, p};
        }
      });
      {  // sort: pieces in parallel, then pairwise merges (ties fall back to the caller's point index)
        const int nt = (int)std::max<size_t>(1, std::min<size_t>(host_threads(), keyed.size() / 8192 + 1));
        std::vector<size_t> cut(nt + 1);
        for (int t = 0; t <= nt; t++) cut[t] = keyed.size() * t / nt;
        {
          std::vector<std::thread> th;
          for (int t = 1; t < nt; t++) th.emplace_back([&, t] { std::sort(keyed.begin() + cut[t], keyed.begin() + cut[t + 1]); });
          std::sort(keyed.begin() + cut[0], keyed.begin() + cut[1]);
          for (auto& x : th) x.join();
        }
        for (int w = 1; w < nt; w *= 2) {
          std::vector<std::thread> th;
          for (int t = 0; t + w < nt; t += 2 * w)
            th.emplace_back([&, t, w] { std::inplace_merge(keyed.begin() + cut[t], keyed.begin() + cut[t + w], keyed.begin() + cut[std::min(nt, t + 2 * w)]); });
          for (auto& x : th) x.join();
        }
      }
      if (keyed.size() == ba->h_pb_pt.size())
        for (size_t i = 0; i < keyed.size(); i++) ba->h_pb_pt[i] = keyed[i].second;
    }
    for (size_t i = 0; i < ba->h_pb_pt.size(); i++) pt_slot[ba->h_pb_pt[i]] = ba->npb++;
    HIP_TRY(ctx, ba->cb_off.from(cb_off_h.empty() ? std::vector<int>(1, 0) : cb_off_h, s));
    HIP_TRY(ctx, ba->padcol.from(padcol_h.empty() ? std::vector<int>(1, 0) : padcol_h, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
  }
  const int ncb = ba->ncb, nmb = ba->nmb, npb = ba->npb;
  ba->nred = 6 * ncb + 3 * nmb;
  ba->npad = 64 * cdiv(ba->nsys + 1, 64);
  if (ba->plan.n_levels > 0) {
    ba->plan.ldc = 64 * cdiv(ba->nsys + 1 - ba->plan.level[0].b0, 64);
    HIP_TRY(ctx, ba->corners.alloc((size_t)8 * ba->plan.ldc * ba->plan.ldc));   // up to eight K-splits of the corner update (MSFM_CORNER_SPLIT_MAX)
    ba->plan.corners = ba->corners.p;
  }
  ba->has_gps = P->gps_xyz != nullptr;
  ba->gps_weight = P->gps_weight;
  lap("blocks + orders");
  // ---- active observations: eliminated-point rows first (point-major), then the rest ----
  // Every observation of an eliminated point is active (its point block is free), so row i of point block pb is
  // input observation run_first[p] + (i - pt_first[pb]): the arrays are filled in place by a few threads.
  std::vector<int>&o_cam = H.o_cam, &o_model = H.o_model, &o_pt = H.o_pt, &o_cb = H.o_cb, &o_mb = H.o_mb, &o_pb = H.o_pb,
                   &o_cpos = H.o_cpos, &o_pm = H.o_pm, &pt_first = H.pt_first;
  std::vector<double>&o_x = H.o_x, &o_y = H.o_y, &o_w = H.o_w;
  pt_first.assign(npb + 1, 0);
  for (int pb = 0; pb < npb; pb++) pt_first[pb + 1] = pt_first[pb] + (run_first[ba->h_pb_pt[pb] + 1] - run_first[ba->h_pb_pt[pb]]);
  ba->AE = pt_first[npb];
  {
    const size_t AE = (size_t)ba->AE;
    for (auto* v : {&o_cam, &o_model, &o_pt, &o_cb, &o_mb, &o_pb}) v->resize(AE);
    for (auto* v : {&o_x, &o_y, &o_w}) v->resize(AE);
    par_ranges((size_t)npb, host_threads(), [&](int, size_t b0, size_t b1) {
      for (size_t pb = b0; pb < b1; pb++) {
        const int p = ba->h_pb_pt[pb];
        int i = pt_first[pb];
        const double w = P->pt_weight ? P->pt_weight[p] : 1.0;
        for (int o = run_first[p]; o < run_first[p + 1]; o++, i++) {
          const int c = P->obs_cam[o], m = P->cam_model_of_cam[c];
          const bool cm = is_mut(P->cam_mutable, c);
          o_cam[i] = c; o_model[i] = m; o_pt[i] = p;
          o_cb[i] = cm ? cam_slot[c] : -1;
          o_mb[i] = (cm && is_mut(P->model_mutable, m)) ? model_slot[m] : -1;
          o_pb[i] = (int)pb;
          o_x[i] = P->obs_xy[2 * (size_t)o]; o_y[i] = P->obs_xy[2 * (size_t)o + 1];
          o_w[i] = w;
        }
      }
    });
  }
  // pass 1: observations of frozen points by free cameras, in input order (only with a point mask)
  if (P->pt_mutable)
    for (int o = 0; o < No; o++) {
      const int c = P->obs_cam[o], p = P->obs_pt[o], m = P->cam_model_of_cam[c];
      if (is_mut(P->pt_mutable, p) || !is_mut(P->cam_mutable, c)) continue;
      o_cam.push_back(c); o_model.push_back(m); o_pt.push_back(p);
      o_cb.push_back(cam_slot[c]);
      o_mb.push_back(is_mut(P->model_mutable, m) ? model_slot[m] : -1);
      o_pb.push_back(-1);
      o_x.push_back(P->obs_xy[2 * (size_t)o]); o_y.push_back(P->obs_xy[2 * (size_t)o + 1]);
      o_w.push_back(P->pt_weight ? P->pt_weight[p] : 1.0);
    }
  const int A = ba->A = (int)o_cam.size();
  ba->n_residuals = 2 * A + (ba->has_gps ? 3 * ncb : 0);
  lap("active observations");
  // ---- camera-major positions (a counting sort by camera block that keeps the row order inside a block) ----
  std::vector<int> cam_first(ncb + 1, 0);
  o_cpos.assign(A, -1);
  int NCR = 0;
  std::vector<int>&cpos_pb = H.cpos_pb, &cpos_cb = H.cpos_cb;
  {
    const int nt = host_threads();
    std::vector<std::vector<int>> hist(nt, std::vector<int>(ncb + 1, 0));
    std::vector<size_t> lo(nt, 0), hi(nt, 0);
    par_ranges((size_t)A, nt, [&](int t, size_t i0, size_t i1) {
      lo[t] = i0; hi[t] = i1;
      int* h = hist[t].data();
      for (size_t i = i0; i < i1; i++) if (o_cb[i] >= 0) h[o_cb[i]]++;
    });
    // hist[t][c] -> first position of thread t's rows of camera c
    int run = 0;
    for (int c = 0; c < ncb; c++) {
      cam_first[c] = run;
      for (int t = 0; t < nt; t++) { const int n = hist[t][c]; hist[t][c] = run; run += n; }
    }
    cam_first[ncb] = NCR = run;
    cpos_pb.assign(std::max(1, NCR), -1);
    cpos_cb.assign(std::max(1, NCR), -1);
    par_ranges((size_t)A, nt, [&](int t, size_t, size_t) {
      int* h = hist[t].data();
      for (size_t i = lo[t]; i < hi[t]; i++)
        if (o_cb[i] >= 0) {
          const int pos = h[o_cb[i]]++;
          o_cpos[i] = pos;
          cpos_pb[pos] = o_pb[i];
          cpos_cb[pos] = o_cb[i];
        }
    });
  }
  ba->NCR = NCR;
  lap("  camera positions");
  // ---- (point, intrinsics block) entries ----
  std::vector<int> pm_first(npb + 1, 0), pm_mb;
  o_pm.assign(A, -1);
  {
    // the sorted distinct intrinsics blocks of a point's rows: counted, prefix-summed, then written (two parallel passes)
    auto distinct = [&](int pb, int* tmp) -> int {
      int nt = 0;
      for (int i = pt_first[pb]; i < pt_first[pb + 1]; i++) {
        const int mb = o_mb[i];
        if (mb < 0) continue;
        bool seen = false;
        for (int k = 0; k < nt; k++) seen |= tmp[k] == mb;
        if (!seen) {
          if (nt == 64) return -1;
          tmp[nt++] = mb;
        }
      }
      std::sort(tmp, tmp + nt);
      return nt;
    };
    const int nth = host_threads();
    std::vector<int> too_many(nth, -1);
    par_ranges((size_t)npb, nth, [&](int t, size_t b0, size_t b1) {
      int tmp[64];
      for (size_t pb = b0; pb < b1; pb++) {
        const int nt = distinct((int)pb, tmp);
        if (nt < 0) { if (too_many[t] < 0) too_many[t] = (int)pb; pm_first[pb + 1] = 0; }
        else pm_first[pb + 1] = nt;
      }
    });
    for (int t = 0; t < nth; t++)
      if (too_many[t] >= 0) return msfm_set_error(ctx, MSFM_E_INVAL, "point %d touches more than 64 intrinsics blocks", ba->h_pb_pt[too_many[t]]);
    for (int pb = 0; pb < npb; pb++) pm_first[pb + 1] += pm_first[pb];
    pm_mb.resize(pm_first[npb]);
    par_ranges((size_t)npb, nth, [&](int, size_t b0, size_t b1) {
      int tmp[64];
      for (size_t pb = b0; pb < b1; pb++) {
        const int nt = distinct((int)pb, tmp);
        for (int k = 0; k < nt; k++) pm_mb[pm_first[pb] + k] = tmp[k];
        for (int i = pt_first[pb]; i < pt_first[pb + 1]; i++)
          if (o_mb[i] >= 0)
            for (int k = 0; k < nt; k++) if (tmp[k] == o_mb[i]) o_pm[i] = pm_first[pb] + k;
      }
    });
  }
  const int NPM = ba->NPM = (int)pm_mb.size();
  lap("positions + pm entries");
  // ---- FTF chunks (camera-major rows) ----
  std::vector<int> f_start, f_end, cam_chunk_first(ncb + 1, 0);
  const int fchunk = ftf_chunk(cam_first[ncb]);
  for (int c = 0; c < ncb; c++) {
    cam_chunk_first[c] = (int)f_start.size();
    for (int e = cam_first[c]; e < cam_first[c + 1]; e += fchunk) { f_start.push_back(e); f_end.push_back(std::min(e + fchunk, cam_first[c + 1])); }
  }
  cam_chunk_first[ncb] = (int)f_start.size();
  ba->n_fchunks = (int)f_start.size();
  // cameras of each intrinsics block
  std::vector<int> cb_mb(std::max(1, ncb), -1), mcam_first(nmb + 1, 0), mcam;
  for (int cb = 0; cb < ncb; cb++) {
    const int m = P->cam_model_of_cam[ba->h_cb_cam[cb]];
    cb_mb[cb] = is_mut(P->model_mutable, m) ? model_slot[m] : -1;
  }
  for (int mb = 0; mb < nmb; mb++) {
    mcam_first[mb] = (int)mcam.size();
    for (int cb = 0; cb < ncb; cb++) if (cb_mb[cb] == mb) mcam.push_back(cb);
  }
  mcam_first[nmb] = (int)mcam.size();
  // ---- pair lists, counting-sorted by (block row, block col) ----
  auto build_pairs = [&](int kind, PairJobs& J, int nout) -> int {
    // kind 0: cam-cam (rows cb_a >= cols cb_b), 1: intr-cam, 2: intr-intr (mb_a >= mb_b)
    const long nrow = kind == 0 ? ncb : nmb, ncol = kind == 2 ? nmb : ncb;
    const size_t nkey = (size_t)(nrow * ncol);
    // counting sort by block key with one histogram per thread (each thread owns a contiguous range of points, so the
    // entries of a block stay in point order); the histograms are capped at 256 MB in total
    int nt = host_threads();
    while (nt > 1 && nkey * sizeof(int) * nt > (size_t)256 << 20) nt--;
    auto visit = [&](size_t pb0, size_t pb1, auto&& emit) {
      for (size_t pb = pb0; pb < pb1; pb++) {
        const int f = pt_first[pb], l = pt_first[pb + 1];
        if (kind == 0) {
          for (int i = f; i < l; i++) {
            if (o_cpos[i] < 0) continue;
            for (int j = f; j < l; j++) {
              if (o_cpos[j] < 0) continue;
              if (o_cb[i] > o_cb[j] || o_cb[i] == o_cb[j]) emit(o_cb[i], o_cb[j], o_cpos[i], o_cpos[j]);
            }
          }
        } else if (kind == 1) {
          for (int e = pm_first[pb]; e < pm_first[pb + 1]; e++)
            for (int j = f; j < l; j++) if (o_cpos[j] >= 0) emit(pm_mb[e], o_cb[j], e, o_cpos[j]);
        } else {
          for (int e = pm_first[pb]; e < pm_first[pb + 1]; e++)
            for (int g = pm_first[pb]; g <= e; g++) emit(pm_mb[e], pm_mb[g], e, g);
        }
      }
    };
    std::vector<std::vector<int>> hist(nt);
    std::vector<size_t> lo(nt, 0), hi(nt, 0);
    par_ranges((size_t)npb, nt, [&](int t, size_t b0, size_t b1) {
      lo[t] = b0; hi[t] = b1;
      hist[t].assign(nkey, 0);
      int* h = hist[t].data();
      visit(b0, b1, [&](int r, int c, int, int) { h[(size_t)r * ncol + c]++; });
    });
    if (kind == 0) lap("  pairs: counted");
    std::vector<long> first(nkey + 1, 0);
    // write cursors per thread, in place of its histogram (32-bit: the total is checked against 2^31 before they are used)
    {
      long run = 0;
      for (size_t k = 0; k < nkey; k++) {
        first[k] = run;
        for (int t = 0; t < nt; t++)
          if (!hist[t].empty()) { const int n = hist[t][k]; hist[t][k] = (int)run; run += n; }
      }
      first[nkey] = run;
    }
    const long total = first[nkey];
    if (total > 0x7fffffffL) return msfm_set_error(ctx, MSFM_E_NOMEM, "pair list too long");
    if (kind == 0) lap("  pairs: offsets");
    std::vector<int>&pa = H.pa, &pbv = H.pb;
    pa.resize((size_t)total); pbv.resize((size_t)total);
    par_ranges((size_t)npb, nt, [&](int t, size_t, size_t) {
      if (hist[t].empty()) return;
      int* fl = hist[t].data();
      visit(lo[t], hi[t], [&](int r, int c, int a, int b) { const int q = fl[(size_t)r * ncol + c]++; pa[q] = a; pbv[q] = b; });
    });
    if (kind == 0) lap("  pairs: filled");
    std::vector<long> key_first;
    std::vector<int> brow, bcol;
    for (long r = 0; r < nrow; r++)
      for (long c = 0; c < ncol; c++) {
        const size_t k = (size_t)r * ncol + c;
        const bool force = (kind == 0 || kind == 2) ? (r == c) : (cb_mb[c] == r);  // FTF-only blocks still need assembling
        if (first[k + 1] > first[k] || force) { key_first.push_back(first[k]); brow.push_back((int)r); bcol.push_back((int)c); }
      }
    key_first.push_back(total);
    if (kind == 0) lap("  pairs: blocks");
    // key_first of forced empty blocks must still be monotone: it is (first[k] == first[k+1]).
    return finish_jobs(ba, J, pa, pbv, key_first, brow, bcol, nout);
  };
  lap("before upload");
  // ---- upload ----
  std::vector<double> gps_cb;
  if (ba->has_gps) {
    gps_cb.resize(3 * (size_t)ncb);
    for (int cb = 0; cb < ncb; cb++) for (int k = 0; k < 3; k++) gps_cb[3 * (size_t)cb + k] = P->gps_xyz[3 * (size_t)ba->h_cb_cam[cb] + k];
  }
#define UP(buf, vec) HIP_TRY(ctx, ba->buf.from(vec, s))
  UP(o_cam, o_cam); UP(o_model, o_model); UP(o_pt, o_pt); UP(o_cb, o_cb); UP(o_mb, o_mb); UP(o_pb, o_pb);
  UP(o_cpos, o_cpos); UP(o_pm, o_pm); UP(o_x, o_x); UP(o_y, o_y); UP(o_w, o_w);
  UP(cb_cam, ba->h_cb_cam); UP(mb_model, ba->h_mb_model); UP(pb_pt, ba->h_pb_pt); UP(cb_mb, cb_mb);
  UP(cpos_pb, cpos_pb); UP(pt_first, pt_first); UP(pm_first, pm_first);
  if (NPM) UP(pm_mb, pm_mb);
  UP(f_start, f_start); UP(f_end, f_end); UP(cam_chunk_first, cam_chunk_first);
  UP(mcam_first, mcam_first);
  if (!mcam.empty()) UP(mcam, mcam);
  if (ba->has_gps) UP(gps, gps_cb);
#undef UP
  HIP_TRY(ctx, hipStreamSynchronize(s));
  lap("uploads");
  MSFM_TRY(build_pairs(0, ba->cc, 36));
  lap("pairs cc");
  MSFM_TRY(build_pairs(1, ba->mc, 18));
  MSFM_TRY(build_pairs(2, ba->mm, 12));
  lap("pairs mc mm");
  return MSFM_OK;
}

MSFM_API int msfm_ba_create(msfm_ctx* ctx, const msfm_ba_problem* P, msfm_ba** out) { return ba_create_impl(ctx, P, false, out); }

int ba_create_impl(msfm_ctx* ctx, const msfm_ba_problem* P, bool bulk_on_device, msfm_ba** out) {
  if (!ctx || !P || !out) return MSFM_E_INVAL;
  *out = nullptr;
  struct ExitLap {   // declared first, destroyed last: the time to the very end of the call, host vectors released
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    bool on = getenv("MSFM_VERBOSE") != nullptr;
    ~ExitLap() { if (on) fprintf(stderr, "msfm: create returned after          %7.2f ms (from entry)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count()); }
  } exit_lap;
  if (P->n_cams <= 0 || P->n_models <= 0 || P->n_points < 0 || P->n_obs < 0 || !P->cam_pose || !P->cam_model ||
      !P->cam_model_of_cam || (P->n_points > 0 && !P->point) ||
      (P->n_obs > 0 && (!P->obs_cam || !P->obs_pt || !P->obs_xy)))
    return msfm_set_error(ctx, MSFM_E_INVAL, "msfm_ba_create: null or empty problem arrays");
  for (int c = 0; c < P->n_cams; c++)
    if (P->cam_model_of_cam[c] < 0 || P->cam_model_of_cam[c] >= P->n_models)
      return msfm_set_error(ctx, MSFM_E_INVAL, "cam_model_of_cam[%d] out of range", c);
  const auto t0 = std::chrono::steady_clock::now();
  const bool verbose = getenv("MSFM_VERBOSE") != nullptr;
  if (verbose) fprintf(stderr, "msfm: create input checked after     %7.2f ms (from entry)\n", std::chrono::duration<double, std::milli>(t0 - exit_lap.t).count());
  auto lap = [&](const char* what) {
    if (verbose && ctx->rank == 0)
      fprintf(stderr, "msfm: create %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  };
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  msfm_ba* ba = new msfm_ba();
  struct Guard { msfm_ba* p; ~Guard() { if (p) msfm_ba_destroy(p); } } guard{ba};
  ba->ctx = ctx;
  ctx->children++;
  ba->world_at_create = ctx->world;
  {
    const char* e = getenv("MSFM_CREATE_HOST");
    const bool on_host = e && atoi(e) != 0;
    if (on_host && bulk_on_device) return msfm_set_error(ctx, MSFM_E_INVAL, "MSFM_CREATE_HOST=1 needs the problem arrays in host memory");
    MSFM_TRY(on_host ? create_structures_host(ctx, P, ba, lap) : create_structures_device(ctx, P, ba, lap, bulk_on_device));
    MSFM_TRY(build_fold_device(ctx, ba));   // from the resident structures: the same tables whichever way those were built
    lap("fold tables");
  }
  const int Nc = ba->Nc, Nm = ba->Nm, Np = ba->Np, ncb = ba->ncb, nmb = ba->nmb, npb = ba->npb, A = ba->A, NCR = ba->NCR, NPM = ba->NPM;
  if (ctx->world > 1 && ncb > 0 && ncb <= 4096) {
    // union over ranks of the camera-camera block structure: a 0/1 matrix, max-reduced once
    std::vector<double> ind((size_t)ncb * ncb, 0.0);
    for (int b = 0; b < ba->cc.n_blocks; b++) ind[(size_t)ba->cc.h_row[b] * ncb + ba->cc.h_col[b]] = 1.0;
    DevBuf<double> dind;
    HIP_TRY(ctx, dind.from(ind, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    {
      const int rc = ctx->allreduce(ctx->allreduce_user, dind.p, ind.size(), MSFM_REDUCE_MAX, (void*)s);
      if (rc != 0) return msfm_set_error(ctx, MSFM_E_DEVICE, "all-reduce hook failed: %d", rc);
    }
    HIP_TRY(ctx, hipMemcpyAsync(ind.data(), dind.p, sizeof(double) * ind.size(), hipMemcpyDeviceToHost, s));
    MSFM_TRY(msfm_stream_wait_bounded(ctx, s, "the all-reduce of the block structure"));
    std::vector<int> ur, uc;
    for (int r = 0; r < ncb; r++)
      for (int c = 0; c <= r; c++)
        if (ind[(size_t)r * ncb + c] != 0.0) { ur.push_back(r); uc.push_back(c); }
    ba->n_ublk = (int)ur.size();
    HIP_TRY(ctx, ba->u_row.from(ur, s));
    HIP_TRY(ctx, ba->u_col.from(uc, s));
    HIP_TRY(ctx, ba->pack.alloc((size_t)std::max(1, ba->n_ublk) * 36 + (size_t)(ba->nsys - ba->mo + 1) * ba->npad));
    HIP_TRY(ctx, hipStreamSynchronize(s));
  }
  const size_t As = std::max(1, A);
#define AL(buf, n) HIP_TRY(ctx, ba->buf.alloc((size_t)std::max<size_t>(1, (n))))
  AL(cam, 6 * (size_t)Nc); AL(model, 3 * (size_t)Nm); AL(pt, 3 * (size_t)std::max(1, Np));
  AL(cam_c, 6 * (size_t)Nc); AL(model_c, 3 * (size_t)Nm); AL(pt_c, 3 * (size_t)std::max(1, Np));
  AL(rot, 4 * (size_t)Nc); AL(rot_c, 4 * (size_t)Nc);
  const size_t Atail = (size_t)std::max(1, A - ba->AE);   // rows of frozen points: the only ones whose linearisation is stored
  AL(lin_r, 2 * Atail); AL(lin_Jc, 12 * Atail); AL(lin_Jm, 6 * Atail);
  AL(T, 18 * (size_t)NCR); AL(Tu, 6 * (size_t)NCR);
  AL(cm_pt, (size_t)NCR); AL(cm_xyw, 3 * (size_t)NCR); AL(cm_X, 3 * (size_t)NCR); AL(cm_Xc, 3 * (size_t)NCR); AL(chunk_cam, (size_t)ba->n_fchunks);
  AL(Tm, 9 * (size_t)NPM); AL(Tmu, 3 * (size_t)NPM);
  AL(scale_c, 6 * (size_t)ncb); AL(scale_m, 3 * (size_t)nmb); AL(scale_p, 3 * (size_t)npb);
  AL(diag_c, 6 * (size_t)ncb); AL(diag_m, 3 * (size_t)nmb); AL(diag_p, 3 * (size_t)npb);
  AL(ptL, 6 * (size_t)npb); AL(ptg, 3 * (size_t)npb);
  AL(f_partial, (size_t)ba->n_fchunks * PSTRIDE); AL(camftf, (size_t)ncb * PSTRIDE); AL(modelsum, 12 * (size_t)nmb);
  AL(M, (size_t)ba->npad * ba->npad); AL(Linv, (size_t)ba->npad * 144); /* 16x16 inverses + full 64x64 block inverses + diagonal blocks of L */ AL(w, ba->npad); AL(z, ba->npad + 8); AL(zsys, 2 * ((size_t)ba->npad + 8));   /* two solution buffers that alternate from solve to solve (k_backsolve_chain) */
  AL(g_r, 3 * (size_t)ncb); AL(g_J, 3 * (size_t)ncb);
  ba->nblk_obs = cdiv(As, 256);
  ba->nblk_pt = cdiv(std::max(1, npb), 32);  // 8 lanes per point
  const size_t npart = (size_t)ba->nblk_obs + ba->nblk_pt + cdiv(std::max(1, ncb), 256) + 64;
  AL(partial, npart); AL(partial2, npart); AL(partial3, npart); AL(partial4, npart);
  AL(gmax_buf, (size_t)ba->nblk_pt + 6 * (size_t)ncb + 3 * (size_t)nmb + 8);
  AL(scal, S_N); AL(sloc, S_N); AL(spec, 8);
  { const char* e = getenv("MSFM_SPEC"); ba->spec_on = !(e && atoi(e) == 0); }
  HIP_TRY(ctx, ba->fail.alloc(4));
#undef AL
  // the first solve's solution buffer starts out "pending"; from then on every solve marks the other one (k_backsolve_chain)
  MSFM_TRY(msfm_chol_fill_pending(ctx, ba->zsys.p, ba->npad));
  if (ba->nred > 0) MSFM_TRY(msfm_chol_ws_create(ctx, ba->npad, &ba->chol_ws));
  {
    // every device buffer a kernel may dereference must exist before the first launch
    const void* must[] = {ba->cam.p, ba->model.p, ba->pt.p, ba->cam_c.p, ba->model_c.p, ba->pt_c.p, ba->lin_r.p, ba->lin_Jc.p,
                          ba->lin_Jm.p, ba->cm_pt.p, ba->cm_xyw.p, ba->chunk_cam.p, ba->T.p, ba->Tu.p, ba->Tm.p, ba->Tmu.p, ba->scale_c.p,
                          ba->scale_m.p, ba->scale_p.p, ba->diag_c.p, ba->diag_m.p, ba->diag_p.p, ba->ptL.p, ba->ptg.p,
                          ba->f_partial.p, ba->camftf.p, ba->modelsum.p, ba->M.p, ba->Linv.p, ba->w.p, ba->z.p, ba->g_r.p,
                          ba->g_J.p, ba->partial.p, ba->partial2.p, ba->partial3.p, ba->gmax_buf.p, ba->scal.p, ba->fail.p};
    for (const void* q : must)
      if (!q) return msfm_set_error(ctx, MSFM_E_NOMEM, "msfm_ba_create: a device buffer was not allocated");
  }
  lap("allocations");
  ba->swrite = ctx->world > 1 ? ba->sloc.p : ba->scal.p;
  HIP_TRY(ctx, hipMemsetAsync(ba->scal.p, 0, sizeof(double) * S_N, s));
  HIP_TRY(ctx, hipMemsetAsync(ba->sloc.p, 0, sizeof(double) * S_N, s));
  HIP_TRY(ctx, hipHostMalloc((void**)&ba->h_scal, 24 * sizeof(double), hipHostMallocMapped));
  memset(ba->h_scal, 0, 24 * sizeof(double));
  HIP_TRY(ctx, hipHostGetDevicePointer((void**)&ba->h_scal_dev, ba->h_scal, 0));
  HIP_TRY(ctx, hipHostMalloc((void**)&ba->h_fail, 4 * sizeof(int)));
  HIP_TRY(ctx, hipMemsetAsync(ba->z.p, 0, sizeof(double) * (size_t)(ba->npad + 8), s));  // tail entries are read (times zero) by frozen blocks
  {
    // camera-major statics of the rows and the camera of every k_ftf chunk
    DevBuf<int> cm_row;
    HIP_TRY(ctx, cm_row.alloc((size_t)std::max(1, NCR)));
    HIP_TRY(ctx, hipMemsetAsync(ba->cm_pt.p, 0, sizeof(int) * (size_t)std::max(1, NCR), s));
    HIP_TRY(ctx, hipMemsetAsync(ba->cm_xyw.p, 0, sizeof(double) * 3 * (size_t)std::max(1, NCR), s));
    HIP_TRY(ctx, hipMemsetAsync(cm_row.p, 0, sizeof(int) * (size_t)std::max(1, NCR), s));
    const size_t ncr = (size_t)std::max(1, NCR);
    if (A > 0)
      hipLaunchKernelGGL(k_cam_rows, dim3(cdiv(A, 256)), dim3(256), 0, s, A, ba->o_cpos.p, ba->o_pt.p, ba->o_x.p, ba->o_y.p, ba->o_w.p, ba->cm_pt.p,
                         ba->cm_xyw.p, ba->cm_xyw.p + ncr, ba->cm_xyw.p + 2 * ncr, cm_row.p);
    if (ba->n_fchunks)
      hipLaunchKernelGGL(k_chunk_cam, dim3(cdiv(ba->n_fchunks, 256)), dim3(256), 0, s, ba->n_fchunks, ba->f_start.p, cm_row.p, ba->o_cam.p, ba->o_model.p,
                         ba->o_cb.p, ba->o_mb.p, ba->chunk_cam.p);
    HIP_TRY(ctx, hipStreamSynchronize(s));   // (cm_row goes back to the pool)
  }
  HIP_TRY(ctx, hipMemsetAsync(ba->T.p, 0, sizeof(double) * std::max<size_t>(1, 18 * (size_t)NCR), s));
  HIP_TRY(ctx, hipMemsetAsync(ba->Tu.p, 0, sizeof(double) * std::max<size_t>(1, 6 * (size_t)NCR), s));
  HIP_TRY(ctx, hipMemcpyAsync(ba->cam.p, P->cam_pose, sizeof(double) * 6 * (size_t)Nc, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipMemcpyAsync(ba->model.p, P->cam_model, sizeof(double) * 3 * (size_t)Nm, hipMemcpyHostToDevice, s));
  if (Np) HIP_TRY(ctx, hipMemcpyAsync(ba->pt.p, P->point, sizeof(double) * 3 * (size_t)Np, bulk_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  lap("done");
  ba->setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  guard.p = nullptr;
  *out = ba;
  return MSFM_OK;
}

MSFM_API int msfm_ba_get_layout(const msfm_ba* ba, msfm_ba_layout* out) {
  if (!ba || !out) return MSFM_E_INVAL;
  memset(out, 0, sizeof *out);
  out->reduced_order = ba->nred;
  out->system_order = ba->nsys;
  const msfm_chol_plan& pl = ba->plan;
  const int K = pl.n_levels > 0 ? pl.level[0].K : 0;
  out->n_domains = K ? K : 1;
  int launches = 0;
  for (int k = 0; k < K; k++) out->domain_cols[k] = pl.level[0].node[k].end - pl.level[0].node[k].begin;
  for (int l = 0; l < pl.n_levels; l++) {
    int maxp = 0;
    for (int k = 0; k < pl.level[l].K; k++) maxp = std::max(maxp, (pl.level[l].node[k].end - pl.level[l].node[k].begin) / 64);
    launches += maxp + 1;   // the chains + the deferred corner update
    out->level_nodes[l] = pl.level[l].K;
    out->level_begin[l] = pl.level[l].begin;
  }
  out->n_levels = pl.n_levels;
  const int root_begin = pl.n_levels > 0 ? pl.level[pl.n_levels - 1].b0 : 0;
  out->root_cols = ba->nsys - root_begin;
  out->separator_cols = K ? ba->nsys - pl.level[0].b0 : ba->nsys;
  out->panel_launches = launches + cdiv(out->root_cols, 64);
  if (!K) out->domain_cols[0] = 0;
  out->cc_entries = ba->cc.n_pairs;
  out->mc_entries = ba->mc.n_pairs;
  if (ba->fold.on) {
    out->cc_entries_folded = ba->fold.n_entries;
    out->fold_slots = ba->fold.n_slots;
    out->fold_passes = ba->fold.n_pass;
    if (ba->fold.mc_on) { out->mc_entries_folded = ba->fold.mc_entries_folded; out->fold_mc_slots = ba->fold.n_diag; }
  }
  return MSFM_OK;
}

MSFM_API int msfm_ba_upload_params(msfm_ba* ba, const double* cam_pose, const double* cam_model, const double* point) {
  if (!ba) return MSFM_E_INVAL;
  msfm_ctx* ctx = ba->ctx;
  hipStream_t s = ctx->stream;
  if (cam_pose) HIP_TRY(ctx, hipMemcpyAsync(ba->cam.p, cam_pose, sizeof(double) * 6 * (size_t)ba->Nc, hipMemcpyHostToDevice, s));
  if (cam_model) HIP_TRY(ctx, hipMemcpyAsync(ba->model.p, cam_model, sizeof(double) * 3 * (size_t)ba->Nm, hipMemcpyHostToDevice, s));
  if (point && ba->Np) HIP_TRY(ctx, hipMemcpyAsync(ba->pt.p, point, sizeof(double) * 3 * (size_t)ba->Np, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return MSFM_OK;
}

MSFM_API int msfm_ba_download_params(msfm_ba* ba, double* cam_pose, double* cam_model, double* point) {
  if (!ba) return MSFM_E_INVAL;
  msfm_ctx* ctx = ba->ctx;
  hipStream_t s = ctx->stream;
  if (cam_pose) HIP_TRY(ctx, hipMemcpyAsync(cam_pose, ba->cam.p, sizeof(double) * 6 * (size_t)ba->Nc, hipMemcpyDeviceToHost, s));
  if (cam_model) HIP_TRY(ctx, hipMemcpyAsync(cam_model, ba->model.p, sizeof(double) * 3 * (size_t)ba->Nm, hipMemcpyDeviceToHost, s));
  if (point && ba->Np) HIP_TRY(ctx, hipMemcpyAsync(point, ba->pt.p, sizeof(double) * 3 * (size_t)ba->Np, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return MSFM_OK;
}

// ---- phases ----------------------------------------------------------------------------
static BaPtrs make_ptrs(msfm_ba* ba, bool candidate, double huber) {
  BaPtrs P;
  P.A = ba->A; P.AE = ba->AE; P.ncb = ba->ncb; P.nmb = ba->nmb; P.npb = ba->npb; P.NCR = ba->NCR;
  P.cam = candidate ? ba->cam_c.p : ba->cam.p;
  P.rot = candidate ? ba->rot_c.p : ba->rot.p;
  P.model = candidate ? ba->model_c.p : ba->model.p;
  P.pt = candidate ? ba->pt_c.p : ba->pt.p;
  P.o_cam = ba->o_cam.p; P.o_model = ba->o_model.p; P.o_pt = ba->o_pt.p; P.o_cb = ba->o_cb.p; P.o_mb = ba->o_mb.p;
  P.o_pb = ba->o_pb.p; P.o_cpos = ba->o_cpos.p; P.o_pm = ba->o_pm.p;
  P.o_x = ba->o_x.p; P.o_y = ba->o_y.p; P.o_w = ba->o_w.p;
  P.lin_r = ba->lin_r.p; P.lin_Jc = ba->lin_Jc.p; P.lin_Jm = ba->lin_Jm.p;
  P.scale_c = ba->scale_c.p; P.scale_m = ba->scale_m.p; P.scale_p = ba->scale_p.p;
  P.huber = huber;
  return P;
}

static int allreduce(msfm_ba* ba, double* buf, size_t count, int op) {
  msfm_ctx* ctx = ba->ctx;
  if (ctx->world <= 1) return MSFM_OK;
  KTimer t(ctx, count > 4096 ? "ba_allreduce_system" : "ba_allreduce_small");
  const int rc = ctx->allreduce(ctx->allreduce_user, buf, count, op, (void*)ctx->stream);
  if (rc != 0) return msfm_set_error(ctx, MSFM_E_DEVICE, "all-reduce hook failed: %d", rc);
  return MSFM_OK;
}

// cost (and, with jac, the stored linearisation) at x or at the candidate -> scal[slot]
// (local partial; summed over ranks by the caller together with the other scalars)
static int run_evaluate(msfm_ba* ba, bool candidate, bool jac, double huber, int slot, bool with_fail = false, const double* spec = nullptr) {
  msfm_ctx* ctx = ba->ctx;
  hipStream_t s = ctx->stream;
  const bool lead = ctx->rank == 0;
  // every rank needs the GPS rows' Jacobian (k_cam_post uses g_J via camftf only on the lead), but
  // only the lead rank counts their cost
  const int nb = ba->nblk_obs, ng = ba->has_gps ? cdiv(ba->ncb, 256) : 0;
  BaPtrs P = make_ptrs(ba, candidate, huber);
  if (jac) {
    // The rows of the eliminated points are linearised inside the next k_point (run_assemble), which also writes their
    // camera-major copies and cost partials [0, nblk_pt); here only the rows of frozen points (their blocks
    // [nblk_pt, nblk_pt + ntail)) and the GPS rows.  The sum into `slot` follows k_point.
    KTimer t(ctx, "ba_linearize");
    const int ntail = cdiv(ba->A - ba->AE, 256);
    if (ntail) hipLaunchKernelGGL(k_linearize<true>, dim3(ntail), dim3(256), 0, s, P, ba->AE, ba->partial.p + ba->nblk_pt, spec);
    if (ng) hipLaunchKernelGGL(k_gps<true>, dim3(ng), dim3(256), 0, s, ba->ncb, ba->cb_cam.p, P.cam, ba->gps.p, ba->gps_weight, huber, ba->scale_c.p, ba->g_r.p, ba->g_J.p, ba->partial.p + ba->nblk_pt + ntail, spec);
    ba->lin_pending = true;
    ba->lin_huber = huber;
    (void)slot;   // S_XCOST
    return MSFM_OK;
  }
  {
    KTimer t(ctx, "ba_cost");
    hipLaunchKernelGGL(k_linearize<false>, dim3(nb), dim3(256), 0, s, P, 0, ba->partial.p, (const double*)nullptr);
    if (ng) hipLaunchKernelGGL(k_gps<false>, dim3(ng), dim3(256), 0, s, ba->ncb, ba->cb_cam.p, P.cam, ba->gps.p, ba->gps_weight, huber, ba->scale_c.p, ba->g_r.p, ba->g_J.p, ba->partial.p + nb, (const double*)nullptr);
    ReduceJobs rj;
    rj.count = 1;
    rj.job[0] = {ba->partial.p, nb + (lead ? ng : 0), slot, 0};
    rj.fail = with_fail ? ba->fail.p : nullptr;   // after the solve: the factorisation's / finiteness failure bits are final here
    rj.fail_slot = S_FAIL;
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, s, rj, ba->swrite);
  }
  return MSFM_OK;
}

// The point kernel: linearises the rows of the eliminated points at x (or, `candidate`, at the candidate: the launch that
// msfm_ba_run enqueues ahead of the host's view of a step, gated and given its radius by `spec`), eliminates the points.
static void launch_point(msfm_ba* ba, const msfm_ba_options* opt, double radius, bool reuse_diag, int mode, bool store_rows, bool candidate,
                         const double* spec) {
  msfm_ctx* ctx = ba->ctx;
  KTimer t(ctx, "ba_point");
  PointPtrs Q;
  Q.B = make_ptrs(ba, candidate, ba->lin_huber);
  Q.npb = ba->npb; Q.NCR = std::max(1, ba->NCR); Q.pt_first = ba->pt_first.p;
  Q.pm_first = ba->pm_first.p; Q.pm_mb = ba->pm_mb.p;
  Q.diag_p = ba->diag_p.p; Q.ptL = ba->ptL.p; Q.ptg = ba->ptg.p;
  Q.T = ba->T.p; Q.Tu = ba->Tu.p; Q.Tm = ba->Tm.p; Q.Tmu = ba->Tmu.p;
  Q.radius = radius; Q.dmin = opt->min_lm_diagonal; Q.dmax = opt->max_lm_diagonal;
  Q.reuse_diag = reuse_diag; Q.mode = mode; Q.fail = ba->fail.p;
  Q.store_rows = store_rows ? 1 : 0; Q.cost_partial = ba->partial.p;
  const FoldTables& F = ba->fold;
  Q.fold_wg = F.on ? F.wg_fold.p : nullptr; Q.fold_ovf_off = F.ovf_off.p; Q.fold_wg_pass_first = F.wg_pass_first.p; Q.fold_slot_rank = F.slot_rank.p;
  Q.fold_pass = F.pass.p; Q.fold_stream = F.stream.p; Q.fold_partial = F.partial.p;
  Q.fold_mc_partial = (F.on && F.mc_on) ? F.mc_partial.p : nullptr;
  Q.spec = spec;
  { static const bool keep = getenv("MSFM_KEEP_T") != nullptr && atoi(getenv("MSFM_KEEP_T")) != 0; Q.keep_T = keep ? 1 : 0; }
  { static const bool d = !(getenv("MSFM_TU_DIRECT") != nullptr && atoi(getenv("MSFM_TU_DIRECT")) == 0); Q.tu_direct = d ? 1 : 0; }
  hipLaunchKernelGGL(k_point, dim3(ba->nblk_pt), dim3(256), 0, ctx->stream, Q, ba->gmax_buf.p);
}

// point kernel + camera sums (+, for mode 0, the reduced system in M).  point_enqueued: the point kernel of this
// linearisation point is already in the stream (msfm_ba_run enqueued it ahead of the step's read-back).
static int run_assemble(msfm_ba* ba, const msfm_ba_options* opt, double radius, bool reuse_diag, int mode, bool point_enqueued = false) {
  msfm_ctx* ctx = ba->ctx;
  hipStream_t s = ctx->stream;
  const int ncb = ba->ncb, nmb = ba->nmb;
  const int lead = ctx->rank == 0 ? 1 : 0;
  const bool store_rows = ba->lin_pending;   // new linearisation point (else: the same rows again, new radius)
  ba->lin_pending = false;
  if (!point_enqueued) launch_point(ba, opt, radius, reuse_diag, mode, store_rows, false, nullptr);
  // The Schur pair products read only what k_point wrote (T, Tm, Tmu) and write their own partials; the per-camera sums
  // (k_ftf ... k_modelsum, with the multi-rank exchange of camftf) read T.u and the camera rows.  Outside profiling runs
  // the pair products therefore go to a second stream beside them and the two meet again at the assembly (with the
  // per-class timers on, everything stays on the main stream so that the timings remain per kernel).
  // (order: the two small lists first - they share the chip with k_ftf, which needs the bandwidth as much as the
  // camera x camera list does, so those two would only slow each other down - and the large list last, beside the
  // latency-bound k_camftf / k_modelsum / k_reduce tail of the main stream)
  auto launch_pairs = [&](hipStream_t sp) {
    if (ba->fold.on && ba->fold.mc_on) {
      if (ba->fold.mc_n_live)
        hipLaunchKernelGGL((k_pairs<3, 6, false, true>), dim3(cdiv(ba->fold.mc_n_live, 4)), dim3(256), 0, sp, ba->fold.mc_n_live, ba->mc.ch_start.p,
                           ba->mc.ch_end.p, ba->mc.pa.p, ba->mc.pb.p, ba->Tm.p, ba->T.p, (const double*)nullptr, (size_t)std::max(1, ba->NCR),
                           ba->mc.partial.p, ba->fold.mc_live_chunk.p);
    } else if (ba->mc.n_chunks)
      hipLaunchKernelGGL((k_pairs<3, 6, false>), dim3(cdiv(ba->mc.n_chunks, 4)), dim3(256), 0, sp, ba->mc.n_chunks, ba->mc.ch_start.p,
                         ba->mc.ch_end.p, ba->mc.pa.p, ba->mc.pb.p, ba->Tm.p, ba->T.p, (const double*)nullptr, (size_t)std::max(1, ba->NCR), ba->mc.partial.p);
    if (ba->mm.n_chunks)
      hipLaunchKernelGGL((k_pairs<3, 3, true>), dim3(cdiv(ba->mm.n_chunks, 4)), dim3(256), 0, sp, ba->mm.n_chunks, ba->mm.ch_start.p,
                         ba->mm.ch_end.p, ba->mm.pa.p, ba->mm.pb.p, ba->Tm.p, ba->Tm.p, ba->Tmu.p, (size_t)0, ba->mm.partial.p);
    if (ba->fold.on) {
      if (ba->fold.n_live)
        hipLaunchKernelGGL((k_pairs<6, 6, false, true>), dim3(cdiv(ba->fold.n_live, 4)), dim3(256), 0, sp, ba->fold.n_live, ba->cc.ch_start.p,
                           ba->cc.ch_end.p, ba->cc.pa.p, ba->cc.pb.p, ba->T.p, ba->T.p, (const double*)nullptr, (size_t)std::max(1, ba->NCR),
                           ba->cc.partial.p, ba->fold.live_chunk.p);
    } else if (ba->cc.n_chunks)
      hipLaunchKernelGGL((k_pairs<6, 6, false>), dim3(cdiv(ba->cc.n_chunks, 4)), dim3(256), 0, sp, ba->cc.n_chunks, ba->cc.ch_start.p,
                         ba->cc.ch_end.p, ba->cc.pa.p, ba->cc.pb.p, ba->T.p, ba->T.p, (const double*)nullptr, (size_t)std::max(1, ba->NCR), ba->cc.partial.p);
  };
  // Round 3: with the gather path (the three pair kernels fill the chip by themselves) the fork gave nothing (1.277 ms per
  // iteration with it, 1.255-1.273 without); with the fold tables the pair kernels are three short launches and the fork
  // pays again (1.251 -> 1.229 ms per iteration at config 3).  So: forked when the fold tables are on, MSFM_OVERLAP=0 / 1 forces.
  static const char* overlap_env = getenv("MSFM_OVERLAP");
  const bool overlap = overlap_env ? atoi(overlap_env) != 0 : ba->fold.on;
  // (round 4, later: with the fold tables and one intrinsics block everything between k_point and the assembly is ONE launch on
  //  the main stream - k_sums - and nothing is forked)
  static const char* fused_env = getenv("MSFM_FUSED_SUMS");
  // (also for the problems too small for the fold tables - the window of a new camera: their full pair lists are three
  //  latency-bound launches of 13-25 us one after the other; a LARGE problem without fold tables keeps the separate launches,
  //  whose pair kernels want more waves per SIMD than this launch has)
  const bool fused = mode == 0 && ba->n_fchunks > 0 && (ba->fold.on || ba->cc.n_pairs < 262144) && !(fused_env && atoi(fused_env) == 0);
  const bool forked = mode == 0 && !ctx->profile && overlap && !fused;
  // zero fill of the reduced system in front of the assembly (reads nothing: with the fork it runs on the second stream too)
  auto zero_system = [&](hipStream_t sz) -> int {
    const int nb64 = ba->npad / 64;
    if (ctx->world > 1 || nb64 > 256) {
      // (several ranks sum whole rows of M: every entry must be defined)
      HIP_TRY(ctx, hipMemsetAsync(ba->M.p, 0, sizeof(double) * (size_t)ba->npad * ba->npad, sz));
    } else {
      ZeroMap Z;
      Z.nb = nb64;
      for (int b = 0; b < nb64; b++) { Z.lev[b] = 127; Z.lo[b] = 0; Z.hi[b] = 0x7fff; }   // root chain / dense order: couples to everything
      for (int lv = 0; lv < ba->plan.n_levels; lv++)
        for (int k = 0; k < ba->plan.level[lv].K; k++) {
          const msfm_chol_node& nd = ba->plan.level[lv].node[k];
          for (int b = nd.begin / 64; b < nd.end / 64; b++) { Z.lev[b] = (unsigned char)lv; Z.lo[b] = (short)nd.leaf_lo; Z.hi[b] = (short)nd.leaf_hi; }
        }
      hipLaunchKernelGGL(k_zero_system, dim3(nb64 * (nb64 + 1) / 2), dim3(256), 0, sz, ba->M.p, ba->npad, Z);
    }
    return MSFM_OK;
  };
  auto zero_map = [&](ZeroMap& Z) -> int {   // the tiles k_sums fills (0: the plain memset is needed instead)
    const int nb64 = ba->npad / 64;
    if (ctx->world > 1 || nb64 > 256) return 0;
    Z.nb = nb64;
    for (int b = 0; b < nb64; b++) { Z.lev[b] = 127; Z.lo[b] = 0; Z.hi[b] = 0x7fff; }
    for (int lv = 0; lv < ba->plan.n_levels; lv++)
      for (int k = 0; k < ba->plan.level[lv].K; k++) {
        const msfm_chol_node& nd = ba->plan.level[lv].node[k];
        for (int b = nd.begin / 64; b < nd.end / 64; b++) { Z.lev[b] = (unsigned char)lv; Z.lo[b] = (short)nd.leaf_lo; Z.hi[b] = (short)nd.leaf_hi; }
      }
    return nb64 * (nb64 + 1) / 2;
  };
  if (forked) {
    if (!ctx->stream2) {
      HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
      HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
      HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, s));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
    launch_pairs(ctx->stream2);
    MSFM_TRY(zero_system(ctx->stream2));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_join, ctx->stream2));
  }
  // Every way out of this function from here on - the error returns of the collective hook and of HIP_TRY included - must
  // leave the second stream joined: whoever synchronises ctx->stream afterwards (msfm_ba_destroy, the pool) then also
  // waits for the pair kernels that read T / Tm and write the partials.
  struct JoinGuard {
    msfm_ctx* c; hipStream_t s; bool armed;
    ~JoinGuard() { if (armed && hipStreamWaitEvent(s, c->ev_join, 0) != hipSuccess) (void)hipStreamSynchronize(c->stream2); }
  } join{ctx, s, forked};
  double* gmax_c = ba->gmax_buf.p + ba->nblk_pt;
  double* gmax_m = gmax_c + 6 * (size_t)ncb;
  {
    KTimer t(ctx, "ba_ftf");
    if (fused) {
      const size_t ncr = (size_t)std::max(1, ba->NCR);
      SumsArgs a;
      a.n_zero = zero_map(a.Z);
      if (a.n_zero == 0) MSFM_TRY(zero_system(s));
      a.M = ba->M.p; a.ld = ba->npad;
      const bool mc_sparse = ba->fold.on && ba->fold.mc_on;
      a.mc_n = mc_sparse ? ba->fold.mc_n_live : ba->mc.n_chunks; a.mc_start = ba->mc.ch_start.p; a.mc_end = ba->mc.ch_end.p; a.mc_pa = ba->mc.pa.p; a.mc_pb = ba->mc.pb.p;
      a.mc_live = mc_sparse ? ba->fold.mc_live_chunk.p : nullptr; a.mc_partial = ba->mc.partial.p;
      a.mm_n = ba->mm.n_chunks; a.mm_start = ba->mm.ch_start.p; a.mm_end = ba->mm.ch_end.p; a.mm_pa = ba->mm.pa.p; a.mm_pb = ba->mm.pb.p; a.mm_partial = ba->mm.partial.p;
      a.cc_n = ba->fold.on ? ba->fold.n_live : ba->cc.n_chunks; a.cc_start = ba->cc.ch_start.p; a.cc_end = ba->cc.ch_end.p; a.cc_pa = ba->cc.pa.p; a.cc_pb = ba->cc.pb.p;
      a.cc_live = ba->fold.on ? ba->fold.live_chunk.p : nullptr; a.cc_partial = ba->cc.partial.p;
      a.T = ba->T.p; a.Tm = ba->Tm.p; a.Tmu = ba->Tmu.p; a.plane = ncr;
      a.f_n = ba->n_fchunks; a.f_start = ba->f_start.p; a.f_end = ba->f_end.p; a.P = make_ptrs(ba, false, ba->lin_huber);
      a.R = CamRows{ba->cm_pt.p, ba->cm_X.p, ba->cm_xyw.p, ba->cm_xyw.p + ncr, ba->cm_xyw.p + 2 * ncr, ba->chunk_cam.p};
      a.Tu = ba->Tu.p; a.cpos_pb = ba->cpos_pb.p; a.f_partial = ba->f_partial.p;
      a.n_mc_wg = cdiv(a.mc_n, 4); a.n_mm_wg = cdiv(a.mm_n, 4); a.n_cc_wg = cdiv(a.cc_n, 4); a.n_ftf_wg = cdiv(a.f_n, 4);
      hipLaunchKernelGGL(k_sums, dim3(a.n_zero + a.n_mc_wg + a.n_mm_wg + a.n_cc_wg + a.n_ftf_wg), dim3(256), 0, s, a);
    } else if (ba->n_fchunks) {
      const size_t ncr = (size_t)std::max(1, ba->NCR);
      const CamRows R{ba->cm_pt.p, ba->cm_X.p, ba->cm_xyw.p, ba->cm_xyw.p + ncr, ba->cm_xyw.p + 2 * ncr, ba->chunk_cam.p};
      hipLaunchKernelGGL(k_ftf, dim3(cdiv(ba->n_fchunks, 4)), dim3(256), 0, s, ba->n_fchunks, ba->f_start.p, ba->f_end.p,
                         make_ptrs(ba, false, ba->lin_huber), R, ba->Tu.p, ba->cpos_pb.p, ba->f_partial.p);
    }
    if (ncb)
      hipLaunchKernelGGL(k_camftf, dim3(ncb), dim3(128), 0, s, ncb, ba->cam_chunk_first.p, ba->f_partial.p, ba->camftf.p,
                         ba->g_r.p, ba->g_J.p, (ba->has_gps && lead) ? 1 : 0, ctx->world <= 1 ? 1 : 0, ba->diag_c.p, ba->scale_c.p,
                         reuse_diag ? 1 : 0, mode, opt->min_lm_diagonal, opt->max_lm_diagonal, gmax_c);
  }
  if (ncb) MSFM_TRY(allreduce(ba, ba->camftf.p, (size_t)ncb * PSTRIDE, MSFM_REDUCE_SUM));  // per-camera sums over all shards
  {
    KTimer t(ctx, "ba_ftf");
    if (ncb && ctx->world > 1)
      hipLaunchKernelGGL(k_cam_post, dim3(cdiv(6 * ncb, 256)), dim3(256), 0, s, ncb, ba->camftf.p, ba->diag_c.p, ba->scale_c.p,
                         reuse_diag ? 1 : 0, mode, opt->min_lm_diagonal, opt->max_lm_diagonal, gmax_c);
    if (nmb)
      hipLaunchKernelGGL(k_modelsum, dim3(nmb), dim3(64), 0, s, ba->mcam_first.p, ba->mcam.p, ba->camftf.p, ba->modelsum.p,
                         ba->diag_m.p, ba->scale_m.p, reuse_diag ? 1 : 0, mode, opt->min_lm_diagonal, opt->max_lm_diagonal, gmax_m);
  }
  if (mode == 1) return MSFM_OK;
  {
    ReduceJobs rj;
    rj.count = 1;
    rj.job[0] = {ba->gmax_buf.p, ba->nblk_pt + 6 * ncb + 3 * nmb, S_GMAX, 1};
    if (store_rows) {
      // cost at x: k_point's blocks, the frozen points' rows, the GPS rows (lead rank only) - see run_evaluate
      const int ntail = cdiv(ba->A - ba->AE, 256), ng = (ba->has_gps && lead) ? cdiv(ncb, 256) : 0;
      rj.job[1] = {ba->partial.p, ba->nblk_pt + ntail + ng, S_XCOST, 0};
      rj.count = 2;
    }
    rj.fail = ba->fail.p;
    rj.fail_slot = S_FAIL;
    hipLaunchKernelGGL(k_reduce, dim3(rj.count), dim3(1024), 0, s, rj, ba->swrite);
  }
  if (forked) {
    join.armed = false;
    const hipError_t je = hipStreamWaitEvent(s, ctx->ev_join, 0);
    if (je != hipSuccess) { (void)hipStreamSynchronize(ctx->stream2); HIP_TRY(ctx, je); }
  } else if (!fused) {
    KTimer t(ctx, "ba_schur_pairs");
    launch_pairs(s);
  }
  {
    KTimer t(ctx, "ba_assemble");
    if (!forked && !fused) MSFM_TRY(zero_system(s));   // (forked: done on the second stream beside the per-camera sums, joined above)
    AsmArgs aa;
    aa.n_cc = ba->cc.n_blocks; aa.n_mc = ba->mc.n_blocks; aa.n_mm = ba->mm.n_blocks; aa.n_rhs = cdiv(6 * ncb, 64);
    aa.cc_row = ba->cc.blk_row.p; aa.cc_col = ba->cc.blk_col.p; aa.cc_first = (ba->fold.on && ba->fold.all) ? nullptr : ba->cc.blk_chunk_first.p; aa.cc_partial = ba->cc.partial.p;
    aa.cc_fold_first = ba->fold.on ? ba->fold.blk_range.p : nullptr; aa.cc_live = ba->fold.on ? ba->fold.blk_live.p : nullptr; aa.cc_fold_partial = ba->fold.partial.p;
    aa.mc_row = ba->mc.blk_row.p; aa.mc_col = ba->mc.blk_col.p; aa.mc_partial = ba->mc.partial.p;
    {
      const bool fmc = ba->fold.on && ba->fold.mc_on;
      aa.mc_first = (fmc && ba->fold.mc_all) ? nullptr : ba->mc.blk_chunk_first.p;
      aa.mc_fold_range = fmc ? ba->fold.mc_range.p : nullptr; aa.mc_fold_partial = ba->fold.mc_partial.p;
    }
    aa.mm_row = ba->mm.blk_row.p; aa.mm_col = ba->mm.blk_col.p; aa.mm_first = ba->mm.blk_chunk_first.p; aa.mm_partial = ba->mm.partial.p;
    aa.cb_mb = ba->cb_mb.p; aa.cb_off = ba->cb_off.p; aa.camftf = ba->camftf.p; aa.diag_c = ba->diag_c.p; aa.modelsum = ba->modelsum.p;
    aa.diag_m = ba->diag_m.p; aa.radius = radius; aa.ncb = ncb; aa.mo = ba->mo; aa.n = ba->nsys; aa.ld = ba->npad; aa.lead = lead; aa.M = ba->M.p;
    aa.n_padcol = ctx->world <= 1 ? ba->n_padcol : 0;
    aa.padcol = ba->padcol.p;
    const int nasm = aa.n_cc + aa.n_mc + aa.n_mm + aa.n_rhs + cdiv(aa.n_padcol, 64);
    if (nasm) hipLaunchKernelGGL(k_asm_all, dim3(nasm), dim3(64), 0, s, aa);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return msfm_set_error(ctx, MSFM_E_DEVICE, "assemble launch: %s", hipGetErrorString(e));
  if (ctx->world > 1) {
    if (ba->n_ublk > 0) {
      // camera-camera blocks packed (union structure), intrinsics rows + rhs row as one dense slab
      const int nb = cdiv(ba->n_ublk * 36, 256);
      // ... and the intrinsics rows + rhs row (contiguous in M) behind them: one sum for the whole system
      const size_t nblk = (size_t)ba->n_ublk * 36, nslab = (size_t)(ba->nsys - ba->mo + 1) * ba->npad;
      double* slab = ba->M.p + (size_t)ba->mo * ba->npad;
      hipLaunchKernelGGL(k_pack_blocks<true>, dim3(nb), dim3(256), 0, s, ba->n_ublk, ba->u_row.p, ba->u_col.p, ba->cb_off.p, ba->M.p, ba->npad, ba->pack.p);
      HIP_TRY(ctx, hipMemcpyAsync(ba->pack.p + nblk, slab, sizeof(double) * nslab, hipMemcpyDeviceToDevice, s));
      MSFM_TRY(allreduce(ba, ba->pack.p, nblk + nslab, MSFM_REDUCE_SUM));
      hipLaunchKernelGGL(k_pack_blocks<false>, dim3(nb), dim3(256), 0, s, ba->n_ublk, ba->u_row.p, ba->u_col.p, ba->cb_off.p, ba->M.p, ba->npad, ba->pack.p);
      HIP_TRY(ctx, hipMemcpyAsync(slab, ba->pack.p + nblk, sizeof(double) * nslab, hipMemcpyDeviceToDevice, s));
    } else {
      // rows [0, nred] of M (S and the rhs row) are contiguous: one sum over ranks
      MSFM_TRY(allreduce(ba, ba->M.p, (size_t)(ba->nsys + 1) * ba->npad, MSFM_REDUCE_SUM));
    }
  }
  // identity on the padding columns (after the exchange: they are not part of it)
  if (ba->n_padcol && ctx->world > 1) hipLaunchKernelGGL(k_pad_diag, dim3(cdiv(ba->n_padcol, 256)), dim3(256), 0, s, ba->n_padcol, ba->padcol.p, ba->M.p, ba->npad);
  return MSFM_OK;
}

__global__ void k_scal_pack(const double* __restrict__ sloc, double* __restrict__ scal, int rank, int world) {
  const int i = threadIdx.x;
  if (i < S_GMAX) scal[i] = sloc[i];
  if (i < world) scal[S_RANK0 + i] = i == rank ? sloc[S_GMAX] : 0.0;
}
__global__ void k_scal_unpack(double* __restrict__ scal, int world) {
  double g = 0.0;  // gradient norms are non-negative
  for (int r = 0; r < world; r++) g = fmax(g, scal[S_RANK0 + r]);
  scal[S_GMAX] = g;
}

// all scalars of the iteration over the ranks, in one sum (see the enum)
static int reduce_scalars(msfm_ba* ba) {
  msfm_ctx* ctx = ba->ctx;
  if (ctx->world <= 1) return MSFM_OK;
  if (S_RANK0 + ctx->world > S_N) return msfm_set_error(ctx, MSFM_E_INVAL, "too many ranks for the scalar block");
  hipLaunchKernelGGL(k_scal_pack, dim3(1), dim3(64), 0, ctx->stream, ba->sloc.p, ba->scal.p, ctx->rank, ctx->world);
  MSFM_TRY(allreduce(ba, ba->scal.p, (size_t)S_RANK0 + ctx->world, MSFM_REDUCE_SUM));
  hipLaunchKernelGGL(k_scal_unpack, dim3(1), dim3(1), 0, ctx->stream, ba->scal.p, ctx->world);
  return MSFM_OK;
}

static void publish_scalars(msfm_ba* ba, const LmDecide& D) {
  msfm_ctx* ctx = ba->ctx;
  const unsigned long long seq = ++ba->scal_seq;
  hipLaunchKernelGGL(k_publish_scalars, dim3(1), dim3(64), 0, ctx->stream, ba->scal.p, ba->h_scal_dev, seq, ba->fail.p, D, ba->spec.p);
}
static int wait_scalars(msfm_ba* ba) {
  msfm_ctx* ctx = ba->ctx;
  const unsigned long long seq = ba->scal_seq;
  // spin on the sequence number: the wake-up of a blocking wait costs more than the whole hand-over.  Bounded: a wedged kernel,
  // or a peer rank that left the loop so that a collective never completes, must surface as an error code, not as a host thread
  // spinning forever; a launch or execution error surfaces through the stream query that accompanies the clock check.
  static const double limit_s = [] { const char* e = getenv("MSFM_SYNC_TIMEOUT_S"); const double v = e ? atof(e) : 120.0; return v > 0 ? v : 120.0; }();
  const volatile unsigned long long* flag = reinterpret_cast<const volatile unsigned long long*>(ba->h_scal + H_SEQ);
  unsigned long long spins = 0;
  const auto t0 = std::chrono::steady_clock::now();   // taken once: the deadline never re-arms, whatever the poll count does
  while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
    if ((++spins & 0x3FFFFull) == 0) {   // look at the stream and the clock every 262 144 polls only
      const hipError_t q = hipStreamQuery(ctx->stream);
      if (q != hipSuccess && q != hipErrorNotReady) HIP_TRY(ctx, q);
      if (q == hipSuccess && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq)
        return msfm_set_error(ctx, MSFM_E_DEVICE, "the iteration's scalars did not arrive although the stream is idle");
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s)
        return msfm_set_error(ctx, MSFM_E_DEVICE, "device did not finish an LM iteration within %.0f s (MSFM_SYNC_TIMEOUT_S)", limit_s);
    }
  }
  return MSFM_OK;
}

// factor + solve + candidate + model cost change + candidate cost
static int run_solve(msfm_ba* ba, const msfm_ba_options* opt) {
  msfm_ctx* ctx = ba->ctx;
  hipStream_t s = ctx->stream;
  const int ncb = ba->ncb, nmb = ba->nmb, npb = ba->npb;
  const bool lead = ctx->rank == 0;
  const double* zsolved = ba->zsys.p;
  if (ba->nred > 0) {
    double* const zcur = ba->zsys.p + (size_t)ba->zflip * (ba->npad + 8);
    double* const znext = ba->zsys.p + (size_t)(ba->zflip ^ 1) * (ba->npad + 8);
    const int rc = msfm_chol_factor_solve(ctx, ba->M.p, ba->npad, ba->nsys, ba->Linv.p, ba->w.p, zcur, ba->fail.p,
                                          ba->plan.n_levels > 0 ? &ba->plan : nullptr, znext, ba->chol_ws);
    if (rc != MSFM_OK) {
      // the call may have stopped before the solve kernel marked znext: neither half can be trusted to be "pending"
      // any more (a stale half would be taken for published values by the next solve) - mark both again
      (void)msfm_chol_fill_pending(ctx, ba->zsys.p, 2 * (ba->npad + 8));
      return rc;
    }
    ba->zflip ^= 1;   // only a completed solve hands the other half over
    zsolved = zcur;
  }
  {
    KTimer t(ctx, "ba_backsub");
    const int nbu = cdiv(std::max(1, ba->nred), 256), nbp = ba->nblk_pt;
    int off = 0;
    const double wrep = lead ? 1.0 : 0.0;
    // partial2 = |dx|^2 partials, partial3 = |x|^2 partials, partial = model cost partials
    if (ba->nred > 0) {
      hipLaunchKernelGGL(k_update_params, dim3(nbu), dim3(256), 0, s, ncb, nmb, ba->cb_cam.p, ba->mb_model.p, ba->cb_off.p, ba->mo, zsolved,
                         ba->scale_c.p, ba->scale_m.p, ba->cam.p, ba->cam_c.p, ba->model.p, ba->model_c.p, ba->z.p, ba->fail.p,
                         ba->partial2.p + off, ba->partial3.p + off, wrep);
      off += nbu;
    }
    int moff = 0;
    if (npb) {
      BackPtrs Q;
      Q.B = make_ptrs(ba, false, ba->lin_huber);
      Q.npb = npb; Q.ncb = ncb; Q.pt_first = ba->pt_first.p; Q.pb_pt = ba->pb_pt.p;
      Q.ptL = ba->ptL.p; Q.z = ba->z.p; Q.pt_c = ba->pt_c.p; Q.cm_Xc = ba->cm_Xc.p;
      Q.Nc = ba->Nc; Q.cam_c = ba->cam_c.p; Q.rot_c = ba->rot_c.p;
      hipLaunchKernelGGL(k_backsub, dim3(nbp), dim3(256), 0, s, Q, ba->partial.p, ba->partial2.p + off, ba->partial3.p + off);
      off += nbp; moff += nbp;
    } else {
      hipLaunchKernelGGL(k_rot_cache, dim3(cdiv(std::max(1, ba->Nc), 256)), dim3(256), 0, s, ba->Nc, ba->cam_c.p, ba->rot_c.p);
    }
    const int ngps = (ba->has_gps && lead) ? ncb : 0;
    const int nrest = (ba->A - ba->AE) + ngps;
    const char* tail_env = getenv("MSFM_FUSED_TAIL");   // (read per call: the tests switch it inside one process)
    if (!(tail_env && atoi(tail_env) == 0)) {
      // model cost change of the remaining rows + cost at the candidate (all rows, GPS rows) in one launch, one reduction for
      // the four sums (the failure bits of the factorisation / finiteness checks are final here)
      t.stop();   // (the class ba_backsub ends here: what follows is timed as ba_cost)
      TailArgs a;
      a.n_mcc = nrest ? cdiv(nrest, 256) : 0;
      a.A = ba->A; a.AE = ba->AE; a.ncb = ncb; a.o_cb = ba->o_cb.p; a.o_mb = ba->o_mb.p; a.lin_r = ba->lin_r.p; a.lin_Jc = ba->lin_Jc.p; a.lin_Jm = ba->lin_Jm.p;
      a.z = ba->z.p; a.has_gps = ngps ? 1 : 0; a.g_r = ba->g_r.p; a.g_J = ba->g_J.p; a.mcc_partial = ba->partial.p + moff;
      moff += a.n_mcc;
      a.P = make_ptrs(ba, /*candidate=*/true, opt->huber_delta);
      a.n_cost = ba->nblk_obs; a.cost_partial = ba->partial4.p;
      a.n_gps = ba->has_gps ? cdiv(ncb, 256) : 0;
      a.cb_cam = ba->cb_cam.p; a.cam_c = a.P.cam; a.gps = ba->gps.p; a.gps_weight = ba->gps_weight; a.huber = opt->huber_delta; a.scale_c = ba->scale_c.p;
      a.g_r_w = ba->g_r.p; a.g_J_w = ba->g_J.p;   // (not written: WRITE_JAC is false)
      KTimer t2(ctx, "ba_cost");
      hipLaunchKernelGGL(k_tail, dim3(a.n_mcc + a.n_cost + a.n_gps), dim3(256), 0, s, a);
      ReduceJobs rj;
      rj.count = 4;
      rj.job[0] = {ba->partial.p, moff, S_MCC, 0};
      rj.job[1] = {ba->partial2.p, off, S_DX2, 0};
      rj.job[2] = {ba->partial3.p, off, S_X2, 0};
      rj.job[3] = {ba->partial4.p, a.n_cost + (lead ? a.n_gps : 0), S_COST, 0};
      rj.fail = ba->fail.p;
      rj.fail_slot = S_FAIL;
      hipLaunchKernelGGL(k_reduce, dim3(4), dim3(1024), 0, s, rj, ba->swrite);
      hipError_t e2 = hipGetLastError();
      if (e2 != hipSuccess) return msfm_set_error(ctx, MSFM_E_DEVICE, "solve launch: %s", hipGetErrorString(e2));
      return MSFM_OK;
    }
    if (nrest) {
      hipLaunchKernelGGL(k_mcc_rest, dim3(cdiv(nrest, 256)), dim3(256), 0, s, ba->A, ba->AE, ncb, ba->o_cb.p, ba->o_mb.p, ba->lin_r.p,
                         ba->lin_Jc.p, ba->lin_Jm.p, ba->z.p, ngps ? 1 : 0, ba->g_r.p, ba->g_J.p, ba->partial.p + moff);
      moff += cdiv(nrest, 256);
    }
    ReduceJobs rj;
    rj.count = 3;
    rj.job[0] = {ba->partial.p, moff, S_MCC, 0};
    rj.job[1] = {ba->partial2.p, off, S_DX2, 0};
    rj.job[2] = {ba->partial3.p, off, S_X2, 0};
    rj.fail = nullptr;
    rj.fail_slot = S_FAIL;
    hipLaunchKernelGGL(k_reduce, dim3(3), dim3(1024), 0, s, rj, ba->swrite);
  }
  MSFM_TRY(run_evaluate(ba, /*candidate=*/true, /*jac=*/false, opt->huber_delta, S_COST, /*with_fail=*/true));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return msfm_set_error(ctx, MSFM_E_DEVICE, "solve launch: %s", hipGetErrorString(e));
  return MSFM_OK;
}

MSFM_API int msfm_ba_run(msfm_ba* ba, const msfm_ba_options* opt, msfm_ba_summary* sum) {
  if (!ba || !opt || !sum) return MSFM_E_INVAL;
  msfm_ctx* ctx = ba->ctx;
  if (ctx->world != ba->world_at_create)
    return msfm_set_error(ctx, MSFM_E_INVAL, "msfm_ctx_set_allreduce must be called before msfm_ba_create");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  struct Events {   // destroyed on every exit path
    hipEvent_t a = nullptr, b = nullptr;
    ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
  } evs;
  hipEvent_t& ev0 = evs.a;
  hipEvent_t& ev1 = evs.b;
  const auto run_t0 = std::chrono::steady_clock::now();
  const bool verbose = getenv("MSFM_VERBOSE") != nullptr && ctx->rank == 0;
  auto lap = [&](const char* what) {
    if (verbose) fprintf(stderr, "msfm: run %-31s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - run_t0).count());
  };
  HIP_TRY(ctx, hipEventCreate(&ev0));
  HIP_TRY(ctx, hipEventCreate(&ev1));
  HIP_TRY(ctx, hipEventRecord(ev0, s));
  sum->num_residuals = ba->n_residuals;
  sum->num_reduced_params = ba->nred;
  sum->num_successful_steps = sum->num_unsuccessful_steps = 0;
  sum->setup_ms = ba->setup_ms;
  int rows = 0;
  auto push = [&](const msfm_ba_iteration& it) {
    if (sum->iterations && rows < sum->iterations_capacity) sum->iterations[rows] = it;
    rows++;
  };
  const int ncb = ba->ncb, nmb = ba->nmb, npb = ba->npb;
  if (ncb) hipLaunchKernelGGL(k_fill, dim3(cdiv(6 * ncb, 256)), dim3(256), 0, s, 6 * ncb, 1.0, ba->scale_c.p);
  if (nmb) hipLaunchKernelGGL(k_fill, dim3(cdiv(3 * nmb, 256)), dim3(256), 0, s, 3 * nmb, 1.0, ba->scale_m.p);
  if (npb) hipLaunchKernelGGL(k_fill, dim3(cdiv(3 * npb, 256)), dim3(256), 0, s, 3 * npb, 1.0, ba->scale_p.p);
  hipLaunchKernelGGL(k_rot_cache, dim3(cdiv(std::max(1, ba->Nc), 256)), dim3(256), 0, s, ba->Nc, ba->cam.p, ba->rot.p);
  {
    // the candidate buffers start as copies of x: the steps overwrite every active block, the inactive ones never change
    // (x and its candidate swap roles at every accepted step, so both must hold them)
    const size_t n0 = 6 * (size_t)ba->Nc, n1 = 3 * (size_t)ba->Nm, n2 = 3 * (size_t)ba->Np;
    hipLaunchKernelGGL(k_copy3, dim3(cdiv((long)std::max(n0, std::max(n1, n2)), 256)), dim3(256), 0, s, n0, ba->cam.p, ba->cam_c.p, n1,
                       ba->model.p, ba->model_c.p, n2, ba->pt.p, ba->pt_c.p);
  }
  if (ba->NCR > 0)
    hipLaunchKernelGGL(k_cam_points, dim3(cdiv(ba->NCR, 256)), dim3(256), 0, s, ba->NCR, ba->cm_pt.p, ba->pt.p, ba->cm_X.p, ba->cm_Xc.p);
  lap("scales reset");
  hipLaunchKernelGGL(k_zero_int, dim3(1), dim3(1), 0, s, ba->fail.p);
  double radius = opt->initial_trust_region_radius, decrease_factor = 2.0;
  bool reuse_diag = false;
  // ---- IterationZero ----
  MSFM_TRY(run_evaluate(ba, false, true, opt->huber_delta, S_XCOST));
  lap("first evaluate enqueued");
  if (opt->jacobi_scaling) {
    // squared column norms of the corrected, unscaled Jacobian -> scaling -> re-linearise scaled
    MSFM_TRY(run_assemble(ba, opt, radius, false, /*mode=*/1));
    if (ncb) hipLaunchKernelGGL(k_make_scale, dim3(cdiv(6 * ncb, 256)), dim3(256), 0, s, 6 * ncb, ba->diag_c.p, ba->scale_c.p);
    if (nmb) hipLaunchKernelGGL(k_make_scale, dim3(cdiv(3 * nmb, 256)), dim3(256), 0, s, 3 * nmb, ba->diag_m.p, ba->scale_m.p);
    if (npb) hipLaunchKernelGGL(k_make_scale, dim3(cdiv(3 * npb, 256)), dim3(256), 0, s, 3 * npb, ba->diag_p.p, ba->scale_p.p);
    MSFM_TRY(run_evaluate(ba, false, true, opt->huber_delta, S_XCOST));
  }
  int iteration = 0, num_invalid = 0, termination = 0;
  lap("iteration zero enqueued");
  // The reduced system and the trust-region step computed from it are enqueued back to back and
  // their scalars read with ONE host synchronisation per LM iteration: the step is speculative
  // only in that a gradient-tolerance stop discards it (it writes the candidate buffers only).
  // The step's verdict (valid / tolerances / accept + new radius / reject) is formed on the device by k_publish_scalars,
  // and what follows an accepted step - the linearisation at the candidate, i.e. the rows of frozen points, the GPS rows and
  // k_point - is enqueued BEHIND it before the host has seen anything (`spec`: those launches end at once unless the step was
  // accepted and take the radius from the device).  The host mirrors the verdict, swaps its buffer names and carries on
  // behind k_point, so no launch waits for the read-back (it was a 9-13 us hole in front of every k_point).
  double x_cost = 0.0;
  bool spec_inflight = false;
  auto assemble_and_step = [&](bool point_enqueued) -> int {
    LmDecide D;
    D.fresh = ba->lin_pending ? 1 : 0;
    MSFM_TRY(run_assemble(ba, opt, radius, reuse_diag, 0, point_enqueued));
    D.on = (iteration < opt->max_num_iterations && radius > opt->min_trust_region_radius) ? 1 : 0;
    if (D.on) MSFM_TRY(run_solve(ba, opt));
    MSFM_TRY(reduce_scalars(ba));
    D.x_cost = x_cost; D.radius = radius;
    D.min_relative_decrease = opt->min_relative_decrease; D.parameter_tolerance = opt->parameter_tolerance;
    D.function_tolerance = opt->function_tolerance; D.max_radius = opt->max_trust_region_radius; D.min_radius = opt->min_trust_region_radius;
    publish_scalars(ba, D);
    spec_inflight = false;
    if (D.on && ba->spec_on) {
      // x_{k+1} would be the present candidate buffers
      MSFM_TRY(run_evaluate(ba, /*candidate=*/true, /*jac=*/true, opt->huber_delta, S_XCOST, false, ba->spec.p));
      launch_point(ba, opt, radius, false, 0, true, /*candidate=*/true, ba->spec.p);
      ba->lin_pending = false;   // becomes true again if the host finds the step accepted
      spec_inflight = true;
    }
    MSFM_TRY(wait_scalars(ba));
    return MSFM_OK;
  };
  MSFM_TRY(assemble_and_step(false));
  x_cost = ba->h_scal[S_XCOST];
  msfm_ba_iteration it;
  memset(&it, 0, sizeof it);
  it.cost = x_cost; it.gradient_max_norm = ba->h_scal[S_GMAX]; it.trust_region_radius = radius;
  it.step_is_valid = 1; it.step_is_successful = 1;
  sum->initial_cost = x_cost;
  for (;;) {
    if (it.step_is_successful && iteration > 0) sum->num_successful_steps++;
    it.trust_region_radius = radius;
    push(it);
    if (verbose && iteration < 6) lap("iteration read back");
    if (opt->progress_to_stdout && ctx->rank == 0)
      printf("%4d  cost %.6e  change %.3e  |grad| %.3e  |step| %.3e  rho %.3e  radius %.3e\n", iteration, it.cost,
             it.cost_change, it.gradient_max_norm, it.step_norm, it.relative_decrease, radius);
    if (iteration >= opt->max_num_iterations) { termination = MSFM_BA_NO_CONVERGENCE; break; }
    if (it.gradient_max_norm <= opt->gradient_tolerance) { termination = MSFM_BA_CONVERGENCE_GRADIENT; break; }
    if (radius <= opt->min_trust_region_radius) { termination = MSFM_BA_MIN_RADIUS; break; }
    const double prev_gmax = it.gradient_max_norm;
    memset(&it, 0, sizeof it);
    iteration++;
    // ---- ComputeTrustRegionStep ----
    // (already enqueued with the reduced system; S_FAIL carries both the 3x3 point-block and the
    // Cholesky / finiteness failures, and any of them makes the step invalid)
    if ((long)ba->h_scal[S_FAIL] >= MSFM_FAIL_SYNC)
      return msfm_set_error(ctx, MSFM_E_DEVICE, "a bounded in-kernel wait of the back substitution ran out (MSFM_SYNC_TIMEOUT_S)");
    const int code = (int)ba->h_scal[H_CODE];   // the device's verdict on the step (k_publish_scalars); mirrored here
    const double mcc = ba->h_scal[S_MCC], cand_cost = ba->h_scal[S_COST], dx2 = ba->h_scal[S_DX2], x2 = ba->h_scal[S_X2];
    if (code < LM_INVALID || code > LM_REJECT) return msfm_set_error(ctx, MSFM_E_DEVICE, "no verdict for the step (code %d)", code);
    it.step_is_valid = code != LM_INVALID;
    bool relinearise = false;
    if (!it.step_is_valid) {
      if (++num_invalid >= opt->max_num_consecutive_invalid_steps) { termination = MSFM_BA_FAILURE; break; }
      radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diag = true;
      it.cost = x_cost; it.gradient_max_norm = prev_gmax;
      sum->num_unsuccessful_steps++;
    } else {
      num_invalid = 0;
      it.step_norm = std::sqrt(dx2);
      (void)x2;
      if (code == LM_PARAM_TOL) { termination = MSFM_BA_CONVERGENCE_PARAMETER; break; }
      it.cost_change = x_cost - cand_cost;
      if (code == LM_FUNC_TOL) { termination = MSFM_BA_CONVERGENCE_FUNCTION; break; }
      it.relative_decrease = (x_cost - cand_cost) / mcc;
      if (code == LM_ACCEPT) {
        ba->cam.swap(ba->cam_c); ba->model.swap(ba->model_c); ba->pt.swap(ba->pt_c); ba->rot.swap(ba->rot_c); ba->cm_X.swap(ba->cm_Xc);
        relinearise = true;
        it.step_is_successful = 1;
        radius = ba->h_scal[H_RADIUS];   // = min(max_radius, radius / max(1/3, 1 - (2 rho - 1)^3)), formed on the device
        decrease_factor = 2.0; reuse_diag = false;
      } else {
        it.step_is_successful = 0; it.cost = cand_cost; it.gradient_max_norm = prev_gmax;
        radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diag = true;
        sum->num_unsuccessful_steps++;
      }
    }
    // ---- next reduced system (new Jacobian after a successful step, else new radius only) ----
    // (the failure bits were cleared by k_publish_scalars when it handed the last iteration's scalars over)
    bool point_enqueued = false;
    if (relinearise) {
      if (spec_inflight && radius > opt->min_trust_region_radius) {
        // the launches enqueued ahead found the same verdict on the device and are running
        ba->lin_pending = true;
        ba->lin_huber = opt->huber_delta;
        point_enqueued = true;
      } else {
        MSFM_TRY(run_evaluate(ba, false, true, opt->huber_delta, S_XCOST));
      }
    }
    MSFM_TRY(assemble_and_step(point_enqueued));
    if (relinearise) {
      x_cost = ba->h_scal[S_XCOST];
      it.cost = x_cost;
      it.gradient_max_norm = ba->h_scal[S_GMAX];
    }
  }
  sum->termination = termination;
  sum->num_iterations = rows - 1;
  sum->final_cost = x_cost;
  lap("loop done");
  HIP_TRY(ctx, hipEventRecord(ev1, s));
  HIP_TRY(ctx, hipEventSynchronize(ev1));
  lap("synchronised");
  float ms = 0;
  HIP_TRY(ctx, hipEventElapsedTime(&ms, ev0, ev1));
  sum->solve_ms = ms;
  return MSFM_OK;
}

MSFM_API int msfm_ba_solve(msfm_ctx* ctx, msfm_ba_problem* problem, const msfm_ba_options* options, msfm_ba_summary* summary) {
  if (!ctx || !problem || !options || !summary) return MSFM_E_INVAL;
  msfm_ba* ba = nullptr;
  MSFM_TRY(msfm_ba_create(ctx, problem, &ba));
  int rc = msfm_ba_run(ba, options, summary);
  if (rc == MSFM_OK) rc = msfm_ba_download_params(ba, problem->cam_pose, problem->cam_model, problem->point);
  msfm_ba_destroy(ba);
  return rc;
}
