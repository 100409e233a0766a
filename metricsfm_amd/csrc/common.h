// Internal header of libmsfm (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <cstdlib>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/msfm.h"

#define MSFM_API extern "C" __attribute__((visibility("default")))

struct msfm_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  // second stream for work that does not depend on what the main stream is doing (the Schur pair products run beside
  // the per-camera sums: both only read what k_point wrote); created on first use, fork / join by the two events
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  std::string err;
  // multi-GPU hook
  msfm_allreduce_fn allreduce = nullptr;
  void* allreduce_user = nullptr;
  int rank = 0, world = 1;
  // contexts (of this or of other processes) that use this device at the same time, as far as the host has said so
  // (msfm_ctx_create_multi with a shared device; MSFM_DEVICE_SHARE for several processes on one GPU): kernels whose workgroups
  // wait for each other inside a launch must leave room for the others' resident workgroups
  int device_share = 1;
  // per-kernel-class timing (HIP events on `stream`)
  bool profile = false;
  struct Stat { std::string name; uint64_t launches = 0; double ms = 0; };
  std::vector<Stat> stats;
  struct Pending { int stat; hipEvent_t a, b; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> event_pool;
  // host work arrays of msfm_ba_create, kept between calls (ba.hip): a C3-sized problem touches ~150 MB of them, and
  // mapping + unmapping that much fresh memory on every call cost a quarter of the setup time
  void* ba_scratch = nullptr;
  void (*ba_scratch_free)(void*) = nullptr;
  // objects created from this context that are still alive (descriptor sets, match results, resident problems);
  // msfm_ctx_destroy refuses while there are any, and the last child of an orphaned context destroys it
  int children = 0;
  bool orphaned = false;
  // native collective (msfm_ctx_init_rccl): librccl handle, communicator and the entry points resolved from it
  void* rccl_lib = nullptr;
  void* rccl_comm = nullptr;
  void* rccl_allreduce = nullptr;
  void* rccl_comm_destroy = nullptr;
  void* rccl_error_string = nullptr;
};
void msfm_ctx_child_released(msfm_ctx* ctx);

int msfm_set_error(msfm_ctx* ctx, int code, const char* fmt, ...);

#define HIP_TRY(ctx, expr)                                                                      \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return msfm_set_error((ctx), MSFM_E_DEVICE, "%s:%d %s -> %s", __FILE__, __LINE__, #expr,  \
                            hipGetErrorString(e_));                                             \
  } while (0)

#define MSFM_TRY(expr)            \
  do {                            \
    int rc_ = (expr);             \
    if (rc_ != MSFM_OK) return rc_; \
  } while (0)

// Scoped kernel-class timer: records two events around a group of launches when
// profiling is on; resolved lazily in msfm_ctx_profile_get.
struct KTimer {
  msfm_ctx* ctx;
  int idx = -1;
  int count = 1;   // kernel launches between the two events (a chain of dependent launches is timed as a whole: an event
                   // pair around every single launch would put its own few microseconds into each of them)
  hipEvent_t a = nullptr, b = nullptr;
  KTimer(msfm_ctx* c, const char* name);
  void stop();   // records the closing event now (the destructor then does nothing)
  ~KTimer();
};

// Device memory comes from a per-process cache of freed blocks (ctx.hip): a bundle adjustment of the incremental loop
// allocates ~60 buffers / > 1 GB, and a fresh VRAM allocation costs far more than the hipMalloc call itself (the first
// kernels that touch it wait 10-20 ms at config 3).  Blocks are returned only after their stream has been synchronised.
// Host wait for a stream that carries a collective: bounded (MSFM_SYNC_TIMEOUT_S, default 120 s) - a peer rank that never
// joins must surface as MSFM_E_DEVICE, not as a host thread inside hipStreamSynchronize for good (ctx.hip).
int msfm_stream_wait_bounded(msfm_ctx* ctx, hipStream_t s, const char* what);
// Single-process multi-GPU (multi.hip): the communicator of this context was aborted (ncclCommAbort frees it) - forget it.
void msfm_ctx_forget_rccl(msfm_ctx* ctx);
hipError_t msfm_pool_alloc(void** p, size_t bytes, size_t* capacity);
void msfm_pool_free(void* p, size_t capacity);
void msfm_pool_trim(int device);

// Simple owning device buffer.
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  size_t cap = 0;  // bytes of the underlying block
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void swap(DevBuf& o) {  // exchanges the blocks (same element count expected by the callers), capacities included
    std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap);
  }
  void release() {
    if (p) msfm_pool_free(p, cap);
    p = nullptr;
    n = 0;
    cap = 0;
  }
  hipError_t alloc(size_t count) {
    release();
    n = count;
    if (count == 0) return hipSuccess;
    void* q = nullptr;
    const hipError_t e = msfm_pool_alloc(&q, count * sizeof(T), &cap);
    p = static_cast<T*>(q);
    if (e != hipSuccess) { p = nullptr; n = 0; cap = 0; }
    return e;
  }
  hipError_t upload(const T* h, size_t count, hipStream_t s) {
    if (count == 0) return hipSuccess;
    return hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, s);
  }
  hipError_t from(const std::vector<T>& v, hipStream_t s) {
    hipError_t e = alloc(v.size());
    if (e != hipSuccess) return e;
    return upload(v.data(), v.size(), s);
  }
};

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Host-side helper: the index structures of a 10^6-observation problem (msfm_ba_create) and the per-pair iteration tables
// of the RANSAC (geo.hip) are built by a few threads (MSFM_HOST_THREADS, default min(hardware threads, 8));
// fn(t, begin, end) gets one contiguous range per thread.
static inline int host_threads() {
  static const int n = [] {
    const char* e = getenv("MSFM_HOST_THREADS");
    const int v = e ? atoi(e) : std::min(8, (int)std::thread::hardware_concurrency());
    return std::max(1, std::min(v, 64));
  }();
  return n;
}
template <class F>
static inline void par_ranges(size_t n, int nt, F&& fn, size_t grain = 4096) {   // at least `grain` items per thread
  nt = (int)std::max<size_t>(1, std::min<size_t>(nt, n / grain + 1));
  if (nt == 1) { fn(0, (size_t)0, n); return; }
  std::vector<std::thread> th;
  th.reserve(nt - 1);
  for (int t = 1; t < nt; t++) th.emplace_back([&fn, t, n, nt] { fn(t, n * t / nt, n * (t + 1) / nt); });
  fn(0, (size_t)0, n / nt);
  for (auto& x : th) x.join();
}

// ---- device-array cores shared with chain.hip (the resident chain from match codes to a bundle adjustment) ----
struct msfm_fransac_options;
int geo_fransac_dev(msfm_ctx* ctx, int n_pairs, const int* h_offsets, const int* d_off, const float* d1, const float* d2,
                    const msfm_fransac_options* opt, double* dF, uint8_t* d_in, int* d_nin, uint8_t* d_ok);
int geo_epipolar_batch_dev(msfm_ctx* ctx, int total, const int* d_pair_of, const float* d1, const float* d2, const double* dF,
                           const uint8_t* d_ok, double th, uint8_t* d_in);
struct msfm_track_dev {   // CSR tracks on the device: observations of a track in ascending image order
  DevBuf<int> off, img, feat;
  int n_tracks = 0, n_obs = 0;
};
// What chain.hip needs of a match result and its descriptor set (both defined in knn.hip)
struct msfm_match_result;
struct MatchView {
  msfm_ctx* ctx;
  int n_images, n_pairs;
  long total_q;
  const int* pairs;              // host [n_pairs][2]
  const int* out_off;            // host [n_pairs]: first code of the pair
  const int* nq;                 // host [n_pairs]: query features of the pair
  const int32_t* code;           // device [total_q]
  const int* n_all; const int* n_good;   // device [n_pairs]
  std::vector<int> count;        // features per image
  std::vector<const float*> kp;  // device [count][2] per image, nullptr: no keypoints uploaded
  bool slam;
};
int match_result_view(msfm_match_result* R, MatchView* out);   // MSFM_E_INVAL when the result is stale or orphaned

struct TrackPtrs {   // tri.hip: CSR tracks + cameras as the reference keeps them, device pointers
  int n_tracks;
  const int *off, *cam;
  const double *xy, *R, *t, *c, *fk;
};
int tri_midpoint_dev(msfm_ctx* ctx, const TrackPtrs& T, double th_error, double th_angle, double* dX, double* dmse, uint8_t* dok);
struct msfm_ba_problem;
struct msfm_ba;
// msfm_ba_create with the bulk arrays of the problem (obs_cam, obs_pt, obs_xy, point, pt_weight, pt_mutable) in DEVICE memory
int ba_create_impl(msfm_ctx* ctx, const msfm_ba_problem* P, bool bulk_on_device, msfm_ba** out);
int tracks_build_dev(msfm_ctx* ctx, int n_images, const std::vector<int>& feat_off, const int* d_nf, const int* d_fo, int n_pairs,
                     const int* d_pair, const int* d_moff, const int* d_match, int M, msfm_track_dev* out);

// Elimination structure of the reduced system (chol.hip): a nested-dissection tree of the camera graph laid out level
// by level.  Level 0 holds the leaf domains, level 1 the deepest separators, ... ; the nodes of one level are mutually
// uncoupled (their panel chains share launches), each is 64-aligned (identity padding inside) and couples only to its own
// descendants and to later columns.  Columns from the last level's b0 on (root separator + intrinsics) form the final
// dense chain.  n_levels = 0: plain dense order.  A node's descendants in a lower level are the nodes whose leaf interval
// lies inside its own (tree order inside every level makes them contiguous).
// corners: scratch for the deferred separator x separator updates, `nsplit` buffers of ldc x ldc doubles.
// bits of the device-side failure word of a solve: 1 = reduced system not positive definite, 2 = a 3x3 point block, 4 = non-finite step;
// MSFM_FAIL_SYNC = a bounded in-kernel wait ran out (the host turns it into MSFM_E_DEVICE)
#define MSFM_FAIL_SYNC (1 << 20)
#define MSFM_CORNER_MAX_BLOCKS 128   // 64-column blocks behind the leaf level that k_corner_syrk's range table holds
struct msfm_chol_node { int begin, end, leaf_lo, leaf_hi; };
struct msfm_chol_level { int K = 0; msfm_chol_node node[8]; int begin = 0, b0 = 0; };
struct msfm_chol_plan {
  int n_levels = 0;
  msfm_chol_level level[3];
  double* corners = nullptr;
  int ldc = 0;
};
struct msfm_chol_ws;   // hand-off state of the persistent panel chain (chol.hip): flags, hand-off buffers, ticket counters
int msfm_chol_ws_create(msfm_ctx* ctx, int npad, msfm_chol_ws** out);
void msfm_chol_ws_destroy(msfm_chol_ws* ws);
int msfm_chol_factor_solve(msfm_ctx* ctx, double* M, int npad, int n, double* work, double* w, double* z, int* fail,
                           const msfm_chol_plan* plan, double* z_next = nullptr, msfm_chol_ws* ws = nullptr);
int msfm_chol_fill_pending(msfm_ctx* ctx, double* z, int npad);   // "not solved yet" marks of k_backsolve_chain
