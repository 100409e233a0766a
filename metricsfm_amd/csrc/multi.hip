// Single-process multi-GPU context (include/msfm.h, last section): one msfm_ctx per device, one host thread per device
// INSIDE the library, and the communicator between them.
//
// The reference's entry point is one process (SfM/test/test_sfm/test_sfm.cc:22-70) that calls ceres::Solve from a single
// thread (optimizer.cc:133).  The one-process-per-GPU form (msfm_ctx_init_rccl under torch.distributed.run) cannot be reached
// from there; this form can: the caller hands over the whole problem, the split happens here.
//  * distinct devices: RCCL's ncclCommInitAll (one communicator per device, xGMI between them); every rank's LM loop runs in
//    its own host thread and calls ncclAllReduce on its own communicator and stream (the hook of msfm_ctx_init_rccl);
//  * contexts that share ONE device (a one-GPU box: how this path is tested): an in-process reduction - the threads meet at
//    a host barrier, every rank sums its slice of the buffers of all ranks in rank order on the device, they meet again and
//    copy the result back.  RCCL refuses two ranks on one device, and needs none here.
// Splits: points (with their observations) for the bundle adjustment, tracks for triangulation / reprojection, the idx1-major
// pair list for matching - the partitions of metricsfm_amd/shard.py, restated.
#include "common.h"
#include <atomic>
#include <condition_variable>
#include <dlfcn.h>
#include <memory>
#include <mutex>
#include <thread>
#include "rccl_iface.h"

int msfm_ctx_adopt_rccl(msfm_ctx* ctx, void* lib, void* comm, int rank, int world);   // ctx.hip

namespace {
struct LocalComm {
  std::mutex m;
  std::condition_variable cv;
  int n = 0, arrived = 0;
  unsigned long gen = 0;
  bool aborted = false;
  std::vector<double*> buf;
  DevBuf<double> scratch;
};
// all ranks meet; false when a rank has given up (its error return must not leave the others waiting)
bool meet(LocalComm& L) {
  std::unique_lock<std::mutex> lk(L.m);
  if (L.aborted) return false;
  const unsigned long g = L.gen;
  if (++L.arrived == L.n) { L.arrived = 0; L.gen++; L.cv.notify_all(); return true; }
  L.cv.wait(lk, [&] { return L.gen != g || L.aborted; });
  return !L.aborted;
}
void give_up(LocalComm& L) {
  std::lock_guard<std::mutex> lk(L.m);
  L.aborted = true;
  L.cv.notify_all();
}
struct PtrTable { const double* p[16]; };
__global__ __launch_bounds__(256) void k_multi_reduce(PtrTable T, int n, size_t lo, size_t hi, int op, double* __restrict__ out) {
  const size_t i = lo + (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= hi) return;
  double s = T.p[0][i];
  for (int r = 1; r < n; r++) s = op == MSFM_REDUCE_MAX ? fmax(s, T.p[r][i]) : s + T.p[r][i];   // rank order: every rank gets the same bits
  out[i] = s;
}
}  // namespace

struct msfm_multi {
  int n = 0;
  std::vector<int> device;
  std::vector<msfm_ctx*> ctx;
  bool shared = false;          // every context on one device: in-process reduction
  LocalComm local;
  struct Hook { msfm_multi* mc; int rank; };
  std::vector<Hook> hooks;
  void* rccl_lib = nullptr;
  // distinct devices: the communicators (owned by the contexts) and ncclCommAbort.  A rank that fails while its peers are inside
  // ncclAllReduce - or behind it in a stream wait - would leave them there: the first failing rank aborts EVERY communicator
  // (the collective kernels of the peers end, their bounded waits see an error), the multi context is then broken for good.
  std::vector<void*> comms;
  msfm_rccl::comm_abort_t comm_abort = nullptr;
  std::mutex abort_m;
  bool broken = false;
  std::string err;
};

namespace {
int local_allreduce(void* user, double* buf, size_t count, int op, void* stream) {
  auto* h = static_cast<msfm_multi::Hook*>(user);
  msfm_multi* mc = h->mc;
  LocalComm& L = mc->local;
  const int r = h->rank, n = mc->n;
  hipStream_t s = (hipStream_t)stream;
  if (hipStreamSynchronize(s) != hipSuccess) { give_up(L); return 1; }   // this rank's partial is complete
  L.buf[r] = buf;
  if (r == 0 && L.scratch.n < count && L.scratch.alloc(std::max<size_t>(count, 1 << 16)) != hipSuccess) { give_up(L); return 2; }
  if (!meet(L)) return 3;
  PtrTable T;
  for (int q = 0; q < 16; q++) T.p[q] = q < n ? L.buf[q] : nullptr;
  const size_t lo = count * r / n, hi = count * (r + 1) / n;
  if (hi > lo) hipLaunchKernelGGL(k_multi_reduce, dim3(cdiv((long)(hi - lo), 256)), dim3(256), 0, s, T, n, lo, hi, op, L.scratch.p);
  if (hipStreamSynchronize(s) != hipSuccess) { give_up(L); return 4; }
  if (!meet(L)) return 5;
  if (hipMemcpyAsync(buf, L.scratch.p, sizeof(double) * count, hipMemcpyDeviceToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
    give_up(L);
    return 6;
  }
  if (!meet(L)) return 7;   // nobody writes the scratch of the next call before everybody has its copy
  return 0;
}

int fail(msfm_multi* mc, int code, const std::string& what) {
  mc->err = what;
  return code;
}

void abort_comms(msfm_multi* mc) {
  std::lock_guard<std::mutex> lk(mc->abort_m);
  if (mc->broken || mc->comms.empty()) return;
  mc->broken = true;
  for (void* c : mc->comms)
    if (c && mc->comm_abort) (void)mc->comm_abort(c);
}

// (test hook) MSFM_MULTI_FAIL_RANK=r: rank r of the next collective-bearing call returns an error before it joins anything
int injected_failure(msfm_multi* mc, int r) {
  const char* e = getenv("MSFM_MULTI_FAIL_RANK");
  if (!e || atoi(e) != r) return MSFM_OK;
  return msfm_set_error(mc->ctx[r], MSFM_E_INVAL, "injected failure (MSFM_MULTI_FAIL_RANK)");
}

// fn(rank) on one host thread per context; the first error wins, the other ranks are released from their barriers (shared
// device) or from their collectives (communicators aborted)
template <class F>
int on_all_ranks(msfm_multi* mc, F&& fn, bool collective = false) {
  if (mc->broken) return fail(mc, MSFM_E_DEVICE, "the communicators of this multi context were aborted after a rank failed; create a new one");
  {
    std::lock_guard<std::mutex> lk(mc->local.m);
    mc->local.aborted = false;
    mc->local.arrived = 0;
  }
  std::vector<int> rc(mc->n, MSFM_OK);
  std::vector<std::thread> th;
  std::atomic<int> first_failed{-1};   // the rank whose error came FIRST is the cause; the ranks it released fail after it
  for (int r = 0; r < mc->n; r++)
    th.emplace_back([&, r] {
      (void)hipSetDevice(mc->device[r]);
      rc[r] = fn(r);
      if (rc[r] != MSFM_OK) {
        int none = -1;
        first_failed.compare_exchange_strong(none, r);
        give_up(mc->local);
        if (collective && !mc->shared && mc->n > 1) abort_comms(mc);   // (calls without a collective leave the communicators alone)
      }
    });
  for (auto& t : th) t.join();
  if (mc->broken)   // ncclCommAbort has freed the communicators: the contexts must not destroy them again
    for (msfm_ctx* c : mc->ctx) msfm_ctx_forget_rccl(c);
  const int r = first_failed.load();
  if (r >= 0) return fail(mc, rc[r], "rank " + std::to_string(r) + ": " + msfm_last_error(mc->ctx[r]));
  return MSFM_OK;
}

// contiguous ranges [cut[r], cut[r+1]) of n items whose cumulative cost is cum[0..n)
std::vector<int> cuts_by_cost(const std::vector<double>& cum, int n, int world) {
  std::vector<int> cut(world + 1, 0);
  const double total = n ? cum[n - 1] : 0.0;
  for (int r = 1; r < world; r++) cut[r] = (int)(std::lower_bound(cum.begin(), cum.begin() + n, total * r / world) - cum.begin());
  cut[world] = n;
  for (int r = 1; r <= world; r++) cut[r] = std::min(n, std::max(cut[r], cut[r - 1]));
  return cut;
}
}  // namespace

MSFM_API int msfm_ctx_create_multi(int n_gpus, const int* devices, msfm_multi** out) {
  if (!out || n_gpus < 1 || n_gpus > 16) return MSFM_E_INVAL;
  *out = nullptr;
  std::unique_ptr<msfm_multi> mc(new msfm_multi());
  mc->n = n_gpus;
  for (int r = 0; r < n_gpus; r++) mc->device.push_back(devices ? devices[r] : r);
  bool all_same = true, all_distinct = true;
  for (int r = 0; r < n_gpus; r++)
    for (int q = 0; q < r; q++) {
      if (mc->device[r] != mc->device[q]) all_same = false; else all_distinct = false;
    }
  if (n_gpus > 1 && !all_same && !all_distinct) return MSFM_E_INVAL;   // either one context per device, or all of them on one
  mc->shared = n_gpus > 1 && all_same;
  auto cleanup = [&]() { for (msfm_ctx* c : mc->ctx) msfm_ctx_destroy(c); mc->ctx.clear(); };
  for (int r = 0; r < n_gpus; r++) {
    msfm_ctx* c = nullptr;
    const int rc = msfm_ctx_create(mc->device[r], &c);
    if (rc != MSFM_OK) { cleanup(); return rc; }
    c->device_share = all_same ? n_gpus : 1;
    mc->ctx.push_back(c);
  }
  mc->hooks.resize(n_gpus);
  mc->local.n = n_gpus;
  mc->local.buf.assign(n_gpus, nullptr);
  if (n_gpus == 1) { *out = mc.release(); return MSFM_OK; }
  if (mc->shared) {
    for (int r = 0; r < n_gpus; r++) {
      mc->hooks[r] = msfm_multi::Hook{mc.get(), r};
      const int rc = msfm_ctx_set_allreduce(mc->ctx[r], local_allreduce, &mc->hooks[r], r, n_gpus);
      if (rc != MSFM_OK) { cleanup(); return rc; }
    }
  } else {
    // one communicator per device, created together (ncclCommInitAll: the single-process form of RCCL)
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      mc->rccl_lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (mc->rccl_lib) break;
    }
    auto init_all = mc->rccl_lib ? reinterpret_cast<msfm_rccl::comm_init_all_t>(dlsym(mc->rccl_lib, "ncclCommInitAll")) : nullptr;
    if (!init_all) { cleanup(); return MSFM_E_DEVICE; }
    auto comm_destroy = reinterpret_cast<msfm_rccl::comm_destroy_t>(dlsym(mc->rccl_lib, "ncclCommDestroy"));
    mc->comm_abort = reinterpret_cast<msfm_rccl::comm_abort_t>(dlsym(mc->rccl_lib, "ncclCommAbort"));
    if (!comm_destroy || !mc->comm_abort) { cleanup(); return MSFM_E_DEVICE; }
    std::vector<void*> comms(n_gpus, nullptr);
    if (init_all(comms.data(), n_gpus, mc->device.data()) != msfm_rccl::SUCCESS) { cleanup(); return MSFM_E_DEVICE; }
    for (int r = 0; r < n_gpus; r++) {
      const int rc = msfm_ctx_adopt_rccl(mc->ctx[r], mc->rccl_lib, comms[r], r, n_gpus);   // the context owns (and destroys) its communicator
      if (rc != MSFM_OK) {
        for (int q = r; q < n_gpus; q++) (void)comm_destroy(comms[q]);   // the ones no context has taken over
        cleanup();
        return rc;
      }
    }
    mc->comms = comms;
  }
  *out = mc.release();
  return MSFM_OK;
}

MSFM_API void msfm_multi_destroy(msfm_multi* mc) {
  if (!mc) return;
  for (size_t r = 0; r < mc->ctx.size(); r++) {
    (void)hipSetDevice(mc->device[r]);
    msfm_ctx_destroy(mc->ctx[r]);
  }
  if (!mc->device.empty()) (void)hipSetDevice(mc->device[0]);
  mc->local.scratch.release();
  delete mc;
}

MSFM_API int msfm_multi_size(const msfm_multi* mc) { return mc ? mc->n : 0; }
MSFM_API msfm_ctx* msfm_multi_ctx(msfm_multi* mc, int rank) { return (mc && rank >= 0 && rank < mc->n) ? mc->ctx[rank] : nullptr; }
MSFM_API const char* msfm_multi_last_error(const msfm_multi* mc) { return mc ? mc->err.c_str() : "null multi context"; }

// ---- bundle adjustment: points split over the contexts (shard.point_ranges restated) ----
MSFM_API int msfm_multi_ba_solve(msfm_multi* mc, msfm_ba_problem* P, const msfm_ba_options* opt, msfm_ba_summary* summary) {
  if (!mc || !P || !opt || !summary) return MSFM_E_INVAL;
  if (mc->n == 1) {
    const int rc = msfm_ba_solve(mc->ctx[0], P, opt, summary);
    return rc == MSFM_OK ? rc : fail(mc, rc, msfm_last_error(mc->ctx[0]));
  }
  const int Np = P->n_points, No = P->n_obs, W = mc->n;
  if (Np < 0 || No < 0 || (No > 0 && (!P->obs_pt || !P->obs_cam))) return fail(mc, MSFM_E_INVAL, "msfm_multi_ba_solve: bad problem");
  for (int o = 0; o < No; o++)
    if (P->obs_pt[o] < 0 || P->obs_pt[o] >= Np || (o > 0 && P->obs_pt[o] < P->obs_pt[o - 1]))
      return fail(mc, MSFM_E_INVAL, "msfm_multi_ba_solve: obs_pt must be non-decreasing and inside [0, n_points)");
  // cost of a point: what the masks leave of it (PartialBundleAdjustment, sfm_incremental.cc:917-945): a free point the
  // square of its free-camera rows plus its rows, a frozen point only its free-camera rows
  std::vector<long> k(Np, 0), kc(Np, 0);
  std::vector<int> first(Np + 1, 0);
  for (int o = 0; o < No; o++) {
    k[P->obs_pt[o]]++;
    if (!P->cam_mutable || P->cam_mutable[P->obs_cam[o]]) kc[P->obs_pt[o]]++;
  }
  for (int p = 0; p < Np; p++) first[p + 1] = first[p] + (int)k[p];
  std::vector<double> cum(std::max(1, Np), 0.0);
  double acc = 0;
  for (int p = 0; p < Np; p++) {
    const bool pm = !P->pt_mutable || P->pt_mutable[p];
    acc += pm ? (double)(kc[p] * kc[p] + 4 * k[p]) : (double)(4 * kc[p]);
    cum[p] = acc;
  }
  const std::vector<int> cut = cuts_by_cost(cum, Np, W);
  // every rank solves with its own copy of the camera / intrinsics arrays (all end identical; rank 0 works on the caller's)
  std::vector<std::vector<double>> cam(W), model(W);
  std::vector<std::vector<int32_t>> opt_(W);
  std::vector<std::vector<msfm_ba_iteration>> its(W);
  std::vector<msfm_ba_summary> sums(W);
  const int rc = on_all_ranks(mc, [&](int r) -> int {
    const int lo = cut[r], hi = cut[r + 1], o0 = first[lo], o1 = first[hi];
    msfm_ba_problem S = *P;
    if (r > 0) {
      cam[r].assign(P->cam_pose, P->cam_pose + 6 * (size_t)P->n_cams);
      model[r].assign(P->cam_model, P->cam_model + 3 * (size_t)P->n_models);
      S.cam_pose = cam[r].data(); S.cam_model = model[r].data();
    }
    opt_[r].resize(std::max(1, o1 - o0));
    for (int o = o0; o < o1; o++) opt_[r][o - o0] = P->obs_pt[o] - lo;
    S.n_points = hi - lo; S.n_obs = o1 - o0;
    S.point = P->point ? P->point + 3 * (size_t)lo : nullptr;
    S.obs_cam = P->obs_cam + o0; S.obs_pt = opt_[r].data(); S.obs_xy = P->obs_xy + 2 * (size_t)o0;
    S.pt_weight = P->pt_weight ? P->pt_weight + lo : nullptr;
    S.pt_mutable = P->pt_mutable ? P->pt_mutable + lo : nullptr;
    msfm_ba_summary* sm = &sums[r];
    if (r == 0) sm = summary;
    else {
      *sm = msfm_ba_summary();
      its[r].assign((size_t)std::max(1, summary->iterations_capacity), msfm_ba_iteration());
      sm->iterations = summary->iterations ? its[r].data() : nullptr;
      sm->iterations_capacity = summary->iterations ? summary->iterations_capacity : 0;
    }
    MSFM_TRY(injected_failure(mc, r));
    return msfm_ba_solve(mc->ctx[r], &S, opt, sm);
  }, true);
  return rc;
}

// ---- triangulation / reprojection: tracks split by observation count (shard.track_ranges restated) ----
namespace {
template <class Call>
int split_tracks(msfm_multi* mc, const msfm_tracks* T, Call&& call) {
  if (!mc || !T || T->n_tracks < 0 || !T->track_off) return MSFM_E_INVAL;
  const int n = T->n_tracks, W = mc->n;
  if (W == 1 || n == 0) {
    const int rc = call(0, *T, 0);
    return rc == MSFM_OK ? rc : fail(mc, rc, msfm_last_error(mc->ctx[0]));
  }
  std::vector<double> cum(n);
  for (int t = 0; t < n; t++) cum[t] = (double)(T->track_off[t + 1] - T->track_off[0]);
  const std::vector<int> cut = cuts_by_cost(cum, n, W);
  std::vector<std::vector<int32_t>> off(W);
  return on_all_ranks(mc, [&](int r) -> int {
    const int lo = cut[r], hi = cut[r + 1];
    if (hi == lo) return MSFM_OK;
    const int o0 = T->track_off[lo];
    off[r].resize(hi - lo + 1);
    for (int t = lo; t <= hi; t++) off[r][t - lo] = T->track_off[t] - o0;
    msfm_tracks S = *T;
    S.n_tracks = hi - lo; S.track_off = off[r].data(); S.track_cam = T->track_cam + o0; S.track_xy = T->track_xy + 2 * (size_t)o0;
    return call(r, S, lo);
  });
}
}  // namespace

MSFM_API int msfm_multi_triangulate_midpoint_batch(msfm_multi* mc, const msfm_tracks* T, double th_error, double th_angle, double* X, double* mse,
                                                    uint8_t* ok) {
  return split_tracks(mc, T, [&](int r, const msfm_tracks& S, int lo) {
    return msfm_triangulate_midpoint_batch(mc->ctx[r], &S, th_error, th_angle, X + 3 * (size_t)lo, mse + lo, ok + lo);
  });
}
MSFM_API int msfm_multi_triangulate_dlt_batch(msfm_multi* mc, const msfm_tracks* T, double th_error, double th_angle, double* X, double* mse,
                                               uint8_t* ok) {
  return split_tracks(mc, T, [&](int r, const msfm_tracks& S, int lo) {
    return msfm_triangulate_dlt_batch(mc->ctx[r], &S, th_error, th_angle, X + 3 * (size_t)lo, mse + lo, ok + lo);
  });
}
MSFM_API int msfm_multi_reproject_mse_batch(msfm_multi* mc, const msfm_tracks* T, const double* X, double* mse) {
  return split_tracks(mc, T, [&](int r, const msfm_tracks& S, int lo) { return msfm_reproject_mse_batch(mc->ctx[r], &S, X + 3 * (size_t)lo, mse + lo); });
}

// ---- matching: the idx1-major pair list in contiguous slices balanced by M1 * M2 (shard.shard_pairs restated) ----
MSFM_API int msfm_multi_match_pairs(msfm_multi* mc, int n_images, const float* const* desc, const int* count, int dim, const int* pairs, int n_pairs,
                                    float ratio_good, float ratio_all, int32_t* const* code, int* n_all, int* n_good) {
  if (!mc || n_images < 1 || !desc || !count || n_pairs < 0 || (n_pairs > 0 && (!pairs || !code))) return MSFM_E_INVAL;
  for (int p = 0; p < n_pairs; p++)
    if (pairs[2 * p] < 0 || pairs[2 * p] >= n_images || pairs[2 * p + 1] < 0 || pairs[2 * p + 1] >= n_images)
      return fail(mc, MSFM_E_INVAL, "msfm_multi_match_pairs: image index out of range");
  std::vector<double> cum(std::max(1, n_pairs));
  double acc = 0;
  for (int p = 0; p < n_pairs; p++) { acc += (double)count[pairs[2 * p]] * count[pairs[2 * p + 1]] + 1.0; cum[p] = acc; }
  const std::vector<int> cut = cuts_by_cost(cum, n_pairs, mc->n);
  return on_all_ranks(mc, [&](int r) -> int {
    const int lo = cut[r], hi = cut[r + 1];
    if (hi == lo) return MSFM_OK;
    msfm_ctx* ctx = mc->ctx[r];
    msfm_descset* set = nullptr;
    MSFM_TRY(msfm_descset_create(ctx, n_images, dim, &set));
    std::vector<char> used(n_images, 0);
    for (int p = lo; p < hi; p++) { used[pairs[2 * p]] = 1; used[pairs[2 * p + 1]] = 1; }
    int rc = MSFM_OK;
    for (int i = 0; i < n_images && rc == MSFM_OK; i++)
      if (used[i]) rc = msfm_descset_upload(set, i, desc[i], count[i]);   // only the images this slice touches
    msfm_match_result* res = nullptr;
    if (rc == MSFM_OK) rc = msfm_match_pairs(set, pairs + 2 * (size_t)lo, hi - lo, ratio_good, ratio_all, 0, &res);
    if (rc == MSFM_OK && (n_all || n_good)) rc = msfm_match_result_counts(res, n_all ? n_all + lo : nullptr, n_good ? n_good + lo : nullptr);
    for (int p = lo; p < hi && rc == MSFM_OK; p++)
      if (count[pairs[2 * p + 1]] > 0) rc = msfm_match_result_fetch(res, p - lo, code[p], nullptr, nullptr);
    msfm_match_result_destroy(res);
    msfm_descset_destroy(set);
    return rc;
  });
}
