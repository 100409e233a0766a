// Context, error reporting and per-kernel-class timing for libmsfm.
#include "common.h"

int msfm_set_error(msfm_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}

MSFM_API int msfm_version(void) { return MSFM_VERSION; }

MSFM_API int msfm_ctx_create(int device, msfm_ctx** out) {
  if (!out) return MSFM_E_INVAL;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return MSFM_E_DEVICE;  // no CPU fallback
  msfm_ctx* ctx = new msfm_ctx();
  if (device < 0) {
    if (hipGetDevice(&device) != hipSuccess) device = 0;
  }
  if (device >= n || hipSetDevice(device) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return MSFM_E_DEVICE;
  }
  ctx->device = device;
  {
    const char* e = getenv("MSFM_DEVICE_SHARE");   // several processes on this device (the multi-rank tests on a one-GPU box)
    ctx->device_share = e ? std::max(1, atoi(e)) : 1;
  }
  *out = ctx;
  return MSFM_OK;
}

// ---- cache of freed device blocks (see common.h) ----
// The pool itself records the capacity and the owning device of every block it has handed out (one map lookup per
// alloc / free, ~60 per bundle adjustment): the value a caller passes to msfm_pool_free is only cross-checked, never
// trusted, and a block always returns to the free list of the device it was allocated on, whatever device the freeing
// thread has current.  (A capacity mix-up in a buffer swap once put a 32 KB block on the 64 KB list; the next kernel
// that received it wrote out of bounds.)
#include <map>
#include <mutex>
namespace {
struct Pool {
  std::mutex mu;
  struct Live { size_t capacity; int device; };
  std::map<void*, Live> live;   // blocks handed out
  std::multimap<std::pair<int, size_t>, void*> free_blocks;  // (device, capacity) -> block
  size_t cached_bytes = 0;
  size_t mismatches = 0;
  size_t limit() {
    static const size_t v = [] { const char* e = getenv("MSFM_POOL_MB"); return (size_t)(e ? atol(e) : 16384) << 20; }();
    return v;
  }
  bool debug() {
    static const bool v = getenv("MSFM_POOL_DEBUG") != nullptr;
    return v;
  }
};
Pool& pool() { static Pool* p = new Pool(); return *p; }  // intentionally never destroyed (no hipFree at process exit)
size_t round_capacity(size_t bytes) {
  if (bytes <= 4096) return 4096;
  if (bytes <= ((size_t)2 << 20)) { size_t c = 4096; while (c < bytes) c <<= 1; return c; }   // powers of two up to 2 MB
  return (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);                            // then multiples of 2 MB
}
}  // namespace

hipError_t msfm_pool_alloc(void** p, size_t bytes, size_t* capacity) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const size_t want = round_capacity(bytes);
  Pool& P = pool();
  {
    std::lock_guard<std::mutex> g(P.mu);
    // the smallest cached block that fits and wastes at most half of itself
    auto it = P.free_blocks.lower_bound({dev, want});
    if (it != P.free_blocks.end() && it->first.first == dev && it->first.second <= 2 * want) {
      *p = it->second;
      *capacity = it->first.second;
      P.cached_bytes -= it->first.second;
      P.free_blocks.erase(it);
      if (P.live.count(*p)) { fprintf(stderr, "msfm pool: block %p handed out twice\n", *p); abort(); }
      P.live[*p] = Pool::Live{*capacity, dev};
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(p, want);
  if (e != hipSuccess) {  // give the cache back to the driver and try once more
    msfm_pool_trim(dev);
    (void)hipGetLastError();
    e = hipMalloc(p, want);
  }
  *capacity = want;
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> g(P.mu);
    if (P.live.count(*p)) { fprintf(stderr, "msfm pool: hipMalloc returned a live block %p\n", *p); abort(); }
    if (P.debug())
      for (auto& kv : P.free_blocks) if (kv.second == *p) { fprintf(stderr, "msfm pool: hipMalloc returned a cached block %p\n", *p); abort(); }
    P.live[*p] = Pool::Live{want, dev};
  }
  return e;
}

void msfm_pool_free(void* p, size_t capacity) {
  if (!p) return;
  Pool& P = pool();
  int dev = 0;
  {
    std::lock_guard<std::mutex> g(P.mu);
    auto it = P.live.find(p);
    if (it == P.live.end()) {
      // not one of ours (or freed twice): never cache it
      fprintf(stderr, "msfm pool: free of unknown block %p (capacity %zu)\n", p, capacity);
      if (P.debug()) abort();
      return;
    }
    if (it->second.capacity != capacity) {
      P.mismatches++;
      fprintf(stderr, "msfm pool: block %p freed with capacity %zu, allocated with %zu\n", p, capacity, it->second.capacity);
      if (P.debug()) abort();
    }
    capacity = it->second.capacity;   // the recorded values win
    dev = it->second.device;
    P.live.erase(it);
    if (capacity > 0 && P.cached_bytes + capacity <= P.limit()) {
      P.free_blocks.insert({{dev, capacity}, p});
      P.cached_bytes += capacity;
      return;
    }
  }
  int cur = 0;
  (void)hipGetDevice(&cur);
  if (cur != dev) (void)hipSetDevice(dev);
  (void)hipFree(p);
  if (cur != dev) (void)hipSetDevice(cur);
}

void msfm_pool_trim(int device) {
  Pool& P = pool();
  std::vector<void*> drop;
  {
    std::lock_guard<std::mutex> g(P.mu);
    for (auto it = P.free_blocks.begin(); it != P.free_blocks.end();) {
      if (it->first.first == device) { drop.push_back(it->second); P.cached_bytes -= it->first.second; it = P.free_blocks.erase(it); }
      else ++it;
    }
  }
  int cur = 0;
  (void)hipGetDevice(&cur);
  if (cur != device) (void)hipSetDevice(device);
  for (void* q : drop) (void)hipFree(q);
  if (cur != device) (void)hipSetDevice(cur);
}

MSFM_API void msfm_ctx_destroy(msfm_ctx* ctx) {
  if (!ctx) return;
  if (ctx->children > 0) {
    // descriptor sets, match results and resident problems hold ctx->stream: destroying the context under them would
    // leave their destroy functions with a dangling pointer.  Keep the context (a leak, reported) instead.
    if (getenv("MSFM_VERBOSE")) fprintf(stderr, "msfm_ctx_destroy: %d object(s) created from this context are still alive; context kept until they go\n", ctx->children);
    ctx->orphaned = true;
    return;
  }
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->rccl_comm && ctx->rccl_comm_destroy) {
    (void)reinterpret_cast<int (*)(void*)>(ctx->rccl_comm_destroy)(ctx->rccl_comm);
    ctx->rccl_comm = nullptr;
  }
  msfm_pool_trim(ctx->device);
  if (ctx->ba_scratch && ctx->ba_scratch_free) ctx->ba_scratch_free(ctx->ba_scratch);
  for (auto& p : ctx->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
  if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);   // before its events go
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

void msfm_ctx_child_released(msfm_ctx* ctx) {
  if (!ctx) return;
  ctx->children--;
  if (ctx->children == 0 && ctx->orphaned) { ctx->orphaned = false; msfm_ctx_destroy(ctx); }
}

MSFM_API const char* msfm_last_error(const msfm_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }
MSFM_API void* msfm_ctx_stream(msfm_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

MSFM_API int msfm_ctx_synchronize(msfm_ctx* ctx) {
  if (!ctx) return MSFM_E_INVAL;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MSFM_OK;
}

MSFM_API int msfm_ctx_set_allreduce(msfm_ctx* ctx, msfm_allreduce_fn fn, void* user, int rank, int world_size) {
  if (!ctx || world_size < 1 || rank < 0 || rank >= world_size) return MSFM_E_INVAL;
  if (world_size > 1 && !fn) return msfm_set_error(ctx, MSFM_E_INVAL, "world_size > 1 needs an all-reduce hook");
  ctx->allreduce = fn;
  ctx->allreduce_user = user;
  ctx->rank = rank;
  ctx->world = world_size;
  return MSFM_OK;
}

// ---- native RCCL collective ------------------------------------------------------------
// rccl.h is not included on purpose (the library must build and load where RCCL is absent); the few types used are declared
// in rccl_iface.h and checked against the real header by a compile-time test.
#include <chrono>
#include <dlfcn.h>
#include <thread>
#include "rccl_iface.h"
namespace {
using RcclId = msfm_rccl::UniqueId;
typedef msfm_rccl::get_unique_id_t rccl_get_unique_id_t;
typedef msfm_rccl::comm_init_rank_t rccl_comm_init_rank_t;
typedef msfm_rccl::all_reduce_t rccl_all_reduce_t;
typedef msfm_rccl::comm_destroy_t rccl_comm_destroy_t;
typedef msfm_rccl::error_string_t rccl_error_string_t;
enum { RCCL_SUM = msfm_rccl::SUM, RCCL_MAX = msfm_rccl::MAX, RCCL_FLOAT64 = msfm_rccl::FLOAT64 };

void* rccl_open(msfm_ctx* ctx) {
  if (ctx->rccl_lib) return ctx->rccl_lib;
  // a copy already mapped into the process (torch ships one) is found first by its soname
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    ctx->rccl_lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (ctx->rccl_lib) break;
  }
  return ctx->rccl_lib;
}

int rccl_hook(void* user, double* buf, size_t count, int op, void* stream) {
  msfm_ctx* ctx = static_cast<msfm_ctx*>(user);
  const int rc = reinterpret_cast<rccl_all_reduce_t>(ctx->rccl_allreduce)(buf, buf, count, RCCL_FLOAT64, op == MSFM_REDUCE_MAX ? RCCL_MAX : RCCL_SUM,
                                                                           ctx->rccl_comm, stream);
  return rc;   // ncclSuccess = 0
}
}  // namespace

MSFM_API int msfm_rccl_get_unique_id(msfm_ctx* ctx, unsigned char id[MSFM_RCCL_ID_BYTES]) {
  if (!ctx || !id) return MSFM_E_INVAL;
  if (!rccl_open(ctx)) return msfm_set_error(ctx, MSFM_E_DEVICE, "librccl not found: %s", dlerror());
  auto fn = reinterpret_cast<rccl_get_unique_id_t>(dlsym(ctx->rccl_lib, "ncclGetUniqueId"));
  if (!fn) return msfm_set_error(ctx, MSFM_E_DEVICE, "ncclGetUniqueId not found in librccl");
  RcclId u;
  const int rc = fn(&u);
  if (rc != 0) return msfm_set_error(ctx, MSFM_E_DEVICE, "ncclGetUniqueId failed: %d", rc);
  memcpy(id, u.internal, MSFM_RCCL_ID_BYTES);
  return MSFM_OK;
}

MSFM_API int msfm_ctx_init_rccl(msfm_ctx* ctx, const unsigned char id[MSFM_RCCL_ID_BYTES], int rank, int world_size) {
  if (!ctx || !id || world_size < 1 || rank < 0 || rank >= world_size) return MSFM_E_INVAL;
  if (ctx->rccl_comm) return msfm_set_error(ctx, MSFM_E_INVAL, "this context already owns a communicator");
  if (!rccl_open(ctx)) return msfm_set_error(ctx, MSFM_E_DEVICE, "librccl not found: %s", dlerror());
  auto init = reinterpret_cast<rccl_comm_init_rank_t>(dlsym(ctx->rccl_lib, "ncclCommInitRank"));
  ctx->rccl_allreduce = dlsym(ctx->rccl_lib, "ncclAllReduce");
  ctx->rccl_comm_destroy = dlsym(ctx->rccl_lib, "ncclCommDestroy");
  ctx->rccl_error_string = dlsym(ctx->rccl_lib, "ncclGetErrorString");
  if (!init || !ctx->rccl_allreduce || !ctx->rccl_comm_destroy) return msfm_set_error(ctx, MSFM_E_DEVICE, "librccl lacks an entry point");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  RcclId u;
  memcpy(u.internal, id, MSFM_RCCL_ID_BYTES);
  void* comm = nullptr;
  const int rc = init(&comm, world_size, u, rank);
  if (rc != 0) {
    const char* txt = ctx->rccl_error_string ? reinterpret_cast<rccl_error_string_t>(ctx->rccl_error_string)(rc) : "";
    return msfm_set_error(ctx, MSFM_E_DEVICE, "ncclCommInitRank failed: %d %s", rc, txt);
  }
  ctx->rccl_comm = comm;
  return msfm_ctx_set_allreduce(ctx, rccl_hook, ctx, rank, world_size);
}

// A communicator created elsewhere (msfm_ctx_create_multi: ncclCommInitAll makes one per device in a single call): the
// context takes it over - ncclAllReduce on its stream becomes the reduction hook, msfm_ctx_destroy releases it.
int msfm_ctx_adopt_rccl(msfm_ctx* ctx, void* lib, void* comm, int rank, int world) {
  if (!ctx || !lib || !comm) return MSFM_E_INVAL;
  if (ctx->rccl_comm) return msfm_set_error(ctx, MSFM_E_INVAL, "this context already owns a communicator");
  ctx->rccl_lib = lib;
  ctx->rccl_allreduce = dlsym(lib, "ncclAllReduce");
  ctx->rccl_comm_destroy = dlsym(lib, "ncclCommDestroy");
  ctx->rccl_error_string = dlsym(lib, "ncclGetErrorString");
  if (!ctx->rccl_allreduce || !ctx->rccl_comm_destroy) return msfm_set_error(ctx, MSFM_E_DEVICE, "librccl lacks an entry point");
  ctx->rccl_comm = comm;
  return msfm_ctx_set_allreduce(ctx, rccl_hook, ctx, rank, world);
}

void msfm_ctx_forget_rccl(msfm_ctx* ctx) {
  if (!ctx) return;
  ctx->rccl_comm = nullptr;
  ctx->allreduce = nullptr;
  ctx->allreduce_user = nullptr;
}

int msfm_stream_wait_bounded(msfm_ctx* ctx, hipStream_t s, const char* what) {
  static const double limit_s = [] { const char* e = getenv("MSFM_SYNC_TIMEOUT_S"); const double v = e ? atof(e) : 120.0; return v > 0 ? v : 120.0; }();
  const auto t0 = std::chrono::steady_clock::now();
  unsigned long polls = 0;
  for (;;) {
    const hipError_t q = hipStreamQuery(s);
    if (q == hipSuccess) return MSFM_OK;
    if (q != hipErrorNotReady) HIP_TRY(ctx, q);
    if ((++polls & 0xfful) == 0) {
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (dt > limit_s)
        return msfm_set_error(ctx, MSFM_E_DEVICE, "%s did not complete within %.0f s (MSFM_SYNC_TIMEOUT_S): a peer rank may have left", what, limit_s);
      if (dt > 2e-3) std::this_thread::sleep_for(std::chrono::microseconds(200));   // a short wait spins, a long one does not burn a core
    }
  }
}

MSFM_API int msfm_ctx_allreduce(msfm_ctx* ctx, double* buf_dev, size_t count, int op) {
  if (!ctx || (count > 0 && !buf_dev) || (op != MSFM_REDUCE_SUM && op != MSFM_REDUCE_MAX)) return MSFM_E_INVAL;
  if (!ctx->allreduce) return ctx->world <= 1 ? MSFM_OK : msfm_set_error(ctx, MSFM_E_INVAL, "no collective installed");
  if (count == 0) return MSFM_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int rc = ctx->allreduce(ctx->allreduce_user, buf_dev, count, op, (void*)ctx->stream);
  if (rc != 0) return msfm_set_error(ctx, MSFM_E_DEVICE, "all-reduce failed: %d", rc);
  return MSFM_OK;
}

// ---- profiling -------------------------------------------------------------------------
static hipEvent_t get_event(msfm_ctx* ctx) {
  if (!ctx->event_pool.empty()) {
    hipEvent_t e = ctx->event_pool.back();
    ctx->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

KTimer::KTimer(msfm_ctx* c, const char* name) : ctx(c) {
  if (!ctx || !ctx->profile) return;
  for (size_t i = 0; i < ctx->stats.size(); i++)
    if (ctx->stats[i].name == name) { idx = (int)i; break; }
  if (idx < 0) {
    if ((int)ctx->stats.size() >= MSFM_MAX_KERNEL_STATS) return;
    msfm_ctx::Stat s;
    s.name = name;
    ctx->stats.push_back(s);
    idx = (int)ctx->stats.size() - 1;
  }
  a = get_event(ctx);
  b = get_event(ctx);
  (void)hipEventRecord(a, ctx->stream);
}

void KTimer::stop() {
  if (idx < 0) return;
  (void)hipEventRecord(b, ctx->stream);
  ctx->stats[idx].launches += count;
  ctx->pending.push_back({idx, a, b});
  idx = -1;
}

KTimer::~KTimer() { stop(); }

static void resolve_pending(msfm_ctx* ctx) {
  if (ctx->pending.empty()) return;
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& p : ctx->pending) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) ctx->stats[p.stat].ms += ms;
    ctx->event_pool.push_back(p.a);
    ctx->event_pool.push_back(p.b);
  }
  ctx->pending.clear();
}

MSFM_API int msfm_ctx_profile_enable(msfm_ctx* ctx, int enable) {
  if (!ctx) return MSFM_E_INVAL;
  resolve_pending(ctx);
  ctx->profile = enable != 0;
  return MSFM_OK;
}

MSFM_API int msfm_ctx_profile_reset(msfm_ctx* ctx) {
  if (!ctx) return MSFM_E_INVAL;
  resolve_pending(ctx);
  ctx->stats.clear();
  return MSFM_OK;
}

MSFM_API int msfm_ctx_profile_get(msfm_ctx* ctx, msfm_kernel_stat* stats, int cap, int* n_out) {
  if (!ctx || !n_out) return MSFM_E_INVAL;
  resolve_pending(ctx);
  int n = 0;
  for (auto& s : ctx->stats) {
    if (n >= cap) break;
    memset(&stats[n], 0, sizeof stats[n]);
    strncpy(stats[n].name, s.name.c_str(), sizeof(stats[n].name) - 1);
    stats[n].launches = s.launches;
    stats[n].total_ms = s.ms;
    n++;
  }
  *n_out = n;
  return MSFM_OK;
}
