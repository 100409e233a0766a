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
  *out = ctx;
  return MSFM_OK;
}

MSFM_API void msfm_ctx_destroy(msfm_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& p : ctx->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
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

KTimer::~KTimer() {
  if (idx < 0) return;
  (void)hipEventRecord(b, ctx->stream);
  ctx->stats[idx].launches++;
  ctx->pending.push_back({idx, a, b});
}

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
