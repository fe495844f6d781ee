// C-ABI of the MI355X-native mgf hot path (include/mgf_hip.h): host orchestration of the
// hand-written kernels in kernels.h.  No CPU compute fallback exists: every compute entry point
// needs a HIP device and returns MGF_ERR_HIP without one.
#include <stdarg.h>

#include <algorithm>
#include <cmath>
#include <memory>
#include <utility>

#include "common.h"
#include "kernels.h"
#include "scene_io.h"

using namespace mgf;

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
namespace mgf {
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace mgf
extern "C" const char* mgf_last_error(void) { return g_err; }
extern "C" const char* mgf_version(void) { return "mgf-hip 0.1 (gfx950)"; }
extern "C" mgf_params mgf_default_params(void) {
  mgf_params p;
  p.baumgarte = 0.2f;               // solver.rs:278
  p.penetration_slop = 0.05f;       // solver.rs:277
  p.persistent_threshold_sq = 0.5f; // manifold.rs:38
  p.collision_epsilon = kCollisionEps;
  p.fat_margin = 0.25f;             // world.rs:181
  return p;
}

static inline unsigned nblk(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }
static mgf_status fail(mgf_status s, const char* msg) { set_error("%s", msg); return s; }
#ifdef MGF_DEBUG_LAUNCH  // (development build, tools/build_variant.sh: every launch waited for and named - the last line printed before a device fault is the launch in front of the faulting one)
#define LAUNCH_CHECK() do { MGF_HIP_TRY(hipGetLastError()); MGF_HIP_TRY(hipDeviceSynchronize()); fprintf(stderr, "[mgf] ok %s:%d\n", __FILE__, __LINE__); } while (0)
#else
#define LAUNCH_CHECK() MGF_HIP_TRY(hipGetLastError())
#endif

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
extern "C" mgf_status mgf_ctx_create(int device, mgf_ctx** out) {
  if (!out) return fail(MGF_ERR_INVALID, "out is NULL");
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0) {
    set_error("no HIP device available (%s); mgf-hip has no CPU fallback", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return MGF_ERR_HIP;
  }
  if (device < 0 || device >= count) return fail(MGF_ERR_INVALID, "device index out of range");
  MGF_HIP_TRY(hipSetDevice(device));
  std::unique_ptr<mgf_ctx> c(new mgf_ctx());
  c->device = device;
  MGF_HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  c->pinned_bytes = 1 << 16;
  MGF_HIP_TRY(hipHostMalloc(&c->pinned, c->pinned_bytes, hipHostMallocDefault));
  hipDeviceProp_t prop;
  MGF_HIP_TRY(hipGetDeviceProperties(&prop, device));
  c->num_cus = prop.multiProcessorCount;
  *out = c.release();
  return MGF_OK;
}
extern "C" mgf_status mgf_ctx_set_stream(mgf_ctx* ctx, void* stream) {
  if (!ctx) return fail(MGF_ERR_INVALID, "ctx is NULL");
  MGF_HIP_TRY(hipSetDevice(ctx->device));
  MGF_HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  ctx->stream = static_cast<hipStream_t>(stream);
  ctx->own_stream = false;
  return MGF_OK;
}
static void ctx_retain(mgf_ctx* ctx) { if (ctx) ++ctx->refs; }
static void ctx_release(mgf_ctx* ctx) {
  if (!ctx || --ctx->refs > 0) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  ctx->stream = nullptr;
  if (ctx->aux) { (void)hipStreamSynchronize(ctx->aux); (void)hipStreamDestroy(ctx->aux); ctx->aux = nullptr; }
  if (ctx->prim_tmp) (void)hipFree(ctx->prim_tmp);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  delete ctx;
}
// The creator's reference goes; handles made from the context that are still alive keep its streams and its struct until they are freed
// (tools/r06/exit_probe.py: an interpreter's finalisation freed a context ahead of its worlds - mgf_world_free then read a deleted struct).
extern "C" void mgf_ctx_destroy(mgf_ctx* ctx) {
  if (!ctx || ctx->closed) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  ctx->closed = true;
  if (!ctx->own_stream) ctx->stream = nullptr;  // (the caller's stream - mgf_ctx_set_stream - may be gone before the last handle is freed)
  ctx_release(ctx);
}

static mgf_status ctx_bind(mgf_ctx* ctx) {
  if (!ctx) return fail(MGF_ERR_HIP, "no device context: this entry point computes on the GPU (mgf-hip has no CPU fallback)");
  if (ctx->closed) return fail(MGF_ERR_INVALID, "the context was destroyed (mgf_ctx_destroy): its handles can only be freed");
  MGF_HIP_TRY(hipSetDevice(ctx->device));
  (void)hipGetLastError();  // drop any stale sticky error from unrelated earlier calls
  return MGF_OK;
}
template <class T>
static mgf_status h2d(mgf_ctx* ctx, T* dst, const T* src, size_t n) {
  if (n == 0) return MGF_OK;
  MGF_HIP_TRY(hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  MGF_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return MGF_OK;
}
template <class T>
static mgf_status d2h(mgf_ctx* ctx, T* dst, const T* src, size_t n) {
  if (n == 0) return MGF_OK;
  MGF_HIP_TRY(hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
  MGF_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return MGF_OK;
}

// append `add` behind the first old_n elements of a device array
template <class T>
static mgf_status append(mgf_ctx* ctx, DBuf<T>& buf, size_t old_n, const std::vector<T>& add) {
  MGF_TRY(buf.ensure(old_n + add.size(), ctx->stream, true, old_n));
  return h2d(ctx, buf.p + old_n, add.data(), add.size());
}

extern "C" mgf_status mgf_exclusive_scan_u32(mgf_ctx* ctx, const uint32_t* in, int64_t n, uint32_t* out) {
  MGF_TRY(ctx_bind(ctx));
  if (n < 0 || (n && (!in || !out))) return fail(MGF_ERR_INVALID, "bad argument");
  if (n == 0) return MGF_OK;
  DBuf<uint32_t> d_in, d_out;
  MGF_TRY(d_in.ensure((size_t)n, ctx->stream)); MGF_TRY(d_out.ensure((size_t)n, ctx->stream));
  MGF_TRY(h2d(ctx, d_in.p, in, (size_t)n));
  MGF_TRY(prim_exclusive_scan_u32(ctx, d_in.p, d_out.p, (size_t)n));
  return d2h(ctx, out, d_out.p, (size_t)n);
}

#include "host_trees.inc"
#include "host_geom_io.inc"
#include "host_single_shot.inc"
#include "host_world.inc"
#include "host_perm.inc"
#include "host_tick.inc"
#include "host_tiles.inc"
#include "host_solver.inc"
#include "host_boundary.inc"
#include "host_tiles_native.inc"
