// Host-side plumbing shared by the C-ABI translation units: error handling, the per-GPU
// context (device + stream), and growable device buffers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/mgf_hip.h"

namespace mgf {

void set_error(const char* fmt, ...);  // thread-local message behind mgf_last_error()

#define MGF_HIP_TRY(expr)                                                                  \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      mgf::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return (_e == hipErrorOutOfMemory) ? MGF_ERR_OOM : MGF_ERR_HIP;                      \
    }                                                                                      \
  } while (0)

#define MGF_TRY(expr)                 \
  do {                                \
    mgf_status _s = (expr);           \
    if (_s != MGF_OK) return _s;      \
  } while (0)

}  // namespace mgf

struct mgf_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = true;    // false after mgf_ctx_set_stream: the caller's stream
  void* prim_tmp = nullptr;  // rocPRIM temporary storage
  size_t prim_tmp_bytes = 0;
  void* pinned = nullptr;    // small pinned staging area for read-backs
  size_t pinned_bytes = 0;
  int num_cus = 256;
  // (r06) handles made from a context keep it alive: mgf_ctx_destroy marks it closed and drops the creator's reference, the streams and the struct go
  // with the last handle - whatever order a garbage collector (or a Rust scope) frees them in.  Calls on a closed context fail with MGF_ERR_INVALID.
  int refs = 1;
  bool closed = false;
  hipStream_t aux = nullptr; // a second stream for work that runs BESIDE the tick's main chain of launches (r06: the terrain kernels of a capsule world beside its pair search), created on first use
};

namespace mgf {

// Growable device array.  grow() keeps contents when keep = true.
template <class T>
struct DBuf {
  T* p = nullptr;
  size_t cap = 0;
  DBuf() = default;
  DBuf(const DBuf&) = delete;
  DBuf& operator=(const DBuf&) = delete;
  ~DBuf() { if (p) (void)hipFree(p); }
  mgf_status ensure(size_t n, hipStream_t s, bool keep = false, size_t keep_n = 0) {
    if (n <= cap) return MGF_OK;
    size_t ncap = cap ? cap : 256;
    while (ncap < n) ncap += ncap / 2 + 256;
    T* np = nullptr;
    MGF_HIP_TRY(hipMalloc((void**)&np, ncap * sizeof(T)));
    if (keep && p && keep_n) {
      hipError_t e = hipMemcpyAsync(np, p, keep_n * sizeof(T), hipMemcpyDeviceToDevice, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      if (e != hipSuccess) { (void)hipFree(np); set_error("device copy failed: %s", hipGetErrorString(e)); return MGF_ERR_HIP; }
    }
    if (p) {
      (void)hipStreamSynchronize(s);
      (void)hipFree(p);
    }
    p = np;
    cap = ncap;
    return MGF_OK;
  }
  size_t bytes() const { return cap * sizeof(T); }
};

// Library primitive (rocPRIM) — prims.hip.  The device-wide scan is a stock primitive; every
// physics kernel is hand-written (kernels.h).
// out[0..n] = exclusive prefix sum of in[0..n-1], out[n] = total (in must have n+1 readable slots; slot n is ignored)
mgf_status prim_exclusive_scan_u32(mgf_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n_plus_1);
mgf_status prim_sort_pairs_u32(mgf_ctx* ctx, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, size_t n,
                               unsigned end_bit);

}  // namespace mgf
