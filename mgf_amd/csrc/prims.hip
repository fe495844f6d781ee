// rocPRIM-backed device-wide scan (a library primitive, kept in its own TU so the hand-written kernels
// recompile quickly).
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "common.h"

namespace mgf {

static mgf_status ensure_tmp(mgf_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->prim_tmp_bytes) return MGF_OK;
  if (ctx->prim_tmp) { (void)hipStreamSynchronize(ctx->stream); (void)hipFree(ctx->prim_tmp); ctx->prim_tmp = nullptr; ctx->prim_tmp_bytes = 0; }
  size_t nb = bytes + bytes / 2 + 4096;
  MGF_HIP_TRY(hipMalloc(&ctx->prim_tmp, nb));
  ctx->prim_tmp_bytes = nb;
  return MGF_OK;
}

mgf_status prim_exclusive_scan_u32(mgf_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n_plus_1) {
  if (n_plus_1 == 0) return MGF_OK;
  size_t bytes = 0;
  MGF_HIP_TRY(rocprim::exclusive_scan(nullptr, bytes, in, out, 0u, n_plus_1, rocprim::plus<uint32_t>(), ctx->stream));
  MGF_TRY(ensure_tmp(ctx, bytes));
  MGF_HIP_TRY(rocprim::exclusive_scan(ctx->prim_tmp, bytes, in, out, 0u, n_plus_1, rocprim::plus<uint32_t>(), ctx->stream));
  return MGF_OK;
}

// (keys, values) sorted by the key bits [0, end_bit); stable
mgf_status prim_sort_pairs_u32(mgf_ctx* ctx, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, size_t n,
                               unsigned end_bit) {
  if (n == 0) return MGF_OK;
  size_t bytes = 0;
  MGF_HIP_TRY(rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, end_bit, ctx->stream));
  MGF_TRY(ensure_tmp(ctx, bytes));
  MGF_HIP_TRY(rocprim::radix_sort_pairs(ctx->prim_tmp, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, end_bit, ctx->stream));
  return MGF_OK;
}

}  // namespace mgf
