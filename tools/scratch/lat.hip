// development aid: latency of the load flavours the polling wave could use (one wave per block, dependent loads)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ inline __amdgpu_buffer_rsrc_t rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00027000); }
template <int MODE>
__global__ void k_lat(const uint32_t* buf, uint32_t words_per_block, uint32_t reps, uint32_t same_line, uint64_t* out, int busy_waves) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t* base = buf + (size_t)blockIdx.x * words_per_block;
  if (wave > 0) {  // background traffic: streaming plain loads over the block's region
    float acc = 0;
    for (uint32_t r = 0; r < reps * 8; ++r) {
      const uint32_t o = ((r * 64u + lane) * 32u + wave * 8u) % words_per_block;
      acc += __builtin_nontemporal_load(reinterpret_cast<const float*>(base + o));
    }
    if (acc == 123.f) out[1000 + threadIdx.x] = 1;
    return;
  }
  __amdgpu_buffer_rsrc_t rb = rsrc(base);
  uint32_t off = lane * 32u;  // words: every lane its own 128-B line
  const uint64_t t0 = wall_clock64();
  uint32_t sink = 0;
  for (uint32_t r = 0; r < reps; ++r) {
    uint32_t v;
    if (MODE == 0) { v4f g = __builtin_amdgcn_raw_buffer_load_b128(rb, (int)(off * 4u), 0, 0); v = __float_as_uint(g.x); }
    else if (MODE == 1) { v4f g = __builtin_amdgcn_raw_buffer_load_b128(rb, (int)(off * 4u), 0, 16 /* sc1 */); v = __float_as_uint(g.x); }
    else if (MODE == 2) { v4f g = __builtin_amdgcn_raw_buffer_load_b128(rb, (int)(off * 4u), 0, 17 /* sc0 sc1 */); v = __float_as_uint(g.x); }
    else if (MODE == 3) { v4f g = __builtin_amdgcn_raw_buffer_load_b128(rb, (int)(off * 4u), 0, 1 /* sc0 */); v = __float_as_uint(g.x); }
    else { v = __hip_atomic_load(base + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    sink += v;
    // dependent address: the loaded value is 0; next line (or the same line)
    off = same_line ? off + v : (off + 64u * 32u + v) % words_per_block;
  }
  const uint64_t t1 = wall_clock64();
  if (lane == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = sink; }
}
template <int MODE> void run(const char* name, uint32_t* d, uint64_t* dout, uint32_t wpb, int blocks, int threads, uint32_t same) {
  const uint32_t reps = 2000;
  k_lat<MODE><<<blocks, threads>>>(d, wpb, reps, same, dout, threads / 64 - 1);
  hipDeviceSynchronize();
  std::vector<uint64_t> h(2 * blocks);
  hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0; for (int b = 0; b < blocks; ++b) s += (double)h[2 * b];
  printf("%-14s blocks %3d waves/block %2d %s: %.0f ns per dependent load\n", name, blocks, threads / 64, same ? "same line " : "fresh lines", s / blocks / reps * 10.0);
}
int main() {
  const int blocks = 256; const uint32_t wpb = 1u << 20;  // 4 MB per block: 1 GB total (beyond L2 and MALL)
  uint32_t* d; uint64_t* dout;
  hipMalloc(&d, (size_t)blocks * wpb * 4); hipMemset(d, 0, (size_t)blocks * wpb * 4); hipMalloc(&dout, 8 * 4096 * 2);
  for (int threads : {64, 768}) for (uint32_t same : {1u, 0u}) for (int nb : {1, 256}) {
    run<0>("plain", d, dout, wpb, nb, threads, same);
    run<3>("sc0", d, dout, wpb, nb, threads, same);
    run<1>("sc1", d, dout, wpb, nb, threads, same);
    run<2>("sc0 sc1", d, dout, wpb, nb, threads, same);
    run<4>("atomic agent", d, dout, wpb, nb, threads, same);
  }
  return 0;
}
