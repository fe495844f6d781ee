// development aid: what clock64() (s_memtime) counts against wall_clock64() (100 MHz) under different loads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k_alu(uint64_t* out, uint32_t n, float seed) {
  const uint64_t c0 = clock64(), w0 = wall_clock64();
  float a = seed + threadIdx.x, b = 1.0001f;
  for (uint32_t i = 0; i < n; ++i) { a = a * b + 0.5f; b = b * 0.99999f + 1e-6f; }
  const uint64_t c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[3 * blockIdx.x] = c1 - c0; out[3 * blockIdx.x + 1] = w1 - w0; out[3 * blockIdx.x + 2] = (uint64_t)a; }
}
__global__ void k_lds(uint64_t* out, uint32_t n) {  // dependent LDS round trips, one wave busy, the others idle-spinning with s_sleep
  __shared__ uint32_t s[1024];
  for (uint32_t i = threadIdx.x; i < 1024; i += blockDim.x) s[i] = (i * 17u + 1u) & 1023u;
  __syncthreads();
  const uint64_t c0 = clock64(), w0 = wall_clock64();
  uint32_t x = threadIdx.x & 1023u;
  if (threadIdx.x < 64) { for (uint32_t i = 0; i < n; ++i) x = s[x]; }
  else { for (uint32_t i = 0; i < n / 4; ++i) { __builtin_amdgcn_s_sleep(2); x += s[x] & 1u; } }
  const uint64_t c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[3 * blockIdx.x] = c1 - c0; out[3 * blockIdx.x + 1] = w1 - w0; out[3 * blockIdx.x + 2] = x; }
}
int main() {
  uint64_t* d; hipMalloc(&d, 8 * 3 * 4096);
  std::vector<uint64_t> h(3 * 4096);
  auto rep = [&](const char* name, int blocks, uint32_t n) {
    hipDeviceSynchronize(); hipMemcpy(h.data(), d, 8 * 3 * blocks, hipMemcpyDeviceToHost);
    double c = 0, w = 0; for (int b = 0; b < blocks; ++b) { c += h[3 * b]; w += h[3 * b + 1]; }
    printf("%-28s blocks %4d: %.1f clock64 ticks per us (%.0f us per block), %.1f ticks per inner step\n", name, blocks, c / (w * 0.01), w * 0.01 / blocks, c / blocks / n);
  };
  for (int rounds = 0; rounds < 2; ++rounds) {
    k_alu<<<256, 256>>>(d, 2000000, 1.0f); rep("ALU loop, 256 x 256", 256, 2000000);
    k_alu<<<2048, 1024>>>(d, 400000, 1.0f); rep("ALU loop, 2048 x 1024", 2048, 400000);
    k_alu<<<1, 64>>>(d, 2000000, 1.0f); rep("ALU loop, 1 wave", 1, 2000000);
    k_lds<<<256, 768>>>(d, 200000); rep("LDS chase, 256 x 768", 256, 200000);
    k_lds<<<1, 768>>>(d, 200000); rep("LDS chase, 1 x 768", 1, 200000);
    k_lds<<<256, 64>>>(d, 200000); rep("LDS chase, 256 x 64", 256, 200000);
  }
  return 0;
}
