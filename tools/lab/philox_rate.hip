// Lab: what one Philox4x32 round costs with two multiply forms -- v_mul_hi_u32 + v_mul_lo_u32 (two quarter-rate instructions per
// product) against v_mad_u64_u32 (both halves from one instruction).  hipcc --offload-arch=gfx950 -O3 philox_rate.hip -o philox_rate.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <int FORM, int ROUNDS>
__device__ __forceinline__ void philox(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    uint32_t hi0, lo0, hi1, lo1;
    if (FORM == 0) {
      hi0 = __umulhi(0xD2511F53u, c0); lo0 = 0xD2511F53u * c0;
      hi1 = __umulhi(0xCD9E8D57u, c2); lo1 = 0xCD9E8D57u * c2;
    } else {
      const uint64_t p0 = (uint64_t)c0 * 0xD2511F53ull, p1 = (uint64_t)c2 * 0xCD9E8D57ull;
      hi0 = (uint32_t)(p0 >> 32); lo0 = (uint32_t)p0; hi1 = (uint32_t)(p1 >> 32); lo1 = (uint32_t)p1;
    }
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
template <int FORM, int ROUNDS>
__global__ void __launch_bounds__(256) rate(uint32_t* out, int iters, uint32_t seed) {
  uint32_t acc = 0;
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    uint32_t c0 = t, c1 = i, c2 = 7, c3 = 0x9E3779B9u;
    philox<FORM, ROUNDS>(c0, c1, c2, c3, seed, seed ^ 0x5555u);
    acc ^= c0 ^ c1 ^ c2 ^ c3;
  }
  out[t] = acc;
}
template <int FORM, int ROUNDS>
static void run(const char* name, uint32_t* out) {
  const int blocks = 256 * 8, iters = 4096;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  rate<FORM, ROUNDS><<<blocks, 256>>>(out, iters, 12345u);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) rate<FORM, ROUNDS><<<blocks, 256>>>(out, iters, 12345u + r);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  uint32_t h[4]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
  const double calls = (double)blocks * 256 * iters;
  // cycles per call and wave: calls / 64 wave-calls spread over 1024 SIMDs at 2.4 GHz
  printf("%-28s %8.3f ms   %6.2f Gcall/s   ~%5.1f SIMD cycles per wave-call   (check %08x)\n", name, ms, calls / ms * 1e-6,
         ms * 1e-3 * 2.4e9 * 1024.0 / (calls / 64.0), h[0] ^ h[1] ^ h[2] ^ h[3]);
}
int main() {
  uint32_t* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  run<0, 10>("mul_hi + mul_lo, 10 rounds", out);
  run<1, 10>("mad_u64_u32,     10 rounds", out);
  run<0, 7>("mul_hi + mul_lo,  7 rounds", out);
  run<1, 7>("mad_u64_u32,      7 rounds", out);
  return 0;
}
