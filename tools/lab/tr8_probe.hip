#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v2i __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2i* lds_v2i_ptr;
// LDS image: 64 rows x 64 bytes; byte at (r, c) = r * 64 + c  (mod 256 -> store r in one run and c in another)
__global__ void probe(uint32_t* out, int mode, int addr_mode) {
  __shared__ __attribute__((aligned(16))) uint8_t img[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) img[i] = mode ? (uint8_t)(i / 64) : (uint8_t)(i % 64);
  __syncthreads();
  const int lane = threadIdx.x;
  const int g = lane >> 4, l = lane & 15;
  int r, c;
  if (addr_mode == 0) { r = 8 * g + (l >> 1); c = 8 * (l & 1); }        // lane 2q+p of group: row q, cols 8p..8p+7 (8 rows x 16 cols)
  else { r = 8 * g + (l & 7); c = 8 * (l >> 3); }                        // lane 8p+q
  const uint8_t* a = img + r * 64 + c;
  v2i v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i_ptr)(uintptr_t)(uint32_t)(uintptr_t)a);
  out[2 * lane] = v[0]; out[2 * lane + 1] = v[1];
}
int main() {
  uint32_t* d; hipMalloc(&d, 64 * 8);
  uint32_t h[128];
  for (int am = 0; am < 2; ++am)
    for (int mode = 0; mode < 2; ++mode) {
      probe<<<1, 64>>>(d, mode, am);
      hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      printf("addr_mode %d, %s of each received byte:\n", am, mode ? "ROW" : "COL");
      for (int lane = 0; lane < 64; ++lane) {
        printf(" lane %2d:", lane);
        for (int b = 0; b < 8; ++b) printf(" %2d", (h[2 * lane + b / 4] >> (8 * (b & 3))) & 255);
        printf("\n");
        if (lane == 17) { printf(" ...\n"); lane = 47; }
      }
    }
  return 0;
}
