// Developer probe (not product): what does ONE CU take in per second, and through which path?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/lab/fill_probe.hip -o tools/lab/fill_probe.bin && ./tools/lab/fill_probe.bin
// "settle whether ~70 GB/s per CU is an LDS-DMA limit or a TCP limit".  Every workgroup streams the operand
// tiles of a GEMM K loop -- `rows` rows of 128 bytes per K step, row stride `ld` bytes, advancing 128 bytes per step -- either
//   mode 0: buffer_load_dwordx4 ... lds  (the GEMM kernels' path: L2 -> TCP -> LDS, no registers), `depth` K steps in flight
//   mode 1: buffer_load_dwordx4 into VGPRs (L2 -> TCP -> registers), same addresses, same depth
//   mode 2: mode 0 with CONTIGUOUS 1-KiB pieces (row stride = 128 B: whole 1-KiB runs per wave instruction)
//   mode 3: mode 1 with contiguous pieces
// from a region that is shared by all workgroups (L2-hot), private per workgroup inside 64 MB (Infinity Cache) or inside 2 GB
// (HBM).  Nothing is computed; the rate is bytes / kernel time / active CUs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef unsigned u4_t __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
  else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if constexpr (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
  else if constexpr (N == 36) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
  else if constexpr (N == 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
  else static_assert(N == 0, "add the literal");
}

struct Args {
  const char* buf; size_t region_stride;   // byte distance between the regions of consecutive region indices
  int n_regions;                            // blockIdx.x % n_regions picks the region
  int ld;                                   // row stride in bytes
  int steps;                                // K steps per workgroup
  int wrap;                                 // K steps after which the stream restarts at column 0 (ld / 128)
  unsigned* sink;
};

// P = 1-KiB pieces per wave per K step; NW waves; D = K steps in flight; LDS = to LDS or to registers; CONTIG = 1-KiB runs
template <int P, int NW, int D, bool LDS, bool CONTIG>
__global__ __launch_bounds__(NW * 64) void probe(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* base = a.buf + (size_t)(blockIdx.x % a.n_regions) * a.region_stride;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 0x7fffffff, 0x00020000);
  unsigned off[P];
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int slot = (i * NW + wave) * 64 + lane;          // 16-byte slot of the tile image
    if (CONTIG) off[i] = (unsigned)slot * 16u;              // rows of 128 B back to back
    else off[i] = (unsigned)((slot >> 3) * a.ld + (((slot & 7) ^ ((slot >> 3) & 7)) << 4));   // the GEMM kernels' swizzled source
  }
  const int step_bytes = CONTIG ? P * NW * 1024 : 128;
  u4_t acc = {0, 0, 0, 0};
  u4_t regs[D][P];
  auto issue = [&](int k, int st) {
    const int kk = k % a.wrap;
#pragma unroll
    for (int i = 0; i < P; ++i) {
      if constexpr (LDS)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_ptr)(smem + st * (P * NW * 1024) + (i * NW + wave) * 1024), 16, (int)off[i], kk * step_bytes, 0, 0);
      else
        regs[st][i] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off[i], kk * step_bytes, 0);
    }
  };
  // static ring of D stages: stage index must be a compile-time constant for the register variant -> unrolled by D
#pragma unroll
  for (int d = 0; d < D - 1; ++d) issue(d, d);
  for (int k0 = 0; k0 < a.steps; k0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int k = k0 + d;
      issue(k + D - 1, (d + D - 1) % D);
      wait_vmcnt<(D - 1) * P>();
      if constexpr (LDS) {
        __builtin_amdgcn_s_barrier();                        // the GEMM loop's one barrier per K step
      } else {
#pragma unroll
        for (int i = 0; i < P; ++i) acc ^= regs[d][i];
      }
    }
  }
  wait_vmcnt<0>();
  if (!LDS && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) a.sink[0] = 1;
  if (LDS) { __syncthreads(); if (reinterpret_cast<unsigned*>(smem)[tid] == 0x12345u) a.sink[1] = 1; }
}

static char* g_buf; static unsigned* g_sink;
static hipEvent_t e0, e1;

template <int P, int NW, int D, bool LDS, bool CONTIG>
static double run(int grid, int n_regions, size_t region_stride, int ld, int steps) {
  Args a; a.buf = g_buf; a.region_stride = region_stride; a.n_regions = n_regions; a.ld = ld; a.steps = steps; a.wrap = CONTIG ? 64 : ld / 128; a.sink = g_sink;
  if (CONTIG) a.wrap = (int)(region_stride / (size_t)(P * NW * 1024)) > 0 ? (int)(region_stride / (size_t)(P * NW * 1024)) : 1;
  const size_t lds = LDS ? (size_t)D * P * NW * 1024 : 0;
  auto kern = probe<P, NW, D, LDS, CONTIG>;
  if (lds > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, 0, a);
  hipEventRecord(e0, 0);
  const int reps = 5;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, 0, a);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) return -1;
  const double bytes = (double)grid * steps * P * NW * 1024.0 * reps;
  return bytes / (ms * 1e-3) / 1e9;        // GB/s, whole chip
}

int main() {
  const size_t total = (size_t)2 << 30;
  hipMalloc(&g_buf, total); hipMalloc(&g_sink, 64);
  hipMemset(g_buf, 1, total); hipMemset(g_sink, 0, 64);
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int steps = 1536;
  struct Fp { const char* name; int n_regions; size_t stride; };
  // a region = 192 rows x ld bytes (GEMM-shaped) or 1.5 MB of contiguous bytes
  printf("GEMM-shaped tile stream (192 rows x 128 B per K step = 24 KB, row stride ld), 8 waves, 3 pieces per wave\n");
  printf("%-34s %6s %6s | %9s %9s %9s | %9s %9s %9s\n", "footprint", "grid", "ld", "dma d=2", "dma d=3", "dma d=4", "reg d=2", "reg d=3", "reg d=4");
  for (int ld : {1536, 2048, 6144, 1536 + 128}) {
    const size_t region = (size_t)192 * ld;
    std::vector<Fp> fps = {{"shared by all WGs (L2 hot)", 1, region}, {"8 regions (one per XCD, L2 hot)", 8, region},
                           {"private, 64 MB in all (MALL)", (int)((64u << 20) / region), region}, {"private, 2 GB in all (HBM)", (int)(total / region), region}};
    for (const Fp& f : fps)
      for (int grid : {64, 256, 512}) {
        double v[6];
        v[0] = run<3, 8, 2, true, false>(grid, f.n_regions, f.stride, ld, steps);
        v[1] = run<3, 8, 3, true, false>(grid, f.n_regions, f.stride, ld, steps);
        v[2] = run<3, 8, 4, true, false>(grid, f.n_regions, f.stride, ld, steps);
        v[3] = run<3, 8, 2, false, false>(grid, f.n_regions, f.stride, ld, steps);
        v[4] = run<3, 8, 3, false, false>(grid, f.n_regions, f.stride, ld, steps);
        v[5] = run<3, 8, 4, false, false>(grid, f.n_regions, f.stride, ld, steps);
        const double cus = grid < 256 ? grid : 256;
        printf("%-34s %6d %6d | %9.1f %9.1f %9.1f | %9.1f %9.1f %9.1f   GB/s per CU\n", f.name, grid, ld, v[0] / cus, v[1] / cus, v[2] / cus, v[3] / cus, v[4] / cus, v[5] / cus);
      }
  }
  printf("\ncontiguous 1-KiB pieces (24 KB per K step per WG), 8 waves\n");
  for (int grid : {64, 256, 512}) {
    const size_t region = (size_t)1536 << 10;
    for (int nr : {1, 8, 40, 1300}) {
      double v[4];
      v[0] = run<3, 8, 2, true, true>(grid, nr, region, 0, steps);
      v[1] = run<3, 8, 4, true, true>(grid, nr, region, 0, steps);
      v[2] = run<3, 8, 2, false, true>(grid, nr, region, 0, steps);
      v[3] = run<3, 8, 4, false, true>(grid, nr, region, 0, steps);
      const double cus = grid < 256 ? grid : 256;
      printf("regions %5d (%.1f MB) grid %4d | dma d=2 %7.1f d=4 %7.1f | reg d=2 %7.1f d=4 %7.1f   GB/s per CU\n", nr, nr * 1.5, grid, v[0] / cus, v[1] / cus, v[2] / cus, v[3] / cus);
    }
  }
  printf("\nwave-count sweep, shared L2-hot region, ld 1536, 24 KB per K step split over the waves, depth 3\n");
  {
    const size_t region = (size_t)192 * 1536;
    for (int grid : {256, 512}) {
      double a4 = run<6, 4, 3, true, false>(grid, 1, region, 1536, steps), a8 = run<3, 8, 3, true, false>(grid, 1, region, 1536, steps);
      double a16 = run<3, 16, 3, true, false>(grid, 1, (size_t)384 * 1536, 1536, steps / 2);       // 48 KB per K step with 16 waves
      double r4 = run<6, 4, 3, false, false>(grid, 1, region, 1536, steps), r8 = run<3, 8, 3, false, false>(grid, 1, region, 1536, steps);
      const double cus = 256;
      printf("grid %4d | dma 4w %7.1f 8w %7.1f 16w(48KB) %7.1f | reg 4w %7.1f 8w %7.1f   GB/s per CU\n", grid, a4 / cus, a8 / cus, a16 / cus, r4 / cus, r8 / cus);
    }
  }
  return 0;
}
