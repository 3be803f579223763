// Developer microbenchmark for the GEMM kernels (not part of the product or the tests):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Icqa-crct_amd/csrc tools/gemm_lab.cpp \
//         cqa-crct_amd/csrc/gemm.hip cqa-crct_amd/csrc/engine_err.cpp -o gpurun_out/gemm_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "crct_hip.h"
extern "C" int crct_gemm_force_generic(int on);
hipError_t crct_gemm_launch(const CrctGemmArgs& g, hipStream_t s);
static char err[256];
void crct_set_error(const char* fmt, ...) { strcpy(err, fmt); }

struct Shape { const char* name; int M, N, K, ta, tb; int lda = 0, ldb = 0; };

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 50;
  const int only_shape = argc > 2 ? atoi(argv[2]) : -1, only_tile = argc > 3 ? atoi(argv[3]) : -1, only_gen = argc > 4 ? atoi(argv[4]) : -1;
  std::vector<Shape> shapes = {
      {"t.qkv fwd   ", 1600, 2304, 768, 0, 0}, {"t.ffn_up fwd", 1600, 3072, 768, 0, 0}, {"t.ffn_dn fwd", 1600, 768, 3072, 0, 0},
      {"t.out fwd   ", 1600, 768, 768, 0, 0},  {"v.qkv fwd   ", 2880, 3072, 1024, 0, 0}, {"v.ffn fwd   ", 2880, 1024, 1024, 0, 0},
      {"t.ffn_up dg ", 1600, 768, 3072, 0, 1}, {"t.ffn_dn dg ", 1600, 3072, 768, 0, 1}, {"v.qkv dg    ", 2880, 1024, 3072, 0, 1},
      {"t.ffn_up wg ", 3072, 768, 1600, 1, 1}, {"t.ffn_dn wg ", 768, 3072, 1600, 1, 1}, {"v.qkv wg    ", 3072, 1024, 2880, 1, 1},
      {"v.ffn wg    ", 1024, 1024, 2880, 1, 1}, {"big 4096^3  ", 4096, 4096, 4096, 0, 0},
      {"up dg ldb1024", 1600, 768, 3072, 0, 1, 0, 1024}, {"up dg ldb 832", 1600, 768, 3072, 0, 1, 0, 832},
      {"up dg lda3136", 1600, 768, 3072, 0, 1, 3136, 0}, {"dn fw lda3136", 1600, 768, 3072, 0, 0, 3136, 3136},
      {"up dg N=1536 ", 1600, 1536, 3072, 0, 1}, {"dn fw N=1536 ", 1600, 1536, 3072, 0, 0}};
  size_t maxel = (size_t)4096 * 4096;
  unsigned short *A, *B, *C;
  hipMalloc(&A, maxel * 2); hipMalloc(&B, maxel * 2); hipMalloc(&C, maxel * 4);
  std::vector<unsigned short> h(maxel);
  srand(1);
  for (size_t i = 0; i < maxel; ++i) { float f = (rand() / (float)RAND_MAX - 0.5f); unsigned u; memcpy(&u, &f, 4); h[i] = u >> 16; }
  hipMemcpy(A, h.data(), maxel * 2, hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), maxel * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int generic = 0; generic < 2; ++generic) {
    if (only_gen >= 0 && generic != only_gen) continue;
    crct_gemm_force_generic(generic);
    printf("---- %s kernel\n", generic ? "generic (register staged)" : "pipelined (LDS-DMA)");
    for (size_t si = 0; si < shapes.size(); ++si) {
      auto& s = shapes[si];
      if (only_shape >= 0 && (int)si != only_shape) continue;
      for (int tile = 0; tile < (generic ? 4 : 16); ++tile) {
        if (only_tile >= 0 && tile != only_tile) continue;
        CrctGemmArgs g; memset(&g, 0, sizeof(g));
        g.A = A; g.B = B; g.C = C; g.M = s.M; g.N = s.N; g.K = s.K; g.ta = s.ta; g.tb = s.tb;
        g.lda = s.lda ? s.lda : (s.ta ? s.M : s.K); g.ldb = s.ldb ? s.ldb : (s.tb ? s.N : s.K); g.ldc = s.N; g.ld_aux = s.N; g.ld_add = s.N;
        g.tile = tile; g.alpha = 1.f; g.c_is_f32 = s.ta ? 1 : 0;
        for (int i = 0; i < 5; ++i) crct_gemm_launch(g, 0);
        hipEventRecord(e0, 0);
        for (int i = 0; i < iters; ++i) crct_gemm_launch(g, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double us = ms * 1e3 / iters, tf = 2.0 * s.M * s.N * s.K / (us * 1e-6) / 1e12;
        printf("%s M=%5d N=%5d K=%5d tile=%d  %8.2f us  %7.1f TF\n", s.name, s.M, s.N, s.K, tile, us, tf);
      }
    }
  }
  return 0;
}
