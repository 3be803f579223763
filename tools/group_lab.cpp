// Developer microbenchmark: two independent forward GEMMs of a co-attention layer (text side, visual side) launched
//   seq: back to back on one stream     par: on two streams at once     grp: as ONE grouped launch (crct_gemm_bf16_grouped)
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 -Iinclude tools/group_lab.cpp -Lcqa-crct_amd/crct -lcrct_hip \
//         -Wl,-rpath,'$ORIGIN/../cqa-crct_amd/crct' -o tools/group_lab.bin ;  CRCT_GEMM_GROUP=12 ./tools/group_lab.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "crct_hip.h"

struct Shape { const char* name; int M, N, K; };
struct Pair { Shape a, b; };

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 50;
  std::vector<Pair> pairs = {
      {{"t.qkv2", 1600, 3072, 768}, {"v.qkv1", 2880, 3072, 1024}},   {{"t.dense2", 1600, 768, 1024}, {"v.dense1", 2880, 1024, 1024}},
      {{"t.ffn_up", 1600, 3072, 768}, {"v.ffn_up", 2880, 1024, 1024}}, {{"t.ffn_dn", 1600, 768, 3072}, {"v.ffn_dn", 2880, 1024, 1024}},
      {{"t.qkv", 1600, 2304, 768}, {"v.qkv", 2880, 3072, 1024}},     {{"t.out", 1600, 768, 768}, {"v.out", 2880, 1024, 1024}}};
  const size_t maxel = (size_t)4096 * 4096;
  unsigned short *A[2], *B[2]; void* C[2];
  std::vector<unsigned short> h(maxel);
  for (int s = 0; s < 2; ++s) {
    hipMalloc(&A[s], maxel * 2); hipMalloc(&B[s], maxel * 2); hipMalloc(&C[s], maxel * 2);
    srand(1 + s);
    for (size_t i = 0; i < maxel; ++i) { float f = (rand() / (float)RAND_MAX - 0.5f); unsigned u; memcpy(&u, &f, 4); h[i] = u >> 16; }
    hipMemcpy(A[s], h.data(), maxel * 2, hipMemcpyHostToDevice);
    hipMemcpy(B[s], h.data(), maxel * 2, hipMemcpyHostToDevice);
  }
  hipStream_t s0, s1; hipStreamCreate(&s0); hipStreamCreate(&s1);
  hipEvent_t e0, e1, fork, join; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&fork); hipEventCreate(&join);
  for (auto& p : pairs) {
    CrctGemmArgs g[2];
    const Shape* sh[2] = {&p.a, &p.b};
    for (int i = 0; i < 2; ++i) {
      memset(&g[i], 0, sizeof(g[i]));
      g[i].A = A[i]; g[i].B = B[i]; g[i].C = C[i]; g[i].M = sh[i]->M; g[i].N = sh[i]->N; g[i].K = sh[i]->K;
      g[i].lda = sh[i]->K; g[i].ldb = sh[i]->K; g[i].ldc = sh[i]->N; g[i].ld_aux = sh[i]->N; g[i].ld_add = sh[i]->N;
      g[i].tile = -1; g[i].alpha = 1.f;
    }
    float ms[3];
    for (int mode = 0; mode < 3; ++mode) {
      auto once = [&]() {
        if (mode == 0) { crct_gemm_bf16(&g[0], s0); crct_gemm_bf16(&g[1], s0); }
        else if (mode == 1) {
          hipEventRecord(fork, s0); hipStreamWaitEvent(s1, fork, 0);
          crct_gemm_bf16(&g[0], s0); crct_gemm_bf16(&g[1], s1);
          hipEventRecord(join, s1); hipStreamWaitEvent(s0, join, 0);
        } else crct_gemm_bf16_grouped(g, 2, s0);
      };
      for (int i = 0; i < 5; ++i) once();
      hipEventRecord(e0, s0);
      for (int i = 0; i < iters; ++i) once();
      hipEventRecord(e1, s0); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms[mode], e0, e1);
      if (hipGetLastError() != hipSuccess) { printf("launch failed: %s\n", crct_last_error()); return 1; }
    }
    const double gf = 2.0 * (p.a.M * (double)p.a.N * p.a.K + p.b.M * (double)p.b.N * p.b.K) / 1e9;
    printf("%-9s || %-9s  %6.2f GF   seq %7.2f us   par %7.2f us   grp %7.2f us  (%6.1f TF grouped)\n", p.a.name, p.b.name, gf,
           ms[0] * 1e3 / iters, ms[1] * 1e3 / iters, ms[2] * 1e3 / iters, gf / (ms[2] * 1e3 / iters) / 1e3);
  }
  return 0;
}
