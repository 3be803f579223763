// Developer microbenchmark for the GEMM kernels (not part of the product or the tests); links the in-tree library:
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 -Iinclude tools/gemm_lab.cpp -Lcqa-crct_amd/crct -lcrct_hip \
//         -Wl,-rpath,'$ORIGIN/../cqa-crct_amd/crct' -o tools/gemm_lab.bin
//   ./tools/gemm_lab.bin [iters] [shape | -1] [tile | -1] [first_tile] [cold] [max_split_k]
// max_split_k > 1: every forward / data-gradient shape whose output is narrow (N <= 1024) is also timed K-partitioned
// (CrctGemmArgs.split_k = 2 .. max_split_k) with the configurations built for it (4, 9, 12, 15).
// Every configuration is checked against the register-staged kernel on the same operands (max |diff| printed).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "crct_hip.h"

struct Shape { const char* name; int M, N, K, ta, tb; };

static float bf2f(unsigned short v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 50;
  const int only_shape = argc > 2 ? atoi(argv[2]) : -1, only_tile = argc > 3 ? atoi(argv[3]) : -1;
  const int first_tile = argc > 4 ? atoi(argv[4]) : 0;
  // cold > 0: every launch reads a different copy of B (weights) out of a pool larger than the 256 MiB Infinity Cache, and a
  // different copy of A out of 4 -- the state the GEMMs of a training step run in (weights come from HBM, once per step)
  const int cold = argc > 5 ? atoi(argv[5]) : 0;
  const int max_sk = argc > 6 ? atoi(argv[6]) : 1;
  std::vector<Shape> shapes = {
      {"t.qkv fwd   ", 1600, 2304, 768, 0, 0},  {"t.ffn_up fwd", 1600, 3072, 768, 0, 0},  {"t.ffn_dn fwd", 1600, 768, 3072, 0, 0},
      {"t.out fwd   ", 1600, 768, 768, 0, 0},   {"v.qkv fwd   ", 2880, 3072, 1024, 0, 0}, {"v.ffn fwd   ", 2880, 1024, 1024, 0, 0},
      {"t.ffn_up dg ", 1600, 768, 3072, 0, 1},  {"t.ffn_dn dg ", 1600, 3072, 768, 0, 1},  {"v.qkv dg    ", 2880, 1024, 3072, 0, 1},
      {"t.qkv dg    ", 1600, 768, 2304, 0, 1},  {"v.ffn dg    ", 2880, 1024, 1024, 0, 1}, {"t.out dg    ", 1600, 768, 768, 0, 1},
      {"t.ffn_up wg ", 3072, 768, 1600, 1, 1},  {"t.ffn_dn wg ", 768, 3072, 1600, 1, 1},  {"v.qkv wg    ", 3072, 1024, 2880, 1, 1},
      {"v.ffn wg    ", 1024, 1024, 2880, 1, 1}, {"t.qkv wg    ", 2304, 768, 1600, 1, 1},  {"big 4096^3  ", 4096, 4096, 4096, 0, 0},
      {"lc t.up fwd ", 2560, 3072, 768, 0, 0},  {"lc v.qkv fwd", 6400, 3072, 1024, 0, 0}, {"c.qkv2 fwd  ", 1600, 3072, 768, 0, 0},
      {"c.dense2 fwd", 1600, 768, 1024, 0, 0},  {"img emb fwd ", 2880, 1024, 2048, 0, 0},
      // long context (BASELINE configs[3]: B = 64, 100 visual elements, 40 tokens: 2560 text rows, 6400 visual rows)
      {"lc t.qkv fwd", 2560, 2304, 768, 0, 0},  {"lc t.dn fwd ", 2560, 768, 3072, 0, 0},  {"lc t.out fwd", 2560, 768, 768, 0, 0},
      {"lc v.ffn fwd", 6400, 1024, 1024, 0, 0}, {"lc t.up dg  ", 2560, 768, 3072, 0, 1},  {"lc t.dn dg  ", 2560, 3072, 768, 0, 1},
      {"lc v.qkv dg ", 6400, 1024, 3072, 0, 1}, {"lc v.ffn dg ", 6400, 1024, 1024, 0, 1}, {"lc t.qkv dg ", 2560, 768, 2304, 0, 1},
      {"lc v.qkv wg ", 3072, 1024, 6400, 1, 1}, {"lc v.ffn wg ", 1024, 1024, 6400, 1, 1}, {"lc t.up wg  ", 3072, 768, 2560, 1, 1},
      {"lc t.dn wg  ", 768, 3072, 2560, 1, 1},  {"lc c.qkv2 fw", 2560, 3072, 768, 0, 0},  {"lc img emb  ", 6400, 1024, 2048, 0, 0}};
  size_t maxel = (size_t)6400 * 4096;
  unsigned short *A, *B;
  void *C, *Cref;
  hipMalloc(&A, maxel * 2); hipMalloc(&B, maxel * 2); hipMalloc(&C, maxel * 4); hipMalloc(&Cref, maxel * 4);
  std::vector<unsigned short> h(maxel);
  srand(1);
  for (size_t i = 0; i < maxel; ++i) { float f = (rand() / (float)RAND_MAX - 0.5f); unsigned u; memcpy(&u, &f, 4); h[i] = u >> 16; }
  hipMemcpy(A, h.data(), maxel * 2, hipMemcpyHostToDevice);
  srand(2);
  for (size_t i = 0; i < maxel; ++i) { float f = (rand() / (float)RAND_MAX - 0.5f); unsigned u; memcpy(&u, &f, 4); h[i] = u >> 16; }
  hipMemcpy(B, h.data(), maxel * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float* sk_ws; unsigned* sk_cnt;
  hipMalloc(&sk_ws, (size_t)crct_gemm_splitk_ws_elems(6400, 1024, 4) * 4); hipMalloc(&sk_cnt, 65536); hipMemset(sk_cnt, 0, 65536);
  float* scales; hipMalloc(&scales, 8); { float h2[2] = {1.f, 1.f}; hipMemcpy(scales, h2, 8, hipMemcpyHostToDevice); }
  const size_t pool_bytes = cold ? (size_t)640 << 20 : 0;
  char* poolB = nullptr; char* poolA = nullptr;
  if (cold) {
    hipMalloc(&poolB, pool_bytes); hipMalloc(&poolA, (size_t)4 * 40 << 20);
    for (size_t o = 0; o + maxel * 2 <= pool_bytes; o += maxel * 2) hipMemcpy(poolB + o, B, maxel * 2, hipMemcpyDeviceToDevice);
    for (int i = 0; i < 4; ++i) hipMemcpy(poolA + ((size_t)i * 40 << 20), A, (size_t)40 << 20, hipMemcpyDeviceToDevice);
  }
  std::vector<float> hc, hr;
  std::vector<unsigned short> hcb, hrb;
  for (size_t si = 0; si < shapes.size(); ++si) {
    auto& s = shapes[si];
    if (only_shape >= 0 && (int)si != only_shape) continue;
    auto make = [&](int tile, void* out) {
      CrctGemmArgs g; memset(&g, 0, sizeof(g));
      g.A = A; g.B = B; g.C = out; g.M = s.M; g.N = s.N; g.K = s.K; g.ta = s.ta; g.tb = s.tb;
      g.lda = s.ta ? s.M : s.K; g.ldb = s.tb ? s.N : s.K; g.ldc = s.N; g.ld_aux = s.N; g.ld_add = s.N;
      g.tile = tile; g.alpha = 1.f; g.c_is_f32 = s.ta ? 1 : 0;
      return g;
    };
    const size_t nout = (size_t)s.M * s.N;
    const bool f32 = s.ta != 0;
    crct_gemm_force_generic(1);
    { CrctGemmArgs g = make(-1, Cref); if (crct_gemm_bf16(&g, 0)) { printf("ref failed: %s\n", crct_last_error()); return 1; } }
    crct_gemm_force_generic(0);
    hipDeviceSynchronize();
    if (f32) { hr.resize(nout); hipMemcpy(hr.data(), Cref, nout * 4, hipMemcpyDeviceToHost); }
    else { hrb.resize(nout); hipMemcpy(hrb.data(), Cref, nout * 2, hipMemcpyDeviceToHost); }
    for (int tile = first_tile; tile < 72; ++tile)
     for (int sk = 1; sk <= max_sk; ++sk) {
      if ((tile >= 16 && tile < 20) || tile == 36 || tile == 37) continue;
      if (only_tile >= 0 && tile != only_tile) continue;
      if (sk > 1 && (s.ta || s.N > 1024 || !(tile == 4 || tile == 9 || tile == 12 || tile == 15) || s.K / 64 < 2 * sk)) continue;
      CrctGemmArgs g = make(tile, C);
      if (sk > 1) { g.split_k = sk; g.splitk_ws = sk_ws; g.splitk_cnt = sk_cnt; }
      const bool f8 = tile == 20 || tile == 21;          // fp8 forward kernel (2 / 3 stages): same byte buffers read as e4m3, timing only
      if (f8) {
        if (s.ta || s.tb || s.K % 128) continue;
        g.fp8 = 1; g.scale_a = scales; g.scale_b = scales + 1; g.tile = tile; g.lda = s.K; g.ldb = s.K;
      }
      hipMemset(C, 0xff, nout * (f32 ? 4 : 2));
      if (crct_gemm_bf16(&g, 0) != 0) { (void)hipGetLastError(); continue; }      // configuration not built for this mode
      if (hipDeviceSynchronize() != hipSuccess) { printf("tile %d: launch failed: %s\n", tile, hipGetErrorString(hipGetLastError())); return 1; }
      double maxd = 0;
      if (f32) { hc.resize(nout); hipMemcpy(hc.data(), C, nout * 4, hipMemcpyDeviceToHost); for (size_t i = 0; i < nout; ++i) { double d = fabs((double)hc[i] - hr[i]); if (!(d <= maxd)) maxd = d; } }
      else { hcb.resize(nout); hipMemcpy(hcb.data(), C, nout * 2, hipMemcpyDeviceToHost); for (size_t i = 0; i < nout; ++i) { double d = fabs((double)bf2f(hcb[i]) - bf2f(hrb[i])); if (!(d <= maxd)) maxd = d; } }
      const size_t bsz = ((size_t)(s.tb ? s.K : s.N) * g.ldb * 2 + 4095) & ~(size_t)4095;
      const int nb = cold ? (int)(pool_bytes / bsz) : 1;
      auto launch = [&](int i) {
        if (cold) { g.B = poolB + (size_t)(i % nb) * bsz; g.A = poolA + ((size_t)(i % 4) * 40 << 20); }
        crct_gemm_bf16(&g, 0);
      };
      for (int i = 0; i < 5; ++i) launch(i);
      hipEventRecord(e0, 0);
      for (int i = 0; i < iters; ++i) launch(i + 5);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double us = ms * 1e3 / iters, tf = 2.0 * s.M * s.N * s.K / (us * 1e-6) / 1e12;
      printf("%s M=%5d N=%5d K=%5d tile=%2d S=%d  %8.2f us  %7.1f TF  maxdiff %.3g%s\n", s.name, s.M, s.N, s.K, tile, sk, us, tf, maxd,
             (maxd > 0.26 && !f8) ? "  <-- MISMATCH" : "");
      fflush(stdout);
    }
  }
  return 0;
}
