// Developer microbenchmark: forward of one independent (text layer || visual layer) pair of the ViLBERT schedule
//   two : text chain on stream 0, visual chain on stream 1, fork / join once per layer (what the engine does today)
//   grp : ONE stream, the four GEMM pairs as grouped launches, attention / LayerNorm of both sides back to back
//   seq : ONE stream, nothing grouped
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 -Iinclude tools/pair_lab.cpp -Lcqa-crct_amd/crct -lcrct_hip \
//         -Wl,-rpath,'$ORIGIN/../cqa-crct_amd/crct' -o tools/pair_lab.bin ;  CRCT_GEMM_GROUP=12 ./tools/pair_lab.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "crct_hip.h"

struct Side {
  int B, T, H, heads, d, I, M;
  unsigned short *x, *qkv, *ctx, *y1, *x1, *u, *y2, *x2, *Wqkv, *Wo, *Wup, *Wdn;
  float *gamma, *beta, *mean, *rstd;
  uint8_t* mask;
};

static unsigned short* dev_rand(size_t n, int seed) {
  std::vector<unsigned short> h(n);
  srand(seed);
  for (size_t i = 0; i < n; ++i) { float f = (rand() / (float)RAND_MAX - 0.5f) * 0.1f; unsigned u; memcpy(&u, &f, 4); h[i] = u >> 16; }
  unsigned short* p; hipMalloc(&p, n * 2); hipMemcpy(p, h.data(), n * 2, hipMemcpyHostToDevice);
  return p;
}

static Side make(int B, int T, int H, int heads, int I, int seed) {
  Side s; s.B = B; s.T = T; s.H = H; s.heads = heads; s.d = H / heads; s.I = I; s.M = B * T;
  const size_t M = s.M;
  s.x = dev_rand(M * H, seed); s.qkv = dev_rand(M * 3 * H, seed + 1); s.ctx = dev_rand(M * H, seed + 2); s.y1 = dev_rand(M * H, seed + 3);
  s.x1 = dev_rand(M * H, seed + 4); s.u = dev_rand(M * I, seed + 5); s.y2 = dev_rand(M * H, seed + 6); s.x2 = dev_rand(M * H, seed + 7);
  s.Wqkv = dev_rand((size_t)3 * H * H, seed + 8); s.Wo = dev_rand((size_t)H * H, seed + 9); s.Wup = dev_rand((size_t)I * H, seed + 10);
  s.Wdn = dev_rand((size_t)H * I, seed + 11);
  std::vector<float> ones(H, 1.f), zeros(H, 0.f);
  hipMalloc(&s.gamma, H * 4); hipMalloc(&s.beta, H * 4); hipMalloc(&s.mean, M * 4); hipMalloc(&s.rstd, M * 4);
  hipMemcpy(s.gamma, ones.data(), H * 4, hipMemcpyHostToDevice); hipMemcpy(s.beta, zeros.data(), H * 4, hipMemcpyHostToDevice);
  std::vector<uint8_t> m(B * T, 1);
  hipMalloc(&s.mask, B * T); hipMemcpy(s.mask, m.data(), B * T, hipMemcpyHostToDevice);
  return s;
}

static CrctGemmArgs gemm(const void* A, const void* W, void* C, int M, int N, int K) {
  CrctGemmArgs g; memset(&g, 0, sizeof(g));
  g.A = A; g.B = W; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N; g.ld_aux = N; g.ld_add = N; g.tile = -1; g.alpha = 1.f;
  return g;
}

struct Ops { CrctGemmArgs qkv, out, up, dn; };
static Ops ops_of(const Side& s) {
  return {gemm(s.x, s.Wqkv, s.qkv, s.M, 3 * s.H, s.H), gemm(s.ctx, s.Wo, s.y1, s.M, s.H, s.H), gemm(s.x1, s.Wup, s.u, s.M, s.I, s.H),
          gemm(s.u, s.Wdn, s.y2, s.M, s.H, s.I)};
}
static void attn(const Side& s, hipStream_t st) {
  crct_attention_fwd(s.qkv, s.qkv + s.H, s.qkv + 2 * s.H, s.mask, s.ctx, s.B, s.heads, s.T, s.T, s.d, 3 * s.H, 3 * s.H, 3 * s.H, s.H, 0, 1.f, 0, 0, st);
}
static void ln(const Side& s, const void* in, void* out, hipStream_t st) {
  crct_layernorm_fwd(in, s.gamma, s.beta, out, s.mean, s.rstd, s.M, s.H, 1e-12f, 0, 1.f, 0, 0, st);
}
static void chain(const Side& s, Ops& o, hipStream_t st) {
  crct_gemm_bf16(&o.qkv, st); attn(s, st); crct_gemm_bf16(&o.out, st); ln(s, s.y1, s.x1, st);
  crct_gemm_bf16(&o.up, st); crct_gemm_bf16(&o.dn, st); ln(s, s.y2, s.x2, st);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 50;
  Side t = make(80, 20, 768, 16, 3072, 1), v = make(80, 36, 1024, 16, 1024, 100);
  Ops ot = ops_of(t), ov = ops_of(v);
  hipStream_t s0, s1; hipStreamCreate(&s0); hipStreamCreate(&s1);
  hipEvent_t e0, e1, fork, join; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventCreateWithFlags(&fork, hipEventDisableTiming); hipEventCreateWithFlags(&join, hipEventDisableTiming);
  const char* names[3] = {"two streams", "one stream, grouped GEMM pairs", "one stream, sequential"};
  for (int mode = 0; mode < 3; ++mode) {
    auto layer = [&]() {
      if (mode == 0) {
        hipEventRecord(fork, s0); hipStreamWaitEvent(s1, fork, 0);
        chain(t, ot, s0); chain(v, ov, s1);
        hipEventRecord(join, s1); hipStreamWaitEvent(s0, join, 0);
      } else if (mode == 1) {
        CrctGemmArgs p[2];
        p[0] = ot.qkv; p[1] = ov.qkv; crct_gemm_bf16_grouped(p, 2, s0);
        attn(t, s0); attn(v, s0);
        p[0] = ot.out; p[1] = ov.out; crct_gemm_bf16_grouped(p, 2, s0);
        ln(t, t.y1, t.x1, s0); ln(v, v.y1, v.x1, s0);
        p[0] = ot.up; p[1] = ov.up; crct_gemm_bf16_grouped(p, 2, s0);
        p[0] = ot.dn; p[1] = ov.dn; crct_gemm_bf16_grouped(p, 2, s0);
        ln(t, t.y2, t.x2, s0); ln(v, v.y2, v.x2, s0);
      } else { chain(t, ot, s0); chain(v, ov, s0); }
    };
    for (int i = 0; i < 5; ++i) layer();
    hipEventRecord(e0, s0);
    for (int i = 0; i < iters; ++i) layer();
    hipEventRecord(e1, s0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) { printf("failed: %s\n", crct_last_error()); return 1; }
    printf("%-34s %8.2f us per (text layer || visual layer) forward\n", names[mode], ms * 1e3 / iters);
  }
  return 0;
}
