// Scaled-dot-product attention for the short CRCT sequences (<= 112 visual elements, <= 112 text
// tokens, head size 32 / 48 / 64): one 256-thread workgroup per (batch, head); q, k, v, the score
// matrix and the probabilities live in LDS in fp32, all arithmetic is fp32 VALU (the score / PV
// products are < 1 % of the step's FLOPs -- SURVEY.md 8a -- so they do not go to MFMA).
//   P = softmax(q k^T / sqrt(d) + (1 - keymask) * -10000) ; ctx = dropout(P) v
// Reference: BertSelfAttention.forward vilbert.py:392-412, BertImageSelfAttention :522-543,
// BertBiAttention :684-723 (both directions are two calls with q and k/v from different streams).
// The backward pass recomputes P from q, k (nothing but q/k/v is kept from the forward) and
// regenerates the dropout mask from (seed, site, element index).
#include "common.cuh"
#include "crct_internal.h"

namespace {

struct AttnArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; const uint8_t* keymask;
  bf16_t* ctx;
  const bf16_t* dctx; bf16_t* dq; bf16_t* dk; bf16_t* dv;
  int B, heads, Tq, Tk, d;
  long ldq, ldk, ldv, ldo, lddq, lddk, lddv;
  uint32_t thr; float dscale; uint32_t site; uint64_t seed;
  float scale;
};

// cooperative load of rows [T][d] (bf16, row stride ld) into LDS fp32 [T][d+1]
__device__ __forceinline__ void load_tile(float* dst, const bf16_t* src, long ld, int T, int d, int tid) {
  const int cpr = d >> 3;   // 16-byte chunks per row
  for (int c = tid; c < T * cpr; c += 256) {
    const int r = c / cpr, cc = (c % cpr) << 3;
    const uint4 u = *reinterpret_cast<const uint4*>(src + (long)r * ld + cc);
    float* o = dst + r * (d + 1) + cc;
    o[0] = bf2f((bf16_t)(u.x & 0xffff)); o[1] = bf2f((bf16_t)(u.x >> 16));
    o[2] = bf2f((bf16_t)(u.y & 0xffff)); o[3] = bf2f((bf16_t)(u.y >> 16));
    o[4] = bf2f((bf16_t)(u.z & 0xffff)); o[5] = bf2f((bf16_t)(u.z >> 16));
    o[6] = bf2f((bf16_t)(u.w & 0xffff)); o[7] = bf2f((bf16_t)(u.w >> 16));
  }
}
// cooperative store LDS fp32 [T][d+1] * mul -> bf16 rows
__device__ __forceinline__ void store_tile(bf16_t* dst, long ld, const float* src, int T, int d, int tid, float mul) {
  const int cpr = d >> 3;
  for (int c = tid; c < T * cpr; c += 256) {
    const int r = c / cpr, cc = (c % cpr) << 3;
    const float* s = src + r * (d + 1) + cc;
    uint4 u;
    u.x = pack2bf(s[0] * mul, s[1] * mul); u.y = pack2bf(s[2] * mul, s[3] * mul);
    u.z = pack2bf(s[4] * mul, s[5] * mul); u.w = pack2bf(s[6] * mul, s[7] * mul);
    *reinterpret_cast<uint4*>(dst + (long)r * ld + cc) = u;
  }
}

// C[i][j] = sum_c X[i][c] * Y[j][c]   (X: [TX][d+1], Y: [TY][d+1]) -> out[i*ldo + j]
// 16x16 thread grid, each thread owns an AX x AY register tile (rows ty+16a, cols tx+16b).
template <int AX, int AY>
__device__ __forceinline__ void mm_nt(float* out, int ldo, const float* X, const float* Y, int TX, int TY, int d, int tid) {
  const int ty = tid >> 4, tx = tid & 15;
  float acc[AX][AY];
#pragma unroll
  for (int a = 0; a < AX; ++a)
#pragma unroll
    for (int b = 0; b < AY; ++b) acc[a][b] = 0.f;
  const int st = d + 1;
  for (int c = 0; c < d; ++c) {
    float xv[AX], yv[AY];
#pragma unroll
    for (int a = 0; a < AX; ++a) { const int i = ty + 16 * a; xv[a] = i < TX ? X[i * st + c] : 0.f; }
#pragma unroll
    for (int b = 0; b < AY; ++b) { const int j = tx + 16 * b; yv[b] = j < TY ? Y[j * st + c] : 0.f; }
#pragma unroll
    for (int a = 0; a < AX; ++a)
#pragma unroll
      for (int b = 0; b < AY; ++b) acc[a][b] += xv[a] * yv[b];
  }
#pragma unroll
  for (int a = 0; a < AX; ++a)
#pragma unroll
    for (int b = 0; b < AY; ++b) {
      const int i = ty + 16 * a, j = tx + 16 * b;
      if (i < TX && j < TY) out[i * ldo + j] = acc[a][b];
    }
}

// O[i][c] = sum_j P[i][j] * Y[j][c]    (P: [TX][ldp], Y: [TJ][d+1]) -> out [TX][d+1];  c = tx + 16 b, b < d/16
template <int AX>
__device__ __forceinline__ void mm_nn(float* out, const float* P, int ldp, const float* Y, int TX, int TJ, int d, int tid) {
  const int ty = tid >> 4, tx = tid & 15;
  float acc[AX][4];
#pragma unroll
  for (int a = 0; a < AX; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
  const int st = d + 1;
  for (int j = 0; j < TJ; ++j) {
    float pv[AX], yv[4];
#pragma unroll
    for (int a = 0; a < AX; ++a) { const int i = ty + 16 * a; pv[a] = i < TX ? P[i * ldp + j] : 0.f; }
#pragma unroll
    for (int b = 0; b < 4; ++b) { const int c = tx + 16 * b; yv[b] = c < d ? Y[j * st + c] : 0.f; }
#pragma unroll
    for (int a = 0; a < AX; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] += pv[a] * yv[b];
  }
#pragma unroll
  for (int a = 0; a < AX; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int i = ty + 16 * a, c = tx + 16 * b;
      if (i < TX && c < d) out[i * st + c] = acc[a][b];
    }
}

// O[j][c] = sum_i P[i][j] * Y[i][c]    (contraction over the ROWS of P) -> out [TJ][d+1]
template <int AJ>
__device__ __forceinline__ void mm_tn(float* out, const float* P, int ldp, const float* Y, int TI, int TJ, int d, int tid, bool absval) {
  const int ty = tid >> 4, tx = tid & 15;
  float acc[AJ][4];
#pragma unroll
  for (int a = 0; a < AJ; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
  const int st = d + 1;
  for (int i = 0; i < TI; ++i) {
    float pv[AJ], yv[4];
#pragma unroll
    for (int a = 0; a < AJ; ++a) { const int j = ty + 16 * a; pv[a] = j < TJ ? P[i * ldp + j] : 0.f; }
#pragma unroll
    for (int b = 0; b < 4; ++b) { const int c = tx + 16 * b; yv[b] = c < d ? Y[i * st + c] : 0.f; }
#pragma unroll
    for (int a = 0; a < AJ; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] += pv[a] * yv[b];
  }
#pragma unroll
  for (int a = 0; a < AJ; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int j = ty + 16 * a, c = tx + 16 * b;
      if (j < TJ && c < d) out[j * st + c] = acc[a][b];
    }
}

__device__ __forceinline__ uint32_t philox_one(uint64_t seed, uint32_t site, uint64_t idx) {
  const Philox4 p = philox4x32_10(seed, site, idx >> 2);
  const uint32_t k = (uint32_t)idx & 3u;
  return k == 0 ? p.x : (k == 1 ? p.y : (k == 2 ? p.z : p.w));
}

// row softmax over S[Tq][ldS] in place (+ additive key mask), wave per row.
// mode 0 (forward):  S <- dropout(P)           (scaled by 1/(1-p))
// mode 1 (backward): S <- +P if kept, -P if dropped  (sign carries the mask; P >= 0)
__device__ __forceinline__ void softmax_rows(float* S, int ldS, const uint8_t* km, int Tq, int Tk, float scale,
                                             long bh, uint32_t thr, float dscale, uint32_t site, uint64_t seed,
                                             int tid, int mode) {
  const int lane = tid & 63, wave = tid >> 6;
  const long Tkp = (Tk + 3) & ~3;
  for (int i = wave; i < Tq; i += 4) {
    float* row = S + i * ldS;
    const int j0 = lane, j1 = lane + 64;
    float s0 = -INFINITY, s1 = -INFINITY;
    if (j0 < Tk) s0 = row[j0] * scale + (km[j0] ? 0.f : -10000.f);
    if (j1 < Tk) s1 = row[j1] * scale + (km[j1] ? 0.f : -10000.f);
    const float mx = wave_max(fmaxf(s0, s1));
    const float e0 = j0 < Tk ? expf(s0 - mx) : 0.f, e1 = j1 < Tk ? expf(s1 - mx) : 0.f;
    const float inv = 1.0f / wave_sum(e0 + e1);
    float p0 = e0 * inv, p1 = e1 * inv;
    if (thr) {
      const uint64_t base = (uint64_t)(bh * Tq + i) * (uint64_t)Tkp;
      const bool k0 = j0 < Tk ? philox_one(seed, site, base + j0) >= thr : true;
      const bool k1 = j1 < Tk ? philox_one(seed, site, base + j1) >= thr : true;
      if (mode == 0) { p0 = k0 ? p0 * dscale : 0.f; p1 = k1 ? p1 * dscale : 0.f; }
      else { p0 = k0 ? p0 : -p0; p1 = k1 ? p1 : -p1; }
    }
    if (j0 < Tk) row[j0] = p0;
    if (j1 < Tk) row[j1] = p1;
  }
}

template <int AQ, int AK>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
  const int d = a.d, st = d + 1, ldS = a.Tk + 1;
  float* Q = reinterpret_cast<float*>(smem);
  float* K = Q + a.Tq * st;
  float* V = K + a.Tk * st;
  float* S = V + a.Tk * st;
  load_tile(Q, a.q + (long)b * a.Tq * a.ldq + h * d, a.ldq, a.Tq, d, tid);
  load_tile(K, a.k + (long)b * a.Tk * a.ldk + h * d, a.ldk, a.Tk, d, tid);
  load_tile(V, a.v + (long)b * a.Tk * a.ldv + h * d, a.ldv, a.Tk, d, tid);
  __syncthreads();
  mm_nt<AQ, AK>(S, ldS, Q, K, a.Tq, a.Tk, d, tid);
  __syncthreads();
  softmax_rows(S, ldS, a.keymask + (long)b * a.Tk, a.Tq, a.Tk, a.scale, (long)b * a.heads + h, a.thr, a.dscale, a.site, a.seed, tid, 0);
  __syncthreads();
  mm_nn<AQ>(Q, S, ldS, V, a.Tq, a.Tk, d, tid);     // ctx tile overwrites Q (no longer needed)
  __syncthreads();
  store_tile(a.ctx + (long)b * a.Tq * a.ldo + h * d, a.ldo, Q, a.Tq, d, tid, 1.0f);
}

template <int AQ, int AK>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
  const int d = a.d, st = d + 1, ldS = a.Tk + 1;
  const int Tmax = a.Tq > a.Tk ? a.Tq : a.Tk;
  float* X = reinterpret_cast<float*>(smem);      // operand buffer 1: Q, then dO, then Q again
  float* Y = X + Tmax * st;                       // operand buffer 2: K, then V, then K again
  float* Pm = Y + Tmax * st;                      // +-P  (sign = dropout keep mask)
  float* dS = Pm + a.Tq * ldS;                    // dP, then dS
  float* O = dS + a.Tq * ldS;                     // output staging [Tmax][d+1]
  const bf16_t* qg = a.q + (long)b * a.Tq * a.ldq + h * d;
  const bf16_t* kg = a.k + (long)b * a.Tk * a.ldk + h * d;
  const bf16_t* vg = a.v + (long)b * a.Tk * a.ldv + h * d;
  const bf16_t* og = a.dctx + (long)b * a.Tq * a.ldo + h * d;
  // ---- phase a: P = softmax(q k^T)
  load_tile(X, qg, a.ldq, a.Tq, d, tid);
  load_tile(Y, kg, a.ldk, a.Tk, d, tid);
  __syncthreads();
  mm_nt<AQ, AK>(Pm, ldS, X, Y, a.Tq, a.Tk, d, tid);
  __syncthreads();
  softmax_rows(Pm, ldS, a.keymask + (long)b * a.Tk, a.Tq, a.Tk, a.scale, (long)b * a.heads + h, a.thr, a.dscale, a.site, a.seed, tid, 1);
  // ---- phase b: dP = (dO v^T) * mask/keep
  load_tile(X, og, a.ldo, a.Tq, d, tid);
  load_tile(Y, vg, a.ldv, a.Tk, d, tid);
  __syncthreads();
  mm_nt<AQ, AK>(dS, ldS, X, Y, a.Tq, a.Tk, d, tid);
  __syncthreads();
  // ---- phase c/d: per row: delta = sum_j dP*P ; dS = P (dP - delta); Pd = P*mask/keep (for dV) kept in Pm
  {
    const int lane = tid & 63, wave = tid >> 6;
    const float ds = a.thr ? a.dscale : 1.0f;
    for (int i = wave; i < a.Tq; i += 4) {
      float* pr = Pm + i * ldS;
      float* gr = dS + i * ldS;
      const int j0 = lane, j1 = lane + 64;
      float p0 = j0 < a.Tk ? pr[j0] : 0.f, p1 = j1 < a.Tk ? pr[j1] : 0.f;
      float g0 = j0 < a.Tk ? gr[j0] : 0.f, g1 = j1 < a.Tk ? gr[j1] : 0.f;
      const bool k0 = !(__float_as_uint(p0) >> 31), k1 = !(__float_as_uint(p1) >> 31);
      p0 = fabsf(p0); p1 = fabsf(p1);
      g0 = k0 ? g0 * ds : 0.f; g1 = k1 ? g1 * ds : 0.f;          // gradient w.r.t. P (through dropout)
      const float delta = wave_sum(g0 * p0 + g1 * p1);
      if (j0 < a.Tk) { gr[j0] = p0 * (g0 - delta); pr[j0] = k0 ? p0 * ds : 0.f; }
      if (j1 < a.Tk) { gr[j1] = p1 * (g1 - delta); pr[j1] = k1 ? p1 * ds : 0.f; }
    }
  }
  __syncthreads();
  // dV[j][c] = sum_i Pd[i][j] dO[i][c]
  mm_tn<AK>(O, Pm, ldS, X, a.Tq, a.Tk, d, tid, false);
  __syncthreads();
  store_tile(a.dv + (long)b * a.Tk * a.lddv + h * d, a.lddv, O, a.Tk, d, tid, 1.0f);
  // ---- phase e: dQ = dS k * scale ; dK = dS^T q * scale
  load_tile(X, qg, a.ldq, a.Tq, d, tid);
  load_tile(Y, kg, a.ldk, a.Tk, d, tid);
  __syncthreads();
  mm_nn<AQ>(O, dS, ldS, Y, a.Tq, a.Tk, d, tid);
  __syncthreads();
  store_tile(a.dq + (long)b * a.Tq * a.lddq + h * d, a.lddq, O, a.Tq, d, tid, a.scale);
  __syncthreads();
  mm_tn<AK>(O, dS, ldS, X, a.Tq, a.Tk, d, tid, false);
  __syncthreads();
  store_tile(a.dk + (long)b * a.Tk * a.lddk + h * d, a.lddk, O, a.Tk, d, tid, a.scale);
}

inline int tile_class(int T) { return T <= 32 ? 2 : (T <= 48 ? 3 : (T <= 112 ? 7 : -1)); }

template <bool BWD, int AQ, int AK>
hipError_t launch_one(const AttnArgs& a, size_t lds, hipStream_t s) {
  auto kern = BWD ? attn_bwd_kernel<AQ, AK> : attn_fwd_kernel<AQ, AK>;
  static size_t attr_lds = 0;        // raised once per instantiation, outside any stream capture (1st call is eager)
  if (lds > 64 * 1024 && lds > attr_lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_lds = 160 * 1024;
  }
  hipLaunchKernelGGL(kern, dim3(a.B * a.heads), dim3(256), lds, s, a);
  return hipGetLastError();
}

template <bool BWD>
hipError_t dispatch(const AttnArgs& a, size_t lds, hipStream_t s) {
  const int cq = tile_class(a.Tq), ck = tile_class(a.Tk);
#define CASE(Q_, K_) if (cq == Q_ && ck == K_) return launch_one<BWD, Q_, K_>(a, lds, s);
  CASE(2, 2) CASE(2, 3) CASE(2, 7) CASE(3, 2) CASE(3, 3) CASE(3, 7) CASE(7, 2) CASE(7, 3) CASE(7, 7)
#undef CASE
  return hipErrorInvalidValue;
}

int check_args(int Tq, int Tk, int d) {
  CRCT_REQUIRE(Tq >= 1 && Tk >= 1 && Tq <= 112 && Tk <= 112, "attention: Tq=%d Tk=%d must be in [1,112]", Tq, Tk);
  CRCT_REQUIRE(d % 8 == 0 && d >= 8 && d <= 64, "attention: head size %d must be a multiple of 8 in [8,64]", d);
  return 0;
}

}  // namespace

extern "C" int crct_attention_fwd(const void* q, const void* k, const void* v, const uint8_t* keymask, void* ctx, int B,
                                  int heads, int Tq, int Tk, int d, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                                  uint32_t drop_thr, float drop_scale, uint32_t drop_site, uint64_t seed,
                                  crct_stream_t stream) {
  if (int e = check_args(Tq, Tk, d)) return e;
  if (B * heads <= 0) return 0;
  AttnArgs a = {};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.keymask = keymask; a.ctx = (bf16_t*)ctx;
  a.B = B; a.heads = heads; a.Tq = Tq; a.Tk = Tk; a.d = d;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.thr = drop_thr; a.dscale = drop_scale; a.site = drop_site; a.seed = seed;
  a.scale = 1.0f / sqrtf((float)d);
  const size_t lds = sizeof(float) * ((size_t)(Tq + 2 * Tk) * (d + 1) + (size_t)Tq * (Tk + 1));
  CRCT_CHECK_HIP(dispatch<false>(a, lds, (hipStream_t)stream));
  return 0;
}

extern "C" int crct_attention_bwd(const void* q, const void* k, const void* v, const uint8_t* keymask, const void* dctx,
                                  void* dq, void* dk, void* dv, int B, int heads, int Tq, int Tk, int d, int64_t ldq,
                                  int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv,
                                  uint32_t drop_thr, float drop_scale, uint32_t drop_site, uint64_t seed,
                                  crct_stream_t stream) {
  if (int e = check_args(Tq, Tk, d)) return e;
  if (B * heads <= 0) return 0;
  AttnArgs a = {};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.keymask = keymask;
  a.dctx = (const bf16_t*)dctx; a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv;
  a.B = B; a.heads = heads; a.Tq = Tq; a.Tk = Tk; a.d = d;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
  a.thr = drop_thr; a.dscale = drop_scale; a.site = drop_site; a.seed = seed;
  a.scale = 1.0f / sqrtf((float)d);
  const int Tmax = Tq > Tk ? Tq : Tk;
  const size_t lds = sizeof(float) * ((size_t)3 * Tmax * (d + 1) + (size_t)2 * Tq * (Tk + 1));
  CRCT_REQUIRE(lds <= 160 * 1024, "attention_bwd: Tq=%d Tk=%d d=%d needs %zu B of LDS (> 160 KiB)", Tq, Tk, d, lds);
  CRCT_CHECK_HIP(dispatch<true>(a, lds, (hipStream_t)stream));
  return 0;
}
