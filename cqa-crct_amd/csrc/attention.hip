// Scaled-dot-product attention: the entry points of all three implementations (see pick_path below) and the fp32 VALU kernels
// for the short CRCT sequences (<= 112 visual elements, <= 112 text tokens, any head size that is a multiple of 8 up to 64):
// one 256-thread workgroup per (batch, head); q, k, v, the score
// matrix and the probabilities live in LDS in fp32, all arithmetic is fp32 VALU (the score / PV
// products are < 1 % of the step's FLOPs -- SURVEY.md 8a -- so they do not go to MFMA).
//   P = softmax(q k^T / sqrt(d) + (1 - keymask) * -10000) ; ctx = dropout(P) v
// Reference: BertSelfAttention.forward vilbert.py:392-412, BertImageSelfAttention :522-543,
// BertBiAttention :684-723 (both directions are two calls with q and k/v from different streams).
// The backward pass recomputes P from q, k (nothing but q/k/v is kept from the forward) and
// regenerates the dropout mask from (seed, site, element index).
#include <stdlib.h>

#include "common.hip.h"
#include "crct_internal.h"
#include "attention_args.h"

namespace {

// LDS tiles: fp32 [T4][st], st = d + 4 floats (rows 16-byte aligned; 16 consecutive rows land on 16 distinct
// 16-byte bank slots for d = 32 / 48 / 64), T4 = T rounded up to 4 with the padding rows zero-filled, so that
// every inner loop runs in steps of 4 with vector LDS reads and no tail.
__device__ __forceinline__ int up4(int v) { return (v + 3) & ~3; }

// cooperative load of rows [T][d] (bf16, row stride ld) into the LDS tile; rows T..T4-1 are zeroed
__device__ __forceinline__ void load_tile(float* dst, const bf16_t* src, long ld, int T, int d, int tid) {
  const int cpr = d >> 3, st = d + 4, T4 = up4(T);
  for (int c = tid; c < T4 * cpr; c += 256) {
    const int r = c / cpr, cc = (c % cpr) << 3;
    float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
    if (r < T) {
      const uint4 u = *reinterpret_cast<const uint4*>(src + (long)r * ld + cc);
      lo = make_float4(bf2f((bf16_t)(u.x & 0xffff)), bf2f((bf16_t)(u.x >> 16)), bf2f((bf16_t)(u.y & 0xffff)), bf2f((bf16_t)(u.y >> 16)));
      hi = make_float4(bf2f((bf16_t)(u.z & 0xffff)), bf2f((bf16_t)(u.z >> 16)), bf2f((bf16_t)(u.w & 0xffff)), bf2f((bf16_t)(u.w >> 16)));
    }
    float* o = dst + r * st + cc;
    *reinterpret_cast<float4*>(o) = lo;
    *reinterpret_cast<float4*>(o + 4) = hi;
  }
}
// cooperative store LDS tile * mul -> bf16 rows
__device__ __forceinline__ void store_tile(bf16_t* dst, long ld, const float* src, int T, int d, int tid, float mul) {
  const int cpr = d >> 3, st = d + 4;
  for (int c = tid; c < T * cpr; c += 256) {
    const int r = c / cpr, cc = (c % cpr) << 3;
    const float4 lo = *reinterpret_cast<const float4*>(src + r * st + cc), hi = *reinterpret_cast<const float4*>(src + r * st + cc + 4);
    uint4 u;
    u.x = pack2bf(lo.x * mul, lo.y * mul); u.y = pack2bf(lo.z * mul, lo.w * mul);
    u.z = pack2bf(hi.x * mul, hi.y * mul); u.w = pack2bf(hi.z * mul, hi.w * mul);
    *reinterpret_cast<uint4*>(dst + (long)r * ld + cc) = u;
  }
}

// C[i][j] = sum_c X[i][c] * Y[j][c] -> out[i*ldo + j].  16x16 thread grid, AX x AY register tile per thread
// (rows ty + 16a, cols tx + 16b); the contraction runs 4 wide on float4 LDS reads, two steps in flight.
template <int AX, int AY>
__device__ __forceinline__ void mm_nt(float* out, int ldo, const float* X, const float* Y, int TX, int TY, int d, int tid) {
  const int ty = tid >> 4, tx = tid & 15, st = d + 4;
  const int TX4 = up4(TX), TY4 = up4(TY);
  const float* xp[AX];
  const float* yp[AY];
#pragma unroll
  for (int a = 0; a < AX; ++a) xp[a] = X + min(ty + 16 * a, TX4 - 1) * st;
#pragma unroll
  for (int b = 0; b < AY; ++b) yp[b] = Y + min(tx + 16 * b, TY4 - 1) * st;
  float acc[AX][AY];
#pragma unroll
  for (int a = 0; a < AX; ++a)
#pragma unroll
    for (int b = 0; b < AY; ++b) acc[a][b] = 0.f;
#pragma unroll 2
  for (int c = 0; c < d; c += 4) {
    float4 xv[AX], yv[AY];
#pragma unroll
    for (int a = 0; a < AX; ++a) xv[a] = *reinterpret_cast<const float4*>(xp[a] + c);
#pragma unroll
    for (int b = 0; b < AY; ++b) yv[b] = *reinterpret_cast<const float4*>(yp[b] + c);
#pragma unroll
    for (int a = 0; a < AX; ++a)
#pragma unroll
      for (int b = 0; b < AY; ++b)
        acc[a][b] += xv[a].x * yv[b].x + xv[a].y * yv[b].y + xv[a].z * yv[b].z + xv[a].w * yv[b].w;
  }
#pragma unroll
  for (int a = 0; a < AX; ++a)
#pragma unroll
    for (int b = 0; b < AY; ++b) {
      const int i = ty + 16 * a, j = tx + 16 * b;
      if (i < TX && j < TY) out[i * ldo + j] = acc[a][b];
    }
}

// O[i][c] = sum_j P[i][j] * Y[j][c]  (P: [TX4][ldp], columns >= TJ zero; Y: [TJ4][st], rows >= TJ zero) -> out tile
template <int AX>
__device__ __forceinline__ void mm_nn(float* out, const float* P, int ldp, const float* Y, int TX, int TJ, int d, int tid) {
  const int ty = tid >> 4, tx = tid & 15, st = d + 4;
  const int TX4 = up4(TX), TJ4 = up4(TJ);
  const float* pp[AX];
  int cb[4];
#pragma unroll
  for (int a = 0; a < AX; ++a) pp[a] = P + min(ty + 16 * a, TX4 - 1) * ldp;
#pragma unroll
  for (int b = 0; b < 4; ++b) cb[b] = min(tx + 16 * b, d - 1);
  float acc[AX][4];
#pragma unroll
  for (int a = 0; a < AX; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
  for (int j = 0; j < TJ4; j += 4) {
    float4 pv[AX];
    float yv[4][4];
#pragma unroll
    for (int a = 0; a < AX; ++a) pv[a] = *reinterpret_cast<const float4*>(pp[a] + j);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int b = 0; b < 4; ++b) yv[jj][b] = Y[(j + jj) * st + cb[b]];
#pragma unroll
    for (int a = 0; a < AX; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        acc[a][b] += pv[a].x * yv[0][b] + pv[a].y * yv[1][b] + pv[a].z * yv[2][b] + pv[a].w * yv[3][b];
  }
#pragma unroll
  for (int a = 0; a < AX; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int i = ty + 16 * a, c = tx + 16 * b;
      if (i < TX && c < d) out[i * st + c] = acc[a][b];
    }
}

// O[j][c] = sum_i P[i][j] * Y[i][c]  (contraction over the ROWS of P; rows >= TI of P and Y are zero) -> out tile
template <int AJ>
__device__ __forceinline__ void mm_tn(float* out, const float* P, int ldp, const float* Y, int TI, int TJ, int d, int tid) {
  const int ty = tid >> 4, tx = tid & 15, st = d + 4;
  const int TI4 = up4(TI);
  int ja[AJ], cb[4];
#pragma unroll
  for (int a = 0; a < AJ; ++a) ja[a] = min(ty + 16 * a, ldp - 1);
#pragma unroll
  for (int b = 0; b < 4; ++b) cb[b] = min(tx + 16 * b, d - 1);
  float acc[AJ][4];
#pragma unroll
  for (int a = 0; a < AJ; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
  for (int i = 0; i < TI4; i += 4) {
    float pv[4][AJ], yv[4][4];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
#pragma unroll
      for (int a = 0; a < AJ; ++a) pv[ii][a] = P[(i + ii) * ldp + ja[a]];
#pragma unroll
      for (int b = 0; b < 4; ++b) yv[ii][b] = Y[(i + ii) * st + cb[b]];
    }
#pragma unroll
    for (int a = 0; a < AJ; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        acc[a][b] += pv[0][a] * yv[0][b] + pv[1][a] * yv[1][b] + pv[2][a] * yv[2][b] + pv[3][a] * yv[3][b];
  }
#pragma unroll
  for (int a = 0; a < AJ; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int j = ty + 16 * a, c = tx + 16 * b;
      if (j < TJ && c < d) out[j * st + c] = acc[a][b];
    }
}

// 8-lane group reductions (3 butterfly steps instead of 6 for a whole wave)
__device__ __forceinline__ float group8_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1, 64)); v = fmaxf(v, __shfl_xor(v, 2, 64)); v = fmaxf(v, __shfl_xor(v, 4, 64));
  return v;
}
__device__ __forceinline__ float group8_sum(float v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
  return v;
}

// row softmax over S[Tq][ldS] in place (+ additive key mask).  Eight lanes share a row (32 rows per
// 256-thread pass); lane g of the group owns the C = 4*ceil(Tk/32) consecutive columns g*C .. g*C+C-1, so one
// Philox call covers 4 of its elements and the row reductions are 3 shuffle steps.
// mode 0 (forward):  S <- dropout(P)           (scaled by 1/(1-p))
// mode 1 (backward): S <- +P if kept, -P if dropped  (sign carries the mask; P >= 0)
template <int CQ>     // CQ = C / 4 (1..4)
__device__ __forceinline__ void softmax_rows_c(float* S, int ldS, const uint8_t* km, int Tq, int Tk, float scale,
                                               long bh, uint32_t thr, float dscale, uint32_t site, uint64_t seed,
                                               int tid, int mode) {
  constexpr int C = 4 * CQ;
  const int g = tid & 7;
  for (int i = tid >> 3; i < ((Tq + 31) & ~31); i += 32) {      // whole groups stay converged for the shuffles
    const bool live = i < Tq;
    float* row = S + (live ? i : 0) * ldS;
    float v[C];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int j = g * C + c;
      v[c] = (live && j < Tk) ? row[j] * scale + (km[j] ? 0.f : -10000.f) : -INFINITY;
      mx = fmaxf(mx, v[c]);
    }
    mx = group8_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { v[c] = v[c] == -INFINITY ? 0.f : expf(v[c] - mx); sum += v[c]; }
    const float inv = 1.0f / group8_sum(sum);
    if (!live) continue;
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
      const int j = g * C + 4 * q;
      if (j >= ldS) continue;
      float p[4] = {v[4 * q] * inv, v[4 * q + 1] * inv, v[4 * q + 2] * inv, v[4 * q + 3] * inv};
      if (thr && j < Tk) {
        // this quad = keys j .. j + 3 = the keys of MFMA lane group (j >> 2) & 3 in key tile j >> 4: one half of call (pair j >> 5, group)
        const uint32_t kb = attn_keep8(seed, site, bh, Tq, Tk, i, j >> 5, (j >> 2) & 3, thr) >> (((j >> 4) & 1) * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const bool keep = (kb >> c) & 1u;
          p[c] = mode == 0 ? (keep ? p[c] * dscale : 0.f) : (keep ? p[c] : -p[c]);
        }
      }
      // columns Tk..ldS-1 are zero padding for the 4-wide loops (exp of -inf above gave 0)
      *reinterpret_cast<float4*>(row + j) = make_float4(p[0], p[1], p[2], p[3]);
    }
  }
}
__device__ __forceinline__ void softmax_rows(float* S, int ldS, const uint8_t* km, int Tq, int Tk, float scale,
                                             long bh, uint32_t thr, float dscale, uint32_t site, uint64_t seed,
                                             int tid, int mode) {
  if (Tk <= 32) softmax_rows_c<1>(S, ldS, km, Tq, Tk, scale, bh, thr, dscale, site, seed, tid, mode);
  else if (Tk <= 64) softmax_rows_c<2>(S, ldS, km, Tq, Tk, scale, bh, thr, dscale, site, seed, tid, mode);
  else if (Tk <= 96) softmax_rows_c<3>(S, ldS, km, Tq, Tk, scale, bh, thr, dscale, site, seed, tid, mode);
  else softmax_rows_c<4>(S, ldS, km, Tq, Tk, scale, bh, thr, dscale, site, seed, tid, mode);
}

// backward: per row  delta = sum_j dP*P ; dS = P (dP - delta) ; Pm <- P*mask/keep (for dV).  Same 8-lane mapping.
template <int CQ>
__device__ __forceinline__ void softmax_bwd_rows_c(float* Pm, float* dS, int ldS, int Tq, int Tk, float ds, int tid) {
  constexpr int C = 4 * CQ;
  const int g = tid & 7;
  for (int i = tid >> 3; i < ((Tq + 31) & ~31); i += 32) {
    const bool live = i < Tq;
    float* pr = Pm + (live ? i : 0) * ldS;
    float* gr = dS + (live ? i : 0) * ldS;
    float p[C], gq[C];
    bool kp[C];
    float part = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int j = g * C + c;
      const bool in = live && j < Tk;
      const float pv = in ? pr[j] : 0.f, gv = in ? gr[j] : 0.f;
      kp[c] = !(__float_as_uint(pv) >> 31);
      p[c] = fabsf(pv);
      gq[c] = kp[c] ? gv * ds : 0.f;                 // gradient w.r.t. P (through dropout)
      part += gq[c] * p[c];
    }
    const float delta = group8_sum(part);
    if (!live) continue;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int j = g * C + c;
      if (j < ldS) {
        gr[j] = j < Tk ? p[c] * (gq[c] - delta) : 0.f;
        pr[j] = (j < Tk && kp[c]) ? p[c] * ds : 0.f;
      }
    }
  }
}

template <int AQ, int AK>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
  const int d = a.d, st = d + 4, ldS = up4(a.Tk), Tq4 = up4(a.Tq), Tk4 = up4(a.Tk);
  float* Q = reinterpret_cast<float*>(smem);
  float* K = Q + Tq4 * st;
  float* V = K + Tk4 * st;
  float* S = V + Tk4 * st;
  load_tile(Q, a.q + (long)b * a.Tq * a.ldq + h * d, a.ldq, a.Tq, d, tid);
  load_tile(K, a.k + (long)b * a.Tk * a.ldk + h * d, a.ldk, a.Tk, d, tid);
  load_tile(V, a.v + (long)b * a.Tk * a.ldv + h * d, a.ldv, a.Tk, d, tid);
  __syncthreads();
#ifdef CRCT_ATTN_LAB   // timing ablations of tools/attn_lab only (wrong results); never in the shipped library
  if (!(a.dbg & 1))
#endif
  mm_nt<AQ, AK>(S, ldS, Q, K, a.Tq, a.Tk, d, tid);
  __syncthreads();
#ifdef CRCT_ATTN_LAB
  if (!(a.dbg & 2))
#endif
  softmax_rows(S, ldS, a.keymask + (long)b * a.Tk, a.Tq, a.Tk, a.scale, (long)b * a.heads + h, a.thr, a.dscale, a.site, a.seed, tid, 0);
  __syncthreads();
#ifdef CRCT_ATTN_LAB
  if (!(a.dbg & 4))
#endif
  mm_nn<AQ>(Q, S, ldS, V, a.Tq, a.Tk, d, tid);     // ctx tile overwrites Q (no longer needed)
  __syncthreads();
  store_tile(a.ctx + (long)b * a.Tq * a.ldo + h * d, a.ldo, Q, a.Tq, d, tid, 1.0f);
}

template <int AQ, int AK>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
  const int d = a.d, st = d + 4, ldS = up4(a.Tk), Tq4 = up4(a.Tq);
  const int Tmax4 = up4(a.Tq > a.Tk ? a.Tq : a.Tk);
  float* X = reinterpret_cast<float*>(smem);      // operand buffer 1: Q, then dO, then Q again
  float* Y = X + Tmax4 * st;                      // operand buffer 2: K, then V, then K again
  float* O = Y + Tmax4 * st;                      // output staging tile
  float* Pm = O + Tmax4 * st;                     // +-P  (sign = dropout keep mask)
  float* dS = Pm + Tq4 * ldS;                     // dP, then dS
  for (int i = a.Tq * ldS + tid; i < Tq4 * ldS; i += 256) { Pm[i] = 0.f; dS[i] = 0.f; }   // padding rows of the score tiles
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
    const float ds = a.thr ? a.dscale : 1.0f;
    if (a.Tk <= 32) softmax_bwd_rows_c<1>(Pm, dS, ldS, a.Tq, a.Tk, ds, tid);
    else if (a.Tk <= 64) softmax_bwd_rows_c<2>(Pm, dS, ldS, a.Tq, a.Tk, ds, tid);
    else if (a.Tk <= 96) softmax_bwd_rows_c<3>(Pm, dS, ldS, a.Tq, a.Tk, ds, tid);
    else softmax_bwd_rows_c<4>(Pm, dS, ldS, a.Tq, a.Tk, ds, tid);
  }
  __syncthreads();
  // dV[j][c] = sum_i Pd[i][j] dO[i][c]
  mm_tn<AK>(O, Pm, ldS, X, a.Tq, a.Tk, d, tid);
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
  mm_tn<AK>(O, dS, ldS, X, a.Tq, a.Tk, d, tid);
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
  crct_launch(kern, dim3(a.B * a.heads), dim3(256), lds, s, a);
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

// Three implementations behind one pair of entry points:
//   attention_mfma.hip  register-resident MFMA kernels: Tq, Tk <= 112, d in {32, 48, 64}            (the default where it applies)
//   attention_long.hip  key-tile loop, online softmax:  Tq, Tk <= CRCT_ATTN_MAX_LEN, d in {32, 48, 64} (beyond 112; everywhere it
//                       applies after crct_attention_force_long(1))
//   this file           fp32 VALU, everything in LDS:   Tq, Tk <= 112, any d % 8 == 0 up to 64        (other head sizes; everything
//                       it can take after crct_attention_force_valu(1))
int g_force_valu = 0, g_force_long = 0;
enum { PATH_NONE = 0, PATH_MFMA, PATH_LONG, PATH_VALU };
int pick_path(int Tq, int Tk, int d) {
  const bool valu_ok = Tq <= 112 && Tk <= 112;
  if (g_force_valu && valu_ok) return PATH_VALU;
  if (g_force_long && crct_attention_long_ok(Tq, Tk, d)) return PATH_LONG;
  if (crct_attention_mfma_ok(Tq, Tk, d)) return PATH_MFMA;
  if (crct_attention_long_ok(Tq, Tk, d)) return PATH_LONG;
  return valu_ok ? PATH_VALU : PATH_NONE;
}
int check_args(int Tq, int Tk, int d) {
  CRCT_REQUIRE(Tq >= 1 && Tk >= 1 && Tq <= CRCT_ATTN_MAX_LEN && Tk <= CRCT_ATTN_MAX_LEN, "attention: Tq=%d Tk=%d must be in [1,%d]", Tq, Tk,
               CRCT_ATTN_MAX_LEN);
  CRCT_REQUIRE(d % 8 == 0 && d >= 8 && d <= 64, "attention: head size %d must be a multiple of 8 in [8,64]", d);
  CRCT_REQUIRE(pick_path(Tq, Tk, d) != PATH_NONE, "attention: Tq=%d Tk=%d beyond 112 needs a head size of 32, 48 or 64 (got %d)", Tq, Tk, d);
  return 0;
}
bool use_mfma(int Tq, int Tk, int d) { const int p = pick_path(Tq, Tk, d); return p == PATH_MFMA || p == PATH_LONG; }

}  // namespace

extern "C" void crct_attention_force_valu(int on) { g_force_valu = on ? 1 : 0; }
extern "C" void crct_attention_force_long(int on) { g_force_long = on ? 1 : 0; }

extern "C" int crct_attention_quant_ok(int Tq, int Tk, int d) { return use_mfma(Tq, Tk, d) ? 1 : 0; }

static int attention_fwd_impl(const void* q, const void* k, const void* v, const uint8_t* keymask, void* ctx, int B,
                              int heads, int Tq, int Tk, int d, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                              uint32_t drop_thr, float drop_scale, uint32_t drop_site, uint64_t seed, const CrctAttnQuant* qz,
                              crct_stream_t stream) {
  if (int e = check_args(Tq, Tk, d)) return e;
  if (B * heads <= 0) return 0;
  AttnArgs a = {};
  if (qz && qz->ctx_q) {
    CRCT_REQUIRE(use_mfma(Tq, Tk, d), "attention_fwd_q: the fp8 copy exists in the MFMA kernels only (Tq=%d Tk=%d d=%d)", Tq, Tk, d);
    CRCT_REQUIRE(qz->ctx_scale && qz->ctx_amax && ldo % 8 == 0, "attention_fwd_q: scale / amax missing or ldo %% 8 != 0");
    a.ctx_q = (uint8_t*)qz->ctx_q; a.ctx_qscale = qz->ctx_scale; a.ctx_qamax = qz->ctx_amax;
  }
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.keymask = keymask; a.ctx = (bf16_t*)ctx;
  a.B = B; a.heads = heads; a.Tq = Tq; a.Tk = Tk; a.d = d;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
  a.thr = drop_thr; a.dscale = drop_scale; a.site = drop_site; a.seed = seed;
  a.scale = 1.0f / sqrtf((float)d);
  if (qz && qz->row_lse) a.lse = qz->row_lse;          // written by the long-sequence kernels only (CrctAttnQuant)
#ifdef CRCT_ATTN_LAB
  static const int dbg = getenv("CRCT_ATTN_DBG") ? atoi(getenv("CRCT_ATTN_DBG")) : 0;
  a.dbg = dbg;
#endif
  const int path = pick_path(Tq, Tk, d);
  if (path == PATH_MFMA) {
    CRCT_CHECK_HIP(crct_attention_mfma_fwd(a, (hipStream_t)stream));
    return 0;
  }
  if (path == PATH_LONG) {
    CRCT_CHECK_HIP(crct_attention_long_fwd(a, (hipStream_t)stream));
    return 0;
  }
  const int Tq4 = (Tq + 3) & ~3, Tk4 = (Tk + 3) & ~3;
  const size_t lds = sizeof(float) * ((size_t)(Tq4 + 2 * Tk4) * (d + 4) + (size_t)Tq4 * Tk4);
  CRCT_CHECK_HIP(dispatch<false>(a, lds, (hipStream_t)stream));
  return 0;
}
extern "C" int crct_attention_fwd(const void* q, const void* k, const void* v, const uint8_t* keymask, void* ctx, int B,
                                  int heads, int Tq, int Tk, int d, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                                  uint32_t drop_thr, float drop_scale, uint32_t drop_site, uint64_t seed,
                                  crct_stream_t stream) {
  return attention_fwd_impl(q, k, v, keymask, ctx, B, heads, Tq, Tk, d, ldq, ldk, ldv, ldo, drop_thr, drop_scale, drop_site, seed, nullptr, stream);
}
extern "C" int crct_attention_fwd_q(const void* q, const void* k, const void* v, const uint8_t* keymask, void* ctx, int B,
                                    int heads, int Tq, int Tk, int d, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                                    uint32_t drop_thr, float drop_scale, uint32_t drop_site, uint64_t seed, const CrctAttnQuant* qz,
                                    crct_stream_t stream) {
  return attention_fwd_impl(q, k, v, keymask, ctx, B, heads, Tq, Tk, d, ldq, ldk, ldv, ldo, drop_thr, drop_scale, drop_site, seed, qz, stream);
}

static int attention_bwd_impl(const void* q, const void* k, const void* v, const uint8_t* keymask, const void* dctx,
                              void* dq, void* dk, void* dv, int B, int heads, int Tq, int Tk, int d, int64_t ldq,
                              int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv,
                              uint32_t drop_thr, float drop_scale, uint32_t drop_site, uint64_t seed, const CrctAttnQuant* qz,
                              crct_stream_t stream) {
  if (int e = check_args(Tq, Tk, d)) return e;
  if (B * heads <= 0) return 0;
  AttnArgs a = {};
  if (qz && (qz->dq_q || qz->dk_q || qz->dv_q)) {
    CRCT_REQUIRE(use_mfma(Tq, Tk, d), "attention_bwd_q: the fp8 copies exist in the MFMA kernels only (Tq=%d Tk=%d d=%d)", Tq, Tk, d);
    CRCT_REQUIRE((!qz->dq_q || (qz->dq_scale && qz->dq_amax)) && ((!qz->dk_q && !qz->dv_q) || (qz->dkv_scale && qz->dkv_amax)),
                 "attention_bwd_q: scale / amax missing");
    CRCT_REQUIRE(lddq % 8 == 0 && lddk % 8 == 0 && lddv % 8 == 0, "attention_bwd_q: gradient leading dimensions must be multiples of 8");
    a.dq_q = (uint8_t*)qz->dq_q; a.dk_q = (uint8_t*)qz->dk_q; a.dv_q = (uint8_t*)qz->dv_q;
    a.dq_qscale = qz->dq_scale; a.dq_qamax = qz->dq_amax; a.dkv_qscale = qz->dkv_scale; a.dkv_qamax = qz->dkv_amax;
  }
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.keymask = keymask;
  a.dctx = (const bf16_t*)dctx; a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv;
  a.B = B; a.heads = heads; a.Tq = Tq; a.Tk = Tk; a.d = d;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
  a.thr = drop_thr; a.dscale = drop_scale; a.site = drop_site; a.seed = seed;
  a.scale = 1.0f / sqrtf((float)d);
  if (qz && (qz->row_lse || qz->ctx)) {
    CRCT_REQUIRE(qz->row_lse && qz->ctx && qz->ld_ctx % 4 == 0, "attention_bwd_q: row_lse and ctx (ld_ctx %% 4 == 0) come together");
    a.lse = qz->row_lse; a.ctx = (bf16_t*)qz->ctx; a.ldc = qz->ld_ctx;      // read only
  }
  const int path = pick_path(Tq, Tk, d);
  if (path == PATH_MFMA) {
    CRCT_CHECK_HIP(crct_attention_mfma_bwd(a, (hipStream_t)stream));
    return 0;
  }
  if (path == PATH_LONG) {
    CRCT_CHECK_HIP(crct_attention_long_bwd(a, (hipStream_t)stream));
    return 0;
  }
  const int Tq4 = (Tq + 3) & ~3, Tk4 = (Tk + 3) & ~3, Tmax4 = Tq4 > Tk4 ? Tq4 : Tk4;
  const size_t lds = sizeof(float) * ((size_t)3 * Tmax4 * (d + 4) + (size_t)2 * Tq4 * Tk4);
  CRCT_REQUIRE(lds <= 160 * 1024, "attention_bwd: Tq=%d Tk=%d d=%d needs %zu B of LDS (> 160 KiB)", Tq, Tk, d, lds);
  CRCT_CHECK_HIP(dispatch<true>(a, lds, (hipStream_t)stream));
  return 0;
}
extern "C" int crct_attention_bwd(const void* q, const void* k, const void* v, const uint8_t* keymask, const void* dctx,
                                  void* dq, void* dk, void* dv, int B, int heads, int Tq, int Tk, int d, int64_t ldq,
                                  int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv,
                                  uint32_t drop_thr, float drop_scale, uint32_t drop_site, uint64_t seed,
                                  crct_stream_t stream) {
  return attention_bwd_impl(q, k, v, keymask, dctx, dq, dk, dv, B, heads, Tq, Tk, d, ldq, ldk, ldv, ldo, lddq, lddk, lddv, drop_thr, drop_scale,
                            drop_site, seed, nullptr, stream);
}
extern "C" int crct_attention_bwd_q(const void* q, const void* k, const void* v, const uint8_t* keymask, const void* dctx,
                                    void* dq, void* dk, void* dv, int B, int heads, int Tq, int Tk, int d, int64_t ldq,
                                    int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv,
                                    uint32_t drop_thr, float drop_scale, uint32_t drop_site, uint64_t seed, const CrctAttnQuant* qz,
                                    crct_stream_t stream) {
  return attention_bwd_impl(q, k, v, keymask, dctx, dq, dk, dv, B, heads, Tq, Tk, d, ldq, ldk, ldv, ldo, lddq, lddk, lddv, drop_thr, drop_scale,
                            drop_site, seed, qz, stream);
}
