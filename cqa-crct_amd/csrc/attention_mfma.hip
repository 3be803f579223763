// MFMA attention for the short CRCT sequences: ONE WAVE per (batch, head), sequences up to 64 queries x
// 64 keys, head size 32 / 48 / 64.  Same math and the same dropout stream as attention.hip (which stays
// the path for longer sequences):
//   P = softmax(q k^T / sqrt(d) + (1 - keymask) * -10000) ; ctx = dropout(P) v
// Reference: BertSelfAttention.forward vilbert.py:392-412, BertImageSelfAttention :522-543,
// BertBiAttention :684-723.
//
// Layout idea (cdna_hip_programming.md section 3, "an accumulator tile as the next MFMA's operand"):
// v_mfma_f32_16x16x16_bf16 has the SAME lane map for its A/B operands (row|col = lane & 15,
// k = 4 * (lane >> 4) + e) and for its result (col = lane & 15, row = 4 * (lane >> 4) + r).  The score
// tile is therefore computed TRANSPOSED, S^T = K Q^T: a lane then owns one query (its column) and four
// consecutive keys per 16-key tile, so
//   * the softmax reductions over the keys are in-lane plus two cross-lane steps (xor 16, xor 32),
//   * one Philox4x32 call covers the lane's four keys of a tile (same element numbering as attention.hip),
//   * the probabilities, converted to bf16, ARE the B operand of  ctx^T = V^T P^T  with no data movement.
// Operands whose contraction index is a ROW of the row-major global matrix (V in P V, K in dS K, q in
// dS^T q, dO in P^T dO) are read k-major from a row-major LDS image with ds_read_b64_tr_b16; the two
// products of the backward pass that contract over the queries (dV, dK) take P / dS through a small LDS
// image written from the accumulator layout (8 bytes per lane and tile) and read back transposed.
// Rows of all LDS images are padded by 16 bytes: strides of 80 / 112 / 144 bytes put the 16 rows of a
// fragment read on distinct banks.
#include "common.cuh"
#include "crct_internal.h"
#include "attention_args.h"

namespace {

typedef s4_t __attribute__((address_space(3))) * lds_s4_ptr;

__device__ __forceinline__ f4_t mma16(s4_t a, s4_t b, f4_t c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }

// rows [T][16*ND] bf16 (row stride ld) -> LDS image with `rows` rows (rows >= T are zero)
template <int ND>
__device__ __forceinline__ void load_rows(char* img, const bf16_t* src, long ld, int T, int rows, int lane) {
  constexpr int CPR = 2 * ND, STB = 32 * ND + 16;
  for (int c = lane; c < rows * CPR; c += 64) {
    const int r = c / CPR, cc = (c - r * CPR) << 3;
    uint4 u = make_uint4(0u, 0u, 0u, 0u);
    if (r < T) u = *reinterpret_cast<const uint4*>(src + (long)r * ld + cc);
    *reinterpret_cast<uint4*>(img + r * STB + cc * 2) = u;
  }
}
template <int ND>
__device__ __forceinline__ void store_rows(bf16_t* dst, long ld, const char* img, int T, int lane) {
  constexpr int CPR = 2 * ND, STB = 32 * ND + 16;
  for (int c = lane; c < T * CPR; c += 64) {
    const int r = c / CPR, cc = (c - r * CPR) << 3;
    *reinterpret_cast<uint4*>(dst + (long)r * ld + cc) = *reinterpret_cast<const uint4*>(img + r * STB + cc * 2);
  }
}
// fragment X[r0 + (lane & 15)][c0 + 4 (lane >> 4) + e]: contraction along the image's columns
__device__ __forceinline__ s4_t frag_rows(const char* img, int stb, int r0, int c0, int lane) {
  return *reinterpret_cast<const s4_t*>(img + (r0 + (lane & 15)) * stb + (c0 + 4 * (lane >> 4)) * 2);
}
// fragment X[k0 + 4 (lane >> 4) + e][c0 + (lane & 15)]: contraction along the image's rows (transposed read;
// EXEC must be all ones, every lane supplies an in-bounds address)
__device__ __forceinline__ s4_t frag_cols(const char* img, int stb, int k0, int c0, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const char* addr = img + (k0 + 4 * g + q) * stb + (c0 + 4 * p) * 2;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(uintptr_t)(uint32_t)(uintptr_t)addr);
}
// result tile (rows 4g + r, col lane & 15) -> transposed into a row-major image: img[col][row0 + 4g .. + 3]
__device__ __forceinline__ void put_tile_t(char* img, int stb, int row_of_col0, int col_of_row0, f4_t v, int lane) {
  uint2 u;
  u.x = pack2bf(v[0], v[1]); u.y = pack2bf(v[2], v[3]);
  *reinterpret_cast<uint2*>(img + (row_of_col0 + (lane & 15)) * stb + (col_of_row0 + 4 * (lane >> 4)) * 2) = u;
}
__device__ __forceinline__ s4_t pack4(f4_t v) {
  uint2 u;
  u.x = pack2bf(v[0], v[1]); u.y = pack2bf(v[2], v[3]);
  return __builtin_bit_cast(s4_t, u);
}
__device__ __forceinline__ float xmax2(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xsum2(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// P^T tiles of one 16-query column block `it`: in: raw scores S^T (acc), out: probabilities (before dropout) and
// the keep bits of the lane's 4 keys per tile
template <int NK>
__device__ __forceinline__ void softmax_cols(f4_t (&s)[NK], uint32_t& keep, const uint8_t* km, int Tk, int Tq, int i, long bh,
                                             const AttnArgs& a, int lane) {
  const int g = lane >> 4;
  float mx = -INFINITY;
#pragma unroll
  for (int jt = 0; jt < NK; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 16 * jt + 4 * g + r;
      const float v = j < Tk ? s[jt][r] * a.scale + (km[j] ? 0.f : -10000.f) : -INFINITY;
      s[jt][r] = v;
      mx = fmaxf(mx, v);
    }
  mx = xmax2(mx);
  float sum = 0.f;
#pragma unroll
  for (int jt = 0; jt < NK; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = s[jt][r] == -INFINITY ? 0.f : expf(s[jt][r] - mx);
      s[jt][r] = e;
      sum += e;
    }
  const float inv = 1.0f / xsum2(sum);
  keep = 0xffffffffu;
  const long Tkp = (Tk + 3) & ~3;
#pragma unroll
  for (int jt = 0; jt < NK; ++jt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) s[jt][r] *= inv;
    const int j0 = 16 * jt + 4 * g;
    if (a.thr && j0 < Tk && i < Tq) {
      const Philox4 rnd = philox4x32_10(a.seed, a.site, ((uint64_t)(bh * Tq + i) * (uint64_t)Tkp + (uint64_t)j0) >> 2);
      const uint32_t u[4] = {rnd.x, rnd.y, rnd.z, rnd.w};
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (u[r] < a.thr) keep &= ~(1u << (4 * jt + r));
    }
  }
}

template <int NQ, int NK, int ND>
__global__ __launch_bounds__(64) void attn_fwd_mfma(const AttnArgs a) {
  constexpr int STB = 32 * ND + 16;
  __shared__ __attribute__((aligned(16))) char Qs[16 * NQ * STB];
  __shared__ __attribute__((aligned(16))) char Ks[16 * NK * STB];
  __shared__ __attribute__((aligned(16))) char Vs[16 * NK * STB];
  const int lane = threadIdx.x, g = lane >> 4, n = lane & 15;
  const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads, d = 16 * ND;
  const long bh = (long)b * a.heads + h;
  load_rows<ND>(Qs, a.q + (long)b * a.Tq * a.ldq + h * d, a.ldq, a.Tq, 16 * NQ, lane);
  load_rows<ND>(Ks, a.k + (long)b * a.Tk * a.ldk + h * d, a.ldk, a.Tk, 16 * NK, lane);
  load_rows<ND>(Vs, a.v + (long)b * a.Tk * a.ldv + h * d, a.ldv, a.Tk, 16 * NK, lane);
  __syncthreads();
  const uint8_t* km = a.keymask + (long)b * a.Tk;
  s4_t kf[NK][ND];
#pragma unroll
  for (int jt = 0; jt < NK; ++jt)
#pragma unroll
    for (int ks = 0; ks < ND; ++ks) kf[jt][ks] = frag_rows(Ks, STB, 16 * jt, 16 * ks, lane);
  f4_t o[ND][NQ];
#pragma unroll
  for (int it = 0; it < NQ; ++it) {
    s4_t qf[ND];
#pragma unroll
    for (int ks = 0; ks < ND; ++ks) qf[ks] = frag_rows(Qs, STB, 16 * it, 16 * ks, lane);
    f4_t s[NK];
#pragma unroll
    for (int jt = 0; jt < NK; ++jt) {
      s[jt] = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < ND; ++ks) s[jt] = mma16(kf[jt][ks], qf[ks], s[jt]);      // S^T[j][i]
    }
    uint32_t keep;
    softmax_cols<NK>(s, keep, km, a.Tk, a.Tq, 16 * it + n, bh, a, lane);
    s4_t pb[NK];
#pragma unroll
    for (int jt = 0; jt < NK; ++jt) {
      f4_t p;
#pragma unroll
      for (int r = 0; r < 4; ++r) p[r] = ((keep >> (4 * jt + r)) & 1u) ? (a.thr ? s[jt][r] * a.dscale : s[jt][r]) : 0.f;
      pb[jt] = pack4(p);
    }
#pragma unroll
    for (int ct = 0; ct < ND; ++ct) {
      o[ct][it] = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int jt = 0; jt < NK; ++jt) o[ct][it] = mma16(frag_cols(Vs, STB, 16 * jt, 16 * ct, lane), pb[jt], o[ct][it]);   // ctx^T[c][i]
    }
  }
  __syncthreads();                      // every fragment of Q has been read: its image becomes the output staging tile
#pragma unroll
  for (int it = 0; it < NQ; ++it)
#pragma unroll
    for (int ct = 0; ct < ND; ++ct) put_tile_t(Qs, STB, 16 * it, 16 * ct, o[ct][it], lane);
  __syncthreads();
  store_rows<ND>(a.ctx + (long)b * a.Tq * a.ldo + h * d, a.ldo, Qs, a.Tq, lane);
  (void)g;
}

template <int NQ, int NK, int ND>
__global__ __launch_bounds__(64) void attn_bwd_mfma(const AttnArgs a) {
  constexpr int STB = 32 * ND + 16, NX = NQ > NK ? NQ : NK, PSB = 32 * NK + 16;
  __shared__ __attribute__((aligned(16))) char Qs[16 * NQ * STB];
  __shared__ __attribute__((aligned(16))) char Os[16 * NQ * STB];      // dO
  __shared__ __attribute__((aligned(16))) char Ks[16 * NK * STB];
  __shared__ __attribute__((aligned(16))) char Vs[16 * NX * STB];      // V, later the staging tile of dq / dv / dk
  __shared__ __attribute__((aligned(16))) char Pi[16 * NQ * PSB];      // dropout(P)  [i][j]
  __shared__ __attribute__((aligned(16))) char Di[16 * NQ * PSB];      // dS          [i][j]
  const int lane = threadIdx.x, n = lane & 15;
  const int b = blockIdx.x / a.heads, h = blockIdx.x % a.heads, d = 16 * ND;
  const long bh = (long)b * a.heads + h;
  load_rows<ND>(Qs, a.q + (long)b * a.Tq * a.ldq + h * d, a.ldq, a.Tq, 16 * NQ, lane);
  load_rows<ND>(Os, a.dctx + (long)b * a.Tq * a.ldo + h * d, a.ldo, a.Tq, 16 * NQ, lane);
  load_rows<ND>(Ks, a.k + (long)b * a.Tk * a.ldk + h * d, a.ldk, a.Tk, 16 * NK, lane);
  load_rows<ND>(Vs, a.v + (long)b * a.Tk * a.ldv + h * d, a.ldv, a.Tk, 16 * NK, lane);
  __syncthreads();
  const uint8_t* km = a.keymask + (long)b * a.Tk;
  const float ds = a.thr ? a.dscale : 1.0f;
  f4_t dq[ND][NQ];
  {
    s4_t kf[NK][ND], vf[NK][ND];
#pragma unroll
    for (int jt = 0; jt < NK; ++jt)
#pragma unroll
      for (int ks = 0; ks < ND; ++ks) {
        kf[jt][ks] = frag_rows(Ks, STB, 16 * jt, 16 * ks, lane);
        vf[jt][ks] = frag_rows(Vs, STB, 16 * jt, 16 * ks, lane);
      }
#pragma unroll
    for (int it = 0; it < NQ; ++it) {
      s4_t qf[ND], of[ND];
#pragma unroll
      for (int ks = 0; ks < ND; ++ks) {
        qf[ks] = frag_rows(Qs, STB, 16 * it, 16 * ks, lane);
        of[ks] = frag_rows(Os, STB, 16 * it, 16 * ks, lane);
      }
      f4_t p[NK], gp[NK];
#pragma unroll
      for (int jt = 0; jt < NK; ++jt) {
        p[jt] = f4_t{0.f, 0.f, 0.f, 0.f};
        gp[jt] = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < ND; ++ks) {
          p[jt] = mma16(kf[jt][ks], qf[ks], p[jt]);        // S^T[j][i]
          gp[jt] = mma16(vf[jt][ks], of[ks], gp[jt]);      // (dO v^T)^T[j][i]
        }
      }
      uint32_t keep;
      softmax_cols<NK>(p, keep, km, a.Tk, a.Tq, 16 * it + n, bh, a, lane);
      float part = 0.f;
#pragma unroll
      for (int jt = 0; jt < NK; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          gp[jt][r] = ((keep >> (4 * jt + r)) & 1u) ? gp[jt][r] * ds : 0.f;        // gradient w.r.t. P (through dropout)
          part += gp[jt][r] * p[jt][r];
        }
      const float delta = xsum2(part);
      s4_t dsb[NK];
#pragma unroll
      for (int jt = 0; jt < NK; ++jt) {
        f4_t dsv, pd;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dsv[r] = p[jt][r] * (gp[jt][r] - delta);
          pd[r] = ((keep >> (4 * jt + r)) & 1u) ? p[jt][r] * ds : 0.f;
        }
        dsb[jt] = pack4(dsv);
        put_tile_t(Pi, PSB, 16 * it, 16 * jt, pd, lane);
        put_tile_t(Di, PSB, 16 * it, 16 * jt, dsv, lane);
      }
      // dq^T[c][i] = sum_j k[j][c] dS^T[j][i]
#pragma unroll
      for (int ct = 0; ct < ND; ++ct) {
        dq[ct][it] = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jt = 0; jt < NK; ++jt) dq[ct][it] = mma16(frag_cols(Ks, STB, 16 * jt, 16 * ct, lane), dsb[jt], dq[ct][it]);
      }
    }
  }
  __syncthreads();                      // V fragments are consumed; P / dS images are complete
#pragma unroll
  for (int it = 0; it < NQ; ++it)
#pragma unroll
    for (int ct = 0; ct < ND; ++ct) put_tile_t(Vs, STB, 16 * it, 16 * ct, dq[ct][it] * a.scale, lane);
  __syncthreads();
  store_rows<ND>(a.dq + (long)b * a.Tq * a.lddq + h * d, a.lddq, Vs, a.Tq, lane);
  // dv^T[c][j] = sum_i dO[i][c] Pd[i][j] ; dk^T[c][j] = sum_i q[i][c] dS[i][j]
  f4_t dv[ND][NK], dk[ND][NK];
#pragma unroll
  for (int jt = 0; jt < NK; ++jt) {
    s4_t pf[NQ], sf[NQ];
#pragma unroll
    for (int it = 0; it < NQ; ++it) {
      pf[it] = frag_cols(Pi, PSB, 16 * it, 16 * jt, lane);
      sf[it] = frag_cols(Di, PSB, 16 * it, 16 * jt, lane);
    }
#pragma unroll
    for (int ct = 0; ct < ND; ++ct) {
      dv[ct][jt] = f4_t{0.f, 0.f, 0.f, 0.f};
      dk[ct][jt] = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int it = 0; it < NQ; ++it) {
        dv[ct][jt] = mma16(frag_cols(Os, STB, 16 * it, 16 * ct, lane), pf[it], dv[ct][jt]);
        dk[ct][jt] = mma16(frag_cols(Qs, STB, 16 * it, 16 * ct, lane), sf[it], dk[ct][jt]);
      }
    }
  }
  __syncthreads();                      // dq has left the staging tile
#pragma unroll
  for (int jt = 0; jt < NK; ++jt)
#pragma unroll
    for (int ct = 0; ct < ND; ++ct) put_tile_t(Vs, STB, 16 * jt, 16 * ct, dv[ct][jt], lane);
  __syncthreads();
  store_rows<ND>(a.dv + (long)b * a.Tk * a.lddv + h * d, a.lddv, Vs, a.Tk, lane);
  __syncthreads();
#pragma unroll
  for (int jt = 0; jt < NK; ++jt)
#pragma unroll
    for (int ct = 0; ct < ND; ++ct) put_tile_t(Vs, STB, 16 * jt, 16 * ct, dk[ct][jt] * a.scale, lane);
  __syncthreads();
  store_rows<ND>(a.dk + (long)b * a.Tk * a.lddk + h * d, a.lddk, Vs, a.Tk, lane);
}

template <bool BWD, int NQ, int NK, int ND>
hipError_t launch(const AttnArgs& a, hipStream_t s) {
  if constexpr (BWD) hipLaunchKernelGGL((attn_bwd_mfma<NQ, NK, ND>), dim3(a.B * a.heads), dim3(64), 0, s, a);
  else hipLaunchKernelGGL((attn_fwd_mfma<NQ, NK, ND>), dim3(a.B * a.heads), dim3(64), 0, s, a);
  return hipGetLastError();
}
template <bool BWD, int NQ, int NK>
hipError_t pick_d(const AttnArgs& a, hipStream_t s) {
  switch (a.d) {
    case 32: return launch<BWD, NQ, NK, 2>(a, s);
    case 48: return launch<BWD, NQ, NK, 3>(a, s);
    case 64: return launch<BWD, NQ, NK, 4>(a, s);
  }
  return hipErrorInvalidValue;
}
template <bool BWD, int NQ>
hipError_t pick_k(const AttnArgs& a, hipStream_t s) {
  switch ((a.Tk + 15) / 16) {
    case 1: return pick_d<BWD, NQ, 1>(a, s);
    case 2: return pick_d<BWD, NQ, 2>(a, s);
    case 3: return pick_d<BWD, NQ, 3>(a, s);
    case 4: return pick_d<BWD, NQ, 4>(a, s);
  }
  return hipErrorInvalidValue;
}
template <bool BWD>
hipError_t pick_q(const AttnArgs& a, hipStream_t s) {
  switch ((a.Tq + 15) / 16) {
    case 1: return pick_k<BWD, 1>(a, s);
    case 2: return pick_k<BWD, 2>(a, s);
    case 3: return pick_k<BWD, 3>(a, s);
    case 4: return pick_k<BWD, 4>(a, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace

bool crct_attention_mfma_ok(int Tq, int Tk, int d) { return Tq <= 64 && Tk <= 64 && (d == 32 || d == 48 || d == 64); }
hipError_t crct_attention_mfma_fwd(const AttnArgs& a, hipStream_t s) { return pick_q<false>(a, s); }
hipError_t crct_attention_mfma_bwd(const AttnArgs& a, hipStream_t s) { return pick_q<true>(a, s); }
