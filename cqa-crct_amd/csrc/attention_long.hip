// MFMA attention for sequences beyond the 112 x 112 register-resident kernels of attention_mfma.hip: the reference's own PlotQA
// shape (config/plotqa.json:5-6: 124 text tokens x 44 visual elements; options.py:27 defaults to 256 tokens), up to
// CRCT_ATTN_MAX_LEN = 512 queries x 512 keys (the text position table of config/vilbert.json) at head sizes 32 / 48 / 64.  Same math, same -10000 additive key mask, same
// Philox element numbering (and therefore the same dropout masks) as attention.hip / attention_mfma.hip:
//   P = softmax(q k^T / sqrt(d) + (1 - keymask) * -10000) ; ctx = dropout(P) v
// Reference: BertSelfAttention.forward vilbert.py:392-412, BertImageSelfAttention :522-543, BertBiAttention :684-723.
//
// One workgroup of NW waves per (batch, head); the tile counts are RUN-TIME values (one instantiation per head size), the
// loops over 16-key / 16-query tiles keep a constant register footprint.
//
// Forward: K and V of the pair as row-major LDS images; a wave owns 16 queries at a time (Q fragments straight from global
// memory), sweeps the key tiles two at a time with an ONLINE softmax (running maximum m, running sum l, the context
// accumulator rescaled by exp2(m_old - m_new)) and divides by l at the end.  As in attention_mfma.hip the score tile is
// computed transposed (S^T = K Q^T): a lane owns one query and four consecutive keys per tile, its probabilities are the B
// operand of ctx^T = V^T P^T as they lie.
//
// Backward (q, k, v; and, when the caller kept them, the forward's output and row statistics -- see LSE below): Q, dO, K, V as LDS images (beyond 256 x 256 x 64: K, V for phase A and Q, dO
// in the same space for phase B, a wave's OWN tile straight from global memory), then
//   phase A, a wave per QUERY tile: sweep 1 over the key tiles gives the row statistics -- m, 1 / l and
//            delta_i = sum_j P_ij dP_ij, accumulated online like l -- and keeps the dropout bits of the tile row (4 bits per
//            lane and key tile); sweep 2 recomputes S^T and dP^T per key tile, dS^T = P^T (dP^T - delta) and accumulates
//            dq^T += K^T dS^T (dS^T is the B operand as it lies).  The statistics go to LDS.
//   phase B, a wave per KEY tile: for every query tile S^T, dP^T again, P and dS from the statistics; both tiles pass through
//            a 16 x 16 LDS tile of the wave (written from the accumulator layout, read back transposed) to become the B
//            operands of dv^T += dO^T Pd and dk^T += Q^T dS.
//   LSE (the forward wrote lse_i = log2 sum_j exp2(s_ij), and its output is still there -- the step engine keeps both): sweep 1 is
//            dropped.  P_ij = exp2(s_ij - lse_i) needs no running maximum, and delta_i = sum_j P_ij dP_ij = dO_i . O_i (with dropout
//            too: O = (mask o P / (1 - p)) V) is a d-element dot product per query.  A quarter of the kernel's instructions.
// Every output element is accumulated by ONE wave in a fixed order: no atomics, bit-reproducible.  The price is 9 d / 16 MFMAs
// per tile pair instead of the minimal 5 d / 16 -- on < 3 % of the step's arithmetic.
#include "common.hip.h"
#include "crct_internal.h"
#include "attention_args.h"
#include "attention_tiles.hip.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr int SCR_STB = 48;                  // row stride of a wave's 16 x 16 bf16 transposition tile (32 B of data + 16)
constexpr int LDS_CAP = 160 * 1024;

// rows [T][16 * ND] bf16 (row stride ld) -> row-major LDS image of `rows` rows (rows >= T zero), all threads of the workgroup;
// four 16-byte loads in flight per thread (unconditional: a row past T re-reads row T - 1 and is masked to zero)
template <int ND>
__device__ __forceinline__ void load_image(char* img, const bf16_t* __restrict__ src, long ld, int T, int rows, int tid, int nthr) {
  constexpr int CPR = 2 * ND, STB = 32 * ND + 16;
  const int total = rows * CPR;
  for (int c0 = tid; c0 < total; c0 += 4 * nthr) {
    uint4 u[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + k * nthr, r = c / CPR, cc = (c - r * CPR) << 3;
      const uint32_t m = r < T ? 0xffffffffu : 0u;
      const uint4 t = *reinterpret_cast<const uint4*>(src + (long)min(r, T - 1) * ld + cc);
      u[k] = make_uint4(t.x & m, t.y & m, t.z & m, t.w & m);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + k * nthr, r = c / CPR, cc = (c - r * CPR) << 3;
      if (c < total) *reinterpret_cast<uint4*>(img + r * STB + cc * 2) = u[k];
    }
  }
}
// additive key term in the exp2 domain: 0 (attended), -10000 log2 e (masked), -inf (no such key: padding of the last tile)
__device__ __forceinline__ void load_keybias(float* kb, const uint8_t* km, int Tk, int n, int tid, int nthr) {
  for (int j = tid; j < n; j += nthr) kb[j] = j < Tk ? (km[j] ? 0.f : -10000.f * LOG2E) : -INFINITY;
}
// the lane's 8 keep bits of the key-tile PAIR jp for query i (bit 4 u + r: key 16 (2 jp + u) + 4 g + r is kept): one Philox call, the
// numbering all three implementations share (attention_args.h, attn_keep8)
__device__ __forceinline__ uint32_t keep_byte(const AttnArgs& a, long bh, int i, int jp, int g) {
  uint32_t kb = 0xffu;
  if (a.thr && 32 * jp + 4 * g < a.Tk && i < a.Tq) kb = attn_keep8(a.seed, a.site, bh, a.Tq, a.Tk, i, jp, g, a.thr);
  return kb;
}
// Result tile t (lane: column `row` of the transposed product, its rows 4 g + r = columns col0 .. col0 + 3 of the row-major matrix)
// -> 8 bytes of bf16 to global memory, optionally with the fp8 copy of the bf16-rounded values (E4M3: OCP e4m3, else e5m2;
// saturating) and their running maximum in `am` (include/crct_hip.h CrctAttnQuant; same convention as attention_mfma.hip)
template <bool E4M3>
__device__ __forceinline__ void store_tile_t(bf16_t* dst, uint8_t* qdst, long ld, long row, int col0, f4_t t, const float* qscale, float& am) {
  const uint2 pk = make_uint2(pack2bf(t[0], t[1]), pack2bf(t[2], t[3]));
  *reinterpret_cast<uint2*>(dst + row * ld + col0) = pk;
  if (qdst) {
    constexpr float LIM = E4M3 ? 448.f : 57344.f;
    const float qs = qscale[0];
    float f[4] = {bf2f((bf16_t)(pk.x & 0xffff)), bf2f((bf16_t)(pk.x >> 16)), bf2f((bf16_t)(pk.y & 0xffff)), bf2f((bf16_t)(pk.y >> 16))};
#pragma unroll
    for (int r = 0; r < 4; ++r) { am = fmaxf(am, fabsf(f[r])); f[r] = fminf(fmaxf(f[r] * qs, -LIM), LIM); }
    uint32_t o = 0u;
    if constexpr (E4M3) {
      o = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], o, false);
      o = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], o, true);
    } else {
      o = __builtin_amdgcn_cvt_pk_bf8_f32(f[0], f[1], o, false);
      o = __builtin_amdgcn_cvt_pk_bf8_f32(f[2], f[3], o, true);
    }
    *reinterpret_cast<uint32_t*>(qdst + row * ld + col0) = o;
  }
}
__device__ __forceinline__ void wave_amax(float* dst, float am, int lane) {
  am = fmaxf(am, __shfl_xor(am, 32, 64)); am = fmaxf(am, __shfl_xor(am, 16, 64)); am = fmaxf(am, __shfl_xor(am, 8, 64));
  am = fmaxf(am, __shfl_xor(am, 4, 64)); am = fmaxf(am, __shfl_xor(am, 2, 64)); am = fmaxf(am, __shfl_xor(am, 1, 64));
  if (lane == 0) amax_update(dst, am);
}

// LDS bytes: forward = K, V images of NKP = NK rounded up to 2 key tiles + the key bias; backward = Q, dO, K, V images (or, SHARE, one
// pair of images: K, V in phase A, Q, dO in phase B, in the same place) + row statistics (16 B per query) + key bias + one
// transposition tile per wave
inline size_t fwd_lds(int NK, int ND) { const int NKP = (NK + 1) & ~1; return (size_t)32 * NKP * (32 * ND + 16) + 64 * NKP; }
inline size_t bwd_lds(int NQ, int NK, int ND, int NW, bool share) {
  const int rows = share ? 2 * (NQ > NK ? NQ : NK) : 2 * (NQ + NK);      // image tiles of 16 rows
  return (size_t)16 * rows * (32 * ND + 16) + 256 * NQ + 64 * NK + (size_t)NW * 16 * SCR_STB;
}
// + the dropout bits of the whole score matrix, one byte per lane and tile pair (phase A writes, phase B reads): where it fits
inline size_t keep_cache_bytes(int NQ, int NK) { return (size_t)64 * NQ * NK; }

template <int ND, int NW>
__global__ __launch_bounds__(64 * NW) void attn_fwd_long(ATTN_HOT_PARAMS) {
  ATTN_HOT_UNPACK
  constexpr int STB = 32 * ND + 16, d = 16 * ND;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4, wv = tid >> 6;
  const int NQ = (a.Tq + 15) >> 4, NK = (a.Tk + 15) >> 4, NKP = (NK + 1) & ~1;
  const long bh = blockIdx.x;
  const int b = (int)(bh / a.heads), h = (int)(bh % a.heads);
  char* Ks = smem;
  char* Vs = Ks + 16 * NKP * STB;
  float* kbias = reinterpret_cast<float*>(Vs + 16 * NKP * STB);
  load_image<ND>(Ks, a.k + (long)b * a.Tk * a.ldk + h * d, a.ldk, a.Tk, 16 * NKP, tid, 64 * NW);
  load_image<ND>(Vs, a.v + (long)b * a.Tk * a.ldv + h * d, a.ldv, a.Tk, 16 * NKP, tid, 64 * NW);
  load_keybias(kbias, a.keymask + (long)b * a.Tk, a.Tk, 16 * NKP, tid, 64 * NW);
  __syncthreads();
  const float sc = a.scale * LOG2E, ds = a.thr ? a.dscale : 1.0f;
  const bf16_t* qg = a.q + (long)b * a.Tq * a.ldq + h * d;
  float am = 0.f;
  for (int it = wv; it < NQ; it += NW) {
    const int i = 16 * it + n;
    s4_t qf[ND];
#pragma unroll
    for (int ks = 0; ks < ND; ++ks) qf[ks] = frag_rows_global(qg, a.ldq, a.Tq, 16 * it, 16 * ks, lane);
    float m = -INFINITY, l = 0.f;
    f4_t o[ND];
#pragma unroll
    for (int ct = 0; ct < ND; ++ct) o[ct] = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int jt = 0; jt < NKP; jt += 2) {
      f4_t s[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        s[u] = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < ND; ++ks) s[u] = mma16(frag_rows(Ks, STB, 16 * (jt + u), 16 * ks, lane), qf[ks], s[u]);      // S^T[j][i]
      }
      float cm = -INFINITY;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f4_t kb = *reinterpret_cast<const f4_t*>(kbias + 16 * (jt + u) + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[u][r] = fmaf(s[u][r], sc, kb[r]); cm = fmaxf(cm, s[u][r]); }
      }
      const float mn = fmaxf(m, xmax2(cm));          // finite: key 16 jt exists
      const float alpha = __builtin_amdgcn_exp2f(m - mn);
      float ps = 0.f;
      s4_t pb[2];
      const uint32_t kb8 = keep_byte(a, bh, i, jt >> 1, g);          // jt is even: one call for the pair
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const uint32_t nib = (kb8 >> (4 * u)) & 0xfu;
        f4_t p;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(s[u][r] - mn);
          ps += e;
          p[r] = ((nib >> r) & 1u) ? e * ds : 0.f;
        }
        pb[u] = pack4(p);
      }
      l = fmaf(l, alpha, ps);
      m = mn;
#pragma unroll
      for (int ct = 0; ct < ND; ++ct) {
        o[ct] = o[ct] * alpha;
#pragma unroll
        for (int u = 0; u < 2; ++u) o[ct] = mma16(frag_cols(Vs, STB, 16 * (jt + u), 16 * ct, lane), pb[u], o[ct]);        // ctx^T[c][i]
      }
    }
    const float lsum = xsum2(l);
    const float inv = 1.0f / lsum;
    if (i < a.Tq) {
#pragma unroll
      for (int ct = 0; ct < ND; ++ct)
        store_tile_t<true>(a.ctx, a.ctx_q, a.ldo, (long)b * a.Tq + i, h * d + 16 * ct + 4 * g, o[ct] * inv, a.ctx_qscale, am);
      if (a.lse && g == 0) a.lse[bh * a.Tq + i] = m + __builtin_amdgcn_logf(lsum);      // v_log_f32: log2
    }
  }
  if (a.ctx_q) wave_amax(a.ctx_qamax, am, lane);
}

// S^T and dP^T tiles of (key tile jt, the wave's query fragments): s <- scores in the exp2 domain (scaled, key bias added),
// gp <- (dO v^T)^T, unscaled and before the dropout mask
template <int ND>
__device__ __forceinline__ void score_tiles(f4_t& s, f4_t& gp, const char* Ks, const char* Vs, const float* kbias, const s4_t (&qf)[ND],
                                            const s4_t (&of)[ND], int jt, float sc, int lane) {
  constexpr int STB = 32 * ND + 16;
  s = f4_t{0.f, 0.f, 0.f, 0.f};
  gp = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < ND; ++ks) {
    s = mma16(frag_rows(Ks, STB, 16 * jt, 16 * ks, lane), qf[ks], s);
    gp = mma16(frag_rows(Vs, STB, 16 * jt, 16 * ks, lane), of[ks], gp);
  }
  const f4_t kb = *reinterpret_cast<const f4_t*>(kbias + 16 * jt + 4 * (lane >> 4));
#pragma unroll
  for (int r = 0; r < 4; ++r) s[r] = fmaf(s[r], sc, kb[r]);
}

// SHARE = false: all four images resident (up to 256 x 256 x 64) -- every fragment out of LDS.  SHARE = true (beyond that, up to 512 x 512 x 64):
// the images of the two phases share one space and a wave's OWN tile (q / dO in phase A, k / v in phase B) comes straight from global
// memory; costs the PlotQA-shaped step 0.27 ms (18.95 against 18.68) where both fit, hence only where it must.
template <int ND, int NW, bool SHARE, bool LSE>
__global__ __launch_bounds__(64 * NW) void attn_bwd_long(ATTN_HOT_PARAMS) {
  ATTN_HOT_UNPACK
  constexpr int STB = 32 * ND + 16, d = 16 * ND;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4, wv = tid >> 6;
  const int NQ = (a.Tq + 15) >> 4, NK = (a.Tk + 15) >> 4;
  const long bh = blockIdx.x;
  const int b = (int)(bh / a.heads), h = (int)(bh % a.heads);
  // SHARE: phase A reads K, V from LDS (every wave sweeps all key tiles) and its own q / dO fragments from global memory; phase B reads
  // Q, dO from LDS (every wave sweeps all query tiles) and its own k / v fragments from global memory; the images of the two phases
  // share the space: 2 x max(Tq, Tk) rows instead of 2 x (Tq + Tk) -- 512 x 512 x 64 fits the 160 KB exactly
  const int NX = NQ > NK ? NQ : NK;
  char* Ks = smem;
  char* Vs = Ks + 16 * (SHARE ? NX : NK) * STB;
  char* Qs = SHARE ? Ks : Vs + 16 * NK * STB;
  char* Os = SHARE ? Vs : Qs + 16 * NQ * STB;          // dO
  float4* stats = reinterpret_cast<float4*>(SHARE ? Vs + 16 * NX * STB : Os + 16 * NQ * STB);      // per query: m, 1 / l, delta
  float* kbias = reinterpret_cast<float*>(stats + 16 * NQ);
  char* scr = reinterpret_cast<char*>(kbias + 16 * NK) + wv * 16 * SCR_STB;
  // dropout bits of tile pair (it, jt) as phase A's lanes hold them -- the layout phase B needs them in (lane = query 16 it + n, keys
  // 16 jt + 4 g ..): one Philox call per lane and tile pair instead of two (~140 SIMD cycles a wave-call at 7 rounds:
  // tools/lab/philox_rate.hip; a fifth of this issue-bound kernel's instructions with dropout on)
  uint8_t* keepc = reinterpret_cast<uint8_t*>(kbias + 16 * NK) + NW * 16 * SCR_STB;
  const bool cached = a.keep_cache != 0;
  const bf16_t* qg = a.q + (long)b * a.Tq * a.ldq + h * d;
  const bf16_t* og = a.dctx + (long)b * a.Tq * a.ldo + h * d;
  const bf16_t* kg = a.k + (long)b * a.Tk * a.ldk + h * d;
  const bf16_t* vg = a.v + (long)b * a.Tk * a.ldv + h * d;
  if constexpr (!SHARE) {
    load_image<ND>(Qs, qg, a.ldq, a.Tq, 16 * NQ, tid, 64 * NW);
    load_image<ND>(Os, og, a.ldo, a.Tq, 16 * NQ, tid, 64 * NW);
  }
  load_image<ND>(Ks, kg, a.ldk, a.Tk, 16 * NK, tid, 64 * NW);
  load_image<ND>(Vs, vg, a.ldv, a.Tk, 16 * NK, tid, 64 * NW);
  load_keybias(kbias, a.keymask + (long)b * a.Tk, a.Tk, 16 * NK, tid, 64 * NW);
  __syncthreads();
  const float sc = a.scale * LOG2E, ds = a.thr ? a.dscale : 1.0f;
  float am_q = 0.f, am_kv = 0.f;
  // ---------------------------------------------------------------- phase A: statistics and dq of the wave's query tiles
  for (int it = wv; it < NQ; it += NW) {
    const int i = 16 * it + n;
    s4_t qf[ND], of[ND];
#pragma unroll
    for (int ks = 0; ks < ND; ++ks) {
      if constexpr (SHARE) {
        qf[ks] = frag_rows_global(qg, a.ldq, a.Tq, 16 * it, 16 * ks, lane);
        of[ks] = frag_rows_global(og, a.ldo, a.Tq, 16 * it, 16 * ks, lane);
      } else {
        qf[ks] = frag_rows(Qs, STB, 16 * it, 16 * ks, lane);
        of[ks] = frag_rows(Os, STB, 16 * it, 16 * ks, lane);
      }
    }
    float m = -INFINITY, l = 0.f, dl = 0.f;
    uint32_t kw[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};      // keep bits, 4 per key tile (up to 32 key tiles)
    if constexpr (LSE) {
      // the forward's statistics: m <- lse (then exp2(s - m) IS the probability: l = 1), delta from the forward's output
      const bf16_t* cg = a.ctx + ((long)b * a.Tq + min(i, a.Tq - 1)) * a.ldc + h * d + 4 * g;
#pragma unroll
      for (int ks = 0; ks < ND; ++ks) {
        const uint2 ov = *reinterpret_cast<const uint2*>(cg + 16 * ks);
        dl = fmaf(bf2f((bf16_t)(ov.x & 0xffff)), bf2f((bf16_t)of[ks][0]), dl);
        dl = fmaf(bf2f((bf16_t)(ov.x >> 16)), bf2f((bf16_t)of[ks][1]), dl);
        dl = fmaf(bf2f((bf16_t)(ov.y & 0xffff)), bf2f((bf16_t)of[ks][2]), dl);
        dl = fmaf(bf2f((bf16_t)(ov.y >> 16)), bf2f((bf16_t)of[ks][3]), dl);
      }
      // a query past Tq (padding of the last tile: q and dO rows are zero) gets m = +inf: its probabilities are exactly 0 whatever its scores
      m = i < a.Tq ? a.lse[bh * a.Tq + i] : INFINITY;
      l = 0.25f;                                   // xsum2 over the four lane groups below: 1
    }
#pragma unroll
    for (int jo = 0; jo < 4; ++jo) {
      if (!LSE && 8 * jo < NK) {
        uint32_t w = 0u, kb8 = 0xffu;
        const int je = min(8, NK - 8 * jo);
#pragma unroll 1
        for (int ji = 0; ji < je; ++ji) {
          const int jt = 8 * jo + ji;
          f4_t s, gp;
          score_tiles<ND>(s, gp, Ks, Vs, kbias, qf, of, jt, sc, lane);
          const float cm = xmax2(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])));
          const float mn = fmaxf(m, cm);
          const float alpha = __builtin_amdgcn_exp2f(m - mn);
          if (!(ji & 1)) kb8 = keep_byte(a, bh, i, jt >> 1, g);      // jt = 8 jo + ji: even ji = first tile of a pair
          const uint32_t nib = (kb8 >> (4 * (ji & 1))) & 0xfu;
          w |= nib << (4 * ji);
          if (cached) keepc[(it * NK + jt) * 64 + lane] = (uint8_t)nib;
          float ps = 0.f, pd = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(s[r] - mn);
            ps += e;
            pd = fmaf(e, ((nib >> r) & 1u) ? gp[r] * ds : 0.f, pd);
          }
          l = fmaf(l, alpha, ps);
          dl = fmaf(dl, alpha, pd);
          m = mn;
        }
        kw[jo] = w;
      }
    }
    const float inv = 1.0f / xsum2(l);
    const float delta = xsum2(dl) * inv;
    if (g == 0) stats[i] = make_float4(m, inv, delta, 0.f);
    f4_t dq[ND];
#pragma unroll
    for (int ct = 0; ct < ND; ++ct) dq[ct] = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jo = 0; jo < 4; ++jo) {
      if (8 * jo < NK) {
        const uint32_t w = kw[jo];
        uint32_t kb8 = 0xffu;
        const int je = min(8, NK - 8 * jo);
#pragma unroll 1
        for (int ji = 0; ji < je; ++ji) {
          const int jt = 8 * jo + ji;
          f4_t s, gp;
          score_tiles<ND>(s, gp, Ks, Vs, kbias, qf, of, jt, sc, lane);
          uint32_t nib;
          if constexpr (LSE) {                     // no sweep 1: the dropout bits are made (and parked for phase B) here
            if (!(ji & 1)) kb8 = keep_byte(a, bh, i, jt >> 1, g);
            nib = (kb8 >> (4 * (ji & 1))) & 0xfu;
            if (cached) keepc[(it * NK + jt) * 64 + lane] = (uint8_t)nib;
          } else {
            nib = (w >> (4 * ji)) & 0xfu;
          }
          f4_t dsv;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = __builtin_amdgcn_exp2f(s[r] - m) * inv;
            dsv[r] = p * ((((nib >> r) & 1u) ? gp[r] * ds : 0.f) - delta);
          }
          const s4_t dsb = pack4(dsv);
#pragma unroll
          for (int ct = 0; ct < ND; ++ct) dq[ct] = mma16(frag_cols(Ks, STB, 16 * jt, 16 * ct, lane), dsb, dq[ct]);      // dq^T[c][i]
        }
      }
    }
    if (i < a.Tq) {
#pragma unroll
      for (int ct = 0; ct < ND; ++ct)
        store_tile_t<false>(a.dq, a.dq_q, a.lddq, (long)b * a.Tq + i, h * d + 16 * ct + 4 * g, dq[ct] * a.scale, a.dq_qscale, am_q);
    }
  }
  if (a.dq_q) wave_amax(a.dq_qamax, am_q, lane);
  __syncthreads();                          // the statistics of every query tile are in LDS; every wave is done with the K, V images
  if constexpr (SHARE) {
    load_image<ND>(Qs, qg, a.ldq, a.Tq, 16 * NQ, tid, 64 * NW);
    load_image<ND>(Os, og, a.ldo, a.Tq, 16 * NQ, tid, 64 * NW);
    __syncthreads();
  }
  // ---------------------------------------------------------------- phase B: dv and dk of the wave's key tiles
  for (int jt = wv; jt < NK; jt += NW) {
    const int j = 16 * jt + n;
    s4_t kf[ND], vf[ND];
#pragma unroll
    for (int ks = 0; ks < ND; ++ks) {
      if constexpr (SHARE) {
        kf[ks] = frag_rows_global(kg, a.ldk, a.Tk, 16 * jt, 16 * ks, lane);
        vf[ks] = frag_rows_global(vg, a.ldv, a.Tk, 16 * jt, 16 * ks, lane);
      } else {
        kf[ks] = frag_rows(Ks, STB, 16 * jt, 16 * ks, lane);
        vf[ks] = frag_rows(Vs, STB, 16 * jt, 16 * ks, lane);
      }
    }
    const f4_t kb = *reinterpret_cast<const f4_t*>(kbias + 16 * jt + 4 * g);
    f4_t dv[ND], dk[ND];
#pragma unroll
    for (int ct = 0; ct < ND; ++ct) { dv[ct] = f4_t{0.f, 0.f, 0.f, 0.f}; dk[ct] = f4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
    for (int it = 0; it < NQ; ++it) {
      const int i = 16 * it + n;
      f4_t s = f4_t{0.f, 0.f, 0.f, 0.f}, gp = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < ND; ++ks) {
        s = mma16(kf[ks], frag_rows(Qs, STB, 16 * it, 16 * ks, lane), s);
        gp = mma16(vf[ks], frag_rows(Os, STB, 16 * it, 16 * ks, lane), gp);
      }
      const float4 st = stats[i];
      const uint32_t nib = cached ? keepc[(it * NK + jt) * 64 + lane] : (keep_byte(a, bh, i, jt >> 1, g) >> (4 * (jt & 1))) & 0xfu;
      f4_t pdv, dsv;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[r], sc, kb[r]) - st.x) * st.y;
        const bool kp = (nib >> r) & 1u;
        pdv[r] = kp ? p * ds : 0.f;
        dsv[r] = p * ((kp ? gp[r] * ds : 0.f) - st.z);
      }
      // accumulator layout (query = column) -> [query][key] tile in LDS -> B operand of the products that contract over the queries
      put_tile_t(scr, SCR_STB, 0, 0, pdv, lane);
      wave_sync();
      const s4_t pf = frag_cols(scr, SCR_STB, 0, 0, lane);
      wave_sync();
      put_tile_t(scr, SCR_STB, 0, 0, dsv, lane);
      wave_sync();
      const s4_t sf = frag_cols(scr, SCR_STB, 0, 0, lane);
      wave_sync();
#pragma unroll
      for (int ct = 0; ct < ND; ++ct) {
        dv[ct] = mma16(frag_cols(Os, STB, 16 * it, 16 * ct, lane), pf, dv[ct]);       // dv^T[c][j] += dO[i][c] Pd[i][j]
        dk[ct] = mma16(frag_cols(Qs, STB, 16 * it, 16 * ct, lane), sf, dk[ct]);       // dk^T[c][j] += q[i][c] dS[i][j]
      }
    }
    if (j < a.Tk) {
#pragma unroll
      for (int ct = 0; ct < ND; ++ct) {
        store_tile_t<false>(a.dv, a.dv_q, a.lddv, (long)b * a.Tk + j, h * d + 16 * ct + 4 * g, dv[ct], a.dkv_qscale, am_kv);
        store_tile_t<false>(a.dk, a.dk_q, a.lddk, (long)b * a.Tk + j, h * d + 16 * ct + 4 * g, dk[ct] * a.scale, a.dkv_qscale, am_kv);
      }
    }
  }
  if (a.dv_q || a.dk_q) wave_amax(a.dkv_qamax, am_kv, lane);
}

template <bool BWD, int ND, int NW, bool SHARE, bool LSE>
hipError_t launch_nw(const AttnArgs& a, size_t lds, hipStream_t s) {
  auto kern = BWD ? attn_bwd_long<ND, NW, SHARE, LSE> : attn_fwd_long<ND, NW>;
  static bool raised = false;             // first call is eager (outside any stream capture)
  if (lds > 64 * 1024 && !raised) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_CAP);
    if (e != hipSuccess) return e;
    raised = true;
  }
  if (a.ldq > 0x7fffffffL || a.ldk > 0x7fffffffL || a.ldv > 0x7fffffffL) return hipErrorInvalidValue;      // preloaded as 32-bit scalars
  crct_launch(kern, dim3(a.B * a.heads), dim3(64 * NW), lds, s, ATTN_HOT_ARGS(a) a);
  return hipGetLastError();
}
// Waves per (batch, head).  Forward: 8 when there are at least 8 query tiles to hand out, else 4.  Backward: phase A hands out the query
// tiles, phase B the key tiles, so 8 waves only when BOTH sides have 8 tiles -- the co-attention shapes (124 x 44: 8 x 3 tiles) keep
// 3 of 8 waves busy in one of the two phases, and with thousands of (batch, head) pairs queued a CU is better filled by twice as many
// 4-wave workgroups (wave-slot utilisation 0.82 - 0.90 against 0.47 - 0.64; in the step the two are within noise: 19.00 against 19.03 ms)
inline int waves_for(bool bwd, int NQ, int NK) { return (bwd ? (NQ < NK ? NQ : NK) : NQ) >= 8 ? 8 : 4; }
// backward: four resident images where they fit, the shared image pair beyond (bwd_share)
inline bool bwd_share(int NQ, int NK, int ND, int NW) { return bwd_lds(NQ, NK, ND, NW, false) > (size_t)LDS_CAP; }
template <bool BWD, int ND>
hipError_t launch_d(const AttnArgs& a_in, hipStream_t s) {
  AttnArgs a = a_in;
  const int NQ = (a.Tq + 15) >> 4, NK = (a.Tk + 15) >> 4;
  const int NW = waves_for(BWD, NQ, NK);
  const bool share = BWD && bwd_share(NQ, NK, ND, NW);
  size_t lds = BWD ? bwd_lds(NQ, NK, ND, NW, share) : fwd_lds(NK, ND);
  a.keep_cache = 0;
  // ... where it does not lower the number of workgroups a CU's 160 KB hold
  if (BWD && a.thr && lds + keep_cache_bytes(NQ, NK) <= (size_t)LDS_CAP && LDS_CAP / (lds + keep_cache_bytes(NQ, NK)) == LDS_CAP / lds) {
    a.keep_cache = 1;
    lds += keep_cache_bytes(NQ, NK);
  }
  if constexpr (BWD) {
    if (a.lse && a.ctx) {     // the forward's statistics and output are at hand: no statistics sweep
      if (share) return NW == 8 ? launch_nw<true, ND, 8, true, true>(a, lds, s) : launch_nw<true, ND, 4, true, true>(a, lds, s);
      return NW == 8 ? launch_nw<true, ND, 8, false, true>(a, lds, s) : launch_nw<true, ND, 4, false, true>(a, lds, s);
    }
  }
  if (share) return NW == 8 ? launch_nw<BWD, ND, 8, true, false>(a, lds, s) : launch_nw<BWD, ND, 4, true, false>(a, lds, s);
  return NW == 8 ? launch_nw<BWD, ND, 8, false, false>(a, lds, s) : launch_nw<BWD, ND, 4, false, false>(a, lds, s);
}
template <bool BWD>
hipError_t launch(const AttnArgs& a, hipStream_t s) {
  switch (a.d) {
    case 32: return launch_d<BWD, 2>(a, s);
    case 48: return launch_d<BWD, 3>(a, s);
    case 64: return launch_d<BWD, 4>(a, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace

// Both directions must fit (the forward of a shape the backward cannot take is of no use to the step): head size 32 / 48 / 64
// and at most CRCT_ATTN_MAX_LEN queries and keys -- 512 x 512 x 64: forward 146 KB for the K, V images, backward exactly the 160 KB
// (two images + statistics + key bias + eight transposition tiles).
bool crct_attention_long_ok(int Tq, int Tk, int d) {
  if (!(d == 32 || d == 48 || d == 64) || Tq < 1 || Tk < 1 || Tq > CRCT_ATTN_MAX_LEN || Tk > CRCT_ATTN_MAX_LEN) return false;
  const int NQ = (Tq + 15) >> 4, NK = (Tk + 15) >> 4, ND = d / 16;
  return bwd_lds(NQ, NK, ND, waves_for(true, NQ, NK), true) <= (size_t)LDS_CAP && fwd_lds(NK, ND) <= (size_t)LDS_CAP;
}
hipError_t crct_attention_long_fwd(const AttnArgs& a, hipStream_t s) { return launch<false>(a, s); }
hipError_t crct_attention_long_bwd(const AttnArgs& a, hipStream_t s) { return launch<true>(a, s); }
