// MFMA attention for the short CRCT sequences: one, two or four waves per (batch, head) (see launch()), sequences up to 112 queries x
// 112 keys (1-4 or 7 tiles of 16 per side), head size 32 / 48 / 64.  Same math and the same dropout stream as attention.hip (which stays
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
#include "common.hip.h"
#include "crct_internal.h"
#include "attention_args.h"
#include "attention_tiles.hip.h"

//
// A workgroup holds W = 1, 2 or 4 independent (batch, head) pairs, each with its own slice of the dynamic LDS allocation
// (W is chosen per launch so that a CU's 160 KB hold as many pairs as possible) and SP = 1, 2 or 4 waves per pair.
// With SP = 1 the waves never exchange data: phases are separated by wave-level fences only (LDS operations of one wave
// execute in issue order) and a wave past the end of the grid simply exits; with SP > 1 the waves of a pair split the
// query tiles (scores, softmax, dS, dq) and the key tiles (dv, dk) and meet at workgroup barriers between the phases.
static int g_attn_force_split = 0;      // test hook (crct_attention_force_split): 0 = by tile count, 1 / 2 / 4 = that many waves per pair
extern "C" void crct_attention_force_split(int n) { g_attn_force_split = (n == 1 || n == 2 || n == 4) ? n : 0; }

namespace {

// SP == 1: one wave owns a (batch, head) -- a wave-level fence separates its phases; SP > 1: the SP waves that share
// the (batch, head)'s LDS images meet at a workgroup barrier (every wave of the workgroup takes the same path)
template <int SP>
__device__ __forceinline__ void group_sync() {
  if constexpr (SP == 1) wave_sync();
  else __syncthreads();
}

// rows [T][16*ND] bf16 (row stride ld) -> LDS image of 16*NT rows (rows >= T are zero), in two phases so that the global
// loads of SEVERAL matrices are all in flight before the first one is waited for: fetch() issues every load of the
// matrix unconditionally (rows past T re-read row 0 and are zeroed afterwards: no branch, no wait between loads),
// put() stores the registers to the image.  (A plain load -> store loop costs one memory round trip per 64 chunks and
// matrix: 6-14 serial round trips in front of 2-3 us of arithmetic.)
template <int ND, int NT, int SP = 1>
struct RowTile {
  // SP cooperating waves share the chunks of the matrix round-robin (chunk c belongs to wave (c / 64) % SP)
  static constexpr int CPR = 2 * ND, STB = 32 * ND + 16, CHUNKS = 16 * NT * CPR, ITER = (CHUNKS + 64 * SP - 1) / (64 * SP);
  uint4 reg[ITER];
  __device__ __forceinline__ void fetch(const bf16_t* __restrict__ src, long ld, int T, int lane, int part = 0) {
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int c = lane + 64 * (part + SP * i), r = c / CPR, cc = (c - r * CPR) << 3;
      const int rr = min(r, T - 1);                            // c >= CHUNKS implies r >= 16 * NT >= T
      const uint32_t m = r < T ? 0xffffffffu : 0u;             // mask, not a select: the load must stay unconditional
      const uint4 u = *reinterpret_cast<const uint4*>(src + (long)rr * ld + cc);
      reg[i] = make_uint4(u.x & m, u.y & m, u.z & m, u.w & m);
    }
  }
  __device__ __forceinline__ void put(char* img, int lane, int part = 0) const {
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int c = lane + 64 * (part + SP * i), r = c / CPR, cc = (c - r * CPR) << 3;
      if (c < CHUNKS) *reinterpret_cast<uint4*>(img + r * STB + cc * 2) = reg[i];
    }
  }
};
template <int ND, int SP = 1>
__device__ __forceinline__ void store_rows(bf16_t* dst, long ld, const char* img, int T, int lane, int part = 0) {
  constexpr int CPR = 2 * ND, STB = 32 * ND + 16;
  for (int c = lane + 64 * part; c < T * CPR; c += 64 * SP) {
    const int r = c / CPR, cc = (c - r * CPR) << 3;
    *reinterpret_cast<uint4*>(dst + (long)r * ld + cc) = *reinterpret_cast<const uint4*>(img + r * STB + cc * 2);
  }
}
// ... and an fp8 copy of the same rows (BF8: OCP e5m2, else e4m3; saturating) with their maximum into amax_dst
template <int ND, int SP, bool BF8>
__device__ __forceinline__ void store_rows_q(bf16_t* dst, uint8_t* qdst, long ld, const char* img, int T, int lane, int part, float qs,
                                             float* amax_dst) {
  constexpr int CPR = 2 * ND, STB = 32 * ND + 16;
  constexpr float LIM = BF8 ? 57344.f : 448.f;
  float am = 0.f;
  for (int c = lane + 64 * part; c < T * CPR; c += 64 * SP) {
    const int r = c / CPR, cc = (c - r * CPR) << 3;
    const uint4 u = *reinterpret_cast<const uint4*>(img + r * STB + cc * 2);
    *reinterpret_cast<uint4*>(dst + (long)r * ld + cc) = u;
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
    float f[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[2 * j] = bf2f((bf16_t)(w[j] & 0xffff)); f[2 * j + 1] = bf2f((bf16_t)(w[j] >> 16)); }
    uint32_t o[2] = {0u, 0u};
#pragma unroll
    for (int j = 0; j < 8; ++j) { am = fmaxf(am, fabsf(f[j])); f[j] = fminf(fmaxf(f[j] * qs, -LIM), LIM); }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if constexpr (BF8) {
        o[j] = __builtin_amdgcn_cvt_pk_bf8_f32(f[4 * j], f[4 * j + 1], o[j], false);
        o[j] = __builtin_amdgcn_cvt_pk_bf8_f32(f[4 * j + 2], f[4 * j + 3], o[j], true);
      } else {
        o[j] = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * j], f[4 * j + 1], o[j], false);
        o[j] = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * j + 2], f[4 * j + 3], o[j], true);
      }
    }
    *reinterpret_cast<uint2*>(qdst + (long)r * ld + cc) = make_uint2(o[0], o[1]);
  }
  am = fmaxf(am, __shfl_xor(am, 32, 64)); am = fmaxf(am, __shfl_xor(am, 16, 64)); am = fmaxf(am, __shfl_xor(am, 8, 64));
  am = fmaxf(am, __shfl_xor(am, 4, 64)); am = fmaxf(am, __shfl_xor(am, 2, 64)); am = fmaxf(am, __shfl_xor(am, 1, 64));
  if (lane == 0) amax_update(amax_dst, am);
}
// bit 4*jt + r of the result: key 16*jt + 4*g + r exists and is attended (keymask != 0); `valid`: it exists.
// Lane l reads keymask[l] (and [l + 64] for more than 64 keys), two wave ballots give every lane all the keys.
template <int NK>
__device__ __forceinline__ void key_bits(const uint8_t* km, int Tk, int lane, uint32_t& attend, uint32_t& valid) {
  const int g = lane >> 4;
  const uint8_t m0 = km[min(lane, Tk - 1)];
  const uint8_t m1 = NK > 4 ? km[min(lane + 64, Tk - 1)] : (uint8_t)0;
  const uint64_t e0 = __ballot(lane < Tk), e1 = NK > 4 ? __ballot(lane + 64 < Tk) : 0ull;
  const uint64_t a0 = __ballot(m0 != 0) & e0, a1 = NK > 4 ? (__ballot(m1 != 0) & e1) : 0ull;
  attend = 0u; valid = 0u;
#pragma unroll
  for (int jt = 0; jt < NK; ++jt) {
    const int sh = 16 * (jt & 3) + 4 * g;
    attend |= (uint32_t)(((jt < 4 ? a0 : a1) >> sh) & 0xfull) << (4 * jt);
    valid |= (uint32_t)(((jt < 4 ? e0 : e1) >> sh) & 0xfull) << (4 * jt);
  }
}

// P^T tiles of one 16-query column block `it`: in: raw scores S^T (acc), out: probabilities (before dropout) and
// the keep bits of the lane's 4 keys per tile.  exp(x) is evaluated as exp2(x * log2 e) (one v_exp_f32).
template <int NK>
__device__ __forceinline__ void softmax_cols(f4_t (&s)[NK], uint32_t& keep, uint32_t attend, uint32_t valid, int Tk, int Tq, int i,
                                             long bh, const AttnArgs& a, int lane) {
  const int g = lane >> 4;
  const float sc = a.scale * 1.4426950408889634f, off = -10000.f * 1.4426950408889634f;
  float mx = -INFINITY;
#pragma unroll
  for (int jt = 0; jt < NK; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t bit = 1u << (4 * jt + r);
      const float v = (valid & bit) ? s[jt][r] * sc + ((attend & bit) ? 0.f : off) : -INFINITY;
      s[jt][r] = v;
      mx = fmaxf(mx, v);
    }
  mx = xmax2(mx);
  float sum = 0.f;
#pragma unroll
  for (int jt = 0; jt < NK; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = __builtin_amdgcn_exp2f(s[jt][r] - mx);        // exp2(-inf) = 0 for the padding keys
      s[jt][r] = e;
      sum += e;
    }
  const float inv = 1.0f / xsum2(sum);
  keep = 0xffffffffu;
#pragma unroll
  for (int jt = 0; jt < NK; ++jt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) s[jt][r] *= inv;
  }
  // one Philox call per key-tile PAIR: its 8 slices are this lane's keys 16 jt + 4 g + r of the two tiles (attention_args.h, attn_keep8)
#pragma unroll
  for (int jp = 0; jp < (NK + 1) / 2; ++jp) {
    if (a.thr && 32 * jp + 4 * g < Tk && i < Tq) {
      const uint32_t kb = attn_keep8(a.seed, a.site, bh, Tq, Tk, i, jp, g, a.thr);
      keep = (keep & ~(0xffu << (8 * jp))) | (kb << (8 * jp));
    }
  }
}

template <int NQ, int NK, int ND> struct FwdLds { static constexpr int STB = 32 * ND + 16, BYTES = 16 * (NQ + 2 * NK) * STB; };
template <int NQ, int NK, int ND> struct BwdLds {
  static constexpr int STB = 32 * ND + 16, PSB = 32 * NK + 16, NX = NQ > NK ? NQ : NK;
  static constexpr int BYTES = 16 * (2 * NQ + NX) * STB + 16 * NQ * PSB;
};

// W independent (batch, head) pairs per workgroup, SP cooperating waves per pair (query tiles it = part, part + SP, ...):
// the arithmetic per tile is the same for every SP, so are the results bit for bit.
template <int NQ, int NK, int ND, int W, int SP>
__global__ __launch_bounds__(64 * W * SP) void attn_fwd_mfma(ATTN_HOT_PARAMS) {
  ATTN_HOT_UNPACK
  constexpr int STB = FwdLds<NQ, NK, ND>::STB;
  constexpr int NQL = (NQ + SP - 1) / SP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, n = lane & 15;
  const int wv = threadIdx.x >> 6, grp = wv / SP, part = wv % SP;
  const long total = (long)a.B * a.heads;
  long bh = (long)blockIdx.x * W + grp;
  const bool live = bh < total;
  if constexpr (SP == 1) { if (!live) return; }
  else if (!live) bh = total - 1;       // a surplus group of the last workgroup recomputes the last pair and stores nothing
  char* Qs = smem + grp * FwdLds<NQ, NK, ND>::BYTES;
  char* Ks = Qs + 16 * NQ * STB;
  char* Vs = Ks + 16 * NK * STB;
  const int b = (int)(bh / a.heads), h = (int)(bh % a.heads), d = 16 * ND;
  uint32_t attend, valid;
  {
    RowTile<ND, NQ, SP> tq;
    RowTile<ND, NK, SP> tk, tv;
    tq.fetch(a.q + (long)b * a.Tq * a.ldq + h * d, a.ldq, a.Tq, lane, part);
    tk.fetch(a.k + (long)b * a.Tk * a.ldk + h * d, a.ldk, a.Tk, lane, part);
    tv.fetch(a.v + (long)b * a.Tk * a.ldv + h * d, a.ldv, a.Tk, lane, part);
    key_bits<NK>(a.keymask + (long)b * a.Tk, a.Tk, lane, attend, valid);
    tq.put(Qs, lane, part);
    tk.put(Ks, lane, part);
    tv.put(Vs, lane, part);
  }
  group_sync<SP>();
  s4_t kf[NK][ND];
#pragma unroll
  for (int jt = 0; jt < NK; ++jt)
#pragma unroll
    for (int ks = 0; ks < ND; ++ks) kf[jt][ks] = frag_rows(Ks, STB, 16 * jt, 16 * ks, lane);
#pragma unroll
  for (int li = 0; li < NQL; ++li) {
    const int it = part + SP * li;
    if (it < NQ) {
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
      softmax_cols<NK>(s, keep, attend, valid, a.Tk, a.Tq, 16 * it + n, bh, a, lane);
      const float ds = a.thr ? a.dscale : 1.0f;
      s4_t pb[NK];
#pragma unroll
      for (int jt = 0; jt < NK; ++jt) {
        f4_t p;
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = ((keep >> (4 * jt + r)) & 1u) ? s[jt][r] * ds : 0.f;
        pb[jt] = pack4(p);
      }
      // the 16 query rows of this block have been read into qf: their slot in the Q image takes the output tile
#pragma unroll
      for (int ct = 0; ct < ND; ++ct) {
        f4_t o = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jt = 0; jt < NK; ++jt) o = mma16(frag_cols(Vs, STB, 16 * jt, 16 * ct, lane), pb[jt], o);   // ctx^T[c][i]
        put_tile_t(Qs, STB, 16 * it, 16 * ct, o, lane);
      }
    }
  }
  group_sync<SP>();
  if (live) {
    if (a.ctx_q) store_rows_q<ND, SP, false>(a.ctx + (long)b * a.Tq * a.ldo + h * d, a.ctx_q + (long)b * a.Tq * a.ldo + h * d, a.ldo, Qs, a.Tq,
                                             lane, part, a.ctx_qscale[0], a.ctx_qamax);
    else store_rows<ND, SP>(a.ctx + (long)b * a.Tq * a.ldo + h * d, a.ldo, Qs, a.Tq, lane, part);
  }
}

// SP cooperating waves per (batch, head): phase 1 (scores, softmax, dS, dq) by QUERY tiles it = part, part + SP, ...; the two
// products that contract over the queries (dv, dk) by KEY tiles jt = part, part + SP, ... from the P / dS image all waves
// have filled.  Every tile is computed exactly as with one wave, in the same summation order: bit-identical results.
template <int NQ, int NK, int ND, int W, int SP>
__global__ __launch_bounds__(64 * W * SP) void attn_bwd_mfma(ATTN_HOT_PARAMS) {
  ATTN_HOT_UNPACK
  typedef BwdLds<NQ, NK, ND> G;
  constexpr int STB = G::STB, PSB = G::PSB;
  constexpr int NQL = (NQ + SP - 1) / SP, NKL = (NK + SP - 1) / SP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, n = lane & 15;
  const int wv = threadIdx.x >> 6, grp = wv / SP, part = wv % SP;
  const long total = (long)a.B * a.heads;
  long bh = (long)blockIdx.x * W + grp;
  const bool live = bh < total;
  if constexpr (SP == 1) { if (!live) return; }
  else if (!live) bh = total - 1;       // a surplus group of the last workgroup recomputes the last pair and stores nothing
  char* Qs = smem + grp * G::BYTES;
  char* Os = Qs + 16 * NQ * STB;        // dO
  char* Ks = Os + 16 * NQ * STB;        // K (16 * max(NQ, NK) rows: later the staging tile of dq / dv / dk)
  char* Pi = Ks + 16 * G::NX * STB;     // dropout(P) [i][j], then dS [i][j]
  const int b = (int)(bh / a.heads), h = (int)(bh % a.heads), d = 16 * ND;
  const bf16_t* vg = a.v + (long)b * a.Tk * a.ldv + h * d;
  s4_t vf[NK][ND];                      // V is only ever contracted along its columns: fragments straight from global
  uint32_t attend, valid;
  {
    RowTile<ND, NQ, SP> tq, to;
    RowTile<ND, NK, SP> tk;
    tq.fetch(a.q + (long)b * a.Tq * a.ldq + h * d, a.ldq, a.Tq, lane, part);
    to.fetch(a.dctx + (long)b * a.Tq * a.ldo + h * d, a.ldo, a.Tq, lane, part);
    tk.fetch(a.k + (long)b * a.Tk * a.ldk + h * d, a.ldk, a.Tk, lane, part);
#pragma unroll
    for (int jt = 0; jt < NK; ++jt)
#pragma unroll
      for (int ks = 0; ks < ND; ++ks) vf[jt][ks] = frag_rows_global(vg, a.ldv, a.Tk, 16 * jt, 16 * ks, lane);
    key_bits<NK>(a.keymask + (long)b * a.Tk, a.Tk, lane, attend, valid);
    tq.put(Qs, lane, part);
    to.put(Os, lane, part);
    tk.put(Ks, lane, part);
  }
  group_sync<SP>();
  const float ds = a.thr ? a.dscale : 1.0f;
  // Large tile counts (7 x 3 and up: the 100-element sequences of the long-context configuration) would keep K fragments,
  // V fragments, every dq tile and every dS tile live at once -- 380 registers for 7 x 7 x 64, 848 bytes per lane of scratch
  // with the plain code.  LOWREG re-reads the K fragments from their LDS image per query tile; DQ_DIRECT stores each dq tile
  // to global memory as soon as it is complete (8 bytes per lane) instead of holding all of them for a staged row store
  // (also with cooperating waves: the K image, which is the staging tile, is still being read by the other wave).
  constexpr bool LOWREG = NQ * NK >= 21;
  constexpr bool DQ_DIRECT = LOWREG || SP > 1;
  f4_t dq[DQ_DIRECT ? 1 : ND][DQ_DIRECT ? 1 : NQ];
  float dq_am = 0.f;                    // max |dq| of this wave's direct stores (fp8 copy)
  s4_t dsb[NQL][NK];                    // dS^T tiles (bf16) of this wave's query tiles: B operand of dq now, written to the image for dk later
  {
    s4_t kf[LOWREG ? 1 : NK][LOWREG ? 1 : ND];
    if constexpr (!LOWREG) {
#pragma unroll
      for (int jt = 0; jt < NK; ++jt)
#pragma unroll
        for (int ks = 0; ks < ND; ++ks) kf[jt][ks] = frag_rows(Ks, STB, 16 * jt, 16 * ks, lane);
    }
#pragma unroll
    for (int li = 0; li < NQL; ++li) {
      const int it = part + SP * li;
      if (it < NQ) {
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
            s4_t kfr;
            if constexpr (LOWREG) kfr = frag_rows(Ks, STB, 16 * jt, 16 * ks, lane);
            else kfr = kf[jt][ks];
            p[jt] = mma16(kfr, qf[ks], p[jt]);               // S^T[j][i]
            gp[jt] = mma16(vf[jt][ks], of[ks], gp[jt]);      // (dO v^T)^T[j][i]
          }
        }
        uint32_t keep;
        softmax_cols<NK>(p, keep, attend, valid, a.Tk, a.Tq, 16 * it + n, bh, a, lane);
        float psum = 0.f;
#pragma unroll
        for (int jt = 0; jt < NK; ++jt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            gp[jt][r] = ((keep >> (4 * jt + r)) & 1u) ? gp[jt][r] * ds : 0.f;        // gradient w.r.t. P (through dropout)
            psum += gp[jt][r] * p[jt][r];
          }
        const float delta = xsum2(psum);
#pragma unroll
        for (int jt = 0; jt < NK; ++jt) {
          f4_t dsv, pd;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            dsv[r] = p[jt][r] * (gp[jt][r] - delta);
            pd[r] = ((keep >> (4 * jt + r)) & 1u) ? p[jt][r] * ds : 0.f;
          }
          dsb[li][jt] = pack4(dsv);
          put_tile_t(Pi, PSB, 16 * it, 16 * jt, pd, lane);
        }
        // dq^T[c][i] = sum_j k[j][c] dS^T[j][i]
#pragma unroll
        for (int ct = 0; ct < ND; ++ct) {
          f4_t t = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int jt = 0; jt < NK; ++jt) t = mma16(frag_cols(Ks, STB, 16 * jt, 16 * ct, lane), dsb[li][jt], t);
          if constexpr (DQ_DIRECT) {        // element r = column 16 ct + 4 (lane >> 4) + r of query 16 it + n
            const int i = 16 * it + n;
            if (live && i < a.Tq) {
              t = t * a.scale;
              const uint2 pk = make_uint2(pack2bf(t[0], t[1]), pack2bf(t[2], t[3]));
              *reinterpret_cast<uint2*>(a.dq + ((long)b * a.Tq + i) * a.lddq + h * d + 16 * ct + 4 * (lane >> 4)) = pk;
              if (a.dq_q) {                 // e5m2 copy of the bf16-rounded values
                const float f0 = bf2f((bf16_t)(pk.x & 0xffff)), f1 = bf2f((bf16_t)(pk.x >> 16)), f2 = bf2f((bf16_t)(pk.y & 0xffff)), f3 = bf2f((bf16_t)(pk.y >> 16));
                dq_am = fmaxf(fmaxf(dq_am, fmaxf(fabsf(f0), fabsf(f1))), fmaxf(fabsf(f2), fabsf(f3)));
                const float qs = a.dq_qscale[0];
                uint32_t o = 0u;
                o = __builtin_amdgcn_cvt_pk_bf8_f32(fminf(fmaxf(f0 * qs, -57344.f), 57344.f), fminf(fmaxf(f1 * qs, -57344.f), 57344.f), o, false);
                o = __builtin_amdgcn_cvt_pk_bf8_f32(fminf(fmaxf(f2 * qs, -57344.f), 57344.f), fminf(fmaxf(f3 * qs, -57344.f), 57344.f), o, true);
                *reinterpret_cast<uint32_t*>(a.dq_q + ((long)b * a.Tq + i) * a.lddq + h * d + 16 * ct + 4 * (lane >> 4)) = o;
              }
            }
          } else dq[ct][it] = t;
        }
      }
    }
  }
  if constexpr (DQ_DIRECT) {
    if (a.dq_q) {
      dq_am = fmaxf(dq_am, __shfl_xor(dq_am, 32, 64)); dq_am = fmaxf(dq_am, __shfl_xor(dq_am, 16, 64)); dq_am = fmaxf(dq_am, __shfl_xor(dq_am, 8, 64));
      dq_am = fmaxf(dq_am, __shfl_xor(dq_am, 4, 64)); dq_am = fmaxf(dq_am, __shfl_xor(dq_am, 2, 64)); dq_am = fmaxf(dq_am, __shfl_xor(dq_am, 1, 64));
      if (lane == 0) amax_update(a.dq_qamax, dq_am);
    }
  }
  group_sync<SP>();                     // K is consumed (its image becomes the staging tile); the P image is complete
  if constexpr (!DQ_DIRECT) {
#pragma unroll
    for (int it = 0; it < NQ; ++it)
#pragma unroll
      for (int ct = 0; ct < ND; ++ct) put_tile_t(Ks, STB, 16 * it, 16 * ct, dq[ct][it] * a.scale, lane);
    wave_sync();
    if (a.dq_q) store_rows_q<ND, 1, true>(a.dq + (long)b * a.Tq * a.lddq + h * d, a.dq_q + (long)b * a.Tq * a.lddq + h * d, a.lddq, Ks, a.Tq, lane, 0,
                                          a.dq_qscale[0], a.dq_qamax);
    else store_rows<ND>(a.dq + (long)b * a.Tq * a.lddq + h * d, a.lddq, Ks, a.Tq, lane);
  }
  // dv^T[c][j] = sum_i dO[i][c] Pd[i][j]     (this wave's key tiles)
  f4_t acc[ND][NKL];
#pragma unroll
  for (int lj = 0; lj < NKL; ++lj) {
    const int jt = part + SP * lj;
    if (jt < NK) {
      s4_t pf[NQ];
#pragma unroll
      for (int it = 0; it < NQ; ++it) pf[it] = frag_cols(Pi, PSB, 16 * it, 16 * jt, lane);
#pragma unroll
      for (int ct = 0; ct < ND; ++ct) {
        acc[ct][lj] = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int it = 0; it < NQ; ++it) acc[ct][lj] = mma16(frag_cols(Os, STB, 16 * it, 16 * ct, lane), pf[it], acc[ct][lj]);
      }
    }
  }
  group_sync<SP>();                     // dq has left the staging tile, P has been read by every wave: the image now takes dS
#pragma unroll
  for (int li = 0; li < NQL; ++li) {
    const int it = part + SP * li;
    if (it < NQ) {
#pragma unroll
      for (int jt = 0; jt < NK; ++jt)
        *reinterpret_cast<s4_t*>(Pi + (16 * it + n) * PSB + (16 * jt + 4 * (lane >> 4)) * 2) = dsb[li][jt];
    }
  }
#pragma unroll
  for (int lj = 0; lj < NKL; ++lj) {
    const int jt = part + SP * lj;
    if (jt < NK) {
#pragma unroll
      for (int ct = 0; ct < ND; ++ct) put_tile_t(Ks, STB, 16 * jt, 16 * ct, acc[ct][lj], lane);
    }
  }
  group_sync<SP>();
  if (live) {
    if (a.dv_q) store_rows_q<ND, SP, true>(a.dv + (long)b * a.Tk * a.lddv + h * d, a.dv_q + (long)b * a.Tk * a.lddv + h * d, a.lddv, Ks, a.Tk, lane,
                                           part, a.dkv_qscale[0], a.dkv_qamax);
    else store_rows<ND, SP>(a.dv + (long)b * a.Tk * a.lddv + h * d, a.lddv, Ks, a.Tk, lane, part);
  }
  // dk^T[c][j] = sum_i q[i][c] dS[i][j]      (this wave's key tiles)
#pragma unroll
  for (int lj = 0; lj < NKL; ++lj) {
    const int jt = part + SP * lj;
    if (jt < NK) {
      s4_t sf[NQ];
#pragma unroll
      for (int it = 0; it < NQ; ++it) sf[it] = frag_cols(Pi, PSB, 16 * it, 16 * jt, lane);
#pragma unroll
      for (int ct = 0; ct < ND; ++ct) {
        acc[ct][lj] = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int it = 0; it < NQ; ++it) acc[ct][lj] = mma16(frag_cols(Qs, STB, 16 * it, 16 * ct, lane), sf[it], acc[ct][lj]);
      }
    }
  }
  group_sync<SP>();                     // dv has left the staging tile
#pragma unroll
  for (int lj = 0; lj < NKL; ++lj) {
    const int jt = part + SP * lj;
    if (jt < NK) {
#pragma unroll
      for (int ct = 0; ct < ND; ++ct) put_tile_t(Ks, STB, 16 * jt, 16 * ct, acc[ct][lj] * a.scale, lane);
    }
  }
  group_sync<SP>();
  if (live) {
    if (a.dk_q) store_rows_q<ND, SP, true>(a.dk + (long)b * a.Tk * a.lddk + h * d, a.dk_q + (long)b * a.Tk * a.lddk + h * d, a.lddk, Ks, a.Tk, lane,
                                           part, a.dkv_qscale[0], a.dkv_qamax);
    else store_rows<ND, SP>(a.dk + (long)b * a.Tk * a.lddk + h * d, a.lddk, Ks, a.Tk, lane, part);
  }
}

// waves per workgroup: the largest of 4 / 2 / 1 that does not lower the number of waves a CU's LDS can hold
constexpr int waves_per_group(int bytes_per_wave) {
  const int cap = 160 * 1024;
  const int w1 = cap / bytes_per_wave, w2 = cap / (2 * bytes_per_wave) * 2, w4 = cap / (4 * bytes_per_wave) * 4;
  return (w4 >= w1 && w4 >= w2) ? 4 : (w2 >= w1 ? 2 : 1);
}

template <bool BWD, int NQ, int NK, int ND, int SP>
hipError_t launch_sp(const AttnArgs& a, hipStream_t s) {
  constexpr int BYTES = BWD ? BwdLds<NQ, NK, ND>::BYTES : FwdLds<NQ, NK, ND>::BYTES;
  constexpr int W0 = waves_per_group(BYTES);
  // at most 512 threads per workgroup; a pair that needs more than 48 KB gets a workgroup of its own: a 150 KB workgroup can
  // only start on a CU whose LDS is empty, i.e. after the other streams' GEMM workgroups have left it (long context: 11.83 ->
  // 11.75 ms per step)
  constexpr int W = BYTES > 48 * 1024 ? 1 : (W0 * SP > 8 ? 8 / SP : W0);
  static_assert(BYTES % 16 == 0 && BYTES * W <= 160 * 1024, "LDS slice");
  auto kern = BWD ? attn_bwd_mfma<NQ, NK, ND, W, SP> : attn_fwd_mfma<NQ, NK, ND, W, SP>;
  static bool raised = false;           // first call is eager (outside any stream capture)
  if (BYTES * W > 64 * 1024 && !raised) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, BYTES * W);
    if (e != hipSuccess) return e;
    raised = true;
  }
  const int total = a.B * a.heads;
  if (a.ldq > 0x7fffffffL || a.ldk > 0x7fffffffL || a.ldv > 0x7fffffffL) return hipErrorInvalidValue;      // preloaded as 32-bit scalars (ATTN_HOT_ARGS)
  crct_launch(kern, dim3((total + W - 1) / W), dim3(64 * W * SP), BYTES * W, s, ATTN_HOT_ARGS(a) a);
  return hipGetLastError();
}
// Waves per (batch, head): 4 when both sides have at least four tiles, 2 with at least two (each wave then owns whole query
// tiles in the first phase and whole key tiles in the second), else 1.  Long context (B = 64, V = 100, T = 40): 12.28 ms per
// step with one wave, 11.90 with two, 11.83 with four for the 7 x 7-tile launches; configs[1] (2-3 tiles): unchanged.
// crct_attention_force_split(1 / 2 / 4) forces a count (test hook: every count gives the same bits).
template <bool BWD, int NQ, int NK, int ND>
hipError_t launch(const AttnArgs& a, hipStream_t s) {
  const int forced = g_attn_force_split;
  constexpr bool can2 = NQ >= 2 && NK >= 2, can4 = NQ >= 4 && NK >= 4;
  if constexpr (can4) {
    if (forced == 4 || forced == 0) return launch_sp<BWD, NQ, NK, ND, 4>(a, s);
  }
  if constexpr (can2) {
    if (forced != 1) return launch_sp<BWD, NQ, NK, ND, 2>(a, s);
  }
  return launch_sp<BWD, NQ, NK, ND, 1>(a, s);
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
    case 5: case 6: case 7: return pick_d<BWD, NQ, 7>(a, s);     // 65..112 keys: padded to 7 tiles
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
    case 5: case 6: case 7: return pick_k<BWD, 7>(a, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace

bool crct_attention_mfma_ok(int Tq, int Tk, int d) { return Tq <= 112 && Tk <= 112 && (d == 32 || d == 48 || d == 64); }
hipError_t crct_attention_mfma_fwd(const AttnArgs& a, hipStream_t s) { return pick_q<false>(a, s); }
hipError_t crct_attention_mfma_bwd(const AttnArgs& a, hipStream_t s) { return pick_q<true>(a, s); }
