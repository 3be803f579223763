// bf16 MFMA GEMM for gfx950 with fused epilogues -- the dense projection / FFN contraction of the
// CRCT step (reference: every nn.Linear of CRCT/backbone/vilbert.py, e.g. :388-390, :425, :455,
// :468, :662-675, :749-752; their autograd dgrad / wgrad).
//
//   C[M][N] = epilogue( sum_k A'(m,k) * B'(n,k) )
//     TA = 0: A'(m,k) = A[m*lda + k]      TA = 1: A'(m,k) = A[k*lda + m]
//     TB = 0: B'(n,k) = B[n*ldb + k]      TB = 1: B'(n,k) = B[k*ldb + n]
//   forward  y = x W^T      : A = x  (TA=0), B = W  (TB=0)
//   dgrad    dx = dy W      : A = dy (TA=0), B = W  (TB=1, contraction over W's rows)
//   wgrad    dW = dy^T x    : A = dy (TA=1), B = x  (TB=1, contraction over the token rows)
//
// Structure: 256-thread workgroup = 4 waves (2 x 2), tile (32*TM) x (32*TN) x 64, operands staged
// global -> registers -> LDS (register prefetch of the next K tile overlaps the MFMAs),
// v_mfma_f32_16x16x32_bf16 with the operands swapped (rows of D = n) so that every lane owns 4
// consecutive output columns -> 8-byte bf16 / 16-byte fp32 stores and vector bias loads.
// K-contiguous operands are read with ds_read_b128 from an XOR-swizzled [row][64] image;
// operands whose contraction index is the slow axis (TA/TB = 1) are kept as they lie in memory,
// [k][row], and read with ds_read_b64_tr_b16 (gfx950 transposed LDS read) from the 8x32-subtile
// image of cdna_hip_programming.md T10(a) -- no transposed copies of weights or activations
// exist anywhere in HBM.
#include <vector>

#include "common.cuh"
#include "crct_internal.h"

namespace {

constexpr int BK = 64;

// byte offset of 16-byte chunk `ch` (0..7) of row r in the [R][64] bf16 image (128-B rows)
__device__ __forceinline__ int off_rowmajor(int r, int ch) { return r * 128 + ((ch ^ (r & 7)) << 4); }
// byte offset of 16-byte chunk `ch` of k-row `k` in the [64][W] bf16 image, W = 32*WC columns
template <int WC>
__device__ __forceinline__ int off_tr(int k, int ch) {
  return (WC * 512) * (k >> 3) + 512 * (ch >> 2) + 64 * (k & 7) + 16 * ((ch & 3) ^ ((k >> 2) & 3));
}

struct Frag { bf8_t v; };

// fragment for 16 rows r0.. and 32 contraction values k0.. : lane l -> X'(r0 + (l&15), k0 + 8(l>>4) + j)
template <bool T, int WC>
__device__ __forceinline__ bf8_t load_frag(const char* lds, int r0, int k0, int lane) {
  if constexpr (!T) {
    const int r = r0 + (lane & 15);
    const int ch = (k0 >> 3) + (lane >> 4);
    return *reinterpret_cast<const bf8_t*>(lds + off_rowmajor(r, ch));
  } else {
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int k = k0 + 8 * g + q;
    const int ch = (r0 >> 3) + (p >> 1);
    const int a0 = off_tr<WC>(k, ch) + 8 * (p & 1);
    const int a1 = off_tr<WC>(k + 4, ch) + 8 * (p & 1);
    typedef s4_t __attribute__((address_space(3))) * lds_s4_ptr;
    s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(uintptr_t)(uint32_t)(uintptr_t)(lds + a0));
    s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(uintptr_t)(uint32_t)(uintptr_t)(lds + a1));
    s8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf8_t, v);
  }
}

// ---- global -> register staging of one operand tile: R = 32*RC rows (output index) x 64 (contraction)
template <bool T, int RC>
struct Stage {
  static constexpr int NCHUNK = RC * 32 * BK / 8;      // 16-byte chunks in the tile
  static constexpr int PER = NCHUNK / 256;
  uint4 reg[PER];

  __device__ __forceinline__ void load(const bf16_t* __restrict__ X, long ld, int r0, int k0, int R, int K, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + 256 * i;
      int r, k;
      if constexpr (!T) { r = c >> 3; k = (c & 7) << 3; }            // [row][k chunks]
      else { k = c / (RC * 4); r = (c % (RC * 4)) << 3; }           // [k][row chunks]
      const int gr = r0 + r, gk = k0 + k;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (gr < R && gk < K) {
        const bf16_t* src = T ? (X + (long)gk * ld + gr) : (X + (long)gr * ld + gk);
        v = *reinterpret_cast<const uint4*>(src);
      }
      reg[i] = v;
    }
  }
  __device__ __forceinline__ void store(char* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + 256 * i;
      int off;
      if constexpr (!T) off = off_rowmajor(c >> 3, c & 7);
      else off = off_tr<RC>(c / (RC * 4), c % (RC * 4));
      *reinterpret_cast<uint4*>(lds + off) = reg[i];
    }
  }
};

template <int TM, int TN, bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_kernel(const CrctGemmArgs g) {
  constexpr int BM = 32 * TM, BN = 32 * TN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ldsA = smem;
  char* ldsB = smem + BM * BK * 2;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- XCD-aware tile order: blocks that land on one XCD (bid % 8) walk neighbouring tiles of
  // one row-panel, so the panel of A' stays in that XCD's L2 (cdna_hip_programming.md T1, bijective form)
  const int nwg = gridDim.x;
  const int tiles_n = (g.N + BN - 1) / BN;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, j = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
  }
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;

  const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B);

  f4_t acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f4_t{0.f, 0.f, 0.f, 0.f};

  Stage<TA, TM> sa;
  Stage<TB, TN> sb;
  const int nk = (g.K + BK - 1) / BK;
  sa.load(A, g.lda, m0, 0, g.M, g.K, tid);
  sb.load(B, g.ldb, n0, 0, g.N, g.K, tid);

  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    sa.store(ldsA, tid);
    sb.store(ldsB, tid);
    __syncthreads();
    if (kt + 1 < nk) {
      sa.load(A, g.lda, m0, (kt + 1) * BK, g.M, g.K, tid);
      sb.load(B, g.ldb, n0, (kt + 1) * BK, g.N, g.K, tid);
    }
#pragma unroll
    for (int ks = 0; ks < BK; ks += 32) {
      bf8_t fm[TM], fn[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fm[i] = load_frag<TA, TM>(ldsA, wm * (BM / 2) + i * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < TN; ++i) fn[i] = load_frag<TB, TN>(ldsB, wn * (BN / 2) + i * 16, ks, lane);
#pragma unroll
      for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fn[a], fm[b], acc[a][b], 0, 0, 0);
    }
  }

  // ------------------------------------------------------------------ epilogue
  // D rows = n (4 consecutive per lane), D cols = m (lane & 15)
  const float* bias = g.bias;
  const uint32_t thr = g.drop_thr;
  const float dscale = g.drop_scale;
#pragma unroll
  for (int b = 0; b < TM; ++b) {
    const int m = m0 + wm * (BM / 2) + b * 16 + (lane & 15);
    if (m >= g.M) continue;
#pragma unroll
    for (int a = 0; a < TN; ++a) {
      const int n = n0 + wn * (BN / 2) + a * 16 + (lane >> 4) * 4;
      if (n >= g.N) continue;
      float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
      if (g.alpha != 1.0f) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= g.alpha;
      }
      if (bias) {
        const float4 bv = *reinterpret_cast<const float4*>(bias + n);
        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
      }
      if (g.preact_out) {   // keep the pre-activation for the backward pass (bf16)
        uint2 pk = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(g.preact_out) + (long)m * g.ld_aux + n) = pk;
      }
      if (g.act != ACT_NONE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = act_apply(g.act, v[j]);
      }
      if (g.dact_src) {     // multiply by the derivative of an activation (backward through act)
        const uint2 s = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(g.dact_src) + (long)m * g.ld_aux + n);
        v[0] *= act_grad(g.dact, bf2f((bf16_t)(s.x & 0xffff)));
        v[1] *= act_grad(g.dact, bf2f((bf16_t)(s.x >> 16)));
        v[2] *= act_grad(g.dact, bf2f((bf16_t)(s.y & 0xffff)));
        v[3] *= act_grad(g.dact, bf2f((bf16_t)(s.y >> 16)));
      }
      if (thr) {            // inverted dropout, mask regenerated in backward from (seed, site, index)
        const Philox4 r = philox4x32_10(g.seed, g.drop_site, ((uint64_t)m * (uint64_t)g.N + (uint64_t)n) >> 2);
        v[0] = r.x >= thr ? v[0] * dscale : 0.f;
        v[1] = r.y >= thr ? v[1] * dscale : 0.f;
        v[2] = r.z >= thr ? v[2] * dscale : 0.f;
        v[3] = r.w >= thr ? v[3] * dscale : 0.f;
      }
      if (g.addend) {       // residual / upstream-gradient add (bf16)
        const uint2 s = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(g.addend) + (long)m * g.ld_add + n);
        v[0] += bf2f((bf16_t)(s.x & 0xffff));
        v[1] += bf2f((bf16_t)(s.x >> 16));
        v[2] += bf2f((bf16_t)(s.y & 0xffff));
        v[3] += bf2f((bf16_t)(s.y >> 16));
      }
      if (g.c_is_f32) {
        float4* dst = reinterpret_cast<float4*>(reinterpret_cast<float*>(g.C) + (long)m * g.ldc + n);
        float4 o = make_float4(v[0], v[1], v[2], v[3]);
        if (g.accumulate) { const float4 old = *dst; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
        *dst = o;
      } else {
        uint2* dst = reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(g.C) + (long)m * g.ldc + n);
        if (g.accumulate) {
          const uint2 s = *dst;
          v[0] += bf2f((bf16_t)(s.x & 0xffff)); v[1] += bf2f((bf16_t)(s.x >> 16));
          v[2] += bf2f((bf16_t)(s.y & 0xffff)); v[3] += bf2f((bf16_t)(s.y >> 16));
        }
        *dst = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
      }
    }
  }
}

template <int TM, int TN>
hipError_t launch_cfg(const CrctGemmArgs& g, hipStream_t s) {
  constexpr int BM = 32 * TM, BN = 32 * TN;
  const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  const size_t lds = (size_t)(BM + BN) * BK * 2;
  if (!g.ta && !g.tb) hipLaunchKernelGGL((gemm_kernel<TM, TN, false, false>), dim3(tiles), dim3(256), lds, s, g);
  else if (!g.ta && g.tb) hipLaunchKernelGGL((gemm_kernel<TM, TN, false, true>), dim3(tiles), dim3(256), lds, s, g);
  else if (g.ta && g.tb) hipLaunchKernelGGL((gemm_kernel<TM, TN, true, true>), dim3(tiles), dim3(256), lds, s, g);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace

// Tile choice: the CRCT GEMMs are small against 256 CUs (M = 1600 / 2880 rows); take the largest
// tile that still yields at least ~1 workgroup per CU.
extern "C" int crct_gemm_pick_tile(int M, int N) {
  auto tiles = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
  if (tiles(128, 128) >= 224) return 0;
  if (tiles(128, 64) >= 224) return 1;
  if (tiles(64, 128) >= 224) return 2;
  return 3;
}

// ---- optional live profiling: HIP events around every GEMM launch, on the launch stream ----------
// (bench.py: roofline.achieved = algorithmic FLOPs per launch / average launch duration per variant)
namespace {
struct ProfSlot { hipEvent_t a, b; int variant; };
struct Prof {
  bool on = false;
  std::vector<ProfSlot> slots;
  size_t used = 0;
  double flops[12] = {0}; long count[12] = {0};
} g_prof;
}  // namespace

extern "C" int crct_prof_enable(int on) {
  g_prof.on = on != 0;
  return 0;
}
extern "C" int crct_prof_reset(void) {
  g_prof.used = 0;
  for (int i = 0; i < 12; ++i) { g_prof.flops[i] = 0; g_prof.count[i] = 0; }
  return 0;
}
// variant = tile * 3 + {0: fwd (NT), 1: dgrad (tb), 2: wgrad (ta, tb)}.  Synchronises the events.
extern "C" int crct_prof_read(int variant, long* count, double* flops, double* ms) {
  if (variant < 0 || variant >= 12) return 1;
  double t = 0;
  for (size_t i = 0; i < g_prof.used; ++i) {
    if (g_prof.slots[i].variant != variant) continue;
    float e = 0;
    if (hipEventSynchronize(g_prof.slots[i].b) != hipSuccess) return 1;
    if (hipEventElapsedTime(&e, g_prof.slots[i].a, g_prof.slots[i].b) != hipSuccess) return 1;
    t += e;
  }
  *count = g_prof.count[variant]; *flops = g_prof.flops[variant]; *ms = t;
  return 0;
}

hipError_t crct_gemm_launch(const CrctGemmArgs& g, hipStream_t s) {
  if (g.M <= 0 || g.N <= 0) return hipSuccess;
  int t = g.tile >= 0 ? g.tile : crct_gemm_pick_tile(g.M, g.N);
  if (t > 3) t = 3;
  ProfSlot* slot = nullptr;
  if (g_prof.on) {
    if (g_prof.used == g_prof.slots.size()) {
      ProfSlot ns;
      if (hipEventCreate(&ns.a) != hipSuccess || hipEventCreate(&ns.b) != hipSuccess) return hipErrorOutOfMemory;
      g_prof.slots.push_back(ns);
    }
    slot = &g_prof.slots[g_prof.used++];
    slot->variant = t * 3 + (g.ta ? 2 : (g.tb ? 1 : 0));
    g_prof.count[slot->variant] += 1;
    g_prof.flops[slot->variant] += 2.0 * g.M * g.N * g.K;
    hipEventRecord(slot->a, s);
  }
  hipError_t e;
  switch (t) {
    case 0: e = launch_cfg<4, 4>(g, s); break;
    case 1: e = launch_cfg<4, 2>(g, s); break;
    case 2: e = launch_cfg<2, 4>(g, s); break;
    default: e = launch_cfg<2, 2>(g, s); break;
  }
  if (slot) hipEventRecord(slot->b, s);
  return e;
}
