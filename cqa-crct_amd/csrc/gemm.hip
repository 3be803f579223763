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
// Two kernels share the LDS images and the epilogue:
//   * gemm_pipe_kernel / gemm_group_kernel (K % 64 == 0, every GEMM of the full-size step): operand tiles go
//     HBM / L2 -> LDS by buffer_load_dwordx4 ... lds (no VGPR staging), a 2-4 stage ring, ONE raw s_barrier per
//     64-deep K step with a counted vmcnt; fragments are read by inline-asm ds_read_b128 / ds_read_b64_tr_b16 with
//     counted lgkmcnt waits; 4 or 8 waves per workgroup.
//   * gemm_kernel (any K % 8 == 0: tiny configurations, B-row head GEMMs): 4 waves, operands staged
//     global -> registers -> LDS with a register prefetch of the next K tile.
// Both use v_mfma_f32_16x16x32_bf16 with the operands swapped (rows of D = n) so that every lane owns 4
// consecutive output columns.  K-contiguous operands are read with ds_read_b128 from an XOR-swizzled [row][64]
// image; operands whose contraction index is the slow axis (TA/TB = 1) are kept as they lie in memory, [k][row],
// and read with ds_read_b64_tr_b16 (gfx950 transposed LDS read) from the 8x32-subtile image of
// cdna_hip_programming.md T10(a) -- no transposed copies of weights or activations exist anywhere in HBM.
#include <stdlib.h>

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <string>
#include <vector>
#include <utility>

#include "common.hip.h"
#include "crct_internal.h"

namespace {

// Live measurement (crct_prof_*): while a timing slot is armed the GEMM kernels are dispatched through hipExtLaunchKernelGGL
// with a start / stop event pair, which stamps the begin and the end of THAT kernel (what rocprofv3 --kernel-trace reports)
// instead of bracketing the launch with two extra event-record packets on the stream.
thread_local hipEvent_t g_time_start = nullptr, g_time_stop = nullptr;
template <class K, class... A>
inline void launch_kernel(K kern, dim3 grid, dim3 block, size_t lds, hipStream_t s, A... a) {
  if (g_time_start) { crct_stamp_adopt(s, g_time_start, g_time_stop); hipExtLaunchKernelGGL(kern, grid, block, (uint32_t)lds, s, g_time_start, g_time_stop, 0u, a...); }
  else crct_launch(kern, grid, block, lds, s, a...);
}

constexpr int BK = 64;
#ifndef CRCT_GEMM_NT_F32
#define CRCT_GEMM_NT_F32 1
#endif

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

// Fragment reads of the LDS-DMA kernel are inline assembly.  The compiler treats the transposing-read builtin as a
// possible LDS *store*, so behind every buffer_load ... lds still in flight it puts an s_waitcnt vmcnt(0) in front of the
// first such read: the prefetch of the next K tile then lands before the current one is used and nothing overlaps
// (measured: text FFN-up dgrad 37 us, of which 19 us exposed DMA latency; its K-contiguous twin 10 us).  An asm read
// carries no memory operand, so the counted vmcnt of the pipeline is the only wait; the price is that the compiler does
// not count these reads in lgkmcnt either: frag_async_wait<N>() must follow before the registers are used, and
// frag_async_use() pins the consumers behind that wait.  All reads of a K tile are issued in a fixed order so that the
// first half can be waited for alone.
template <int N>
__device__ __forceinline__ void frag_async_wait() {
  static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
#define CRCT_LGKM_CASE(n) else if constexpr (N == n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory");
  if constexpr (N < 0) {}
  CRCT_LGKM_CASE(0) CRCT_LGKM_CASE(1) CRCT_LGKM_CASE(2) CRCT_LGKM_CASE(3) CRCT_LGKM_CASE(4) CRCT_LGKM_CASE(5) CRCT_LGKM_CASE(6)
  CRCT_LGKM_CASE(7) CRCT_LGKM_CASE(8) CRCT_LGKM_CASE(9) CRCT_LGKM_CASE(10) CRCT_LGKM_CASE(11) CRCT_LGKM_CASE(12) CRCT_LGKM_CASE(13)
  CRCT_LGKM_CASE(14) CRCT_LGKM_CASE(15)
#undef CRCT_LGKM_CASE
}
__device__ __forceinline__ void frag_async_use(bf8_t& f) { asm volatile("" : "+v"(f)); }   // orders consumers behind the wait

// ---- fragment reads with precomputed per-lane base addresses and immediate offsets (no address arithmetic in the loop)
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int OFF>
__device__ __forceinline__ bf8_t lds_read_b128_imm(uint32_t a) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
  bf8_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ s4_t lds_read_tr_imm(uint32_t a) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
  s4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
  return v;
}
// Per-lane bases of one operand's fragments inside a stage (byte offsets from the stage start).  WT = 16-row tiles per wave,
// r0w = the wave's first row.  K-contiguous image: base[h] for the two 32-deep halves, tile i adds 2048 bytes.  Transposed
// image: base[2 * (i & 1) + {0: k, 1: k + 4}] at h = 0, tile pair i >> 1 adds 512 bytes and half h adds WC * 2048 (see off_tr).
template <bool T, int WC, int WT>
struct FragBase {
  static constexpr int NB = T ? (WT > 1 ? 4 : 2) : 2;
  uint32_t b[NB];
  __device__ __forceinline__ void init(int img_off, int r0w, int lane) {
    if constexpr (!T) {
#pragma unroll
      for (int h = 0; h < 2; ++h) b[h] = img_off + off_rowmajor(r0w + (lane & 15), 4 * h + (lane >> 4));
    } else {
      const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
#pragma unroll
      for (int par = 0; par < NB / 2; ++par) {
        const int ch = ((r0w + 16 * par) >> 3) + (p >> 1);
        b[2 * par + 0] = img_off + off_tr<WC>(8 * g + q, ch) + 8 * (p & 1);
        b[2 * par + 1] = img_off + off_tr<WC>(8 * g + q + 4, ch) + 8 * (p & 1);
      }
    }
  }
  // fragments of K half H for all WT tiles, from the stage at byte address `stage`
  __device__ __forceinline__ void at(uint32_t stage, uint32_t (&cur)[NB]) const {
#pragma unroll
    for (int j = 0; j < NB; ++j) cur[j] = stage + b[j];
  }
  template <int H>
  static __device__ __forceinline__ void read(const uint32_t (&cur)[NB], bf8_t (&f)[WT]) {
    static_for<WT>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (!T) {
        f[i] = lds_read_b128_imm<i * 2048>(cur[H]);
      } else {
        constexpr int OFF = WC * 2048 * H + 512 * (i >> 1);
        const s4_t lo = lds_read_tr_imm<OFF>(cur[2 * (i & 1) + 0]);
        const s4_t hi = lds_read_tr_imm<OFF>(cur[2 * (i & 1) + 1]);
        const s8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        f[i] = __builtin_bit_cast(bf8_t, v);
      }
    });
  }
};

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

// ------------------------------------------------------------------ XCD-aware tile placement
// Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the XCD group, each with a
// private 4 MiB L2).  The tile grid is cut into gm x gn = 8 rectangles, one per XCD, chosen so that
// the A' and B' panels one XCD touches (rm*BM + rn*BN rows of K) are as few as possible and stay
// L2-resident; block j of XCD x takes the j-th tile of rectangle x.  Placement only affects speed.
struct TileMap { int tiles_m, tiles_n, gn, rm, rn, dbg; };   // dbg: ablation bits of the -DCRCT_GEMM_LAB build (tools/gemm_lab); the shipped library ignores them

// Kernel-argument PRELOAD (round 5).  gfx950 hands the first <= 16 argument dwords to every wave in SGPRs when they are scalars or pointers
// (-mllvm -amdgpu-kernarg-preload-count=16 in the Makefile); a struct passed by value is fetched by scalar loads after the wave has
// started -- from an argument segment the host wrote microseconds ago.  Every GEMM entry kernel therefore takes what its FIRST instructions
// need -- operand pointers and leading dimensions, the problem size, the tile map -- as 14 leading scalars, and the full CrctGemmArgs /
// TileMap behind them for everything the epilogue reads later.  Dependent chain of small kernels: 10.05 -> 9.77 us per launch
// (tools/lab/kernarg_probe.hip); the step: -0.05 ms (profiles/r5_kernarg_preload_ab.txt).
#define GEMM_HOT_PARAMS                                                                                                                    \
  const void* hA, const void* hB, int hlda, int hldb, int hM, int hN, int hK, int t_m, int t_n, int t_gn, int t_rm, int t_rn,             \
      const CrctGemmArgs g_in, const TileMap tmap_in
#define GEMM_HOT_UNPACK                                                                                                                    \
  CrctGemmArgs g = g_in;                                                                                                                   \
  g.A = hA; g.B = hB; g.lda = hlda; g.ldb = hldb; g.M = hM; g.N = hN; g.K = hK;                                                            \
  TileMap tmap = tmap_in;                                                                                                                  \
  tmap.tiles_m = t_m; tmap.tiles_n = t_n; tmap.gn = t_gn; tmap.rm = t_rm; tmap.rn = t_rn;
#define GEMM_HOT_ARGS(g, tmap) (g).A, (g).B, (int)(g).lda, (int)(g).ldb, (g).M, (g).N, (g).K, (tmap).tiles_m, (tmap).tiles_n, (tmap).gn, (tmap).rm, (tmap).rn,
#ifdef CRCT_GEMM_LAB
__device__ __forceinline__ int lab_bits(int dbg) { return dbg; }
#else
__device__ __forceinline__ constexpr int lab_bits(int) { return 0; }      // no environment variable can make a production kernel skip work
#endif

__device__ __forceinline__ bool map_tile(const TileMap& t, int bid, int& tm, int& tn) {
  const int x = bid & 7, j = bid >> 3;
  const int m_lo = (x / t.gn) * t.rm, n_lo = (x % t.gn) * t.rn;
  const int hm = min(t.rm, t.tiles_m - m_lo), hn = min(t.rn, t.tiles_n - n_lo);
  if (hm <= 0 || hn <= 0 || j >= hm * hn) return false;
  tm = m_lo + j / hn;
  tn = n_lo + j % hn;
  return true;
}

inline TileMap make_tile_map(int M, int N, int BM, int BN, int* grid) {
  TileMap t;
  t.tiles_m = (M + BM - 1) / BM; t.tiles_n = (N + BN - 1) / BN;
  long best = -1;
  int bgm = 1;
  for (int gm = 1; gm <= 8; gm *= 2) {
    const int gn = 8 / gm;
    const int rm = (t.tiles_m + gm - 1) / gm, rn = (t.tiles_n + gn - 1) / gn;
    // panel rows held per XCD, plus a penalty for padded (idle) blocks
    const long cost = (long)rm * BM + (long)rn * BN + 4L * ((long)rm * rn * 8 - (long)t.tiles_m * t.tiles_n) * 16;
    if (best < 0 || cost < best) { best = cost; bgm = gm; }
  }
  t.gn = 8 / bgm;
  t.rm = (t.tiles_m + bgm - 1) / bgm; t.rn = (t.tiles_n + t.gn - 1) / t.gn;
  *grid = 8 * t.rm * t.rn;
#ifdef CRCT_GEMM_LAB   // timing ablations exist in the -DCRCT_GEMM_LAB build of tools/gemm_lab only
  static const int dbg = getenv("CRCT_GEMM_DBG") ? atoi(getenv("CRCT_GEMM_DBG")) : 0;
  t.dbg = dbg;
#else
  t.dbg = 0;
#endif
  return t;
}

// ------------------------------------------------------------------ shared epilogue
// D rows = n (4 consecutive per lane), D cols = m (lane & 15)
// (mw0, nw0) = origin of this wave's sub-tile; TM x TN = its 16x16 MFMA tiles
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const CrctGemmArgs& g, f4_t (&acc)[TN][TM], int mw0, int nw0, int lane) {
  const float* bias = g.bias;
  const uint32_t thr = g.drop_thr;
  const float dscale = g.drop_scale;
#pragma unroll
  for (int b = 0; b < TM; ++b) {
    const int m = mw0 + b * 16 + (lane & 15);
    if (m >= g.M) continue;
#pragma unroll
    for (int a = 0; a < TN; ++a) {
      const int n = nw0 + a * 16 + (lane >> 4) * 4;
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
        const uint64_t idx = (uint64_t)m * (uint64_t)g.N + (uint64_t)n;        // n % 4 == 0: this thread's four are one half of a group of 8
        const uint32_t kb = philox_keep8(g.seed, g.drop_site, idx >> 3, thr) >> ((uint32_t)idx & 4u);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ((kb >> j) & 1u) ? v[j] * dscale : 0.f;
      }
      if (g.addend && g.addend_f32) {       // the fp32 residual stream (CrctGemmArgs.addend_f32)
        const float4 s = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(g.addend) + (long)m * g.ld_add + n);
        v[0] += s.x; v[1] += s.y; v[2] += s.z; v[3] += s.w;
      } else if (g.addend) {       // residual / upstream-gradient add (bf16)
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

template <int TM, int TN, bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_kernel(GEMM_HOT_PARAMS) {
  GEMM_HOT_UNPACK
  constexpr int BM = 32 * TM, BN = 32 * TN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ldsA = smem;
  char* ldsB = smem + BM * BK * 2;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  int tile_m, tile_n;
  if (!map_tile(tmap, blockIdx.x, tile_m, tile_n)) return;     // padding block of a short edge region
  const int m0 = tile_m * BM, n0 = tile_n * BN;

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

  gemm_epilogue<TM, TN>(g, acc, m0 + wm * (BM / 2), n0 + wn * (BN / 2), lane);
}

// ------------------------------------------------------------------ LDS-staged epilogue (pipelined kernel)
// The MFMA accumulator layout gives a lane 4 consecutive columns of ONE row, so storing straight from
// registers issues 8-byte accesses scattered over 16 rows per instruction (measured: 5 of 17 us on the
// text FFN-up GEMM).  Instead each wave row-group parks its fp32 accumulators in the (now idle) operand
// ring, and all threads walk the tile as (row, 8-column chunk): every global access of the epilogue --
// bias, saved pre-activation, activation-derivative source, residual / upstream-gradient addend, fp32
// accumulate and the output itself -- is then 16 bytes per lane, consecutive lanes on consecutive chunks
// of a row (full 128-byte row segments).  Same arithmetic, same Philox element indexing as gemm_epilogue.
// As many wave rows per pass as the ring holds (RING bytes): usually the whole tile in ONE pass (2 workgroup barriers
// in all and every thread busy) instead of one pass per wave row.
// OCP e4m3: largest finite value 448; the hardware conversion does not saturate, so clamp first
__device__ __forceinline__ float f8_clamp(float x) { return fminf(fmaxf(x, -448.0f), 448.0f); }

template <int BM, int BN, int WM, int RING>
constexpr int epilogue_passes() {
  for (int p = 1; p <= WM; p *= 2)
    if (WM % p == 0 && (BM / p) * (BN + 4) * 4 <= RING) return p;
  return 0;
}

// NTH: threads that walk the staged tile (default: the WM x WN compute waves; the loader-wave kernels pass their whole workgroup --
// waves beyond WM x WN hold no accumulators, their wm is >= WM and they never park anything)
template <int BM, int BN, int WM, int WN, int WTM, int WTN, int RING, int NTH = 0>
__device__ __forceinline__ void gemm_epilogue_staged(const CrctGemmArgs& g_in, f4_t (&acc)[WTN][WTM], char* smem, int m0, int n0,
                                                     int wm, int wn, int lane, int tid, int dbg = 0) {
#ifdef CRCT_GEMM_LAB   // lab ablations (tools/lab/step_ablate.sh): 64 = no activation / derivative / dropout arithmetic, 128 = no side inputs or outputs
  CrctGemmArgs g = g_in;
  if (dbg & 64) { g.act = ACT_NONE; g.dact_src = nullptr; g.drop_thr = 0; }
  if (dbg & 128) { g.preact_out = nullptr; g.addend = nullptr; g.bias = nullptr; g.dact_src = nullptr; g.q_out = nullptr; }
#else
  const CrctGemmArgs& g = g_in;
#endif
  constexpr int P = epilogue_passes<BM, BN, WM, RING>();
  static_assert(P >= 1, "staging tile must fit into the operand ring");
  constexpr int R = BM / P;                  // rows staged per pass (WM / P wave rows of the wave grid)
  constexpr int WR = BM / WM;                // rows of one wave row
  constexpr int LDC = BN + 4;                // floats; 16-byte aligned rows, +4 breaks the power-of-two stride
  constexpr int CPR = BN / 8;                // 8-column chunks per row
  constexpr int NT = NTH ? NTH : WM * WN * 64;
  float* ct = reinterpret_cast<float*>(smem);
  const float* bias = g.bias;
  const uint32_t thr = g.drop_thr;
  const float dscale = g.drop_scale;
  // fp8 operands were quantised as q = x * scale: the product is divided by both scales (device scalars, delayed scaling)
  const float alpha = g.scale_a ? g.alpha / (g.scale_a[0] * g.scale_b[0]) : g.alpha;
  const float qs = g.q_out ? g.q_scale[0] : 0.f;
  float q_amax = 0.f;
  __syncthreads();                           // every wave is done reading the operand ring
#pragma unroll 1
  for (int pass = 0; pass < P; ++pass) {
    if (wm / (WM / P) == pass) {
#pragma unroll
      for (int b = 0; b < WTM; ++b)
#pragma unroll
        for (int a = 0; a < WTN; ++a) {
          const int r = (wm % (WM / P)) * WR + b * 16 + (lane & 15), c = wn * (BN / WN) + a * 16 + (lane >> 4) * 4;
          *reinterpret_cast<f4_t*>(ct + r * LDC + c) = acc[a][b];
        }
    }
    __syncthreads();
    for (int idx = tid; idx < R * CPR; idx += NT) {
      const int r = idx / CPR, ch = idx % CPR;
      const int m = m0 + pass * R + r, n = n0 + ch * 8;
      if (m >= g.M || n >= g.N) continue;
      const float4 lo = *reinterpret_cast<const float4*>(ct + r * LDC + ch * 8);
      const float4 hi = *reinterpret_cast<const float4*>(ct + r * LDC + ch * 8 + 4);
      float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      if (alpha != 1.0f) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= alpha;
      }
      if (bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(bias + n), b1 = *reinterpret_cast<const float4*>(bias + n + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
      }
      if (g.preact_out) {
        const uint4 pk = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(g.preact_out) + (long)m * g.ld_aux + n) = pk;
      }
      if (g.act != ACT_NONE) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = act_apply(g.act, v[j]);
      }
      if (g.dact_src) {
        const uint4 sv = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(g.dact_src) + (long)m * g.ld_aux + n);
        const uint32_t w[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[2 * j] *= act_grad(g.dact, bf2f((bf16_t)(w[j] & 0xffff)));
          v[2 * j + 1] *= act_grad(g.dact, bf2f((bf16_t)(w[j] >> 16)));
        }
      }
      if (thr) {
        const uint32_t kb = philox_keep8(g.seed, g.drop_site, ((uint64_t)m * (uint64_t)g.N + (uint64_t)n) >> 3, thr);      // n % 8 == 0, N % 8 == 0
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ((kb >> j) & 1u) ? v[j] * dscale : 0.f;
      }
      if (g.addend && g.addend_f32) {       // the fp32 residual stream (CrctGemmArgs.addend_f32)
        const float4* ap = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(g.addend) + (long)m * g.ld_add + n);
        const float4 s0 = ap[0], s1 = ap[1];
        v[0] += s0.x; v[1] += s0.y; v[2] += s0.z; v[3] += s0.w; v[4] += s1.x; v[5] += s1.y; v[6] += s1.z; v[7] += s1.w;
      } else if (g.addend) {
        const uint4 sv = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(g.addend) + (long)m * g.ld_add + n);
        const uint32_t w[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[2 * j] += bf2f((bf16_t)(w[j] & 0xffff)); v[2 * j + 1] += bf2f((bf16_t)(w[j] >> 16)); }
      }
      if (g.q_out) {       // e4m3 copy of the output for the next fp8 GEMM (FFN-up -> FFN-down), amax for the next step's scale
        uint32_t w0 = 0, w1 = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) q_amax = fmaxf(q_amax, fabsf(v[j]));
        if (g.fp8 & 4) {     // e5m2 copy: a gradient that the next data-gradient GEMM reads (largest finite value 57344)
          auto c5 = [](float x) { return fminf(fmaxf(x, -57344.0f), 57344.0f); };
          w0 = __builtin_amdgcn_cvt_pk_bf8_f32(c5(v[0] * qs), c5(v[1] * qs), w0, false);
          w0 = __builtin_amdgcn_cvt_pk_bf8_f32(c5(v[2] * qs), c5(v[3] * qs), w0, true);
          w1 = __builtin_amdgcn_cvt_pk_bf8_f32(c5(v[4] * qs), c5(v[5] * qs), w1, false);
          w1 = __builtin_amdgcn_cvt_pk_bf8_f32(c5(v[6] * qs), c5(v[7] * qs), w1, true);
        } else {
          w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f8_clamp(v[0] * qs), f8_clamp(v[1] * qs), w0, false);
          w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f8_clamp(v[2] * qs), f8_clamp(v[3] * qs), w0, true);
          w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f8_clamp(v[4] * qs), f8_clamp(v[5] * qs), w1, false);
          w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f8_clamp(v[6] * qs), f8_clamp(v[7] * qs), w1, true);
        }
        *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(g.q_out) + (long)m * g.ld_q + n) = make_uint2(w0, w1);
      }
      if (g.c_is_f32) {
#if CRCT_GEMM_NT_F32   // fp32 outputs are weight gradients: written once per step, read by AdamW / the all-reduce much later
        f4_t* dst = reinterpret_cast<f4_t*>(reinterpret_cast<float*>(g.C) + (long)m * g.ldc + n);
        f4_t o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
        if (g.c_cached) {         // ... or pre-LayerNorm sums of the fp32 residual stream, which the next kernel reads: ordinary stores
          dst[0] = o0; dst[1] = o1;
          continue;
        }
        if (g.accumulate) {
          o0 += __builtin_nontemporal_load(dst);
          o1 += __builtin_nontemporal_load(dst + 1);
        }
        __builtin_nontemporal_store(o0, dst);
        __builtin_nontemporal_store(o1, dst + 1);
#else
        float4* dst = reinterpret_cast<float4*>(reinterpret_cast<float*>(g.C) + (long)m * g.ldc + n);
        float4 o0 = make_float4(v[0], v[1], v[2], v[3]), o1 = make_float4(v[4], v[5], v[6], v[7]);
        if (g.accumulate) {
          const float4 p0 = dst[0], p1 = dst[1];
          o0.x += p0.x; o0.y += p0.y; o0.z += p0.z; o0.w += p0.w; o1.x += p1.x; o1.y += p1.y; o1.z += p1.z; o1.w += p1.w;
        }
        dst[0] = o0; dst[1] = o1;
#endif
      } else {
        uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(g.C) + (long)m * g.ldc + n);
        if (g.accumulate) {
          const uint4 sv = *dst;
          const uint32_t w[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[2 * j] += bf2f((bf16_t)(w[j] & 0xffff)); v[2 * j + 1] += bf2f((bf16_t)(w[j] >> 16)); }
        }
        const uint4 pk = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
        *dst = pk;
      }
    }
    if (pass + 1 < P) __syncthreads();       // the staging tile is rewritten by the next group of wave rows
  }
  if (g.q_out && g.q_amax) {                 // one atomic per wave; |x| >= 0, so the integer order of the bits is the float order
    q_amax = wave_max(q_amax);
    if (lane == 0) amax_update(g.q_amax, q_amax);
  }
}

// ====================================================================================== split-K seam
// S workgroups share one output tile.  Each parks its accumulators as they lie in registers (slab layout: 16-byte unit
// (a * WTM + b) * threads + tid, so every store / load instruction of a wave covers 1 KiB of consecutive bytes) with
// WRITE-THROUGH stores (sc1: the bytes are at the memory side once vmcnt has drained -- no release fence, which would write
// back the whole L2), drains, meets its workgroup at a barrier, and ONE lane draws a ticket (agent-scope atomic).  The
// workgroup that draws S - 1 knows the other S - 1 slabs are complete: one agent-scope acquire (drops this CU's L1 lines),
// then it reads ALL S slabs -- its own included, so the summation order is slice 0, 1, .. S - 1 whichever slice it is -- with
// sc1 loads, and carries on into the ordinary epilogue.  Correct for any placement of a tile's slices over CUs / XCDs
// (cdna_hip_programming.md section 5 "Projection GEMM at M = 256" item 2, Guideline 16 R1); the block -> (tile, slice) map only
// keeps a tile's slices on one XCD for speed.  The ticket is put back to 0 by the last arriver (all S adds have happened).
typedef unsigned u4_t __attribute__((ext_vector_type(4)));
template <int BM, int BN, int NW, int WTM, int WTN>
__device__ __forceinline__ bool splitk_reduce(const CrctGemmArgs& g, f4_t (&acc)[WTN][WTM], char* smem, int slice, int tile_lin, int tid) {
  const int S = g.split_k;
  if (S <= 1) return true;
  constexpr int NT = NW * 64;
  constexpr int SLAB = BM * BN * 4;                                   // bytes per slab
  static_assert(WTM * WTN * NT * 16 == SLAB, "slab layout covers the tile");
  char* tile_ws = reinterpret_cast<char*>(g.splitk_ws) + (size_t)tile_lin * S * SLAB;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(tile_ws, 0, S * SLAB, 0x00020000);
#pragma unroll
  for (int a = 0; a < WTN; ++a)
#pragma unroll
    for (int b = 0; b < WTM; ++b)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, acc[a][b]), rw, slice * SLAB + ((a * WTM + b) * NT + tid) * 16, 0, 16 /* sc1 */);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // EVERY storing wave drains its stores ...
  __syncthreads();                                                    // ... before the one lane that signals for all of them
  unsigned* flag = reinterpret_cast<unsigned*>(smem);                 // the operand ring is idle: every wave is past its last fragment read
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(g.splitk_cnt + tile_lin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == (unsigned)(S - 1)) {
      __hip_atomic_store(g.splitk_cnt + tile_lin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    flag[0] = t;
  }
  __syncthreads();
  const unsigned ticket = flag[0];
  if (ticket != (unsigned)(S - 1)) return false;
#pragma unroll
  for (int a = 0; a < WTN; ++a)
#pragma unroll
    for (int b = 0; b < WTM; ++b) acc[a][b] = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int s2 = 0; s2 < S; ++s2) {
    u4_t v[WTN][WTM];
#pragma unroll
    for (int a = 0; a < WTN; ++a)
#pragma unroll
      for (int b = 0; b < WTM; ++b) v[a][b] = __builtin_amdgcn_raw_buffer_load_b128(rw, s2 * SLAB + ((a * WTM + b) * NT + tid) * 16, 0, 16 /* sc1 */);
#pragma unroll
    for (int a = 0; a < WTN; ++a)
#pragma unroll
      for (int b = 0; b < WTM; ++b) acc[a][b] += __builtin_bit_cast(f4_t, v[a][b]);
  }
  return true;
}

// ====================================================================================== pipelined
// LDS-DMA variant (K % 64 == 0): operand tiles go HBM/L2 -> LDS with buffer_load_dwordx4 ... lds
// (no VGPR staging, no ds_write), NS stages deep, ONE raw s_barrier per K-step and a counted
// s_waitcnt vmcnt(N) that leaves the younger tiles in flight across the barrier
// (cdna_hip_programming.md section 5, "Pipelining across barriers").  The CRCT GEMMs give each CU
// about one tile, so there are no other workgroups to hide HBM/L2 latency behind: the prefetch depth
// inside the workgroup is what matters.  LDS images are identical to the generic kernel's; because
// the DMA writes lane-linearly, the swizzle is applied to each lane's SOURCE address (rule 21).
// Rows beyond M / N are fetched through an out-of-range buffer offset (hardware returns zeros).
constexpr unsigned OOB_OFF = 0x80000000u;
typedef __attribute__((address_space(3))) void* lds_void_ptr;

template <bool T, int RC>
__device__ __forceinline__ unsigned dma_src_offset(int slot, int r0, int R, long ld, int dbg = 0) {
  // slot = 16-byte slot index inside the operand image; returns the byte offset of its source chunk at k0 = 0
#ifdef CRCT_GEMM_LAB       // timing ablations of tools/gemm_lab only (wrong results)
  if (T && (dbg & 4)) {     // whole source rows by consecutive lanes
    const int k = slot / (RC * 4), ch = slot % (RC * 4);
    return (unsigned)(((long)k * ld + r0 + ch * 8) * 2);
  }
  if (T && (dbg & 8)) {     // the K-contiguous operand's address stream (rows of 128 B advancing along the row)
    const int r = slot >> 3, ch = slot & 7;
    return (unsigned)((((long)(r0 / 8 + r)) * ld + ch * 8) * 2);
  }
#endif
  if constexpr (!T) {
    const int r = slot >> 3, ch = (slot & 7) ^ (r & 7);
    return (r0 + r < R) ? (unsigned)((((long)(r0 + r)) * ld + ch * 8) * 2) : OOB_OFF;
  } else {
    const int byte = slot << 4;
    const int grp = byte / (RC * 512), rem = byte % (RC * 512);
    const int sub = rem >> 9, rem2 = rem & 511;
    const int k = grp * 8 + (rem2 >> 6);
    const int c3 = (rem2 & 63) >> 4;
    const int ch = sub * 4 + (c3 ^ ((k >> 2) & 3));
    return (r0 + ch * 8 < R) ? (unsigned)(((long)k * ld + r0 + ch * 8) * 2) : OOB_OFF;
  }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
#define CRCT_VMCNT_CASE(n) else if constexpr (N == n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory");
  if constexpr (N < 0) {}
  CRCT_VMCNT_CASE(0) CRCT_VMCNT_CASE(1) CRCT_VMCNT_CASE(2) CRCT_VMCNT_CASE(3) CRCT_VMCNT_CASE(4) CRCT_VMCNT_CASE(5)
  CRCT_VMCNT_CASE(6) CRCT_VMCNT_CASE(7) CRCT_VMCNT_CASE(8) CRCT_VMCNT_CASE(9) CRCT_VMCNT_CASE(10) CRCT_VMCNT_CASE(11)
  CRCT_VMCNT_CASE(12) CRCT_VMCNT_CASE(13) CRCT_VMCNT_CASE(14) CRCT_VMCNT_CASE(15) CRCT_VMCNT_CASE(16) CRCT_VMCNT_CASE(18)
  CRCT_VMCNT_CASE(20) CRCT_VMCNT_CASE(21) CRCT_VMCNT_CASE(24) CRCT_VMCNT_CASE(27) CRCT_VMCNT_CASE(28) CRCT_VMCNT_CASE(30)
  CRCT_VMCNT_CASE(32) CRCT_VMCNT_CASE(36) CRCT_VMCNT_CASE(40) CRCT_VMCNT_CASE(48)
  else static_assert(N == 0, "add the vmcnt literal");
#undef CRCT_VMCNT_CASE
}

// Block tile (32*TM) x (32*TN), WM x WN waves (4 or 8), each wave owning a (BM/WM) x (BN/WN) sub-tile.
// SK: K-partitioned variant (CrctGemmArgs.split_k): workgroup (tile, slice) contracts K tiles [kt0, kt0 + nk) and the last of
// a tile's slices to arrive reduces the slabs (see splitk_reduce).
// PM = 1: register-pipelined main loop (the fragments of K tile k + 1 are read from LDS while the MFMAs of tile k run): what the
// 128 x 64 tiles cannot use -- they sit on the ~70 GB/s per CU L2 -> LDS fill rate either way -- but the larger tiles need: with
// half the fill bytes per FLOP their time is the serial "read fragments, then multiply" of the plain loop.
template <int TM, int TN, int WM, int WN, bool TA, bool TB, int NS, bool SK = false, int PM = 0>
__device__ __forceinline__ void gemm_pipe_body(const CrctGemmArgs& g, const int tile_m, const int tile_n, const int dbg,
                                               const int slice = 0, const int tile_lin = 0) {
  static_assert(PM == 0 || PM == 1, "PM: 0 plain loop, 1 register-pipelined loop");
  constexpr int BM = 32 * TM, BN = 32 * TN, NW = WM * WN;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int PA = A_BYTES / 1024 / NW, PB = B_BYTES / 1024 / NW;     // 1-KiB DMA pieces per wave per K tile
  constexpr int L = PA + PB;
  constexpr int WTM = BM / WM / 16, WTN = BN / WN / 16;                  // 16x16 MFMA tiles per wave
  static_assert(PA >= 1 && PB >= 1 && PA * NW * 1024 == A_BYTES && PB * NW * 1024 == B_BYTES, "tile / wave-grid mismatch");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), 0, (int)OOB_OFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.B), 0, (int)OOB_OFF, 0x00020000);
  unsigned offA[PA], offB[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) offA[i] = dma_src_offset<TA, TM>((i * NW + wave) * 64 + lane, m0, g.M, g.lda, dbg);
#pragma unroll
  for (int i = 0; i < PB; ++i) offB[i] = dma_src_offset<TB, TN>((i * NW + wave) * 64 + lane, n0, g.N, g.ldb, dbg);
  const int stepA = TA ? (int)(64 * g.lda * 2) : 128;     // bytes per K tile
#ifdef CRCT_GEMM_LAB
  const int stepB = TB && !(dbg & 8) ? (int)(64 * g.ldb * 2) : 128;
#else
  const int stepB = TB ? (int)(64 * g.ldb * 2) : 128;
#endif

  f4_t acc[WTN][WTM];
#pragma unroll
  for (int a = 0; a < WTN; ++a)
#pragma unroll
    for (int b = 0; b < WTM; ++b) acc[a][b] = f4_t{0.f, 0.f, 0.f, 0.f};

  // optional row sums of A' (bias gradient of a weight-gradient GEMM): only the workgroups of the first tile column
  // compute them; the WN waves of a wave row share the K sub-steps round-robin, so the extra MFMAs are 1/WN of a wave's
  const bool do_rs = g.rowsum_out != nullptr && tile_n == 0;
  f4_t accb[WTM];
#pragma unroll
  for (int b = 0; b < WTM; ++b) accb[b] = f4_t{0.f, 0.f, 0.f, 0.f};
  bf8_t ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

  int kt0 = 0, nk = g.K / BK;
  if constexpr (SK) {
    const int S = g.split_k, nk_all = nk;
    kt0 = (int)((long)slice * nk_all / S);
    nk = (int)((long)(slice + 1) * nk_all / S) - kt0;
  }
  auto issue = [&](int kt, int st) {
    char* base = smem + st * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < PA; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void_ptr)(base + i * NW * 1024), 16, (int)offA[i], (kt0 + kt) * stepA, 0, 0);
#pragma unroll
    for (int i = 0; i < PB; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void_ptr)(base + A_BYTES + i * NW * 1024), 16, (int)offB[i], (kt0 + kt) * stepB, 0, 0);
  };

  const int npre = nk < NS - 1 ? nk : NS - 1;
  for (int t = 0; t < npre; ++t) issue(t, t);

  // All fragment reads are uncounted asm reads (see frag_async_wait) from per-lane base addresses computed once, with
  // immediate offsets: the only address arithmetic per K tile is adding the stage offset to the <= 8 bases.
  static_assert(BK == 64, "two MFMA K-halves per tile");
  FragBase<TA, TM, WTM> fbA;
  FragBase<TB, TN, WTN> fbB;
  fbA.init(0, wm * (BM / WM), lane);
  fbB.init(A_BYTES, wn * (BN / WN), lane);
  const uint32_t smem_base = (uint32_t)(uintptr_t)smem;
  constexpr int N_HALF = WTM * (TA ? 2 : 1) + WTN * (TB ? 2 : 1);      // LDS reads per half (a transposed fragment takes two)
  auto wait_tile = [&](int t) {        // tile t has landed; the younger tiles issued so far may stay in flight
    const int ahead = (nk - 1 < t + NS - 2 ? nk - 1 : t + NS - 2) - t;
    if (NS >= 6 && ahead >= 4) wait_vmcnt<4 * L>();
    else if (NS >= 5 && ahead >= 3) wait_vmcnt<3 * L>();
    else if (NS >= 4 && ahead >= 2) wait_vmcnt<2 * L>();
    else if (NS >= 3 && ahead >= 1) wait_vmcnt<L>();
    else wait_vmcnt<0>();
  };
  auto request = [&](int stg, bf8_t (&fm)[2][WTM], bf8_t (&fn)[2][WTN]) {
    uint32_t ca[FragBase<TA, TM, WTM>::NB], cb[FragBase<TB, TN, WTN>::NB];
    fbA.at(smem_base + stg * STAGE, ca);
    fbB.at(smem_base + stg * STAGE, cb);
    FragBase<TA, TM, WTM>::template read<0>(ca, fm[0]);
    FragBase<TB, TN, WTN>::template read<0>(cb, fn[0]);
    FragBase<TA, TM, WTM>::template read<1>(ca, fm[1]);
    FragBase<TB, TN, WTN>::template read<1>(cb, fn[1]);
  };
  auto multiply = [&](int h, bf8_t (&fm)[2][WTM], bf8_t (&fn)[2][WTN]) {
#pragma unroll
    for (int i = 0; i < WTM; ++i) frag_async_use(fm[h][i]);
#pragma unroll
    for (int i = 0; i < WTN; ++i) frag_async_use(fn[h][i]);
#pragma unroll
    for (int a = 0; a < WTN; ++a)
#pragma unroll
      for (int b = 0; b < WTM; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fn[h][a], fm[h][b], acc[a][b], 0, 0, 0);
  };
  auto rowsums = [&](int kt, bf8_t (&fm)[2][WTM]) {
    if (do_rs) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if ((((kt << 1) + h) & (WN - 1)) == wn) {
#pragma unroll
          for (int b = 0; b < WTM; ++b) accb[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fm[h][b], accb[b], 0, 0, 0);
        }
    }
  };
  if constexpr (PM == 1) {
  // register-pipelined: while the MFMAs of tile kt run from one register set, the reads of tile kt+1 fill the other;
  // tile t lives in stage t % NS and its stage goes back to the DMA one barrier after its reads have completed
    bf8_t fmA[2][WTM], fnA[2][WTN], fmB[2][WTM], fnB[2][WTN];
    auto step = [&](int kt, int stg, bf8_t (&fm)[2][WTM], bf8_t (&fn)[2][WTN], bf8_t (&fm_n)[2][WTM], bf8_t (&fn_n)[2][WTN]) {
      frag_async_wait<0>();                            // tile kt is in registers
      if (kt + 1 < nk) {
        wait_tile(kt + 1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + NS < nk && !(lab_bits(dbg) & 2)) issue(kt + NS, stg);
        request(stg + 1 == NS ? 0 : stg + 1, fm_n, fn_n);
      }
      multiply(0, fm, fn);
      multiply(1, fm, fn);
      rowsums(kt, fm);
    };
    wait_tile(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (NS - 1 < nk && !(lab_bits(dbg) & 2)) issue(NS - 1, NS - 1);
    request(0, fmA, fnA);
    int stg = 0;
    for (int kt = 0; kt < nk; kt += 2) {
      step(kt, stg, fmA, fnA, fmB, fnB);
      stg = stg + 1 == NS ? 0 : stg + 1;
      if (kt + 1 < nk) {
        step(kt + 1, stg, fmB, fnB, fmA, fnA);
        stg = stg + 1 == NS ? 0 : stg + 1;
      }
    }
  } else {
  int st = 0, st_next = NS - 1;      // stage holding tile kt; stage the next prefetch goes to
  for (int kt = 0; kt < nk; ++kt) {
    wait_tile(kt);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + NS - 1 < nk && !(lab_bits(dbg) & 2)) issue(kt + NS - 1, st_next);
    // fragments of both 32-deep halves of the K tile are requested up front; the MFMAs of the first half run while the
    // reads of the second are still in flight
    bf8_t fm[2][WTM], fn[2][WTN];
    request(st, fm, fn);
    frag_async_wait<(N_HALF <= 15 ? N_HALF : 0)>();
    multiply(0, fm, fn);
    asm volatile("" : "+v"(acc[WTN - 1][WTM - 1]));     // keep the first half's MFMAs in front of the second wait
    frag_async_wait<0>();
    multiply(1, fm, fn);
    rowsums(kt, fm);
    st_next = st;
    st = st + 1 == NS ? 0 : st + 1;
  }
  }
  if (lab_bits(dbg) & 1) {      // ablation (lab build only): keep the accumulators alive, skip the epilogue
#pragma unroll
    for (int a = 0; a < WTN; ++a)
#pragma unroll
      for (int b = 0; b < WTM; ++b) asm volatile("" ::"v"(acc[a][b]));
    return;
  }
  if constexpr (SK) {
    if (!splitk_reduce<BM, BN, NW, WTM, WTN>(g, acc, smem, slice, tile_lin, tid)) return;     // not the last slice of this tile
  }
  if (do_rs) {
    // every row of the 16x16 result holds the same sums: row 0 lives in lanes 0..15, register 0
    static_assert((WN & (WN - 1)) == 0 && WN * BM * 4 <= NS * STAGE, "row-sum staging");
    float* rs = reinterpret_cast<float*>(smem);            // [WN][BM]
    __syncthreads();                                       // every wave is done reading the operand ring
    if (lane < 16) {
#pragma unroll
      for (int b = 0; b < WTM; ++b) rs[wn * BM + wm * (BM / WM) + b * 16 + lane] = accb[b][0];
    }
    __syncthreads();
    for (int i = tid; i < BM; i += NW * 64) {
      if (m0 + i < g.M) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < WN; ++w) v += rs[w * BM + i];  // fixed order: reproducible
        g.rowsum_out[m0 + i] += v;
      }
    }
  }
  gemm_epilogue_staged<BM, BN, WM, WN, WTM, WTN, NS * STAGE>(g, acc, smem, m0, n0, wm, wn, lane, tid, lab_bits(dbg));
}

template <int TM, int TN, int WM, int WN, bool TA, bool TB, int NS, int PM = 0>
__global__ __launch_bounds__(WM * WN * 64) void gemm_pipe_kernel(GEMM_HOT_PARAMS) {
  GEMM_HOT_UNPACK
  if constexpr (!TA) crct_chain_priority();      // forward / data-gradient GEMMs: the data streams (weight gradients stay at 0)
  int tile_m, tile_n;
  if (!map_tile(tmap, blockIdx.x, tile_m, tile_n)) return;     // padding block of a short edge region
  gemm_pipe_body<TM, TN, WM, WN, TA, TB, NS, false, PM>(g, tile_m, tile_n, tmap.dbg);
}

// K-partitioned launch: block j of XCD x is slice j % S of the (j / S)-th tile of that XCD's rectangle -- a tile's slices share
// an L2 (speed only).  Grid = 8 * rm * rn * S.
template <int TM, int TN, int WM, int WN, bool TA, bool TB, int NS>
__global__ __launch_bounds__(WM * WN * 64) void gemm_splitk_kernel(GEMM_HOT_PARAMS) {
  GEMM_HOT_UNPACK
  crct_chain_priority();
  const int S = g.split_k;
  const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
  int tile_m, tile_n;
  if (!map_tile(tmap, ((j / S) << 3) | x, tile_m, tile_n)) return;
  gemm_pipe_body<TM, TN, WM, WN, TA, TB, NS, true>(g, tile_m, tile_n, 0, j % S, tile_m * tmap.tiles_n + tile_n);
}

// ====================================================================================== loader waves (round 4)
// What the plain loop above pays per K step is not arithmetic and not the fill path's bandwidth (tools/lab/fill_probe.hip: a CU
// takes in 124 - 135 GB/s L2-hot through buffer_load ... lds with all 256 CUs streaming, 150 with 64) but the ISSUE of the fill:
// a CU's texture path accepts one 1-KiB piece per ~13 - 16 cycles, the 24 - 32 pieces of a K step are issued by all eight waves
// together right behind the barrier, and every wave sits in that queue (~300 cycles) before it may read a fragment or start an
// MFMA.  Splitting the waves into two half-step-shifted groups (PM = 2) does not help -- the group that loads still waits out
// its own pieces (measured: slower everywhere, profiles/r4_gemm_lab_*).  Here the fill has waves of its own: NL loader waves
// (one per SIMD at NL = 4) own the whole LDS-DMA ring -- all source offsets, every issue, the counted vmcnt wait -- and the
// WM x WN compute waves never execute a vector-memory instruction inside the K loop: barrier, fragment reads, MFMAs.  One
// workgroup barrier per K step as before: a loader arrives when its pieces of tile k have landed, a compute wave when it has
// the fragments of tile k - 1 in registers; behind it the loaders refill stage (k - 1) % NS with tile k + NS - 1.  Same
// fragments, same MFMA order per accumulator: bit-identical to the plain loop.  The loader waves then help to walk the staged
// epilogue (NTH threads).
// PIPE: the compute waves also keep TWO fragment register sets: behind the barrier of K step k they request the fragments of tile k
// and then multiply tile k - 1 out of the other set, so the LDS round trip of a K step hides behind the MFMAs of the previous one
// (what PM = 1 of the plain kernel could not deliver while the same waves also had to issue the DMA).  A stage is then free one
// barrier earlier, the ring holds all NS tiles at the start and the loaders keep NS - 1 in flight.
template <int TM, int TN, int WM, int WN, bool TA, bool TB, int NS, int NL, int PIPE = 0>
__device__ __forceinline__ void gemm_ldr_body(const CrctGemmArgs& g, const int tile_m, const int tile_n, const int dbg = 0) {
  constexpr int BM = 32 * TM, BN = 32 * TN, NW = WM * WN, NTH = (NW + NL) * 64;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int NPA = A_BYTES / 1024, NPB = B_BYTES / 1024;
  constexpr int PAL = NPA / NL, PBL = NPB / NL, L = PAL + PBL;            // pieces per loader wave per K tile
  constexpr int WTM = BM / WM / 16, WTN = BN / WN / 16;
  static_assert(PAL >= 1 && PBL >= 1 && PAL * NL == NPA && PBL * NL == NPB, "tile / loader-wave mismatch");
  static_assert(NS >= 2 && BK == 64, "ring depth");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;                              // compute waves: wm < WM
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nk = g.K / BK;

  f4_t acc[WTN][WTM];
#pragma unroll
  for (int a = 0; a < WTN; ++a)
#pragma unroll
    for (int b = 0; b < WTM; ++b) acc[a][b] = f4_t{0.f, 0.f, 0.f, 0.f};
  const bool do_rs = g.rowsum_out != nullptr && tile_n == 0;
  f4_t accb[WTM];
#pragma unroll
  for (int b = 0; b < WTM; ++b) accb[b] = f4_t{0.f, 0.f, 0.f, 0.f};

  if (wave >= NW) {
    // ------------------------------------------------------------ loader wave lw: pieces lw, lw + NL, ... of both images
    const int lw = wave - NW;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), 0, (int)OOB_OFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.B), 0, (int)OOB_OFF, 0x00020000);
    unsigned offA[PAL], offB[PBL];
#pragma unroll
    for (int i = 0; i < PAL; ++i) offA[i] = dma_src_offset<TA, TM>((i * NL + lw) * 64 + lane, m0, g.M, g.lda);
#pragma unroll
    for (int i = 0; i < PBL; ++i) offB[i] = dma_src_offset<TB, TN>((i * NL + lw) * 64 + lane, n0, g.N, g.ldb);
    const int stepA = TA ? (int)(64 * g.lda * 2) : 128, stepB = TB ? (int)(64 * g.ldb * 2) : 128;
    auto issue = [&](int kt, int st) {
      char* base = smem + st * STAGE + lw * 1024;
#pragma unroll
      for (int i = 0; i < PAL; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void_ptr)(base + i * NL * 1024), 16, (int)offA[i], kt * stepA, 0, 0);
#pragma unroll
      for (int i = 0; i < PBL; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void_ptr)(base + A_BYTES + i * NL * 1024), 16, (int)offB[i], kt * stepB, 0, 0);
    };
    if constexpr (PIPE == 1) {
      const int npre = nk < NS ? nk : NS;
      for (int t = 0; t < npre; ++t) issue(t, t);
      int st_next = 0;                      // the barrier of K step kt (>= 1) frees the stage of tile kt - 1
      for (int kt = 0; kt < nk; ++kt) {
        const int issued = kt == 0 ? NS - 1 : kt + NS - 2;
        const int ahead = (nk - 1 < issued ? nk - 1 : issued) - kt;
        if (NS >= 4 && ahead >= 3) wait_vmcnt<3 * L>();
        else if (NS >= 3 && ahead >= 2) wait_vmcnt<2 * L>();
        else if (ahead >= 1) wait_vmcnt<L>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt >= 1) {
          if (kt - 1 + NS < nk && !(lab_bits(dbg) & 2)) issue(kt - 1 + NS, st_next);
          st_next = st_next + 1 == NS ? 0 : st_next + 1;
        }
      }
    } else {
    const int npre = nk < NS - 1 ? nk : NS - 1;
    for (int t = 0; t < npre; ++t) issue(t, t);
    int st_next = NS - 1;
    for (int kt = 0; kt < nk; ++kt) {
      const int ahead = (nk - 1 < kt + NS - 2 ? nk - 1 : kt + NS - 2) - kt;      // younger tiles that may stay in flight
      if (NS >= 5 && ahead >= 3) wait_vmcnt<3 * L>();
      else if (NS >= 4 && ahead >= 2) wait_vmcnt<2 * L>();
      else if (NS >= 3 && ahead >= 1) wait_vmcnt<L>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + NS - 1 < nk && !(lab_bits(dbg) & 2)) issue(kt + NS - 1, st_next);
      st_next = st_next + 1 == NS ? 0 : st_next + 1;
    }
    }
  } else {
    // ------------------------------------------------------------ compute wave: no vector-memory instruction in the loop
    FragBase<TA, TM, WTM> fbA;
    FragBase<TB, TN, WTN> fbB;
    fbA.init(0, wm * (BM / WM), lane);
    fbB.init(A_BYTES, wn * (BN / WN), lane);
    const uint32_t smem_base = (uint32_t)(uintptr_t)smem;
    constexpr int N_HALF = WTM * (TA ? 2 : 1) + WTN * (TB ? 2 : 1);
    bf8_t ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
    if constexpr (PIPE == 2) {
      // half-step pipelining, no extra registers: the two 32-deep halves of a K tile live in the two halves of ONE fragment set.
      // While the MFMAs of half 0 run the reads of half 1 are in flight, and while those of half 1 run the reads of the NEXT tile's
      // half 0 are -- issued right behind the barrier that says the next tile has landed, which therefore sits between the two MFMA
      // groups of a K tile (a wave reaches it with 512+ cycles of MFMAs still in the pipe).  For the large wave tiles (128 x 64:
      // 32 MFMAs per half) whose fragment sets leave no room for a second copy (PIPE = 1 spills at 256 x 128).
      bf8_t fm[2][WTM], fn[2][WTN];
      auto reads = [&](int stg, auto hc) {
        constexpr int h = decltype(hc)::value;
        uint32_t ca[FragBase<TA, TM, WTM>::NB], cb[FragBase<TB, TN, WTN>::NB];
        fbA.at(smem_base + stg * STAGE, ca);
        fbB.at(smem_base + stg * STAGE, cb);
        FragBase<TA, TM, WTM>::template read<h>(ca, fm[h]);
        FragBase<TB, TN, WTN>::template read<h>(cb, fn[h]);
      };
      auto mul = [&](int kt, int h) {
#pragma unroll
        for (int i = 0; i < WTM; ++i) frag_async_use(fm[h][i]);
#pragma unroll
        for (int i = 0; i < WTN; ++i) frag_async_use(fn[h][i]);
#pragma unroll
        for (int a = 0; a < WTN; ++a)
#pragma unroll
          for (int b = 0; b < WTM; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fn[h][a], fm[h][b], acc[a][b], 0, 0, 0);
        if (do_rs && (((kt << 1) + h) & (WN - 1)) == wn) {
#pragma unroll
          for (int b = 0; b < WTM; ++b) accb[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fm[h][b], accb[b], 0, 0, 0);
        }
      };
      __builtin_amdgcn_s_barrier();                      // tile 0 has landed
      asm volatile("" ::: "memory");
      reads(0, std::integral_constant<int, 0>{});
      int stg = 0;
      for (int kt = 0; kt < nk; ++kt) {
        frag_async_wait<0>();                             // half 0 of tile kt is in registers (requested one MFMA group ago)
        reads(stg, std::integral_constant<int, 1>{});
        mul(kt, 0);
        frag_async_wait<0>();                             // ... and half 1: this wave is done with the stage
        const int nxt = stg + 1 == NS ? 0 : stg + 1;
        if (kt + 1 < nk) {
          __builtin_amdgcn_s_barrier();                   // tile kt + 1 has landed (the loaders waited for it)
          asm volatile("" ::: "memory");
          reads(nxt, std::integral_constant<int, 0>{});
        }
        mul(kt, 1);
        stg = nxt;
      }
    } else if constexpr (PIPE == 1) {
      bf8_t fmA[2][WTM], fnA[2][WTN], fmB[2][WTM], fnB[2][WTN];
      auto mul = [&](int kt, bf8_t (&fm)[2][WTM], bf8_t (&fn)[2][WTN]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int i = 0; i < WTM; ++i) frag_async_use(fm[h][i]);
#pragma unroll
          for (int i = 0; i < WTN; ++i) frag_async_use(fn[h][i]);
#pragma unroll
          for (int a = 0; a < WTN; ++a)
#pragma unroll
            for (int b = 0; b < WTM; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fn[h][a], fm[h][b], acc[a][b], 0, 0, 0);
        }
        if (do_rs) {
#pragma unroll
          for (int h = 0; h < 2; ++h)
            if ((((kt << 1) + h) & (WN - 1)) == wn) {
#pragma unroll
              for (int b = 0; b < WTM; ++b) accb[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fm[h][b], accb[b], 0, 0, 0);
            }
        }
      };
      auto step = [&](int kt, int stg, bf8_t (&fm)[2][WTM], bf8_t (&fn)[2][WTN], bf8_t (&fm_p)[2][WTM], bf8_t (&fn_p)[2][WTN]) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        uint32_t ca[FragBase<TA, TM, WTM>::NB], cb[FragBase<TB, TN, WTN>::NB];
        fbA.at(smem_base + stg * STAGE, ca);
        fbB.at(smem_base + stg * STAGE, cb);
        FragBase<TA, TM, WTM>::template read<0>(ca, fm[0]);
        FragBase<TB, TN, WTN>::template read<0>(cb, fn[0]);
        FragBase<TA, TM, WTM>::template read<1>(ca, fm[1]);
        FragBase<TB, TN, WTN>::template read<1>(cb, fn[1]);
        if (kt > 0) mul(kt - 1, fm_p, fn_p);               // the previous tile's MFMAs run while these reads are in flight
        frag_async_wait<0>();                              // ... and the stage is read out before this wave reaches the next barrier
      };
      int stg = 0;
      for (int kt = 0; kt < nk; kt += 2) {
        step(kt, stg, fmA, fnA, fmB, fnB);
        stg = stg + 1 == NS ? 0 : stg + 1;
        if (kt + 1 < nk) {
          step(kt + 1, stg, fmB, fnB, fmA, fnA);
          stg = stg + 1 == NS ? 0 : stg + 1;
        }
      }
      if (nk & 1) mul(nk - 1, fmA, fnA);
      else mul(nk - 1, fmB, fnB);
    } else {
    int st = 0;
    for (int kt = 0; kt < nk; ++kt) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (lab_bits(dbg) & 32) continue;        // lab ablation: the compute waves only keep the barrier count (pure fill time)
      bf8_t fm[2][WTM], fn[2][WTN];
      uint32_t ca[FragBase<TA, TM, WTM>::NB], cb[FragBase<TB, TN, WTN>::NB];
      fbA.at(smem_base + st * STAGE, ca);
      fbB.at(smem_base + st * STAGE, cb);
      FragBase<TA, TM, WTM>::template read<0>(ca, fm[0]);
      FragBase<TB, TN, WTN>::template read<0>(cb, fn[0]);
      FragBase<TA, TM, WTM>::template read<1>(ca, fm[1]);
      FragBase<TB, TN, WTN>::template read<1>(cb, fn[1]);
      frag_async_wait<(N_HALF <= 15 ? N_HALF : 0)>();
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (h == 1) {
          asm volatile("" : "+v"(acc[WTN - 1][WTM - 1]));     // keep the first half's MFMAs in front of the second wait
          frag_async_wait<0>();
        }
#pragma unroll
        for (int i = 0; i < WTM; ++i) frag_async_use(fm[h][i]);
#pragma unroll
        for (int i = 0; i < WTN; ++i) frag_async_use(fn[h][i]);
#pragma unroll
        for (int a = 0; a < WTN; ++a)
#pragma unroll
          for (int b = 0; b < WTM; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fn[h][a], fm[h][b], acc[a][b], 0, 0, 0);
      }
      if (do_rs) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
          if ((((kt << 1) + h) & (WN - 1)) == wn) {
#pragma unroll
            for (int b = 0; b < WTM; ++b) accb[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fm[h][b], accb[b], 0, 0, 0);
          }
      }
      st = st + 1 == NS ? 0 : st + 1;
    }
    }
  }
  if (do_rs) {
    static_assert((WN & (WN - 1)) == 0 && WN * BM * 4 <= NS * STAGE, "row-sum staging");
    float* rs = reinterpret_cast<float*>(smem);            // [WN][BM]
    __syncthreads();                                       // every compute wave is done reading the operand ring
    if (wave < NW && lane < 16) {
#pragma unroll
      for (int b = 0; b < WTM; ++b) rs[wn * BM + wm * (BM / WM) + b * 16 + lane] = accb[b][0];
    }
    __syncthreads();
    for (int i = tid; i < BM; i += NTH) {
      if (m0 + i < g.M) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < WN; ++w) v += rs[w * BM + i];  // fixed order: reproducible
        g.rowsum_out[m0 + i] += v;
      }
    }
  }
  gemm_epilogue_staged<BM, BN, WM, WN, WTM, WTN, NS * STAGE, NTH>(g, acc, smem, m0, n0, wm, wn, lane, tid);
}

template <int TM, int TN, int WM, int WN, bool TA, bool TB, int NS, int NL, int PIPE = 0>
__global__ __launch_bounds__((WM * WN + NL) * 64) void gemm_ldr_kernel(GEMM_HOT_PARAMS) {
  GEMM_HOT_UNPACK
  if constexpr (!TA) crct_chain_priority();
  int tile_m, tile_n;
  if (!map_tile(tmap, blockIdx.x, tile_m, tile_n)) return;
  gemm_ldr_body<TM, TN, WM, WN, TA, TB, NS, NL, PIPE>(g, tile_m, tile_n, tmap.dbg);
}

// ====================================================================================== fp8 forward (BASELINE configs[4])
// y = x W^T with OCP e4m3 operands (per-tensor delayed scaling), fp32 accumulation, the same fused epilogues.  A 128-deep
// fp8 K tile is a row of 128 BYTES, exactly like a 64-deep bf16 one, so the LDS images, the DMA pieces, the swizzle and the
// fragment addresses are those of the bf16 kernel: a lane's 16-byte fragment read now holds TWO 8-byte MFMA operands
// (v_mfma_f32_16x16x32_fp8_fp8 takes 8 e4m3 per lane).  The k index a lane's bytes stand for is permuted against the
// natural order of the instruction, identically for both operands, which a dot product does not see.  Half the operand
// bytes per K element travel HBM -> L2 -> LDS -> registers; the non-scaled fp8 MFMA itself runs at the bf16 rate.
typedef long l2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned f8_src_offset(int slot, int r0, int R, long ld) {
  const int r = slot >> 3, ch = (slot & 7) ^ (r & 7);
  return (r0 + r < R) ? (unsigned)(((long)(r0 + r)) * ld + ch * 16) : OOB_OFF;
}

// A_BF8: the A operand (a GRADIENT: the data-gradient GEMMs dx = dy W against the transposed e4m3 weight shadow) is OCP e5m2
// MX (round 4): the whole 128-deep K tile of a 16 x 16 output tile as ONE v_mfma_scale_f32_16x16x128_f8f6f4 with every block scale
// 2^0 (e8m0 127) -- gfx950's block-scaled instruction used as a plain fp8 MFMA at TWICE the rate of v_mfma_f32_16x16x32_fp8_fp8
// (32 cycles for K = 128 against 4 x 16; MI355X_MICROARCH.md, Matrix cores).  With 128-byte rows the plain instruction makes this
// kernel MFMA-bound (512 cycles per K tile and SIMD against a fill floor of 384).  A lane's operand is the 32 bytes it already reads
// per K tile -- the 16-byte chunks (lane >> 4) and 4 + (lane >> 4) of its row -- for both operands alike, so the products pair the
// same k as before; only the order of the fp32 additions inside a K tile differs.  Scaling stays per tensor (gemm_epilogue_staged):
// block scales would not buy fidelity here (EXPERIMENTS.md round 4, item 9).
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef int v4i_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v8i_t f8_pair(const bf8_t& lo, const bf8_t& hi) {
  const v4i_t a = __builtin_bit_cast(v4i_t, lo), b = __builtin_bit_cast(v4i_t, hi);
  return v8i_t{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
constexpr int MX_ONE = 0x7F7F7F7F;       // four e8m0 block scales of 2^0

template <int TM, int TN, int WM, int WN, int NS, bool A_BF8 = false, bool MX = false>
__global__ __launch_bounds__(WM * WN * 64) void gemm_f8_kernel(GEMM_HOT_PARAMS) {
  GEMM_HOT_UNPACK
  crct_chain_priority();
  int tile_m, tile_n;
  if (!map_tile(tmap, blockIdx.x, tile_m, tile_n)) return;
  constexpr int BM = 32 * TM, BN = 32 * TN, NW = WM * WN;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int PA = A_BYTES / 1024 / NW, PB = B_BYTES / 1024 / NW, L = PA + PB;
  constexpr int WTM = BM / WM / 16, WTN = BN / WN / 16;
  static_assert(PA >= 1 && PB >= 1 && PA * NW * 1024 == A_BYTES && PB * NW * 1024 == B_BYTES, "tile / wave-grid mismatch");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), 0, (int)OOB_OFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.B), 0, (int)OOB_OFF, 0x00020000);
  unsigned offA[PA], offB[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) offA[i] = f8_src_offset((i * NW + wave) * 64 + lane, m0, g.M, g.lda);
#pragma unroll
  for (int i = 0; i < PB; ++i) offB[i] = f8_src_offset((i * NW + wave) * 64 + lane, n0, g.N, g.ldb);

  f4_t acc[WTN][WTM];
#pragma unroll
  for (int a = 0; a < WTN; ++a)
#pragma unroll
    for (int b = 0; b < WTM; ++b) acc[a][b] = f4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / 128;
  auto issue = [&](int kt, int st) {
    char* base = smem + st * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < PA; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void_ptr)(base + i * NW * 1024), 16, (int)offA[i], kt * 128, 0, 0);
#pragma unroll
    for (int i = 0; i < PB; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void_ptr)(base + A_BYTES + i * NW * 1024), 16, (int)offB[i], kt * 128, 0, 0);
  };
  const int npre = nk < NS - 1 ? nk : NS - 1;
  for (int t = 0; t < npre; ++t) issue(t, t);

  using FA = FragBase<false, TM, WTM>;
  using FB = FragBase<false, TN, WTN>;
  FA fbA;
  FB fbB;
  fbA.init(0, wm * (BM / WM), lane);
  fbB.init(A_BYTES, wn * (BN / WN), lane);
  const uint32_t smem_base = (uint32_t)(uintptr_t)smem;
  constexpr int N_HALF = WTM + WTN;
  auto wait_tile = [&](int t) {        // tile t has landed; the younger tiles issued so far may stay in flight
    const int ahead = (nk - 1 < t + NS - 2 ? nk - 1 : t + NS - 2) - t;
    if (NS >= 4 && ahead >= 2) wait_vmcnt<2 * L>();
    else if (NS >= 3 && ahead >= 1) wait_vmcnt<L>();
    else wait_vmcnt<0>();
  };
  auto multiply = [&](int h, bf8_t (&fm)[2][WTM], bf8_t (&fn)[2][WTN]) {
#pragma unroll
    for (int i = 0; i < WTM; ++i) frag_async_use(fm[h][i]);
#pragma unroll
    for (int i = 0; i < WTN; ++i) frag_async_use(fn[h][i]);
#pragma unroll
    for (int a = 0; a < WTN; ++a) {
      const l2_t bn = __builtin_bit_cast(l2_t, fn[h][a]);
#pragma unroll
      for (int b = 0; b < WTM; ++b) {
        const l2_t am = __builtin_bit_cast(l2_t, fm[h][b]);
        if constexpr (A_BF8) {      // first source = the weight fragment (e4m3), second = the gradient fragment (e5m2)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(bn[0], am[0], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(bn[1], am[1], acc[a][b], 0, 0, 0);
        } else {
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(bn[0], am[0], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(bn[1], am[1], acc[a][b], 0, 0, 0);
        }
      }
    }
  };
  int st = 0, st_next = NS - 1;
  for (int kt = 0; kt < nk; ++kt) {
    wait_tile(kt);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + NS - 1 < nk) issue(kt + NS - 1, st_next);
    bf8_t fm[2][WTM], fn[2][WTN];
    uint32_t ca[FA::NB], cb[FB::NB];
    fbA.at(smem_base + st * STAGE, ca);
    fbB.at(smem_base + st * STAGE, cb);
    FA::template read<0>(ca, fm[0]);
    FB::template read<0>(cb, fn[0]);
    FA::template read<1>(ca, fm[1]);
    FB::template read<1>(cb, fn[1]);
    if constexpr (MX) {
      frag_async_wait<0>();
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < WTM; ++i) frag_async_use(fm[h][i]);
#pragma unroll
        for (int i = 0; i < WTN; ++i) frag_async_use(fn[h][i]);
      }
#pragma unroll
      for (int a = 0; a < WTN; ++a) {
        const v8i_t bn = f8_pair(fn[0][a], fn[1][a]);
#pragma unroll
        for (int b = 0; b < WTM; ++b)     // first source = the weight fragment (e4m3), second = the activation (e4m3) / gradient (e5m2) fragment
          acc[a][b] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bn, f8_pair(fm[0][b], fm[1][b]), acc[a][b], 0, A_BF8 ? 1 : 0, 0, MX_ONE, 0, MX_ONE);
      }
    } else {
    frag_async_wait<(N_HALF <= 15 ? N_HALF : 0)>();
    multiply(0, fm, fn);
    asm volatile("" : "+v"(acc[WTN - 1][WTM - 1]));     // keep the first half's MFMAs in front of the second wait
    frag_async_wait<0>();
    multiply(1, fm, fn);
    }
    st_next = st;
    st = st + 1 == NS ? 0 : st + 1;
  }
  gemm_epilogue_staged<BM, BN, WM, WN, WTM, WTN, NS * STAGE>(g, acc, smem, m0, n0, wm, wn, lane, tid);
}

static int g_f8_mx = 1;      // crct_gemm_fp8_scaled_mfma: the fp8 GEMMs on v_mfma_scale_f32_16x16x128_f8f6f4 (unit scales) instead of 16x16x32
}  // namespace
extern "C" int crct_gemm_fp8_scaled_mfma(int on) { const int old = g_f8_mx; if (on >= 0) g_f8_mx = on != 0; return old; }
namespace {

template <int TM, int TN, int WM, int WN, int NS>
hipError_t launch_f8(const CrctGemmArgs& g, hipStream_t s) {
  constexpr int BM = 32 * TM, BN = 32 * TN;
  int tiles = 0;
  const TileMap tmap = make_tile_map(g.M, g.N, BM, BN, &tiles);
  const size_t lds = (size_t)NS * (BM + BN) * 128;
  hipError_t e = hipSuccess;
#define CRCT_LAUNCH_F8(BF8_, MX_)                                                                                          \
  do {                                                                                                                     \
    auto kern = gemm_f8_kernel<TM, TN, WM, WN, NS, BF8_, MX_>;                                                             \
    static bool attr_set = false;                                                                                          \
    if (lds > 64 * 1024 && !attr_set) {                                                                                    \
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
      if (e != hipSuccess) return e;                                                                                       \
      attr_set = true;                                                                                                     \
    }                                                                                                                      \
    launch_kernel(kern, dim3(tiles), dim3(WM * WN * 64), lds, s, GEMM_HOT_ARGS(g, tmap) g, tmap);                                                 \
  } while (0)
  if (g_f8_mx) { if (g.fp8 & 2) CRCT_LAUNCH_F8(true, true); else CRCT_LAUNCH_F8(false, true); }
  else { if (g.fp8 & 2) CRCT_LAUNCH_F8(true, false); else CRCT_LAUNCH_F8(false, false); }
#undef CRCT_LAUNCH_F8
  return hipGetLastError();
}

// the fp8 kernel needs whole 128-deep K tiles and 16-byte aligned rows of both operands
inline bool f8_ok(const CrctGemmArgs& g) {
  if (g.ta || g.tb || g.K % 128 != 0 || g.K < 128 || g.lda % 16 != 0 || g.ldb % 16 != 0) return false;
  if (g.N % 8 != 0 || g.ldc % 8 != 0 || !g.scale_a || !g.scale_b || g.rowsum_out) return false;
  if ((g.preact_out || g.dact_src) && g.ld_aux % 8 != 0) return false;
  if (g.addend && g.ld_add % 8 != 0) return false;
  if (g.q_out && (g.ld_q % 8 != 0 || !g.q_scale)) return false;
  return (long)g.M * g.lda < 0x7f000000L && (long)g.N * g.ldb < 0x7f000000L;
}


// ====================================================================================== fp8 weight gradients (BASELINE configs[4])
// dW[out][in] = dy^T x with the operands AS THE FORWARD / DATA-GRADIENT PASSES LEFT THEM: dy [tokens][out] OCP e5m2 (the copy the
// LayerNorm-backward kernel / the GELU' epilogue wrote for the fp8 data gradient), x [tokens][in] e4m3 (the copy the forward GEMM
// read).  Both are stored with the CONTRACTION index (tokens) as the row index, so a 128-token K tile of an operand is an image of
// 128 rows x BM bytes, and an MFMA operand -- 8 consecutive tokens of one column per lane -- is one ds_read_b64_tr_b8: lane 2q + p
// of a 16-lane group addresses row q, bytes 8p .. 8p + 7 of an 8-row x 16-byte block, lane i receives column i (measured layout,
// tools/lab/tr8_probe.hip).  Image layout in 16-byte slots (row k, 16-byte column chunk ch, W64 = BM / 64):
//   slot = 32 * ((k >> 3) * W64 + (ch >> 2)) + 4 * (k & 7) + ((ch & 3) ^ ((k >> 2) & 1) ^ 2 * ((k >> 3) & 1))
// i.e. 512-byte blocks of 8 rows x 64 bytes; the XOR spreads the 16 rows a 32-lane half reads in one instruction over the 16
// slots of a 256-byte bank row.  Half the operand bytes of the bf16 kernel travel L2 -> LDS per token; the K tail (tokens % 128)
// is zero-filled by the buffer range check, rows beyond the tensor likewise.
template <int W64>
__device__ __forceinline__ int f8t_slot(int k, int ch) {
  return 32 * ((k >> 3) * W64 + (ch >> 2)) + 4 * (k & 7) + ((ch & 3) ^ ((k >> 2) & 1) ^ (2 * ((k >> 3) & 1)));
}
template <int W64>
__device__ __forceinline__ unsigned f8t_src_offset(int slot, int c0, int C, long ld, int k_lim) {
  const int blk = slot >> 5, sb = slot & 31, kb = blk / W64, cb = blk % W64, kq = sb >> 2;
  const int k = 8 * kb + kq, ch = 4 * cb + ((sb & 3) ^ ((kq >> 2) & 1) ^ (2 * (kb & 1)));
  return (k < k_lim && c0 + 16 * ch < C) ? (unsigned)((long)k * ld + c0 + 16 * ch) : OOB_OFF;
}
template <int OFF>
__device__ __forceinline__ long lds_read_tr8_imm(uint32_t a) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
  long v;
  asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF));
  return v;
}
__device__ __forceinline__ void frag_async_use(long& f) { asm volatile("" : "+v"(f)); }

template <int TM, int TN, int WM, int WN, int NS, bool MX = false>
__device__ __forceinline__ void gemm_f8t_body(const CrctGemmArgs& g, int tile_m, int tile_n) {
  constexpr int BM = 32 * TM, BN = 32 * TN, NW = WM * WN, KT = 128;
  constexpr int A_BYTES = KT * BM, B_BYTES = KT * BN, STAGE = A_BYTES + B_BYTES;
  constexpr int PA = A_BYTES / 1024 / NW, PB = B_BYTES / 1024 / NW, L = PA + PB;
  constexpr int WTM = BM / WM / 16, WTN = BN / WN / 16;
  constexpr int AW = BM / 64, BW = BN / 64;
  static_assert(BM % 64 == 0 && BN % 64 == 0, "64-byte image blocks");
  static_assert(PA >= 1 && PB >= 1 && PA * NW * 1024 == A_BYTES && PB * NW * 1024 == B_BYTES, "tile / wave-grid mismatch");
  static_assert(WTM <= 4 && WTN <= 4 && (BM / WM) % (16 * WTM) == 0, "a wave's columns stay inside one 64-byte block row");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), 0, (int)OOB_OFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.B), 0, (int)OOB_OFF, 0x00020000);
  const int nk = (g.K + KT - 1) / KT, k_tail = g.K - (nk - 1) * KT;       // tokens in the last K tile: 1 .. 128
  unsigned offA[PA], offB[PB], offAt[PA], offBt[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    offA[i] = f8t_src_offset<AW>((i * NW + wave) * 64 + lane, m0, g.M, g.lda, KT);
    offAt[i] = f8t_src_offset<AW>((i * NW + wave) * 64 + lane, m0, g.M, g.lda, k_tail);
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    offB[i] = f8t_src_offset<BW>((i * NW + wave) * 64 + lane, n0, g.N, g.ldb, KT);
    offBt[i] = f8t_src_offset<BW>((i * NW + wave) * 64 + lane, n0, g.N, g.ldb, k_tail);
  }
  f4_t acc[WTN][WTM];
#pragma unroll
  for (int a = 0; a < WTN; ++a)
#pragma unroll
    for (int b = 0; b < WTM; ++b) acc[a][b] = f4_t{0.f, 0.f, 0.f, 0.f};

  auto issue = [&](int kt, int st) {
    char* base = smem + st * STAGE + wave * 1024;
    const bool tail = kt == nk - 1;
    const int sa = kt * KT * (int)g.lda, sb = kt * KT * (int)g.ldb;
#pragma unroll
    for (int i = 0; i < PA; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void_ptr)(base + i * NW * 1024), 16, (int)(tail ? offAt[i] : offA[i]), sa, 0, 0);
#pragma unroll
    for (int i = 0; i < PB; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void_ptr)(base + A_BYTES + i * NW * 1024), 16, (int)(tail ? offBt[i] : offB[i]), sb, 0, 0);
  };
  const int npre = nk < NS - 1 ? nk : NS - 1;
  for (int t = 0; t < npre; ++t) issue(t, t);

  // per-lane fragment bases (K step h = 0; step h adds 2048 * W64 bytes: four 8-row blocks further down, same parities)
  const int fg = lane >> 4, fq = (lane & 15) >> 1, fp = lane & 1;
  uint32_t bA[WTM], bB[WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i) bA[i] = 16 * f8t_slot<AW>(8 * fg + fq, (wm * (BM / WM)) / 16 + i) + 8 * fp;
#pragma unroll
  for (int i = 0; i < WTN; ++i) bB[i] = A_BYTES + 16 * f8t_slot<BW>(8 * fg + fq, (wn * (BN / WN)) / 16 + i) + 8 * fp;
  const uint32_t smem_base = (uint32_t)(uintptr_t)smem;
  constexpr int NR = WTM + WTN;
  static_assert(2 * NR <= 15, "two K steps of fragment reads in flight");
  auto wait_tile = [&](int t) {
    const int ahead = (nk - 1 < t + NS - 2 ? nk - 1 : t + NS - 2) - t;
    if (NS >= 4 && ahead >= 2) wait_vmcnt<2 * L>();
    else if (NS >= 3 && ahead >= 1) wait_vmcnt<L>();
    else wait_vmcnt<0>();
  };
  int st = 0, st_next = NS - 1;
  for (int kt = 0; kt < nk; ++kt) {
    wait_tile(kt);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + NS - 1 < nk) issue(kt + NS - 1, st_next);
    const uint32_t stage = smem_base + st * STAGE;
    long fa[4][WTM], fb[4][WTN];
    if constexpr (MX) {       // the four 32-token steps of the K tile as ONE scaled MFMA per output tile (see gemm_f8_kernel)
      static_for<4>([&](auto hc) {
        constexpr int h = decltype(hc)::value;
        static_for<WTM>([&](auto ic) { constexpr int i = decltype(ic)::value; fa[h][i] = lds_read_tr8_imm<h * 2048 * AW>(stage + bA[i]); });
        static_for<WTN>([&](auto ic) { constexpr int i = decltype(ic)::value; fb[h][i] = lds_read_tr8_imm<h * 2048 * BW>(stage + bB[i]); });
      });
      frag_async_wait<0>();
#pragma unroll
      for (int h = 0; h < 4; ++h) {
#pragma unroll
        for (int i = 0; i < WTM; ++i) frag_async_use(fa[h][i]);
#pragma unroll
        for (int i = 0; i < WTN; ++i) frag_async_use(fb[h][i]);
      }
      auto quad = [](long a, long b, long c, long d) {
        return v8i_t{(int)a, (int)(a >> 32), (int)b, (int)(b >> 32), (int)c, (int)(c >> 32), (int)d, (int)(d >> 32)};
      };
#pragma unroll
      for (int a = 0; a < WTN; ++a) {
        const v8i_t xb = quad(fb[0][a], fb[1][a], fb[2][a], fb[3][a]);
#pragma unroll
        for (int b = 0; b < WTM; ++b)     // first source = the activation fragment (e4m3), second = the gradient fragment (e5m2)
          acc[a][b] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(xb, quad(fa[0][b], fa[1][b], fa[2][b], fa[3][b]), acc[a][b], 0, 1, 0, MX_ONE, 0, MX_ONE);
      }
      st_next = st;
      st = st + 1 == NS ? 0 : st + 1;
      continue;
    }
    static_for<4>([&](auto hc) {
      constexpr int h = decltype(hc)::value;
      static_for<WTM>([&](auto ic) { constexpr int i = decltype(ic)::value; fa[h][i] = lds_read_tr8_imm<h * 2048 * AW>(stage + bA[i]); });
      static_for<WTN>([&](auto ic) { constexpr int i = decltype(ic)::value; fb[h][i] = lds_read_tr8_imm<h * 2048 * BW>(stage + bB[i]); });
      if constexpr (h >= 1) {             // multiply step h - 1 while step h's reads are in flight
        frag_async_wait<NR>();
#pragma unroll
        for (int i = 0; i < WTM; ++i) frag_async_use(fa[h - 1][i]);
#pragma unroll
        for (int i = 0; i < WTN; ++i) frag_async_use(fb[h - 1][i]);
#pragma unroll
        for (int a = 0; a < WTN; ++a)
#pragma unroll
          for (int b = 0; b < WTM; ++b)     // first source = the activation fragment (e4m3), second = the gradient fragment (e5m2)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(fb[h - 1][a], fa[h - 1][b], acc[a][b], 0, 0, 0);
        asm volatile("" : "+v"(acc[WTN - 1][WTM - 1]));
      }
    });
    frag_async_wait<0>();
#pragma unroll
    for (int i = 0; i < WTM; ++i) frag_async_use(fa[3][i]);
#pragma unroll
    for (int i = 0; i < WTN; ++i) frag_async_use(fb[3][i]);
#pragma unroll
    for (int a = 0; a < WTN; ++a)
#pragma unroll
      for (int b = 0; b < WTM; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(fb[3][a], fa[3][b], acc[a][b], 0, 0, 0);
    st_next = st;
    st = st + 1 == NS ? 0 : st + 1;
  }
  gemm_epilogue_staged<BM, BN, WM, WN, WTM, WTN, NS * STAGE>(g, acc, smem, m0, n0, wm, wn, lane, tid);
}

template <int TM, int TN, int WM, int WN, int NS, bool MX = false>
__global__ __launch_bounds__(WM * WN * 64) void gemm_f8t_kernel(GEMM_HOT_PARAMS) {
  GEMM_HOT_UNPACK
  int tile_m, tile_n;
  if (!map_tile(tmap, blockIdx.x, tile_m, tile_n)) return;
  gemm_f8t_body<TM, TN, WM, WN, NS, MX>(g, tile_m, tile_n);
}

template <int TM, int TN, int WM, int WN, int NS>
hipError_t launch_f8t(const CrctGemmArgs& g, hipStream_t s) {
  constexpr int BM = 32 * TM, BN = 32 * TN;
  int tiles = 0;
  const TileMap tmap = make_tile_map(g.M, g.N, BM, BN, &tiles);
  const size_t lds = (size_t)NS * (BM + BN) * 128;
  hipError_t e = hipSuccess;
#define CRCT_LAUNCH_F8T(MX_)                                                                                               \
  do {                                                                                                                     \
    auto kern = gemm_f8t_kernel<TM, TN, WM, WN, NS, MX_>;                                                                  \
    static bool attr_set = false;                                                                                          \
    if (lds > 64 * 1024 && !attr_set) {                                                                                    \
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
      if (e != hipSuccess) return e;                                                                                       \
      attr_set = true;                                                                                                     \
    }                                                                                                                      \
    launch_kernel(kern, dim3(tiles), dim3(WM * WN * 64), lds, s, GEMM_HOT_ARGS(g, tmap) g, tmap);                                                 \
  } while (0)
  if (g_f8_mx) CRCT_LAUNCH_F8T(true);
  else CRCT_LAUNCH_F8T(false);
#undef CRCT_LAUNCH_F8T
  return hipGetLastError();
}

// the fp8 weight-gradient kernel: both operands token-major bytes with 16-byte aligned rows, fp32 or bf16 result, no fused row sums
inline bool f8t_ok(const CrctGemmArgs& g) {
  if (!g.ta || !g.tb || !(g.fp8 & 2) || g.K < 1 || g.lda % 16 != 0 || g.ldb % 16 != 0 || g.M % 16 != 0 || g.N % 16 != 0) return false;
  if (((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15) || g.ldc % 8 != 0 || !g.scale_a || !g.scale_b || g.rowsum_out || g.q_out) return false;
  if (g.preact_out || g.dact_src || g.addend || g.drop_thr || g.bias) return false;
  return (long)g.K * g.lda < 0x7f000000L && (long)g.K * g.ldb < 0x7f000000L;
}

// ---- grouped launch: up to 8 independent GEMMs of the same mode in ONE grid (the weight gradients of one
// layer: 4-6 small problems that individually leave most CUs idle).  Block -> (problem, tile) by prefix table.
constexpr int GROUP_MAX = 8;
struct GroupArgs {
  int n;
  int tile_begin[GROUP_MAX + 1];       // first block of every problem (multiples of 8: a problem starts on XCD 0)
  TileMap map[GROUP_MAX];              // per-problem XCD-aware block -> tile map, as for single launches
  // concat != 0 (the weight gradients of a layer): the tiles of ALL problems form one list -- problem after problem, inside a
  // problem along its shorter grid dimension first -- and XCD x (= block % 8) takes the x-th run of per_xcd consecutive tiles.  A
  // run then lies inside one or two problems and covers a compact rectangle of each: an XCD's private L2 fetches the operand
  // panels of ~1/8 of the group instead of 1/8 of EVERY problem (fabric reads of a text layer's group 143 -> ~50 MB).
  int concat, per_xcd;
  int tile_prefix[GROUP_MAX + 1];      // tiles before every problem (concat mode)
  CrctGemmArgs p[GROUP_MAX];           // the problems with their full epilogues (ta / tb are the kernel's template arguments)
};
// block -> (problem, tile) of a grouped launch; false: a padding block
__device__ __forceinline__ bool group_pick(const GroupArgs& ga, int bid, int& pi, int& tm, int& tn) {
  pi = 0;
  if (ga.concat) {
    const int j = bid >> 3, t = (bid & 7) * ga.per_xcd + j;
    if (j >= ga.per_xcd || t >= ga.tile_prefix[ga.n]) return false;
#pragma unroll
    for (int i = 1; i < GROUP_MAX; ++i)
      if (i < ga.n && t >= ga.tile_prefix[i]) pi = i;
    const int local = t - ga.tile_prefix[pi], nm = ga.map[pi].tiles_m, nn = ga.map[pi].tiles_n;
    if (nn <= nm) { tm = local / nn; tn = local - tm * nn; }
    else { tn = local / nm; tm = local - tn * nm; }
    return true;
  }
#pragma unroll
  for (int i = 1; i < GROUP_MAX; ++i)
    if (i < ga.n && bid >= ga.tile_begin[i]) pi = i;
  return map_tile(ga.map[pi], bid - ga.tile_begin[pi], tm, tn);
}
inline void group_concat(GroupArgs& ga, int* grid) {      // host side of concat mode
  int tiles = 0;
  for (int i = 0; i < ga.n; ++i) { ga.tile_prefix[i] = tiles; tiles += ga.map[i].tiles_m * ga.map[i].tiles_n; }
  ga.tile_prefix[ga.n] = tiles;
  ga.concat = 1;
  ga.per_xcd = (tiles + 7) / 8;
  *grid = 8 * ga.per_xcd;
}

// A grid smaller than the tile count (launch_group's max_wgs) makes the workgroups persistent: workgroup b takes tiles b, b + grid,
// ... -- the weight gradients then occupy at most `grid` CUs' LDS at a time and the data-gradient chain that runs beside them
// finds free CUs at once (grid a multiple of 8: a workgroup's tiles stay on its XCD's rectangle).
template <int TM, int TN, int WM, int WN, bool TA, bool TB, int NS, int PM = 0>
__global__ __launch_bounds__(WM * WN * 64) void gemm_group_kernel(const GroupArgs ga) {
  const int total = ga.concat ? 8 * ga.per_xcd : ga.tile_begin[ga.n];
  for (int bid = blockIdx.x; bid < total; bid += gridDim.x) {
    int pi, tm, tn;
    if (group_pick(ga, bid, pi, tm, tn)) {
      CrctGemmArgs g = ga.p[pi];
      g.ta = TA; g.tb = TB;
      gemm_pipe_body<TM, TN, WM, WN, TA, TB, NS, false, PM>(g, tm, tn, lab_bits(ga.map[pi].dbg));
    }
    if (bid + (int)gridDim.x < total) __syncthreads();      // the next tile's first DMA reuses the ring the epilogue has just read
  }
}

// the same with the loader-wave body (configurations 48 / 53 / 58 / 59: 128 x 128 tiles)
template <int TM, int TN, int WM, int WN, bool TA, bool TB, int NS, int NL, int PIPE>
__global__ __launch_bounds__((WM * WN + NL) * 64) void gemm_group_ldr_kernel(const GroupArgs ga) {
  const int total = ga.concat ? 8 * ga.per_xcd : ga.tile_begin[ga.n];
  for (int bid = blockIdx.x; bid < total; bid += gridDim.x) {
    int pi, tm, tn;
    if (group_pick(ga, bid, pi, tm, tn)) {
      CrctGemmArgs g = ga.p[pi];
      g.ta = TA; g.tb = TB;
      gemm_ldr_body<TM, TN, WM, WN, TA, TB, NS, NL, PIPE>(g, tm, tn);
    }
    if (bid + (int)gridDim.x < total) __syncthreads();
  }
}

static int g_group_max_wgs = 0;        // crct_gemm_group_max_workgroups: 0 = one workgroup per tile
static int g_group_wgrad_cfg = 4;      // configuration of a layer's grouped weight gradients: 4 = 128 x 128 plain loop (48 / 53 / 58 / 59 / 68: loader-wave variants)
static int g_group_concat = 0;         // crct_gemm_group_concat: grouped weight gradients as ONE tile list over the XCDs (GroupArgs.concat); measured: no gain
extern "C" int crct_gemm_group_concat(int on) { g_group_concat = on != 0; return 0; }
// configuration of the grouped weight-gradient launches (4 / 48 / 53 / 58 / 59 / 68: all 128 x 128 tiles); returns the previous one
extern "C" int crct_gemm_group_wgrad_config(int cfg) {
  const int old = g_group_wgrad_cfg;
  if (cfg == 4 || cfg == 48 || cfg == 53 || cfg == 58 || cfg == 59 || cfg == 68) g_group_wgrad_cfg = cfg;
  return old;
}
extern "C" int crct_gemm_group_max_workgroups(int n) { g_group_max_wgs = n > 0 ? (n + 7) / 8 * 8 : 0; return 0; }
// Workgroups of a grouped bf16 weight-gradient launch (round 4).  The groups run on a side stream BESIDE the data-gradient chain, and
// with one workgroup per tile (288 for a text layer's FFN pair) they take every CU the chain's 156-workgroup GEMMs leave and a share
// of the ones they use.  A persistent grid of about `target` workgroups that walks the tile list in whole rounds -- rounds =
// ceil(tiles / target), grid = ceil(tiles / rounds) rounded up to the 8 XCDs -- leaves the chain ~150 CUs: 7.45 -> 7.30 - 7.39 ms at
// configs[1] with 96 workgroups for 288 tiles (3 rounds; 88 = 4 rounds: 7.48, 104 / 112 = 3 rounds on more CUs: 7.39 / 7.43, 128:
// 7.55; profiles/r4_wgrad_workgroups_ab.txt).  0 = off.  An explicit crct_gemm_group_max_workgroups overrides it.  The step engine
// decides per launch (engine.cpp, Run::flush_wgrads): only where a throttled side stream cannot become the critical path.
static int g_group_target_wgs = 0;       // the library's default for direct callers; the step engine passes its own (crct_gemm_launch_grouped_wgs)
static int g_group_target_now = 0;       // the target of the launch in progress
extern "C" int crct_gemm_group_target_workgroups(int n) { const int old = g_group_target_wgs; g_group_target_wgs = n > 0 ? n : 0; return old; }
static int group_grid(int total, bool wgrad_bf16) {
  if (g_group_max_wgs > 0) return g_group_max_wgs < total ? g_group_max_wgs : total;
  if (!wgrad_bf16 || g_group_target_now <= 0 || total <= g_group_target_now) return total;
  const int rounds = (total + g_group_target_now - 1) / g_group_target_now;
  const int grid = ((total + rounds - 1) / rounds + 7) / 8 * 8;
  return grid < total ? grid : total;
}

template <int TM, int TN, int WM, int WN, int NS, int PM = 0>
hipError_t launch_group(const CrctGemmArgs* gs, int n, hipStream_t s) {
  constexpr int BM = 32 * TM, BN = 32 * TN;
  GroupArgs ga = {};
  ga.n = n;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    const CrctGemmArgs& g = gs[i];
    ga.tile_begin[i] = total;
    int grid = 0;
    ga.map[i] = make_tile_map(g.M, g.N, BM, BN, &grid);
    total += grid;
    ga.p[i] = g;
  }
  ga.tile_begin[n] = total;
#ifdef CRCT_GEMM_LAB   // ablation bits for the grouped launches alone (CRCT_GEMM_DBG covers every GEMM)
  { static const int gd = getenv("CRCT_GEMM_DBG_GROUP") ? atoi(getenv("CRCT_GEMM_DBG_GROUP")) : -1; if (gd >= 0) for (int i = 0; i < n; ++i) ga.map[i].dbg = gd; }
#endif
  if (gs[0].ta && g_group_concat) group_concat(ga, &total);      // the weight gradients of a layer
  const size_t lds = (size_t)NS * (BM + BN) * BK * 2;
  hipError_t e = hipSuccess;
#define CRCT_LAUNCH_GROUP(TA_, TB_)                                                                                        \
  do {                                                                                                                     \
    auto kern = gemm_group_kernel<TM, TN, WM, WN, TA_, TB_, NS, PM>;                                                       \
    static bool attr_set = false;                                                                                          \
    if (lds > 64 * 1024 && !attr_set) {                                                                                    \
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
      if (e != hipSuccess) return e;                                                                                       \
      attr_set = true;                                                                                                     \
    }                                                                                                                      \
    launch_kernel(kern, dim3(group_grid(total, gs[0].ta && gs[0].tb && !gs[0].fp8)), dim3(WM * WN * 64), lds, s, ga); \
  } while (0)
  if (gs[0].ta && gs[0].tb) CRCT_LAUNCH_GROUP(true, true);
  else if (!gs[0].ta && gs[0].tb) CRCT_LAUNCH_GROUP(false, true);
  else if (!gs[0].ta && !gs[0].tb) CRCT_LAUNCH_GROUP(false, false);
  else return hipErrorInvalidValue;
#undef CRCT_LAUNCH_GROUP
  return hipGetLastError();
}

template <int TM, int TN, int WM, int WN, int NS, int NL, int PIPE>
hipError_t launch_group_ldr(const CrctGemmArgs* gs, int n, hipStream_t s) {
  constexpr int BM = 32 * TM, BN = 32 * TN;
  GroupArgs ga = {};
  ga.n = n;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    ga.tile_begin[i] = total;
    int grid = 0;
    ga.map[i] = make_tile_map(gs[i].M, gs[i].N, BM, BN, &grid);
    ga.map[i].dbg = 0;
    total += grid;
    ga.p[i] = gs[i];
  }
  ga.tile_begin[n] = total;
  if (gs[0].ta && g_group_concat) group_concat(ga, &total);
  const size_t lds = (size_t)NS * (BM + BN) * BK * 2;
  hipError_t e = hipSuccess;
#define CRCT_LAUNCH_GROUP_LDR(TA_, TB_)                                                                                    \
  do {                                                                                                                     \
    auto kern = gemm_group_ldr_kernel<TM, TN, WM, WN, TA_, TB_, NS, NL, PIPE>;                                             \
    static bool attr_set = false;                                                                                          \
    if (lds > 64 * 1024 && !attr_set) {                                                                                    \
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
      if (e != hipSuccess) return e;                                                                                       \
      attr_set = true;                                                                                                     \
    }                                                                                                                      \
    launch_kernel(kern, dim3(group_grid(total, gs[0].ta && gs[0].tb && !gs[0].fp8)), dim3((WM * WN + NL) * 64), lds, s, ga); \
  } while (0)
  if (gs[0].ta && gs[0].tb) CRCT_LAUNCH_GROUP_LDR(true, true);
  else if (!gs[0].ta && gs[0].tb) CRCT_LAUNCH_GROUP_LDR(false, true);
  else if (!gs[0].ta && !gs[0].tb) CRCT_LAUNCH_GROUP_LDR(false, false);
  else return hipErrorInvalidValue;
#undef CRCT_LAUNCH_GROUP_LDR
  return hipGetLastError();
}

// the fp8 weight gradients of a layer in one grid (same block -> (problem, tile) table, gemm_f8t_body per tile)
template <int TM, int TN, int WM, int WN, int NS, bool MX = false>
__global__ __launch_bounds__(WM * WN * 64) void gemm_f8t_group_kernel(const GroupArgs ga) {
  const int total = ga.concat ? 8 * ga.per_xcd : ga.tile_begin[ga.n];
  for (int bid = blockIdx.x; bid < total; bid += gridDim.x) {
    int pi, tm, tn;
    if (group_pick(ga, bid, pi, tm, tn)) gemm_f8t_body<TM, TN, WM, WN, NS, MX>(ga.p[pi], tm, tn);
    if (bid + (int)gridDim.x < total) __syncthreads();
  }
}
template <int TM, int TN, int WM, int WN, int NS>
hipError_t launch_group_f8t(const CrctGemmArgs* gs, int n, hipStream_t s) {
  constexpr int BM = 32 * TM, BN = 32 * TN;
  GroupArgs ga = {};
  ga.n = n;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    ga.tile_begin[i] = total;
    int grid = 0;
    ga.map[i] = make_tile_map(gs[i].M, gs[i].N, BM, BN, &grid);
    ga.map[i].dbg = 0;
    total += grid;
    ga.p[i] = gs[i];
  }
  ga.tile_begin[n] = total;
  if (g_group_concat) group_concat(ga, &total);
  const size_t lds = (size_t)NS * (BM + BN) * 128;
  hipError_t e = hipSuccess;
#define CRCT_LAUNCH_F8TG(MX_)                                                                                              \
  do {                                                                                                                     \
    auto kern = gemm_f8t_group_kernel<TM, TN, WM, WN, NS, MX_>;                                                            \
    static bool attr_set = false;                                                                                          \
    if (lds > 64 * 1024 && !attr_set) {                                                                                    \
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
      if (e != hipSuccess) return e;                                                                                       \
      attr_set = true;                                                                                                     \
    }                                                                                                                      \
    launch_kernel(kern, dim3(group_grid(total, false)), dim3(WM * WN * 64), lds, s, ga); \
  } while (0)
  if (g_f8_mx) CRCT_LAUNCH_F8TG(true);
  else CRCT_LAUNCH_F8TG(false);
#undef CRCT_LAUNCH_F8TG
  return hipGetLastError();
}

// split-K launcher: only the configurations the step uses it with are instantiated (forward and data gradient)
template <int TM, int TN, int WM, int WN, int NS>
hipError_t launch_splitk(const CrctGemmArgs& g, hipStream_t s) {
  constexpr int BM = 32 * TM, BN = 32 * TN;
  int tiles = 0;
  const TileMap tmap = make_tile_map(g.M, g.N, BM, BN, &tiles);
  const size_t lds = (size_t)NS * (BM + BN) * BK * 2;
  hipError_t e = hipSuccess;
#define CRCT_LAUNCH_SK(TB_)                                                                                                \
  do {                                                                                                                     \
    auto kern = gemm_splitk_kernel<TM, TN, WM, WN, false, TB_, NS>;                                                        \
    static bool attr_set = false;                                                                                          \
    if (lds > 64 * 1024 && !attr_set) {                                                                                    \
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
      if (e != hipSuccess) return e;                                                                                       \
      attr_set = true;                                                                                                     \
    }                                                                                                                      \
    launch_kernel(kern, dim3(tiles * g.split_k), dim3(WM * WN * 64), lds, s, GEMM_HOT_ARGS(g, tmap) g, tmap);                                     \
  } while (0)
  if (g.ta) return hipErrorInvalidValue;
  if (g.tb) CRCT_LAUNCH_SK(true);
  else CRCT_LAUNCH_SK(false);
#undef CRCT_LAUNCH_SK
  return hipGetLastError();
}

template <int TM, int TN, int WM, int WN, int NS, int PM = 0>
hipError_t launch_pipe(const CrctGemmArgs& g, hipStream_t s) {
  constexpr int BM = 32 * TM, BN = 32 * TN;
  int tiles = 0;
  const TileMap tmap = make_tile_map(g.M, g.N, BM, BN, &tiles);
  const size_t lds = (size_t)NS * (BM + BN) * BK * 2;
  hipError_t e = hipSuccess;
#define CRCT_LAUNCH_PIPE(TA_, TB_)                                                                                         \
  do {                                                                                                                     \
    auto kern = gemm_pipe_kernel<TM, TN, WM, WN, TA_, TB_, NS, PM>;                                                        \
    static bool attr_set = false;                                                                                          \
    if (lds > 64 * 1024 && !attr_set) {                                                                                    \
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
      if (e != hipSuccess) return e;                                                                                       \
      attr_set = true;                                                                                                     \
    }                                                                                                                      \
    launch_kernel(kern, dim3(tiles), dim3(WM * WN * 64), lds, s, GEMM_HOT_ARGS(g, tmap) g, tmap);                      \
  } while (0)
  if (!g.ta && !g.tb) CRCT_LAUNCH_PIPE(false, false);
  else if (!g.ta && g.tb) CRCT_LAUNCH_PIPE(false, true);
  else if (g.ta && g.tb) CRCT_LAUNCH_PIPE(true, true);
  else return hipErrorInvalidValue;
#undef CRCT_LAUNCH_PIPE
  return hipGetLastError();
}

template <int TM, int TN, int WM, int WN, int NS, int NL, int PIPE = 0>
hipError_t launch_ldr(const CrctGemmArgs& g, hipStream_t s) {
  constexpr int BM = 32 * TM, BN = 32 * TN;
  int tiles = 0;
  const TileMap tmap = make_tile_map(g.M, g.N, BM, BN, &tiles);
  const size_t lds = (size_t)NS * (BM + BN) * BK * 2;
  hipError_t e = hipSuccess;
#define CRCT_LAUNCH_LDR(TA_, TB_)                                                                                          \
  do {                                                                                                                     \
    auto kern = gemm_ldr_kernel<TM, TN, WM, WN, TA_, TB_, NS, NL, PIPE>;                                                   \
    static bool attr_set = false;                                                                                          \
    if (lds > 64 * 1024 && !attr_set) {                                                                                    \
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
      if (e != hipSuccess) return e;                                                                                       \
      attr_set = true;                                                                                                     \
    }                                                                                                                      \
    launch_kernel(kern, dim3(tiles), dim3((WM * WN + NL) * 64), lds, s, GEMM_HOT_ARGS(g, tmap) g, tmap);                                          \
  } while (0)
  if (!g.ta && !g.tb) CRCT_LAUNCH_LDR(false, false);
  else if (!g.ta && g.tb) CRCT_LAUNCH_LDR(false, true);
  else if (g.ta && g.tb) CRCT_LAUNCH_LDR(true, true);
  else return hipErrorInvalidValue;
#undef CRCT_LAUNCH_LDR
  return hipGetLastError();
}

// the DMA path needs whole K tiles, 32-bit source offsets, and 16-byte aligned rows
inline bool pipe_ok(const CrctGemmArgs& g) {
  if (g.K % BK != 0 || g.K < BK) return false;
  if (g.N % 8 != 0 || g.ldc % 8 != 0) return false;                       // 16-byte epilogue accesses
  if ((g.preact_out || g.dact_src) && g.ld_aux % 8 != 0) return false;
  if (g.addend && g.ld_add % 8 != 0) return false;
  const long spanA = g.ta ? (long)g.K * g.lda : (long)g.M * g.lda;
  const long spanB = g.tb ? (long)g.K * g.ldb : (long)g.N * g.ldb;
  return spanA * 2 < 0x7f000000L && spanB * 2 < 0x7f000000L;
}

template <int TM, int TN>
hipError_t launch_cfg(const CrctGemmArgs& g, hipStream_t s) {
  constexpr int BM = 32 * TM, BN = 32 * TN;
  int tiles = 0;
  const TileMap tmap = make_tile_map(g.M, g.N, BM, BN, &tiles);
  const size_t lds = (size_t)(BM + BN) * BK * 2;
  if (!g.ta && !g.tb) launch_kernel(gemm_kernel<TM, TN, false, false>, dim3(tiles), dim3(256), lds, s, GEMM_HOT_ARGS(g, tmap) g, tmap);
  else if (!g.ta && g.tb) launch_kernel(gemm_kernel<TM, TN, false, true>, dim3(tiles), dim3(256), lds, s, GEMM_HOT_ARGS(g, tmap) g, tmap);
  else if (g.ta && g.tb) launch_kernel(gemm_kernel<TM, TN, true, true>, dim3(tiles), dim3(256), lds, s, GEMM_HOT_ARGS(g, tmap) g, tmap);
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

// Configuration of the LDS-DMA kernel per call, from tools/gemm_lab measurements on MI355X at the
// CRCT shapes (profiles/gemm_lab_r1.txt): every GEMM here is a 1-2 wave problem whose time is set by
// L2->LDS traffic and prefetch latency, so 8-wave workgroups with 2 resident per CU win.
//   12: 128x64, 8 waves (4x2), 2 stages    15: the same with 3 stages (narrow output, long K)    3: 64x64, 4 waves, 4 stages
//   grouped weight gradients: 4 = 128x128, 8 waves (2x4), 3 stages
// Shape classes of the step's forward / data-gradient GEMMs and the configuration each one runs with (ids of the switch in
// crct_gemm_launch).  Per-site overrides for A/B runs of the whole step go through crct_engine_set_site_policy (CrctGemmArgs.tile):
// no environment variable changes what a shipped kernel launch does.
//   tw   text rows (M <= 2000), wide output (N >= 2304)          FFN-up / QKV forward, FFN-down data gradient
//   tn   text rows, narrow output (N <= 1024), K <= 1024          attention-output / dense2 forward and data gradient
//   tnl  text rows, narrow output, long K (> 1024)                FFN-down forward, FFN-up / QKV data gradient
//   vw   visual rows (M > 2000), wide output                      visual / co-attention QKV forward
//   vm   visual rows, N <= 1024, K <= 1024                        visual projections and FFN, both directions
//   vml  visual rows, N <= 1024, long K                           QKV data gradient, image-embedding forward (128 x 128 tiles: id 4)
// Round 2 tried a tile-flexible variant of the LDS-DMA kernel (160 x 128, 96 x 64, 192 x 192 ... tiles that cover M = 1600 /
// 2880 in ONE round of workgroups, 4 or 8 waves, 3-4 stages, in-place inline-asm MFMAs, half-tile register double buffering;
// commit 11e1801).  Stand-alone with cold weights it won 10-17 % on several shapes (profiles/r2_gemm_lab_cold.txt), in the
// step EVERY class lost 0.1-0.4 ms (profiles/r2_gemm_flex_step_ab.txt): its 108-160 KB of LDS allow one workgroup per CU,
// so the kernels of the other internal streams can no longer share the CUs.  The 48-72 KB configurations below stay.
// Round 4: the classes are keyed by the ROW COUNT of the output in three buckets -- S (<= 2000 rows: the text stream of configs[1]),
// M (<= 4000: the visual stream of configs[1], the text stream of the long-context configs[3]), L (> 4000: its visual stream) --
// times the three column / contraction classes above (w: N >= 2304, n: N <= 1024 and K <= 1024, nl: N <= 1024 and K > 1024).
// Round 3 classified by M <= 2000 alone, so the long-context text GEMMs (2560 rows) ran with the visual stream's choices and
// its 6400-row visual GEMMs with tiles picked for 2880 rows.  The table is filled from in-step sweeps (bench.py --class-policy,
// profiles/r4_longctx_class_sweep.txt); crct_gemm_class_config overrides an entry for such a sweep.
enum { CLS_W, CLS_N, CLS_NL, CLS_PER_BUCKET };
enum { MB_S, MB_M, MB_L, MB_COUNT };
constexpr int CLS_COUNT = MB_COUNT * CLS_PER_BUCKET;
// 128 x 64 tiles everywhere they were, now with 3 stages (15) instead of 2 (12): re-swept in the step AFTER the weight gradients
// became a persistent grid of ~96 workgroups (the data-gradient chain then has ~160 CUs to itself and a deeper ring pays): 7.43 - 7.45 ->
// 7.31 - 7.33 ms at configs[1], 11.74 -> 11.68 long context, forced exchange 8.04 -> 7.87 (profiles/r4_class_resweep.txt)
static int g_class_table[CLS_COUNT] = {15, 15, 15,      // S: text rows of configs[1] (and, below, every narrow GEMM of text width)
                                       15, 15, 4,       // M: visual rows of configs[1] (nl: 128x128, 3 stages: 7.61 -> 7.56 ms)
                                       9, 50, 50};      // L: long-context visual rows: 256x128 tiles with loader waves for the narrow outputs
                                                        //    (in-step sweep, profiles/r4_longctx_class_sweep.txt: 12.11 -> 12.04 each, 11.95 -> 11.79 ms
                                                        //    together with the text-width rule below; every other entry measured neutral or worse).
                                                        //    Round 6: wide outputs 15 -> 9 (128x128, 8 waves, 2 stages) for the 9 920 text rows of the
                                                        //    reference's PlotQA shape (B 80 x 124 tokens: FFN-up / QKV forward, FFN-down data gradient):
                                                        //    19.12 -> 18.69 ms per step there, 11.57 -> 11.57 at configs[3] (profiles/r6_plotqa_class_sweep.txt)
extern "C" int crct_gemm_class_config(int cls, int cfg) {
  if (cls < 0 || cls >= CLS_COUNT) return -1;
  const int old = g_class_table[cls];
  if (cfg >= 0 && cfg <= 71) g_class_table[cls] = cfg;
  return old;
}
static int pick_pipe_config(const CrctGemmArgs& g) {
  if (g.M <= 96) return 3;                                          // head / regressor GEMMs: B rows
  if (g.ta) return ((long)g.M * g.N <= 1024L * 1024L) ? 3 : 9;      // single weight gradient (grouped ones: crct_gemm_launch_grouped)
  int mb = g.M <= 2000 ? MB_S : (g.M <= 4000 ? MB_M : MB_L);
  const bool wide = g.N >= 2304, longk = g.K > 1024;
  // a narrow output of TEXT width (N < 1024: H = 768) in the M bucket is the long-context text stream (2560 rows), not the visual
  // stream of configs[1] (2880 rows x 1024): it takes the text entries (the visual nl choice, 128 x 128 tiles, leaves 120 tiles)
  if (mb == MB_M && !wide && g.N < 1024) mb = MB_S;
  const int t = g_class_table[mb * CLS_PER_BUCKET + (wide ? CLS_W : (longk ? CLS_NL : CLS_N))];
  return (t < 0 || t > 71) ? ((g.N <= 1024 && g.K >= 2048) ? 15 : 12) : t;
}

// ---- optional live profiling: begin / end stamps of every GEMM kernel, on the launch stream ----------
// (bench.py: roofline.achieved = algorithmic FLOPs per launch / average launch duration, per variant and per model site)
namespace {
struct ProfSlot { hipEvent_t a, b; int variant; int nsite; short site[GROUP_MAX]; double fl[GROUP_MAX]; };   // site[] = site * 3 + kind
struct Prof {
  bool on = false;
  std::vector<ProfSlot> slots;
  size_t used = 0;
  static constexpr int NV = 225;       // (72 LDS-DMA / fp8 / register-staged configuration ids + spare) x {fwd, dgrad, wgrad}
  double flops[NV] = {0}; long count[NV] = {0};
  bool log_on = false;
  std::vector<CrctLaunchRec> log;
} g_prof;

// (an fp8 data gradient reads a TRANSPOSED weight shadow, so it has the forward's operand layout: CrctGemmArgs.fp8 bit 3 says what it is)
inline int kind_of(const CrctGemmArgs& g) { return g.ta ? CRCT_KIND_WGRAD : ((g.tb || (g.fp8 & 8)) ? CRCT_KIND_DGRAD : CRCT_KIND_FWD); }
inline int site_of(const CrctGemmArgs& g) { return (g.site > 0 && g.site < CRCT_SITE_COUNT) ? g.site : 0; }

ProfSlot* prof_begin(int variant, const CrctGemmArgs* gs, int n) {
#ifdef CRCT_NO_PROF_HOOKS      // A/B build (tools/ab_lib.sh): the hooks compiled out, to show what they cost a launch when they are off
  return nullptr;
#endif
  if (!g_prof.on) return nullptr;
  if (g_prof.used == g_prof.slots.size()) {
    ProfSlot ns;
    if (hipEventCreate(&ns.a) != hipSuccess || hipEventCreate(&ns.b) != hipSuccess) return nullptr;
    g_prof.slots.push_back(ns);
  }
  ProfSlot* slot = &g_prof.slots[g_prof.used++];
  slot->variant = variant;
  slot->nsite = n;
  g_prof.count[variant] += 1;
  for (int i = 0; i < n; ++i) {
    const double fl = 2.0 * gs[i].M * gs[i].N * gs[i].K;
    slot->site[i] = (short)(site_of(gs[i]) * 3 + kind_of(gs[i]));
    slot->fl[i] = fl;
    g_prof.flops[variant] += fl;
  }
  g_time_start = slot->a; g_time_stop = slot->b;
  return slot;
}
void log_launch(const CrctGemmArgs* gs, int n, int cfg, int grid) {
#ifdef CRCT_NO_PROF_HOOKS
  return;
#endif
  if (!g_prof.log_on) return;
  CrctLaunchRec r;
  r.site = n == 1 ? site_of(gs[0]) : -1; r.kind = kind_of(gs[0]); r.M = gs[0].M; r.N = gs[0].N; r.K = gs[0].K;
  r.cfg = cfg; r.split_k = gs[0].split_k > 1 ? gs[0].split_k : 1; r.grid = grid; r.n_problems = n; r.flops = 0;
  for (int i = 0; i < n; ++i) r.flops += 2.0 * gs[i].M * gs[i].N * gs[i].K;
  g_prof.log.push_back(r);
}
}  // namespace

static bool g_force_generic = false;
// test hook: route every GEMM through the register-staged kernel (parity of both code paths)
extern "C" int crct_gemm_force_generic(int on) { g_force_generic = on != 0; return 0; }

extern "C" int crct_prof_enable(int on) {      // 1: the GEMM kernels (per configuration / per site); 2: also every other kernel of the library (crct_prof_stamp_*)
  g_prof.on = on != 0;
  crct_stamp_enable(on == 2);
  return 0;
}
extern "C" int crct_prof_reset(void) {
  crct_stamp_reset();
  g_prof.used = 0;
  for (int i = 0; i < Prof::NV; ++i) { g_prof.flops[i] = 0; g_prof.count[i] = 0; }
  return 0;
}
static int slot_ms(const ProfSlot& sl, float* ms) {
  if (hipEventSynchronize(sl.b) != hipSuccess) return 1;
  return hipEventElapsedTime(ms, sl.a, sl.b) != hipSuccess;
}
// variant = config * 3 + {0: fwd (NT), 1: dgrad (tb), 2: wgrad (ta, tb)}; config 0..15 = LDS-DMA kernel, 16..19 = register-staged
// kernel tiles 0..3, 20 / 21 = fp8.  Synchronises the events.
extern "C" int crct_prof_read(int variant, long* count, double* flops, double* ms) {
  if (variant < 0 || variant >= Prof::NV) return 1;
  double t = 0;
  for (size_t i = 0; i < g_prof.used; ++i) {
    if (g_prof.slots[i].variant != variant) continue;
    float e = 0;
    if (slot_ms(g_prof.slots[i], &e)) return 1;
    t += e;
  }
  *count = g_prof.count[variant]; *flops = g_prof.flops[variant]; *ms = t;
  return 0;
}
extern "C" int crct_prof_read_site(int site, int kind, long* count, double* flops, double* ms, int* apportioned) {
  if (site < 0 || site >= CRCT_SITE_COUNT || kind < 0 || kind > 2) return 1;
  const short key = (short)(site * 3 + kind);
  long n = 0; double fl = 0, t = 0; int app = 0;
  for (size_t i = 0; i < g_prof.used; ++i) {
    const ProfSlot& sl = g_prof.slots[i];
    double mine = 0, all = 0;
    int hits = 0;
    for (int j = 0; j < sl.nsite; ++j) { all += sl.fl[j]; if (sl.site[j] == key) { mine += sl.fl[j]; ++hits; } }
    if (!hits) continue;
    float e = 0;
    if (slot_ms(sl, &e)) return 1;
    n += hits; fl += mine;
    t += all > 0 ? (double)e * mine / all : 0.0;
    if (sl.nsite > 1) app = 1;
  }
  *count = n; *flops = fl; *ms = t;
  if (apportioned) *apportioned = app;
  return 0;
}
extern "C" int crct_launch_log_enable(int on) {
  g_prof.log_on = on != 0;
  if (on) g_prof.log.clear();
  return 0;
}
extern "C" int crct_launch_log_count(void) { return (int)g_prof.log.size(); }
extern "C" int crct_launch_log_read(int i, CrctLaunchRec* out) {
  if (i < 0 || i >= (int)g_prof.log.size() || !out) return 1;
  *out = g_prof.log[i];
  return 0;
}

// split-K needs the LDS-DMA kernel; the slab space is sized for the larger of the tiles it is built with (128 x 128)
extern "C" int64_t crct_gemm_splitk_ws_elems(int M, int N, int split_k) {
  if (split_k <= 1) return 0;
  const int64_t tm = (M + 127) / 128, tn = (N + 127) / 128;
  return tm * tn * 128 * 128 * (int64_t)split_k;
}
extern "C" int crct_gemm_splitk_tickets(int M, int N) { return ((M + 127) / 128) * ((N + 63) / 64); }
static bool splitk_ok(const CrctGemmArgs& g) {
  return g.split_k > 1 && g.split_k <= 8 && !g.ta && !(g.fp8 & 1) && !g.rowsum_out && g.splitk_ws && g.splitk_cnt && g.K / BK >= 2 * g.split_k;
}

// tile (rows x columns) of a kernel configuration id of the LDS-DMA kernels (the switch in crct_gemm_launch)
static bool cfg_tile(int t, int& bm, int& bn) {
  switch (t) {
    case 0: case 4: case 8: case 9: case 32: case 33: case 48: case 49: case 53: case 58: case 59: case 62: case 68: case 70: bm = 128; bn = 128; return true;
    case 1: case 10: case 12: case 13: case 14: case 15: case 46: case 47: case 51: case 52: case 54: case 56: case 57: case 61: case 63: case 64: case 69:
      bm = 128; bn = 64; return true;
    case 2: case 11: bm = 64; bn = 128; return true;
    case 3: bm = 64; bn = 64; return true;
    case 5: bm = 128; bn = 256; return true;
    case 6: case 7: case 34: case 50: case 55: case 60: case 65: case 66: case 67: case 71: bm = 256; bn = 128; return true;
    case 22: case 23: case 30: case 31: bm = 160; bn = 128; return true;
    case 24: case 25: case 35: bm = 160; bn = 96; return true;
    case 26: case 27: bm = 96; bn = 64; return true;
    case 28: bm = 160; bn = 64; return true;
    case 29: bm = 64; bn = 96; return true;
  }
  return false;
}
// the configuration id crct_gemm_launch resolves for a bf16 (non-fp8) problem, and whether it runs on the LDS-DMA kernels
static int resolve_config(const CrctGemmArgs& g, bool& pipe) {
  pipe = pipe_ok(g) && !g_force_generic;
  int t = g.tile >= 0 ? g.tile : (pipe ? pick_pipe_config(g) : crct_gemm_pick_tile(g.M, g.N));
  if (t > 15 && !(((t >= 22 && t <= 35) || (t >= 46 && t <= 71)) && pipe)) t = 12;
  if (t > 3 && !pipe) t = crct_gemm_pick_tile(g.M, g.N);
  if (!pipe && t == 0) t = 1;
  return t;
}

hipError_t crct_gemm_launch(const CrctGemmArgs& g_in, hipStream_t s) {
  if (g_in.M <= 0 || g_in.N <= 0) return hipSuccess;
  // the leading dimensions of the operands travel to the kernel as preloaded 32-bit scalars (GEMM_HOT_ARGS)
  if (g_in.lda < 0 || g_in.ldb < 0 || g_in.lda > 0x7fffffffLL || g_in.ldb > 0x7fffffffLL) return hipErrorInvalidValue;
  CrctGemmArgs g = g_in;
  const bool is_f8 = (g.fp8 & 1) != 0;                         // bit 0: fp8 operands; bits 1 / 2 qualify the A operand / the q_out copy
  if (is_f8 && g.ta) {                                         // fp8 weight gradient (token-major operands): ids 36 (3 stages) / 37 (2 stages)
    if (!f8t_ok(g)) return hipErrorInvalidValue;
    const int t8 = g.tile == 37 ? 37 : 36;
    prof_begin(t8 * 3 + kind_of(g), &g, 1);
    const hipError_t e8 = t8 == 37 ? launch_f8t<4, 4, 2, 4, 2>(g, s) : launch_f8t<4, 4, 2, 4, 3>(g, s);
    g_time_start = g_time_stop = nullptr;
    log_launch(&g, 1, t8, 0);
    return e8;
  }
  if (is_f8 && !f8_ok(g)) return hipErrorInvalidValue;
  const bool pipe = is_f8 || (pipe_ok(g) && !g_force_generic);
  if (g.q_out && !(pipe && g.q_scale && g.ld_q % 8 == 0)) return hipErrorInvalidValue;      // the fp8 output copy lives in the staged epilogue
  if (g.rowsum_out && !pipe) return hipErrorNotSupported;       // row sums exist in the LDS-DMA kernel only
  int t = g.tile >= 0 ? g.tile : (pipe ? pick_pipe_config(g) : crct_gemm_pick_tile(g.M, g.N));
  if (t > 15 && !(((t >= 22 && t <= 35) || (t >= 46 && t <= 71)) && pipe && !is_f8)) t = 12;
  // fp8 forward: the tile of the bf16 kernel, 2 stages (id 20) or 3 for the narrow long-K GEMMs (id 21)
  if (is_f8) t = (g.tile == 20 || g.tile == 21) ? g.tile : ((g.N <= 1024 && g.K >= 2048) ? 21 : 20);
  if (t > 3 && !pipe) t = crct_gemm_pick_tile(g.M, g.N);
  if (!pipe && t == 0) t = 1;      // the register-staged 128x128 instantiation is 4x slower than 128x64 (measured)
  // K-partitioned launch: the four configurations it is built for; anything else runs unsplit (same function, other summation order)
  if (g.split_k > 1 && !(pipe && splitk_ok(g) && (t == 4 || t == 9 || t == 12 || t == 15))) g.split_k = 0;
  if (g.split_k <= 1) g.split_k = 0;
  prof_begin((pipe ? t : 16 + (t & 3)) * 3 + kind_of(g), &g, 1);
  hipError_t e;
  int grid = 0;
  if (is_f8) {
    e = t == 21 ? launch_f8<4, 2, 4, 2, 3>(g, s) : launch_f8<4, 2, 4, 2, 2>(g, s);
  } else if (pipe && g.split_k) {
    switch (t) {
      case 4: e = launch_splitk<4, 4, 2, 4, 3>(g, s); break;
      case 9: e = launch_splitk<4, 4, 2, 4, 2>(g, s); break;
      case 15: e = launch_splitk<4, 2, 4, 2, 3>(g, s); break;
      default: e = launch_splitk<4, 2, 4, 2, 2>(g, s); break;
    }
  } else if (pipe) {
    switch (t) {
      case 0: e = launch_pipe<4, 4, 2, 2, 3>(g, s); break;
      case 1: e = launch_pipe<4, 2, 2, 2, 4>(g, s); break;
      case 2: e = launch_pipe<2, 4, 2, 2, 4>(g, s); break;
      case 3: e = launch_pipe<2, 2, 2, 2, 4>(g, s); break;
      case 4: e = launch_pipe<4, 4, 2, 4, 3>(g, s); break;     // 128x128, 8 waves
      case 5: e = launch_pipe<4, 8, 2, 4, 3>(g, s); break;     // 128x256, 8 waves
      case 6: e = launch_pipe<8, 4, 4, 2, 3>(g, s); break;     // 256x128, 8 waves
      case 7: e = launch_pipe<8, 4, 4, 2, 2>(g, s); break;     // 256x128, 8 waves, 2 stages
      case 8: e = launch_pipe<4, 4, 2, 4, 4>(g, s); break;     // 128x128, 8 waves, 4 stages
      case 9: e = launch_pipe<4, 4, 2, 4, 2>(g, s); break;     // 128x128, 8 waves, 2 stages (2 blocks / CU)
      case 10: e = launch_pipe<4, 2, 2, 4, 3>(g, s); break;    // 128x64, 8 waves
      case 11: e = launch_pipe<2, 4, 2, 4, 3>(g, s); break;    // 64x128, 8 waves
      case 13: e = launch_pipe<4, 2, 4, 2, 4>(g, s); break;    // 128x64, 8 waves (4x2), 4 stages: long K, one block per CU
      case 14: e = launch_pipe<4, 2, 4, 2, 6>(g, s); break;    // 128x64, 8 waves (4x2), 6 stages
      case 15: e = launch_pipe<4, 2, 4, 2, 3>(g, s); break;    // 128x64, 8 waves (4x2), 3 stages
      // 160-row tiles: M = 1600 (text) and 2880 (visual) are exact multiples, 4 waves with 80 x (BN / 2) wave tiles
      case 22: e = launch_pipe<5, 4, 2, 2, 2>(g, s); break;    // 160x128, 4 waves, 2 stages (72 KB: two per CU)
      case 23: e = launch_pipe<5, 4, 2, 2, 3>(g, s); break;    // 160x128, 4 waves, 3 stages
      case 24: e = launch_pipe<5, 3, 2, 2, 3>(g, s); break;    // 160x96, 4 waves, 3 stages
      case 25: e = launch_pipe<5, 3, 2, 2, 2>(g, s); break;    // 160x96, 4 waves, 2 stages
      case 26: e = launch_pipe<3, 2, 2, 2, 3>(g, s); break;    // 96x64, 4 waves, 3 stages
      case 27: e = launch_pipe<3, 2, 2, 2, 4>(g, s); break;    // 96x64, 4 waves, 4 stages
      case 28: e = launch_pipe<5, 2, 2, 2, 3>(g, s); break;    // 160x64, 4 waves, 3 stages
      case 29: e = launch_pipe<2, 3, 2, 2, 4>(g, s); break;    // 64x96, 4 waves, 4 stages
      // register-pipelined main loop (PM = 1), one workgroup per CU
      case 30: e = launch_pipe<5, 4, 2, 2, 3, 1>(g, s); break; // 160x128, 4 waves, 3 stages
      case 31: e = launch_pipe<5, 4, 2, 2, 2, 1>(g, s); break; // 160x128, 4 waves, 2 stages
      case 32: e = launch_pipe<4, 4, 2, 2, 3, 1>(g, s); break; // 128x128, 4 waves, 3 stages
      case 33: e = launch_pipe<4, 4, 2, 4, 3, 1>(g, s); break; // 128x128, 8 waves, 3 stages
      case 34: e = launch_pipe<8, 4, 4, 2, 3, 1>(g, s); break; // 256x128, 8 waves, 3 stages
      case 35: e = launch_pipe<5, 3, 2, 2, 3, 1>(g, s); break; // 160x96, 4 waves, 3 stages
      // loader waves (round 4): WM x WN compute waves + NL waves that own the LDS-DMA ring
      case 46: e = launch_ldr<4, 2, 4, 2, 3, 4>(g, s); break;  // 128x64, 8 + 4 waves, 3 stages (72 KB)
      case 47: e = launch_ldr<4, 2, 4, 2, 2, 4>(g, s); break;  // 128x64, 8 + 4 waves, 2 stages (48 KB)
      case 48: e = launch_ldr<4, 4, 2, 4, 3, 4>(g, s); break;  // 128x128, 8 (2x4) + 4 waves, 3 stages (96 KB)
      case 49: e = launch_ldr<4, 4, 2, 4, 2, 4>(g, s); break;  // 128x128, 8 + 4 waves, 2 stages (64 KB)
      case 50: e = launch_ldr<8, 4, 4, 2, 3, 4>(g, s); break;  // 256x128, 8 (4x2: 64x64 wave tiles) + 4 waves, 3 stages (144 KB)
      case 51: e = launch_ldr<4, 2, 4, 2, 4, 4>(g, s); break;  // 128x64, 8 + 4 waves, 4 stages (96 KB)
      case 52: e = launch_ldr<4, 2, 4, 2, 3, 2>(g, s); break;  // 128x64, 8 + 2 waves, 3 stages
      case 53: e = launch_ldr<4, 4, 2, 2, 3, 4>(g, s); break;  // 128x128, 4 (2x2: 64x64 wave tiles) + 4 waves, 3 stages
      case 54: e = launch_ldr<4, 2, 2, 2, 3, 2>(g, s); break;  // 128x64, 4 (2x2: 64x32 wave tiles) + 2 waves, 3 stages
      case 55: e = launch_ldr<8, 4, 4, 2, 2, 4>(g, s); break;  // 256x128, 8 + 4 waves, 2 stages (96 KB)
      // loader waves + two fragment register sets in the compute waves (PIPE)
      case 56: e = launch_ldr<4, 2, 2, 2, 3, 2, true>(g, s); break;  // 128x64, 4 (64x32 wave tiles) + 2 waves, 3 stages (72 KB)
      case 57: e = launch_ldr<4, 2, 4, 2, 3, 4, true>(g, s); break;  // 128x64, 8 + 4 waves, 3 stages
      case 58: e = launch_ldr<4, 4, 2, 2, 3, 4, true>(g, s); break;  // 128x128, 4 (64x64 wave tiles) + 4 waves, 3 stages (96 KB)
      case 59: e = launch_ldr<4, 4, 2, 4, 3, 4, true>(g, s); break;  // 128x128, 8 (64x32) + 4 waves, 3 stages
      case 61: e = launch_ldr<4, 2, 2, 2, 4, 2, true>(g, s); break;  // 128x64, 4 + 2 waves, 4 stages (96 KB)
      case 62: e = launch_ldr<4, 4, 2, 2, 2, 4, true>(g, s); break;  // 128x128, 4 + 4 waves, 2 stages (64 KB: two per CU)
      case 63: e = launch_ldr<4, 2, 2, 2, 3, 4, true>(g, s); break;  // 128x64, 4 + 4 waves, 3 stages
      case 64: e = launch_ldr<4, 2, 2, 2, 2, 2, true>(g, s); break;  // 128x64, 4 + 2 waves, 2 stages (48 KB: three per CU)
      case 60: e = launch_ldr<8, 4, 2, 2, 3, 4>(g, s); break;        // 256x128, 4 (128x64 wave tiles) + 4 waves, 3 stages (144 KB)
      // loader waves + half-step pipelining in the compute waves (PIPE = 2: no extra registers)
      case 66: e = launch_ldr<8, 4, 2, 2, 3, 4, 2>(g, s); break;     // 256x128, 4 + 4 waves, 3 stages (144 KB)
      case 67: e = launch_ldr<8, 4, 4, 2, 3, 4, 2>(g, s); break;     // 256x128, 8 + 4 waves, 3 stages
      case 68: e = launch_ldr<4, 4, 2, 2, 3, 4, 2>(g, s); break;     // 128x128, 4 + 4 waves, 3 stages (96 KB)
      case 69: e = launch_ldr<4, 2, 2, 2, 3, 2, 2>(g, s); break;     // 128x64, 4 + 2 waves, 3 stages (72 KB)
      case 70: e = launch_ldr<4, 4, 2, 4, 3, 4, 2>(g, s); break;     // 128x128, 8 + 4 waves, 3 stages
      case 71: e = launch_ldr<8, 4, 2, 2, 2, 4, 2>(g, s); break;     // 256x128, 4 + 4 waves, 2 stages (96 KB)
      case 65: e = launch_ldr<8, 4, 2, 2, 2, 4>(g, s); break;        // 256x128, 4 + 4 waves, 2 stages (96 KB)
      default: e = launch_pipe<4, 2, 4, 2, 2>(g, s); break;    // 128x64, 8 waves (4x2), 2 stages
    }
  } else {
    switch (t) {
      case 0: e = launch_cfg<4, 4>(g, s); break;
      case 1: e = launch_cfg<4, 2>(g, s); break;
      case 2: e = launch_cfg<2, 4>(g, s); break;
      default: e = launch_cfg<2, 2>(g, s); break;
    }
  }
  g_time_start = g_time_stop = nullptr;
  log_launch(&g, 1, pipe ? t : 16 + (t & 3), grid);
  return e;
}

// Grouped launch of n <= 8 bf16 GEMMs with their full epilogues that share (ta, tb) and satisfy the LDS-DMA kernel's
// requirements; anything else is launched one by one.  Two uses: the weight gradients of a layer (ta = tb = 1), and the
// forward / data-gradient GEMMs of the text and the visual side of a co-attention layer or of two independent layers
// (n = 2): one grid, one ramp, the tiles of both problems packed over the CUs.
hipError_t crct_gemm_launch_grouped(const CrctGemmArgs* gs, int n, hipStream_t s) { return crct_gemm_launch_grouped_wgs(gs, n, s, g_group_target_wgs); }
hipError_t crct_gemm_launch_grouped_wgs(const CrctGemmArgs* gs, int n, hipStream_t s, int target_wgs) {
  g_group_target_now = target_wgs;
  bool all_f8t = n >= 2 && n <= GROUP_MAX;
  for (int i = 0; all_f8t && i < n; ++i) all_f8t = (gs[i].fp8 & 1) && gs[i].ta && f8t_ok(gs[i]);
  if (all_f8t) {                    // the fp8 weight gradients of a layer
    const int t8 = gs[0].tile == 37 ? 37 : 36;
    prof_begin(t8 * 3 + CRCT_KIND_WGRAD, gs, n);
    const hipError_t e8 = t8 == 37 ? launch_group_f8t<4, 4, 2, 4, 2>(gs, n, s) : launch_group_f8t<4, 4, 2, 4, 3>(gs, n, s);
    g_time_start = g_time_stop = nullptr;
    log_launch(gs, n, t8, 0);
    return e8;
  }
  bool ok = n >= 2 && n <= GROUP_MAX && !g_force_generic;
  for (int i = 0; ok && i < n; ++i) {
    const CrctGemmArgs& g = gs[i];
    ok = pipe_ok(g) && g.ta == gs[0].ta && g.tb == gs[0].tb && !(g.fp8 & 1) && !g.q_out && g.M > 96 && (g.ta || !g.rowsum_out);
  }
  if (!ok) {
    for (int i = 0; i < n; ++i) {
      hipError_t e = crct_gemm_launch(gs[i], s);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }
  // weight gradients: 128x128, 8 waves, 3 stages; forward / dgrad pairs: 128x128, 8 waves, 2 stages; CrctGemmArgs.tile of the first
  // problem may pick the other one (crct_engine_set_site_policy: A/B runs)
  const int t0 = gs[0].tile;
  const int cfg = (t0 == 4 || t0 == 9 || t0 == 48 || t0 == 53 || t0 == 58 || t0 == 59 || t0 == 68) ? t0 : (gs[0].ta ? g_group_wgrad_cfg : 9);
  prof_begin(cfg * 3 + kind_of(gs[0]), gs, n);
  hipError_t e;
  switch (cfg) {
    case 9: e = launch_group<4, 4, 2, 4, 2>(gs, n, s); break;
    case 48: e = launch_group_ldr<4, 4, 2, 4, 3, 4, false>(gs, n, s); break;
    case 53: e = launch_group_ldr<4, 4, 2, 2, 3, 4, false>(gs, n, s); break;
    case 58: e = launch_group_ldr<4, 4, 2, 2, 3, 4, true>(gs, n, s); break;
    case 59: e = launch_group_ldr<4, 4, 2, 4, 3, 4, true>(gs, n, s); break;
    case 68: e = launch_group_ldr<4, 4, 2, 2, 3, 4, 2>(gs, n, s); break;
    default: e = launch_group<4, 4, 2, 4, 3>(gs, n, s); break;
  }
  g_time_start = g_time_stop = nullptr;
  log_launch(gs, n, cfg, 0);
  return e;
}
