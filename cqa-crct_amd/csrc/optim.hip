// Fused multi-tensor AdamW over the flat fp32 parameter / gradient / moment buffers.
// Semantics: torch.optim.AdamW as the reference constructs it (CRCT/utils.py:228-249: one group per
// tensor, lr by language_weights.json, weight_decay 0 for bias / LayerNorm, betas (0.9, 0.999),
// eps 1e-8) -- decoupled decay p *= 1 - lr*wd, then the bias-corrected Adam update -- plus the
// refresh of the bf16 weight shadow the GEMMs read.  HBM-bound: 16 B read + 12 B (+2 B) written per
// parameter; each workgroup owns up to 4096 contiguous elements of ONE tensor (no divergence on
// lr / wd), float4 accesses.
#include <stdlib.h>

#include "common.hip.h"
#include "crct_internal.h"

// non-temporal streaming of the optimizer state: measured 8.73 -> 8.60 ms per step (the next forward keeps its operands in
// L2 / Infinity Cache while 7 GB of state stream past)
#ifndef CRCT_ADAMW_NT
#define CRCT_ADAMW_NT 1
#endif
namespace {
constexpr int ADAMW_CHUNK = 4096;
#ifdef CRCT_GEMM_LAB
__global__ __launch_bounds__(1024) void lab_spin_kernel(long long ticks) {      // ticks of 10 ns; no clock reads, no memory instruction at all
  const long long n = ticks * 24 / (127 * 64);                                  // s_sleep 127 = 127 x 64 cycles at ~2.4 GHz
  for (long long i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
}
#endif

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ pb,
                                                    const int64_t* __restrict__ seg_off, const int64_t* __restrict__ seg_len,
                                                    const float* __restrict__ seg_lr, const float* __restrict__ seg_wd,
                                                    const int32_t* __restrict__ blk_seg, const int64_t* __restrict__ blk_off,
                                                    float beta1, float beta2, float eps, float inv_bc1, float inv_sqrt_bc2,
                                                    const float* __restrict__ inv_scale_dev, const CrctAmpState amp, const CrctFp8Shadow f8,
                                                    int n_blk, int zero_g, const bf16_t* __restrict__ g16) {
 // loss scaling (torch.amp.GradScaler): a step whose gradients held an inf / nan is skipped as a whole, the scale divides
 // the gradients, and the step count behind the bias corrections is the device counter that only advances on real steps
 if (amp.found_inf && amp.found_inf[0] != 0.f) return;
 if (amp.step) {
   const float st = (float)amp.step[0];
   inv_bc1 = 1.0f / (1.0f - powf(beta1, st));
   inv_sqrt_bc2 = 1.0f / sqrtf(1.0f - powf(beta2, st));
 }
 __shared__ uint32_t tile[64][17];          // e4m3 bytes of one 64 x 64 weight tile (transposed shadow), 68-byte rows
 for (int blk = blockIdx.x; blk < n_blk; blk += gridDim.x) {
  const int sgi = blk_seg[blk];
  const int64_t off = blk_off[blk];
  int64_t base = seg_off[sgi] + off;
  int64_t n = seg_len[sgi] - off;
  if (n > ADAMW_CHUNK) n = ADAMW_CHUNK;
  // A weight [out][in] that also keeps a TRANSPOSED e4m3 shadow (fp8 data gradients) is walked tile by tile instead of chunk by
  // chunk: chunk number c of the tensor is the 64 x 64 tile (c / (in / 64), c % (in / 64)), so that the workgroup holds whole
  // columns of the tile and can write them as 16-byte runs of the transposed copy.  Everything else in the update is elementwise.
  const int tin = (f8.qt && f8.seg_in && f8.seg_t_base && f8.seg_t_ld) ? f8.seg_in[sgi] : 0;
  int64_t row_stride = 64, t_base = 0;
  int t_out = 0;
  if (tin > 0) {
    const int tiles_in = tin >> 6;
    const int64_t c = off / ADAMW_CHUNK;
    const int tr = (int)(c / tiles_in), tc = (int)(c % tiles_in);
    t_out = f8.seg_t_ld[sgi];                      // rows of the whole (possibly fused) weight = row length of its transposed copy
    base = seg_off[sgi] + (int64_t)tr * 64 * tin + (int64_t)tc * 64;
    row_stride = tin;
    t_base = f8.seg_t_base[sgi] + (int64_t)tc * 64 * t_out + (int64_t)tr * 64;
  }
  const float lr = seg_lr[sgi], wd = seg_wd[sgi];
  const float decay = 1.0f - lr * wd, step_size = lr * inv_bc1;
  float gsc = inv_scale_dev ? inv_scale_dev[0] : 1.0f;
  if (amp.grad_scale) gsc /= amp.grad_scale[0];
  // e4m3 shadow of the weights the fp8 forward GEMMs read (BASELINE configs[4]): written with the tensor's CURRENT scale
  // (from the amax one step back: delayed scaling, saturating), while max |w| of the new values feeds the next scale
  const int qslot = f8.q ? f8.seg_slot[sgi] : -1;
  const float qs = qslot >= 0 ? f8.scale[qslot] : 0.f;
  float qmax = 0.f;
  for (int64_t i = (int64_t)threadIdx.x * 4; i < n; i += 1024) {
    const int64_t e = base + (i >> 6) * row_stride + (i & 63);      // row_stride == 64: the plain chunk, e = base + i
    if (i + 4 <= n) {
#if CRCT_ADAMW_NT      // streamed once per step: non-temporal, so the 7 GB do not evict the forward's operands from L2 / MALL
      const f4_t pv = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(p + e));
      f4_t gv;
      if (g16) {      // the exchanged bf16 gradient as it lies in the communication buffer (8 bytes per 4 elements)
        typedef unsigned u2_t __attribute__((ext_vector_type(2)));
        const u2_t raw = __builtin_nontemporal_load(reinterpret_cast<const u2_t*>(g16 + e));
        gv = f4_t{bf2f((bf16_t)(raw[0] & 0xffff)), bf2f((bf16_t)(raw[0] >> 16)), bf2f((bf16_t)(raw[1] & 0xffff)), bf2f((bf16_t)(raw[1] >> 16))};
      } else {
        gv = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(g + e));
      }
      const f4_t mv = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(m + e));
      const f4_t vv = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(v + e));
#else
      const f4_t pv = *reinterpret_cast<const f4_t*>(p + e);
      f4_t gv = g16 ? f4_t{bf2f(g16[e]), bf2f(g16[e + 1]), bf2f(g16[e + 2]), bf2f(g16[e + 3])} : *reinterpret_cast<const f4_t*>(g + e);
      const f4_t mv = *reinterpret_cast<const f4_t*>(m + e), vv = *reinterpret_cast<const f4_t*>(v + e);
#endif
      float pa[4] = {pv[0], pv[1], pv[2], pv[3]}, ga[4] = {gv[0] * gsc, gv[1] * gsc, gv[2] * gsc, gv[3] * gsc};
      float ma[4] = {mv[0], mv[1], mv[2], mv[3]}, va[4] = {vv[0], vv[1], vv[2], vv[3]};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        pa[k] *= decay;
        ma[k] = beta1 * ma[k] + (1.0f - beta1) * ga[k];
        va[k] = beta2 * va[k] + (1.0f - beta2) * ga[k] * ga[k];
        pa[k] -= step_size * ma[k] / (sqrtf(va[k]) * inv_sqrt_bc2 + eps);
      }
#if CRCT_ADAMW_NT
      __builtin_nontemporal_store(f4_t{pa[0], pa[1], pa[2], pa[3]}, reinterpret_cast<f4_t*>(p + e));
      __builtin_nontemporal_store(f4_t{ma[0], ma[1], ma[2], ma[3]}, reinterpret_cast<f4_t*>(m + e));
      __builtin_nontemporal_store(f4_t{va[0], va[1], va[2], va[3]}, reinterpret_cast<f4_t*>(v + e));
#else
      *reinterpret_cast<float4*>(p + e) = make_float4(pa[0], pa[1], pa[2], pa[3]);
      *reinterpret_cast<float4*>(m + e) = make_float4(ma[0], ma[1], ma[2], ma[3]);
      *reinterpret_cast<float4*>(v + e) = make_float4(va[0], va[1], va[2], va[3]);
#endif
      if (pb) *reinterpret_cast<uint2*>(pb + e) = make_uint2(pack2bf(pa[0], pa[1]), pack2bf(pa[2], pa[3]));
      if (qslot >= 0) {
        qmax = fmaxf(fmaxf(qmax, fmaxf(fabsf(pa[0]), fabsf(pa[1]))), fmaxf(fabsf(pa[2]), fabsf(pa[3])));
        uint32_t w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(pa[0] * qs, -448.f), 448.f), fminf(fmaxf(pa[1] * qs, -448.f), 448.f), w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(pa[2] * qs, -448.f), 448.f), fminf(fmaxf(pa[3] * qs, -448.f), 448.f), w, true);
        *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(f8.q) + e) = w;
        if (tin > 0) tile[i >> 6][(i & 63) >> 2] = w;
      }
      if (zero_g) __builtin_nontemporal_store(f4_t{0.f, 0.f, 0.f, 0.f}, reinterpret_cast<f4_t*>(g + e));
    } else {
      for (int64_t k = i; k < n; ++k) {
        const int64_t q = base + k;
        float pa = p[q] * decay;
        const float ga = (g16 ? bf2f(g16[q]) : g[q]) * gsc;
        const float ma = beta1 * m[q] + (1.0f - beta1) * ga;
        const float va = beta2 * v[q] + (1.0f - beta2) * ga * ga;
        pa -= step_size * ma / (sqrtf(va) * inv_sqrt_bc2 + eps);
        p[q] = pa; m[q] = ma; v[q] = va;
        if (pb) pb[q] = f2bf(pa);
        if (zero_g) g[q] = 0.f;
      }
    }
  }
  if (qslot >= 0) {          // fp8-shadowed tensors are multiples of 4 elements (Linear weights): the scalar tail never holds them
    qmax = wave_max(qmax);
    if ((threadIdx.x & 63) == 0) amax_update(f8.amax + (long)qslot * CRCT_FP8_AMAX_LANES, qmax);
  }
  if (tin > 0) {             // uniform over the workgroup (one tensor per chunk)
    __syncthreads();
    const uint8_t* tb = reinterpret_cast<const uint8_t*>(&tile[0][0]);
    const int c = threadIdx.x >> 2, ch = (threadIdx.x & 3) * 16;      // transposed row = in index c, 16 consecutive out indices
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      w[k] = (uint32_t)tb[(ch + 4 * k) * 68 + c] | ((uint32_t)tb[(ch + 4 * k + 1) * 68 + c] << 8) |
             ((uint32_t)tb[(ch + 4 * k + 2) * 68 + c] << 16) | ((uint32_t)tb[(ch + 4 * k + 3) * 68 + c] << 24);
    *reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(f8.qt) + t_base + (int64_t)c * t_out + ch) = make_uint4(w[0], w[1], w[2], w[3]);
    __syncthreads();
  }
 }
}
__global__ void adamw_advance_kernel(int32_t* step, const float* found_inf) {
  if (threadIdx.x == 0 && (!found_inf || found_inf[0] == 0.f)) step[0] += 1;
}
// g[off[r] + blk_off .. ) = 0 over the chunk table of crct_adamw_plan (same chunking as the update itself)
__global__ __launch_bounds__(256) void zero_runs_kernel(float* __restrict__ g, const int64_t* __restrict__ run_off,
                                                        const int64_t* __restrict__ run_len, const int32_t* __restrict__ blk_seg,
                                                        const int64_t* __restrict__ blk_off, int n_blk) {
  for (int blk = blockIdx.x; blk < n_blk; blk += gridDim.x) {
    const int r = blk_seg[blk];
    const int64_t off = blk_off[blk], base = run_off[r] + off;
    int64_t n = run_len[r] - off;
    if (n > ADAMW_CHUNK) n = ADAMW_CHUNK;
    const int64_t head = (4 - (base & 3)) & 3;            // elements in front of the first 16-byte boundary
    for (int64_t i = threadIdx.x; i < (head < n ? head : n); i += 256) g[base + i] = 0.f;
    const int64_t nv = n > head ? (n - head) / 4 : 0;
    for (int64_t i = threadIdx.x; i < nv; i += 256)
      __builtin_nontemporal_store(f4_t{0.f, 0.f, 0.f, 0.f}, reinterpret_cast<f4_t*>(g + base + head) + i);
    for (int64_t i = head + nv * 4 + threadIdx.x; i < n; i += 256) g[base + i] = 0.f;
  }
}
}  // namespace

extern "C" int crct_zero_runs(float* base, const int64_t* off, const int64_t* len, const int32_t* blk_seg,
                              const int64_t* blk_off, int64_t n_blk, crct_stream_t stream) {
  CRCT_REQUIRE(base && off && len && blk_seg && blk_off, "zero_runs: null argument");
  if (n_blk <= 0) return 0;
  const long grid = n_blk > 1024 ? 1024 : n_blk;
  crct_launch(zero_runs_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, base, off, len, blk_seg, blk_off,
                     (int)n_blk);
  CRCT_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int64_t crct_adamw_plan(const int64_t* seg_len, int n_seg, int32_t* blk_seg, int64_t* blk_off, int64_t cap) {
  int64_t nb = 0;
  for (int s = 0; s < n_seg; ++s)
    for (int64_t o = 0; o < seg_len[s]; o += ADAMW_CHUNK) {
      if (blk_seg && blk_off && nb < cap) { blk_seg[nb] = s; blk_off[nb] = o; }
      ++nb;
    }
  return nb;
}

extern "C" int crct_adamw_step(float* p, float* g, float* m, float* v, void* p_bf16, const int64_t* seg_off,
                               const int64_t* seg_len, const float* seg_lr, const float* seg_wd, const int32_t* blk_seg,
                               const int64_t* blk_off, int64_t n_blk, float beta1, float beta2, float eps, int step,
                               const float* inv_scale_dev, const CrctAmpState* amp_state, const CrctFp8Shadow* fp8_shadow,
                               int max_workgroups, int zero_grads, const void* g_bf16, crct_stream_t stream) {
  CRCT_REQUIRE(step >= 1, "adamw: step must be >= 1 (got %d)", step);
  CrctAmpState amp = {nullptr, nullptr, nullptr};
  if (amp_state) amp = *amp_state;
  CrctFp8Shadow f8 = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (fp8_shadow) f8 = *fp8_shadow;
  CRCT_REQUIRE(!f8.q || (f8.seg_slot && f8.scale && f8.amax), "adamw: incomplete fp8 shadow description");
  if (n_blk <= 0) return 0;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  // max_workgroups > 0: a grid-stride launch of at most that many workgroups.  An update that runs BESIDE the next
  // forward on its own stream is throttled this way (256 = one workgroup per CU): at full width it saturates HBM for
  // 1.4 ms and the first layers of the forward crawl (measured: 10.25-10.36 -> 10.03 ms per step).
  const long grid = (max_workgroups > 0 && n_blk > max_workgroups) ? max_workgroups : n_blk;
#ifdef CRCT_GEMM_LAB   // timing only: the update as a kernel of the same grid and duration that touches no memory (what its TRAFFIC costs the forward beside it)
  static const double lab_spin_tbps = getenv("CRCT_LAB_ADAMW_SPIN") ? atof(getenv("CRCT_LAB_ADAMW_SPIN")) : 0.0;
  if (lab_spin_tbps > 0.0) {
    const long long ticks = (long long)((double)n_blk * ADAMW_CHUNK * 30.0 / (lab_spin_tbps * 1e12) * 1e8);
    static const int spin_threads = getenv("CRCT_LAB_ADAMW_SPIN_THREADS") ? atoi(getenv("CRCT_LAB_ADAMW_SPIN_THREADS")) : 256;
    static const int spin_grid = getenv("CRCT_LAB_ADAMW_SPIN_GRID") ? atoi(getenv("CRCT_LAB_ADAMW_SPIN_GRID")) : 0;
    crct_launch(lab_spin_kernel, dim3((unsigned)(spin_grid > 0 ? spin_grid : grid)), dim3(spin_threads), 0, (hipStream_t)stream, ticks);
    return 0;
  }
#endif
  crct_launch(adamw_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)p_bf16,
                     seg_off, seg_len, seg_lr, seg_wd, blk_seg, blk_off, beta1, beta2, eps, (float)(1.0 / bc1),
                     (float)(1.0 / sqrt(bc2)), inv_scale_dev, amp, f8, (int)n_blk, zero_grads, (const bf16_t*)g_bf16);
  CRCT_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int crct_adamw_advance(int32_t* step_dev, const float* found_inf_dev, crct_stream_t stream) {
  CRCT_REQUIRE(step_dev, "adamw_advance: null step counter");
  crct_launch(adamw_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_dev, found_inf_dev);
  CRCT_CHECK_HIP(hipGetLastError());
  return 0;
}
