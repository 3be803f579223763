// Shared device helpers for the CRCT gfx950 kernels (wave64, bf16 storage / fp32 math).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CRCT_WAVE 64

typedef __attribute__((ext_vector_type(4))) short s4_t;
typedef __attribute__((ext_vector_type(8))) short s8_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8_t;
typedef __attribute__((ext_vector_type(4))) float f4_t;
typedef unsigned short bf16_t;   // raw bf16 bits in memory

// ------------------------------------------------------------------ bf16 <-> f32
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even through the hardware convert (keeps NaN a NaN, MI355X_MICROARCH.md)
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

// ------------------------------------------------------------------ wave reductions
// Pure-VALU reductions (DPP inside the 16-lane rows, v_readlane across the four rows); every lane gets the result.
// They deliberately stay OFF the LDS data path: the __shfl_xor butterfly compiles to ds_bpermute_b32, and with other
// waves saturating the LDS pipeline a late 16-lane slice of a bpermute result was observed to overwrite the
// destination register after the compiler had already given that register to another value (tools/embed_stress.py,
// DESIGN.md section 8).
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {      // sum over each row of 16 lanes, in every lane of the row
  v += dpp_f32<0xB1>(v);        // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(v);        // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(v);       // row_half_mirror
  v += dpp_f32<0x140>(v);       // row_mirror
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v));
  v = fmaxf(v, dpp_f32<0x4E>(v));
  v = fmaxf(v, dpp_f32<0x141>(v));
  v = fmaxf(v, dpp_f32<0x140>(v));
  return v;
}
__device__ __forceinline__ float lane_f32(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  return (lane_f32(v, 0) + lane_f32(v, 16)) + (lane_f32(v, 32) + lane_f32(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = row16_max(v);
  return fmaxf(fmaxf(lane_f32(v, 0), lane_f32(v, 16)), fmaxf(lane_f32(v, 32), lane_f32(v, 48)));
}

// ------------------------------------------------------------------ Philox4x32 (counter based RNG), 7 rounds
// Dropout masks are never stored: forward and backward regenerate them from
// (seed, site, element index).  One call yields 4 x u32 = eight 16-bit slices for the elements 8 idx8 .. 8 idx8 + 7 (philox_keep8).
// Rounds: 7 is the smallest count at which Philox4x32 passes BigCrush (Salmon, Moraes, Dror, Shaw: "Parallel Random Numbers: As Easy as
// 1, 2, 3", SC'11, table 2: "Crush-resistant" from 7 rounds on; the library default of 10 is that plus a safety margin).  A dropout mask asks
// for far less than BigCrush does, and every round is two 32 x 32 -> 64-bit multiplies per lane (measured, tools/lab/philox_rate.hip:
// 21 - 22 SIMD cycles per round and wave): all dropout together cost the step 0.25 ms at configs[1] and 0.87 ms at the reference's
// PlotQA shape with 10 rounds and 4 elements per call (bench.py --no-dropout).
#define CRCT_PHILOX_ROUNDS 7
struct Philox4 { uint32_t x, y, z, w; };
// `seed` is either the value itself or, with bit 63 set, the device address of a u64 holding it:
// a captured hipGraph bakes kernel arguments, so the per-step seed is then read from memory that the
// host refreshes before every replay (uniform address -> one scalar load).
#define CRCT_SEED_IN_MEMORY 0x8000000000000000ull
__device__ __forceinline__ Philox4 philox4x32(uint64_t seed, uint32_t site, uint64_t idx4) {
  if (seed & CRCT_SEED_IN_MEMORY) seed = *reinterpret_cast<const uint64_t*>(seed & ~CRCT_SEED_IN_MEMORY);
  uint32_t c0 = (uint32_t)idx4, c1 = (uint32_t)(idx4 >> 32), c2 = site, c3 = 0x9E3779B9u;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < CRCT_PHILOX_ROUNDS; ++r) {
    // 64-bit products: ONE v_mad_u64_u32 each instead of a v_mul_hi_u32 / v_mul_lo_u32 pair (tools/lab/philox_rate.hip: 136 against 158
    // cycles per 7-round wave-call)
    const uint64_t p0 = (uint64_t)c0 * 0xD2511F53ull, p1 = (uint64_t)c2 * 0xCD9E8D57ull;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return Philox4{c0, c1, c2, c3};
}
// Dropout bits, 8 per call: element e of the 8-element group `idx8` (e = element index & 7; every site numbers its elements row-major and
// its groups are 8-aligned) is kept iff the e-th 16-bit slice of the call's 128 random bits is >= the upper half of the 32-bit threshold --
// the drop probability is resolved to 2^-16 (p = 0.1 -> 6553 / 65536: 1e-5 below the scale's 1 / (1 - p), far under anything a mask of
// 10^6 elements can show) and one Philox call serves 8 elements instead of 4.  Bit e of the result = element e is kept.
__device__ __forceinline__ uint32_t philox_keep8(uint64_t seed, uint32_t site, uint64_t idx8, uint32_t thr) {
  const Philox4 p = philox4x32(seed, site, idx8);
  const uint32_t t = thr >> 16;
  return ((p.x & 0xffffu) >= t ? 1u : 0u) | ((p.x >> 16) >= t ? 2u : 0u) | ((p.y & 0xffffu) >= t ? 4u : 0u) | ((p.y >> 16) >= t ? 8u : 0u) |
         ((p.z & 0xffffu) >= t ? 16u : 0u) | ((p.z >> 16) >= t ? 32u : 0u) | ((p.w & 0xffffu) >= t ? 64u : 0u) | ((p.w >> 16) >= t ? 128u : 0u);
}
// keep-threshold for drop probability p: keep iff u32 >= thr (the kernels compare 16-bit slices against thr >> 16: philox_keep8)
__host__ __device__ __forceinline__ uint32_t drop_threshold(float p) {
  double t = (double)p * 4294967296.0;
  if (t <= 0.0) return 0u;
  if (t >= 4294967295.0) return 4294967295u;
  return (uint32_t)t;
}

// ------------------------------------------------------------------ activations
enum { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2, ACT_LEAKY = 3, ACT_TANH = 4 };

// erf-GELU (vilbert.py:111-117) and its derivative.  erf through Abramowitz & Stegun 7.1.26: for z >= 0
//   erf(z) = 1 - t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-z^2),  t = 1 / (1 + p z),  |error| <= 1.5e-7
// -- branch-free, one v_rcp and one v_exp, about a third of the instructions of the library erff (a two-branch minimax polynomial, both
// branches executed by a wave).  Every element of every FFN passes through it twice per step (123 M forward, 123 M backward at
// configs[1]), inside GEMM epilogues whose vector instructions share the SIMDs with other kernels' MFMAs.  The results are rounded to
// bf16 (2^-9 relative): the approximation error is invisible there.  exp(-z^2) with z = x / sqrt(2) is also the Gaussian of the
// derivative: gelu'(x) = Phi(x) + x phi(x) needs ONE exponential for both terms.  -DCRCT_GELU_LIBM_ERF restores erff (A/B builds).
__device__ __forceinline__ float erf_as7126_pos(float z, float e) {      // z >= 0, e = exp(-z * z)
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  return fmaf(-poly, e, 1.0f);
}
#ifdef CRCT_GELU_LIBM_ERF
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float kInvSqrt2Pi = 0.3989422804014327f;
  return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * kInvSqrt2Pi * __expf(-0.5f * x * x);
}
#else
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float er = copysignf(erf_as7126_pos(z, __expf(-z * z)), x);
  return 0.5f * x * (1.0f + er);
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float kInvSqrt2Pi = 0.3989422804014327f;
  const float z = fabsf(x) * 0.70710678118654752f;
  const float e = __expf(-z * z);                      // = exp(-x^2 / 2)
  const float er = copysignf(erf_as7126_pos(z, e), x);
#ifdef CRCT_GELU_GRAD_PERTURB      // tools/lab/libcrct_gelu_perturbed.so only: the build tests/test_kernels_gpu.py must be able to tell from this one
  return CRCT_GELU_GRAD_PERTURB * fmaf(x * kInvSqrt2Pi, e, 0.5f * (1.0f + er));
#else
  return fmaf(x * kInvSqrt2Pi, e, 0.5f * (1.0f + er));
#endif
}
#endif
__device__ __forceinline__ float act_apply(int act, float x) {
  switch (act) {
    case ACT_GELU: return gelu_erf(x);
    case ACT_RELU: return x > 0.f ? x : 0.f;
    case ACT_LEAKY: return x > 0.f ? x : 0.01f * x;
    case ACT_TANH: return tanhf(x);
    default: return x;
  }
}
// derivative given `s`: the saved PRE-activation for GELU, the saved OUTPUT for relu/leaky/tanh
__device__ __forceinline__ float act_grad(int act, float s) {
  switch (act) {
    case ACT_GELU: return gelu_erf_grad(s);
    case ACT_RELU: return s > 0.f ? 1.f : 0.f;
    case ACT_LEAKY: return s > 0.f ? 1.f : 0.01f;
    case ACT_TANH: return 1.f - s * s;
    default: return 1.f;
  }
}

// ------------------------------------------------------------------ running maximum of |x| (fp8 delayed scaling)
// Every amax "word" of the C ABI is CRCT_FP8_AMAX_LANES fp32 words: thousands of waves of one launch report the maximum of a
// tensor, and memory-side atomics to ONE address serialise (measured: LayerNorm 7.7 -> 31 us, GEMM max 190 us with a single
// word, because every wave of a launch still sees the freshly reset 0).  A wave max-es into word (workgroup id % LANES), so
// an address sees a dozen atomics per launch; crct_fp8_update_scales reduces the LANES words.  v >= 0: the integer order
// of the bits is the float order.
#ifndef CRCT_FP8_AMAX_LANES
#define CRCT_FP8_AMAX_LANES 64      // = include/crct_hip.h
#endif
__device__ __forceinline__ void amax_update(float* dst, float v) {
  float* w = dst + (blockIdx.x & (CRCT_FP8_AMAX_LANES - 1));
  if (v > __builtin_nontemporal_load(w)) atomicMax(reinterpret_cast<int*>(w), __float_as_int(v));
}

// Kernels of the two data streams' dependent chains raise their waves' issue priority (s_setprio 3): on a SIMD they share with waves of the
// work that is off the path by dependency -- the grouped weight gradients, AdamW, the column-sum passes, all at the default 0 -- they issue
// first.  Round 5: 7.29 - 7.34 -> 7.24 - 7.26 ms per step for the GEMMs alone (profiles/r5_setprio_ab.txt).  Results are unaffected.
__device__ __forceinline__ void crct_chain_priority() { __builtin_amdgcn_s_setprio(3); }
