// Arguments shared by the three attention implementations (attention.hip: fp32 VALU, any length up to 112;
// attention_mfma.hip: register-resident MFMA kernels, lengths up to 112, head sizes 32 / 48 / 64; attention_long.hip: key-tile loop
// with online softmax, lengths up to CRCT_ATTN_MAX_LEN, head sizes 32 / 48 / 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.hip.h"

struct AttnArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; const uint8_t* keymask;
  bf16_t* ctx;
  const bf16_t* dctx; bf16_t* dq; bf16_t* dk; bf16_t* dv;
  int B, heads, Tq, Tk, d;
  long ldq, ldk, ldv, ldo, lddq, lddk, lddv;
  uint32_t thr; float dscale; uint32_t site; uint64_t seed;
  float scale;
  // fp8 copies of the results (MFMA kernels of both files only; include/crct_hip.h CrctAttnQuant): ctx as e4m3, dq / dk / dv as e5m2, each with
  // the leading dimension of its bf16 twin (in bytes), quantised with *scale, max |.| into *amax (CRCT_FP8_AMAX_LANES words)
  uint8_t* ctx_q; const float* ctx_qscale; float* ctx_qamax;
  uint8_t* dq_q; uint8_t* dk_q; uint8_t* dv_q;
  const float* dq_qscale; float* dq_qamax; const float* dkv_qscale; float* dkv_qamax;
  // Row statistics of the softmax, attention_long.hip only (include/crct_hip.h CrctAttnQuant.row_lse): the forward writes
  // log2 sum_j exp2(s_ij) per (batch, head, query) [B][heads][Tq]; a backward that gets them AND the forward's output (`ctx`, leading
  // dimension ldc) skips its statistics sweep: p = exp2(s - lse), delta_i = dctx_i . ctx_i
  float* lse; long ldc;
  int keep_cache;      // attention_long.hip backward: the dropout bits of phase A are kept in LDS for phase B (set by its launcher)
  int dbg;      // ablation bits, read only by -DCRCT_ATTN_LAB builds (tools/attn_lab); always 0 in the shipped library
};

// Dropout bits of the attention probabilities, shared by the three implementations.  The MFMA kernels give a lane the keys 16 jt + 4 g .. + 3
// of every key tile jt, so the 8 elements of one Philox call (philox_keep8) are numbered to suit THEM: call ((bh Tq + i) NP + P) 4 + g covers
// query i, keys 32 P + 4 g + r (slices 0 - 3) and 32 P + 16 + 4 g + r (slices 4 - 7) -- the lane's keys of the key-tile pair P = jt / 2;
// NP = ceil(Tk / 32).  Returns the 8 keep bits (bit 4 (jt & 1) + r: key 16 jt + 4 g + r).
__device__ __forceinline__ uint32_t attn_keep8(uint64_t seed, uint32_t site, long bh, int Tq, int Tk, int i, int pair, int g, uint32_t thr) {
  const uint64_t np = (uint64_t)((Tk + 31) >> 5);
  return philox_keep8(seed, site, (((uint64_t)(bh * Tq + i)) * np + (uint64_t)pair) * 4u + (uint64_t)g, thr);
}

bool crct_attention_mfma_ok(int Tq, int Tk, int d);
hipError_t crct_attention_mfma_fwd(const AttnArgs& a, hipStream_t s);
hipError_t crct_attention_mfma_bwd(const AttnArgs& a, hipStream_t s);
bool crct_attention_long_ok(int Tq, int Tk, int d);
hipError_t crct_attention_long_fwd(const AttnArgs& a, hipStream_t s);
hipError_t crct_attention_long_bwd(const AttnArgs& a, hipStream_t s);
