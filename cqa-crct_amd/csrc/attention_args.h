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
  int keep_cache;      // attention_long.hip backward: the dropout bits of phase A are kept in LDS for phase B (set by its launcher)
  int dbg;      // ablation bits, read only by -DCRCT_ATTN_LAB builds (tools/attn_lab); always 0 in the shipped library
};

bool crct_attention_mfma_ok(int Tq, int Tk, int d);
hipError_t crct_attention_mfma_fwd(const AttnArgs& a, hipStream_t s);
hipError_t crct_attention_mfma_bwd(const AttnArgs& a, hipStream_t s);
bool crct_attention_long_ok(int Tq, int Tk, int d);
hipError_t crct_attention_long_fwd(const AttnArgs& a, hipStream_t s);
hipError_t crct_attention_long_bwd(const AttnArgs& a, hipStream_t s);
