// Device helpers shared by the MFMA attention kernels (attention_mfma.hip: short sequences, everything in registers;
// attention_long.hip: key-tile loop with online softmax): v_mfma_f32_16x16x16_bf16 fragments out of row-major LDS images.
#pragma once
#include "common.hip.h"
#include "attention_args.h"

namespace {

typedef s4_t __attribute__((address_space(3))) * lds_s4_ptr;

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}


__device__ __forceinline__ f4_t mma16(s4_t a, s4_t b, f4_t c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }

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

// fragment X[r0 + (lane & 15)][c0 + 4 (lane >> 4) + e] straight from global memory (rows >= T read as zero)
__device__ __forceinline__ s4_t frag_rows_global(const bf16_t* src, long ld, int T, int r0, int c0, int lane) {
  const int r = r0 + (lane & 15);
  // unconditional load (last row for the padding rows) and a mask: no branch, the loads stay batched
  const uint2 u = *reinterpret_cast<const uint2*>(src + (long)min(r, T - 1) * ld + c0 + 4 * (lane >> 4));
  const uint32_t m = r < T ? 0xffffffffu : 0u;
  return __builtin_bit_cast(s4_t, make_uint2(u.x & m, u.y & m));
}

// Kernel-argument preload (gemm.hip, GEMM_HOT_PARAMS): what the first instructions need (operand pointers, sizes, leading dimensions) as 15
// leading scalar arguments -- gfx950 hands the first argument dwords to the wave in SGPRs; a struct passed by value is fetched by scalar loads.
#define ATTN_HOT_PARAMS const bf16_t* hq, const bf16_t* hk, const bf16_t* hv, const uint8_t* hkm, int hB, int hheads, int hTq, int hTk, int hldq, int hldk, int hldv, const AttnArgs a_in
#define ATTN_HOT_UNPACK AttnArgs a = a_in; a.q = hq; a.k = hk; a.v = hv; a.keymask = hkm; a.B = hB; a.heads = hheads; a.Tq = hTq; a.Tk = hTk; a.ldq = hldq; a.ldk = hldk; a.ldv = hldv;
#define ATTN_HOT_ARGS(a) (a).q, (a).k, (a).v, (a).keymask, (a).B, (a).heads, (a).Tq, (a).Tk, (int)(a).ldq, (int)(a).ldk, (int)(a).ldv,

}  // namespace
