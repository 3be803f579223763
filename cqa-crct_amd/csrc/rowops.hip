// Row-wise HBM-bound kernels of the CRCT step: LayerNorm fwd/bwd, text / image embedding
// fwd/bwd (sum of gathers + Linear(4) + LayerNorm + dropout), column sums, row softmax, casts.
// One wave (64 lanes) owns one row; every lane keeps NCH 16-byte chunks (8 bf16) of the row in
// registers, so a row is read once and written once.  Statistics are fp32.
// Reference arithmetic: BertLayerNorm vilbert.py:281-294; BertEmbeddingLocation :320-358;
// BertImageEmbeddings :1474-1496.
#include <type_traits>

#include "common.hip.h"
#include "crct_internal.h"

namespace {

constexpr int ROWS_PER_BLOCK = 4;   // waves per 256-thread workgroup

template <int NCH>
struct Row {
  float v[NCH][8];
};

// All loads of the row are issued before the first one is unpacked, and none is conditional: a chunk past the end of
// the row re-reads the last chunk and is masked to zero (a branch around each 16-byte load made the compiler wait for
// every load separately: 4-6 serial memory round trips per row in the LayerNorm kernels).
template <int NCH>
__device__ __forceinline__ void row_load_bf16(Row<NCH>& r, const bf16_t* p, int H, int lane) {
  uint4 u[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) u[i] = *reinterpret_cast<const uint4*>(p + min((lane + 64 * i) * 8, H - 8));
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const uint32_t m = (lane + 64 * i) * 8 < H ? 0xffffffffu : 0u;
    const uint32_t x = u[i].x & m, y = u[i].y & m, z = u[i].z & m, w = u[i].w & m;
    r.v[i][0] = bf2f((bf16_t)(x & 0xffff)); r.v[i][1] = bf2f((bf16_t)(x >> 16));
    r.v[i][2] = bf2f((bf16_t)(y & 0xffff)); r.v[i][3] = bf2f((bf16_t)(y >> 16));
    r.v[i][4] = bf2f((bf16_t)(z & 0xffff)); r.v[i][5] = bf2f((bf16_t)(z >> 16));
    r.v[i][6] = bf2f((bf16_t)(w & 0xffff)); r.v[i][7] = bf2f((bf16_t)(w >> 16));
  }
}
// the two halves of row_load_bf16, for loops that keep the NEXT row's loads in flight while they work on the current one
template <int NCH>
struct RawRow { uint4 u[NCH]; };
template <int NCH>
__device__ __forceinline__ void row_fetch_bf16(RawRow<NCH>& raw, const bf16_t* p, int H, int lane) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) raw.u[i] = *reinterpret_cast<const uint4*>(p + min((lane + 64 * i) * 8, H - 8));
}
template <int NCH>
__device__ __forceinline__ void row_unpack_bf16(Row<NCH>& r, const RawRow<NCH>& raw, int H, int lane) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const uint32_t m = (lane + 64 * i) * 8 < H ? 0xffffffffu : 0u;
    const uint32_t x = raw.u[i].x & m, y = raw.u[i].y & m, z = raw.u[i].z & m, w = raw.u[i].w & m;
    r.v[i][0] = bf2f((bf16_t)(x & 0xffff)); r.v[i][1] = bf2f((bf16_t)(x >> 16));
    r.v[i][2] = bf2f((bf16_t)(y & 0xffff)); r.v[i][3] = bf2f((bf16_t)(y >> 16));
    r.v[i][4] = bf2f((bf16_t)(z & 0xffff)); r.v[i][5] = bf2f((bf16_t)(z >> 16));
    r.v[i][6] = bf2f((bf16_t)(w & 0xffff)); r.v[i][7] = bf2f((bf16_t)(w >> 16));
  }
}
// r <- the values its bf16 copy holds (what row_store_bf16 followed by row_load_bf16 of the same row would give, without the round trip)
template <int NCH>
__device__ __forceinline__ void row_round_bf16(Row<NCH>& r) {
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) r.v[i][j] = bf2f(f2bf(r.v[i][j]));
}
// the same pair for fp32 rows (the pre-LayerNorm sums of the fp32 residual stream)
template <int NCH>
struct RawRowF { float4 a[NCH], b[NCH]; };
template <int NCH>
__device__ __forceinline__ void row_fetch_f32(RawRowF<NCH>& raw, const float* p, int H, int lane) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = min((lane + 64 * i) * 8, H - 8);
    raw.a[i] = *reinterpret_cast<const float4*>(p + c);
    raw.b[i] = *reinterpret_cast<const float4*>(p + c + 4);
  }
}
template <int NCH>
__device__ __forceinline__ void row_unpack_f32(Row<NCH>& r, const RawRowF<NCH>& raw, int H, int lane) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const float m = (lane + 64 * i) * 8 < H ? 1.f : 0.f;
    r.v[i][0] = raw.a[i].x * m; r.v[i][1] = raw.a[i].y * m; r.v[i][2] = raw.a[i].z * m; r.v[i][3] = raw.a[i].w * m;
    r.v[i][4] = raw.b[i].x * m; r.v[i][5] = raw.b[i].y * m; r.v[i][6] = raw.b[i].z * m; r.v[i][7] = raw.b[i].w * m;
  }
}
template <int NCH>
__device__ __forceinline__ void row_store_bf16(const Row<NCH>& r, bf16_t* p, int H, int lane) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < H) {
      uint4 u;
      u.x = pack2bf(r.v[i][0], r.v[i][1]); u.y = pack2bf(r.v[i][2], r.v[i][3]);
      u.z = pack2bf(r.v[i][4], r.v[i][5]); u.w = pack2bf(r.v[i][6], r.v[i][7]);
      *reinterpret_cast<uint4*>(p + c) = u;
    }
  }
}
// OCP e4m3 copy of a row, q = clamp(x * qs, +-448); returns max |x| over this lane's elements (the NEXT step's scale)
__device__ __forceinline__ float f8_clamp(float x) { return fminf(fmaxf(x, -448.0f), 448.0f); }
__device__ __forceinline__ uint2 pack8_fp8(const float (&v)[8], float qs) {
  uint32_t w0 = 0, w1 = 0;
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f8_clamp(v[0] * qs), f8_clamp(v[1] * qs), w0, false);
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f8_clamp(v[2] * qs), f8_clamp(v[3] * qs), w0, true);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f8_clamp(v[4] * qs), f8_clamp(v[5] * qs), w1, false);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f8_clamp(v[6] * qs), f8_clamp(v[7] * qs), w1, true);
  return make_uint2(w0, w1);
}
template <int NCH>
__device__ __forceinline__ float row_store_fp8(const Row<NCH>& r, uint8_t* p, int H, int lane, float qs) {
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < H) {
      // the bf16 copy is what backward and the bf16 GEMMs see: quantise the bf16-rounded value, so both copies agree
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[j] = bf2f(f2bf(r.v[i][j])); amax = fmaxf(amax, fabsf(v[j])); }
      *reinterpret_cast<uint2*>(p + c) = pack8_fp8(v, qs);
    }
  }
  return amax;
}
// OCP e5m2 copy of a row (a GRADIENT that the next data-gradient GEMM reads as its fp8 A operand): q = clamp(x * qs, +-57344)
__device__ __forceinline__ float bf8_clamp(float x) { return fminf(fmaxf(x, -57344.0f), 57344.0f); }
__device__ __forceinline__ uint2 pack8_bf8(const float (&v)[8], float qs) {
  uint32_t w0 = 0, w1 = 0;
  w0 = __builtin_amdgcn_cvt_pk_bf8_f32(bf8_clamp(v[0] * qs), bf8_clamp(v[1] * qs), w0, false);
  w0 = __builtin_amdgcn_cvt_pk_bf8_f32(bf8_clamp(v[2] * qs), bf8_clamp(v[3] * qs), w0, true);
  w1 = __builtin_amdgcn_cvt_pk_bf8_f32(bf8_clamp(v[4] * qs), bf8_clamp(v[5] * qs), w1, false);
  w1 = __builtin_amdgcn_cvt_pk_bf8_f32(bf8_clamp(v[6] * qs), bf8_clamp(v[7] * qs), w1, true);
  return make_uint2(w0, w1);
}
template <int NCH>
__device__ __forceinline__ float row_store_bf8(const Row<NCH>& r, uint8_t* p, int H, int lane, float qs) {
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < H) {
      float v[8];      // the bf16-rounded value, as in row_store_fp8: the bf16 copy (weight gradients) and this one agree
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[j] = bf2f(f2bf(r.v[i][j])); amax = fmaxf(amax, fabsf(v[j])); }
      *reinterpret_cast<uint2*>(p + c) = pack8_bf8(v, qs);
    }
  }
  return amax;
}
template <int NCH>
__device__ __forceinline__ void row_load_f32(Row<NCH>& r, const float* p, int H, int lane) {
  float4 a[NCH], b[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = min((lane + 64 * i) * 8, H - 8);
    a[i] = *reinterpret_cast<const float4*>(p + c);
    b[i] = *reinterpret_cast<const float4*>(p + c + 4);
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const float m = (lane + 64 * i) * 8 < H ? 1.f : 0.f;
    r.v[i][0] = a[i].x * m; r.v[i][1] = a[i].y * m; r.v[i][2] = a[i].z * m; r.v[i][3] = a[i].w * m;
    r.v[i][4] = b[i].x * m; r.v[i][5] = b[i].y * m; r.v[i][6] = b[i].z * m; r.v[i][7] = b[i].w * m;
  }
}
template <int NCH>
__device__ __forceinline__ void row_add_f32(Row<NCH>& r, const float* p, int H, int lane) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < H) {
      const float4 a = *reinterpret_cast<const float4*>(p + c);
      const float4 b = *reinterpret_cast<const float4*>(p + c + 4);
      r.v[i][0] += a.x; r.v[i][1] += a.y; r.v[i][2] += a.z; r.v[i][3] += a.w;
      r.v[i][4] += b.x; r.v[i][5] += b.y; r.v[i][6] += b.z; r.v[i][7] += b.w;
    }
  }
}
// r += W[c][0..3] . loc + b[c]   (a Linear(4, H): txt_location_embeddings / new_loc_emb)
template <int NCH>
__device__ __forceinline__ void row_add_loc_linear(Row<NCH>& r, const float* w, const float* b, const float l[4], int H, int lane) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < H) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 wv = *reinterpret_cast<const float4*>(w + (long)(c + j) * 4);
        r.v[i][j] += b[c + j] + wv.x * l[0] + wv.y * l[1] + wv.z * l[2] + wv.w * l[3];
      }
    }
  }
}

template <int NCH>
__device__ __forceinline__ void row_stats(const Row<NCH>& r, int H, int lane, float eps, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += r.v[i][j];     // out-of-range chunks hold zeros
  mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < H) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = r.v[i][j] - mean; q += d * d; }
    }
  }
  const float var = wave_sum(q) / (float)H;
  rstd = 1.0f / sqrtf(var + eps);
}

// y = gamma * (x - mean) * rstd + beta, optional post-norm dropout keyed by (row*H + col); gamma / beta held in registers
// (loaded once per wave, outside the row loop)
template <int NCH>
__device__ __forceinline__ void row_normalize(Row<NCH>& r, const Row<NCH>& gamma, const Row<NCH>& beta, int H, int lane,
                                              float mean, float rstd, long row, uint32_t thr, float scale,
                                              uint32_t site, uint64_t seed) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < H) {
#pragma unroll
      for (int j = 0; j < 8; ++j) r.v[i][j] = gamma.v[i][j] * ((r.v[i][j] - mean) * rstd) + beta.v[i][j];
      if (thr) {
        const uint32_t kb = philox_keep8(seed, site, ((uint64_t)row * (uint64_t)H + (uint64_t)c) >> 3, thr);      // c % 8 == 0, H % 8 == 0
#pragma unroll
        for (int j = 0; j < 8; ++j) r.v[i][j] = ((kb >> j) & 1u) ? r.v[i][j] * scale : 0.f;
      }
    }
  }
}
// y = gamma * (x - mean) * rstd + beta, optional post-norm dropout keyed by (row*H + col)
template <int NCH>
__device__ __forceinline__ void row_normalize(Row<NCH>& r, const float* gamma, const float* beta, int H, int lane,
                                              float mean, float rstd, long row, uint32_t thr, float scale,
                                              uint32_t site, uint64_t seed) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < H) {
      const float4 g0 = *reinterpret_cast<const float4*>(gamma + c), g1 = *reinterpret_cast<const float4*>(gamma + c + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(beta + c), b1 = *reinterpret_cast<const float4*>(beta + c + 4);
      const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) r.v[i][j] = gg[j] * ((r.v[i][j] - mean) * rstd) + bb[j];
      if (thr) {
        const uint32_t kb = philox_keep8(seed, site, ((uint64_t)row * (uint64_t)H + (uint64_t)c) >> 3, thr);      // c % 8 == 0, H % 8 == 0
#pragma unroll
        for (int j = 0; j < 8; ++j) r.v[i][j] = ((kb >> j) & 1u) ? r.v[i][j] * scale : 0.f;
      }
    }
  }
}

// multiply the row by the dropout keep-mask * scale of (site, seed) keyed by (row*H + col)
template <int NCH>
__device__ __forceinline__ void row_apply_dropmask(Row<NCH>& r, int H, int lane, long row, uint32_t thr, float scale,
                                                   uint32_t site, uint64_t seed) {
  if (!thr) return;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < H) {
      const uint32_t kb = philox_keep8(seed, site, ((uint64_t)row * (uint64_t)H + (uint64_t)c) >> 3, thr);
#pragma unroll
      for (int j = 0; j < 8; ++j) r.v[i][j] = ((kb >> j) & 1u) ? r.v[i][j] * scale : 0.f;
    }
  }
}

// LayerNorm backward of one row.  in: dy (gradient w.r.t. y), x (pre-norm row).  out: dy <- dx,
// xhat in x.  dgamma/dbeta accumulators updated.
template <int NCH>
__device__ __forceinline__ void row_ln_bwd(Row<NCH>& dy, Row<NCH>& x, const Row<NCH>& gamma, int H, int lane,
                                           float mean, float rstd, Row<NCH>& acc_dg, Row<NCH>& acc_db) {
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < H) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (x.v[i][j] - mean) * rstd;
        x.v[i][j] = xh;
        acc_dg.v[i][j] += dy.v[i][j] * xh;
        acc_db.v[i][j] += dy.v[i][j];
        const float g = dy.v[i][j] * gamma.v[i][j];
        dy.v[i][j] = g;
        s1 += g; s2 += g * xh;
      }
    }
  }
  const float c1 = wave_sum(s1) / (float)H, c2 = wave_sum(s2) / (float)H;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < H) {
#pragma unroll
      for (int j = 0; j < 8; ++j) dy.v[i][j] = rstd * (dy.v[i][j] - c1 - x.v[i][j] * c2);
    }
  }
}
template <int NCH>
__device__ __forceinline__ void row_ln_bwd(Row<NCH>& dy, Row<NCH>& x, const float* gamma, int H, int lane,
                                           float mean, float rstd, Row<NCH>& acc_dg, Row<NCH>& acc_db) {
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < H) {
      const float4 g0 = *reinterpret_cast<const float4*>(gamma + c), g1 = *reinterpret_cast<const float4*>(gamma + c + 4);
      const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (x.v[i][j] - mean) * rstd;
        x.v[i][j] = xh;
        acc_dg.v[i][j] += dy.v[i][j] * xh;
        acc_db.v[i][j] += dy.v[i][j];
        const float g = dy.v[i][j] * gg[j];
        dy.v[i][j] = g;
        s1 += g; s2 += g * xh;
      }
    }
  }
  const float c1 = wave_sum(s1) / (float)H, c2 = wave_sum(s2) / (float)H;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < H) {
#pragma unroll
      for (int j = 0; j < 8; ++j) dy.v[i][j] = rstd * (dy.v[i][j] - c1 - x.v[i][j] * c2);
    }
  }
}

template <int NCH>
__device__ __forceinline__ void row_zero(Row<NCH>& r) {
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) r.v[i][j] = 0.f;
}
template <int NCH>
__device__ __forceinline__ void row_acc(Row<NCH>& a, const Row<NCH>& r, float s = 1.f) {
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) a.v[i][j] += r.v[i][j] * s;
}
template <int NCH>
__device__ __forceinline__ void row_atomic_add(const Row<NCH>& r, float* dst, int H, int lane) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < H) {
#pragma unroll
      for (int j = 0; j < 8; ++j) atomicAdd(dst + c + j, r.v[i][j]);
    }
  }
}

template <int NCH>
__device__ __forceinline__ void row_store_f32(const Row<NCH>& r, float* p, int H, int lane) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < H) {
      *reinterpret_cast<float4*>(p + c) = make_float4(r.v[i][0], r.v[i][1], r.v[i][2], r.v[i][3]);
      *reinterpret_cast<float4*>(p + c + 4) = make_float4(r.v[i][4], r.v[i][5], r.v[i][6], r.v[i][7]);
    }
  }
}

// streaming form: the fp32 copy of a LayerNorm output is read two kernels later (the next residual GEMM's epilogue), by every XCD --
// write-through stores leave no dirty L2 lines for the end-of-kernel write-back (-0.1 ms per PlotQA-shaped step, neutral at configs[1])
template <int NCH>
__device__ __forceinline__ void row_store_f32_nt(const Row<NCH>& r, float* p, int H, int lane) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < H) {
      __builtin_nontemporal_store(f4_t{r.v[i][0], r.v[i][1], r.v[i][2], r.v[i][3]}, reinterpret_cast<f4_t*>(p + c));
      __builtin_nontemporal_store(f4_t{r.v[i][4], r.v[i][5], r.v[i][6], r.v[i][7]}, reinterpret_cast<f4_t*>(p + c + 4));
    }
  }
}

// table[p][:] += sum of rows[r][:] over the rows r with idx[r] == p   (small tables: position / type / colour
// embeddings, where thousands of rows hit a few dozen table rows and float atomics serialise).  One workgroup per
// table row: the 256 threads scan idx in a fixed strided order, compact the matching row numbers into LDS through a
// block prefix sum (a fixed order: the result does not depend on timing), then every thread sums its 4 columns over
// the listed rows.  M <= GATHER_MAX_ROWS.
constexpr int GATHER_MAX_ROWS = 15360;     // 4 bytes per row + the 1 KiB of counts stay under the default 64 KiB of dynamic LDS
// Two tables in one launch: workgroups [0, n0) serve (idx, table), the rest (idx1, table1).
__device__ __forceinline__ void gather_sum_body(const float* __restrict__ rows, const int* __restrict__ idx, int M, int H,
                                                float* __restrict__ table, int n0, const int* __restrict__ idx1,
                                                float* __restrict__ table1, int blk, char* smem) {
  int* list = reinterpret_cast<int*>(smem);              // [M]
  __shared__ int cnt[256];
  const int tid = threadIdx.x;
  int p = blk;
  if (p >= n0) { p -= n0; idx = idx1; table = table1; }
  int n = 0;
  for (int r = tid; r < M; r += 256) n += idx[r] == p;
  cnt[tid] = n;
  __syncthreads();
  // exclusive prefix over the 256 counts (Hillis-Steele in LDS)
  for (int o = 1; o < 256; o <<= 1) {
    const int v = tid >= o ? cnt[tid - o] : 0;
    __syncthreads();
    cnt[tid] += v;
    __syncthreads();
  }
  const int total = cnt[255];
  if (total == 0) return;
  int at = cnt[tid] - n;
  for (int r = tid; r < M; r += 256)
    if (idx[r] == p) list[at++] = r;
  __syncthreads();
  for (int c = tid * 4; c < H; c += 1024) {
    float4 acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    int i = 0;
    for (; i + 7 < total; i += 8) {                 // eight rows in flight per thread (a type row can list every token)
      float4 x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const float4*>(rows + (long)list[i + u] * H + c);
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc[u].x += x[u].x; acc[u].y += x[u].y; acc[u].z += x[u].z; acc[u].w += x[u].w; }
    }
    for (; i < total; ++i) {
      const float4 x = *reinterpret_cast<const float4*>(rows + (long)list[i] * H + c);
      acc[i & 7].x += x.x; acc[i & 7].y += x.y; acc[i & 7].z += x.z; acc[i & 7].w += x.w;
    }
    float4* dst = reinterpret_cast<float4*>(table + (long)p * H + c);
    float4 o = *dst;
    o.x += ((acc[0].x + acc[1].x) + (acc[2].x + acc[3].x)) + ((acc[4].x + acc[5].x) + (acc[6].x + acc[7].x));
    o.y += ((acc[0].y + acc[1].y) + (acc[2].y + acc[3].y)) + ((acc[4].y + acc[5].y) + (acc[6].y + acc[7].y));
    o.z += ((acc[0].z + acc[1].z) + (acc[2].z + acc[3].z)) + ((acc[4].z + acc[5].z) + (acc[6].z + acc[7].z));
    o.w += ((acc[0].w + acc[1].w) + (acc[2].w + acc[3].w)) + ((acc[4].w + acc[5].w) + (acc[6].w + acc[7].w));
    *dst = o;
  }
}
__global__ __launch_bounds__(256) void gather_sum_kernel(const float* __restrict__ rows, const int* __restrict__ idx, int M, int H,
                                                         float* __restrict__ table, int n0, const int* __restrict__ idx1,
                                                         float* __restrict__ table1) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gather_sum_body(rows, idx, M, H, table, n0, idx1, table1, blockIdx.x, smem);
}

// word_embeddings gradient without float atomics: table[ids[r]][:] += rows[r][:] summed in a FIXED order.  One wave per token
// row r: if an earlier row carries the same id the wave has nothing to do; otherwise it owns that table row and adds the
// fp32 gradient rows of ALL tokens with this id in increasing row order (ballot over 64 ids at a time).  The 30 522-row
// table is touched by <= B*T rows, most ids occur once, [PAD] / [CLS] / [SEP] a few hundred times: the result is bitwise
// reproducible, which the atomics were not (arrival-order rounding: 1 of ~12 runs ended a 38-step training run on a
// different loss, round-2 tools/det_check.sh).
// One workgroup of 4 waves serves 4 consecutive token rows (one wave each); the ids of ALL rows sit in LDS (int32, one
// coalesced pass), so the scans are LDS reads.  A wave keeps its whole output row in registers (H / 64 columns per lane, 16-byte
// pieces).  Light ids (<= WORD_HEAVY matches, nearly all of them) are summed by the owning wave alone.  A heavy id ([PAD]: a few
// hundred rows) would leave one wave walking hundreds of dependent 3 KB loads while the chip idles (106 us in the round-2
// timeline), so its workgroup shares it: the owner lists the matches in LDS, match e goes to wave (e / 4) % 4, accumulator e % 4
// (four independent rows in flight per wave), and the 4 x 4 partial rows are folded in a fixed tree -- still bit-reproducible.
constexpr int WORD_HEAVY = 8;
template <int NCH>
__device__ __forceinline__ void word_scatter_body(const float* __restrict__ rows, const int64_t* __restrict__ ids, int M, int H,
                                                  float* __restrict__ table, const int blk, char* smem) {
  int* sid = reinterpret_cast<int*>(smem);               // [M]
  int* list = sid + M;                                   // [M]   matches of a heavy id
  float* part = reinterpret_cast<float*>(sid + ((2 * M + 3) & ~3));      // [4][H] per-wave partial rows of a heavy id
  __shared__ int s_cnt[ROWS_PER_BLOCK];
  for (int i = threadIdx.x; i < M; i += 256) sid[i] = (int)ids[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = blk * ROWS_PER_BLOCK + w;
  const int id = r < M ? sid[r] : -1;
  bool earlier = r >= M;
  for (int i = lane; i < r && i < M; i += 64) earlier |= sid[i] == id;
  const bool own = __ballot(earlier) == 0ull;            // no earlier row carries this id: this wave owns the table row
  int cnt = 0;
  if (own)
    for (int base = r & ~63; base < M; base += 64) {
      const int i = base + lane;
      cnt += __builtin_popcountll(__ballot(i >= r && i < M && sid[i] == id));
    }
  if (lane == 0) s_cnt[w] = cnt;
  __syncthreads();
  if (own && cnt <= WORD_HEAVY) {
    Row<NCH> acc;
    row_zero(acc);
    for (int base = r & ~63; base < M; base += 64) {
      const int i = base + lane;
      unsigned long long m = __ballot(i >= r && i < M && sid[i] == id);
      while (m) {                                        // increasing row order: a fixed summation order
        const int j = base + __builtin_ctzll(m);
        m &= m - 1;
        row_add_f32(acc, rows + (long)j * H, H, lane);
      }
    }
    row_add_f32(acc, table + (long)id * H, H, lane);
    row_store_f32(acc, table + (long)id * H, H, lane);
  }
  for (int k = 0; k < ROWS_PER_BLOCK; ++k) {             // workgroup-uniform: s_cnt is shared
    const int n = s_cnt[k];
    if (n <= WORD_HEAVY) continue;
    const int rk = blk * ROWS_PER_BLOCK + k, idk = sid[rk];
    if (w == k) {
      int at = 0;
      for (int base = rk & ~63; base < M; base += 64) {
        const int i = base + lane;
        const bool hit = i >= rk && i < M && sid[i] == idk;
        const unsigned long long m = __ballot(hit);
        if (hit) list[at + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = i;
        at += __builtin_popcountll(m);
      }
    }
    __syncthreads();
    Row<NCH> a0, a1, a2, a3;
    row_zero(a0); row_zero(a1); row_zero(a2); row_zero(a3);
    int e = w * 4;
    for (; e + 3 < n; e += 16) {
      const int j0 = list[e], j1 = list[e + 1], j2 = list[e + 2], j3 = list[e + 3];
      row_add_f32(a0, rows + (long)j0 * H, H, lane);
      row_add_f32(a1, rows + (long)j1 * H, H, lane);
      row_add_f32(a2, rows + (long)j2 * H, H, lane);
      row_add_f32(a3, rows + (long)j3 * H, H, lane);
    }
    if (e < n) row_add_f32(a0, rows + (long)list[e] * H, H, lane);
    if (e + 1 < n) row_add_f32(a1, rows + (long)list[e + 1] * H, H, lane);
    if (e + 2 < n) row_add_f32(a2, rows + (long)list[e + 2] * H, H, lane);
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) a0.v[i][j] = (a0.v[i][j] + a1.v[i][j]) + (a2.v[i][j] + a3.v[i][j]);
    row_store_f32(a0, part + (long)w * H, H, lane);
    __syncthreads();
    if (w == k) {
      Row<NCH> t, u;
      row_load_f32(t, part, H, lane);
      row_load_f32(u, part + H, H, lane);
#pragma unroll
      for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) t.v[i][j] += u.v[i][j];
      row_load_f32(u, part + 2 * (long)H, H, lane);
      Row<NCH> v;
      row_load_f32(v, part + 3 * (long)H, H, lane);
#pragma unroll
      for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) t.v[i][j] += u.v[i][j] + v.v[i][j];
      row_add_f32(t, table + (long)idk * H, H, lane);
      row_store_f32(t, table + (long)idk * H, H, lane);
    }
    __syncthreads();
  }
}

// The same sums (same owner, same order, same bits) WITHOUT the O(M^2 / 64) scans: the kernel that produced the gradient rows has left,
// per token id, first[id] = M - (smallest row with this id) and last[id] = (largest such row) + 1 (integer atomic maxima: order-free),
// in two zero-initialised tables of vocabulary size.  A wave whose row is not its id's first row returns at once; an id that occurs once
// (nearly all of them) costs its owner one row; otherwise the owner scans ids[r .. last] only, straight from global memory -- no LDS
// image of all ids (40 KB per workgroup at 9 920 rows), no "does an earlier row carry my id" scan per row.  The owner puts its two table
// entries back to zero (nobody else reads them any more: a later wave of the same id sees 0 != its own code and returns, as it would
// have anyway).  embed_scatter: 320 -> 35 us at 9 920 rows, 68 -> 17 us at 1 600 (EXPERIMENTS.md round 6).
template <int NCH>
__device__ __forceinline__ void word_scatter_indexed_body(const float* __restrict__ rows, const int64_t* __restrict__ ids, int M, int H,
                                                          float* __restrict__ table, int* __restrict__ first, int* __restrict__ last,
                                                          const int blk, char* smem) {
  int* list = reinterpret_cast<int*>(smem);              // [M]   matches of a heavy id
  float* part = reinterpret_cast<float*>(list + ((M + 3) & ~3));      // [4][H] per-wave partial rows of a heavy id
  __shared__ int s_cnt[ROWS_PER_BLOCK], s_last[ROWS_PER_BLOCK];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = blk * ROWS_PER_BLOCK + w;
  const int id = r < M ? (int)ids[r] : -1;
  const bool own = r < M && first[id] == M - r;
  const int l = own ? last[id] - 1 : r;                  // largest row with this id
  // the ids of (r, l], 256 at a time: four independent loads per lane in flight before the first ballot (the scans are chains of dependent
  // global loads otherwise: 25 round trips for a [PAD] owner at 1 600 rows)
  auto hits4 = [&](int base, unsigned long long (&m)[4]) {
    int v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = base + 64 * u + lane; v[u] = (i > r && i <= l) ? (int)ids[i] : -2; }
#pragma unroll
    for (int u = 0; u < 4; ++u) m[u] = __ballot(v[u] == id);
  };
  int cnt = own ? 1 : 0;
  if (own && l > r)
    for (int base = (r + 1) & ~63; base <= l; base += 256) {
      unsigned long long m[4];
      hits4(base, m);
      cnt += __builtin_popcountll(m[0]) + __builtin_popcountll(m[1]) + __builtin_popcountll(m[2]) + __builtin_popcountll(m[3]);
    }
  if (lane == 0) { s_cnt[w] = cnt; s_last[w] = l; }
  __syncthreads();
  if (own && cnt <= WORD_HEAVY) {
    Row<NCH> acc;
    row_zero(acc);
    row_add_f32(acc, rows + (long)r * H, H, lane);
    if (cnt > 1)
      for (int base = (r + 1) & ~63; base <= l; base += 256) {
        unsigned long long m[4];
        hits4(base, m);
#pragma unroll
        for (int u = 0; u < 4; ++u)
          while (m[u]) {                                 // increasing row order: a fixed summation order
            const int j = base + 64 * u + __builtin_ctzll(m[u]);
            m[u] &= m[u] - 1;
            row_add_f32(acc, rows + (long)j * H, H, lane);
          }
      }
    row_add_f32(acc, table + (long)id * H, H, lane);
    row_store_f32(acc, table + (long)id * H, H, lane);
  }
  if (own && lane == 0) { first[id] = 0; last[id] = 0; }
  for (int k = 0; k < ROWS_PER_BLOCK; ++k) {             // workgroup-uniform: s_cnt is shared
    const int n = s_cnt[k];
    if (n <= WORD_HEAVY) continue;
    const int rk = blk * ROWS_PER_BLOCK + k, idk = (int)ids[rk], lk = s_last[k];
    if (w == k) {                                        // here r == rk, id == idk, l == lk: the owner lists itself and its matches
      int at = 0;
      if (lane == 0) list[0] = rk;
      at = 1;
      for (int base = (rk + 1) & ~63; base <= lk; base += 256) {
        unsigned long long m[4];
        hits4(base, m);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if ((m[u] >> lane) & 1ull) list[at + __builtin_popcountll(m[u] & ((1ull << lane) - 1ull))] = base + 64 * u + lane;
          at += __builtin_popcountll(m[u]);
        }
      }
    }
    __syncthreads();
    Row<NCH> a0, a1, a2, a3;
    row_zero(a0); row_zero(a1); row_zero(a2); row_zero(a3);
    int e = w * 4;
    for (; e + 3 < n; e += 16) {
      const int j0 = list[e], j1 = list[e + 1], j2 = list[e + 2], j3 = list[e + 3];
      row_add_f32(a0, rows + (long)j0 * H, H, lane);
      row_add_f32(a1, rows + (long)j1 * H, H, lane);
      row_add_f32(a2, rows + (long)j2 * H, H, lane);
      row_add_f32(a3, rows + (long)j3 * H, H, lane);
    }
    if (e < n) row_add_f32(a0, rows + (long)list[e] * H, H, lane);
    if (e + 1 < n) row_add_f32(a1, rows + (long)list[e + 1] * H, H, lane);
    if (e + 2 < n) row_add_f32(a2, rows + (long)list[e + 2] * H, H, lane);
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) a0.v[i][j] = (a0.v[i][j] + a1.v[i][j]) + (a2.v[i][j] + a3.v[i][j]);
    row_store_f32(a0, part + (long)w * H, H, lane);
    __syncthreads();
    if (w == k) {
      Row<NCH> t, u;
      row_load_f32(t, part, H, lane);
      row_load_f32(u, part + H, H, lane);
#pragma unroll
      for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) t.v[i][j] += u.v[i][j];
      row_load_f32(u, part + 2 * (long)H, H, lane);
      Row<NCH> v;
      row_load_f32(v, part + 3 * (long)H, H, lane);
#pragma unroll
      for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) t.v[i][j] += u.v[i][j] + v.v[i][j];
      row_add_f32(t, table + (long)idk * H, H, lane);
      row_store_f32(t, table + (long)idk * H, H, lane);
    }
    __syncthreads();
  }
}

template <int NCH>
__global__ __launch_bounds__(256) void word_scatter_kernel(const float* __restrict__ rows, const int64_t* __restrict__ ids, int M, int H,
                                                           float* __restrict__ table) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  word_scatter_body<NCH>(rows, ids, M, H, table, blockIdx.x, smem);
}
// the position / type sums and the word-table scatter of the text embedding's backward in ONE launch (independent outputs, both
// read the per-token gradient rows): the few long gather workgroups first, the word workgroups fill in beside them
template <int NCH>
__global__ __launch_bounds__(256) void embed_scatter_kernel(const float* __restrict__ rows, const int* __restrict__ idx, int M, int H,
                                                            float* __restrict__ d_pos, int n0, const int* __restrict__ idx1,
                                                            float* __restrict__ d_type, int n_gather, const int64_t* __restrict__ ids,
                                                            float* __restrict__ d_word, int* __restrict__ w_first, int* __restrict__ w_last) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)blockIdx.x < n_gather) gather_sum_body(rows, idx, M, H, d_pos, n0, idx1, d_type, blockIdx.x, smem);
  else if (w_first) word_scatter_indexed_body<NCH>(rows, ids, M, H, d_word, w_first, w_last, (int)blockIdx.x - n_gather, smem);
  else word_scatter_body<NCH>(rows, ids, M, H, d_word, (int)blockIdx.x - n_gather, smem);
}

// ------------------------------------------------------------------------------ LayerNorm fwd
// One problem of a LayerNorm launch (the pair kernels below carry two: the text and the visual side of a layer pair).
struct LnFwdP {
  const bf16_t* x; const float* gamma; const float* beta; bf16_t* y; float* mean_o; float* rstd_o; int M, H; float eps;
  uint32_t thr; float scale; uint32_t site; uint64_t seed; uint8_t* q_out; const float* q_scale; float* q_amax;
  int x_f32; float* y_f32;      // the fp32 residual stream: x is fp32 [M][H]; also write y as fp32 (CrctLnFwdArgs)
};
template <int NCH>
__device__ __forceinline__ void ln_fwd_body(const LnFwdP& a, const int blk, const int nblk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int M = a.M, H = a.H;
  Row<NCH> g, b;
  row_load_f32(g, a.gamma, H, lane);
  row_load_f32(b, a.beta, H, lane);
  const float qs = a.q_out ? a.q_scale[0] : 0.f;
  float amax = 0.f;
  const long stride = (long)nblk * ROWS_PER_BLOCK;
  for (long row = (long)blk * ROWS_PER_BLOCK + wave; row < M; row += stride) {
    Row<NCH> r;
    if (a.x_f32) row_load_f32(r, reinterpret_cast<const float*>(a.x) + row * H, H, lane);
    else row_load_bf16(r, a.x + row * H, H, lane);
    float mean, rstd;
    row_stats(r, H, lane, a.eps, mean, rstd);
    row_normalize(r, g, b, H, lane, mean, rstd, row, a.thr, a.scale, a.site, a.seed);
    row_store_bf16(r, a.y + row * H, H, lane);
    if (a.y_f32) row_store_f32_nt(r, a.y_f32 + row * H, H, lane);
    if (a.q_out) amax = fmaxf(amax, row_store_fp8(r, a.q_out + row * H, H, lane, qs));     // the fp8 GEMMs' operand (BASELINE configs[4])
    if (lane == 0) { a.mean_o[row] = mean; a.rstd_o[row] = rstd; }
  }
  if (a.q_out && a.q_amax) {
    amax = wave_max(amax);
    if (lane == 0) amax_update(a.q_amax, amax);
  }
}
// Kernel-argument preload (gemm.hip, GEMM_HOT_PARAMS): the LayerNorm kernels of the data streams' chains take what their first instructions
// need as leading scalar arguments (gfx950 preloads the first argument dwords into SGPRs; a struct passed by value is fetched by scalar
// loads after the wave has started).
template <int NCH>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const bf16_t* x, const float* gamma, const float* beta, bf16_t* y, float* mean_o, float* rstd_o, int M, int H,
                                                       float eps, uint32_t thr, const LnFwdP a_in) {
  LnFwdP a = a_in;
  a.x = x; a.gamma = gamma; a.beta = beta; a.y = y; a.mean_o = mean_o; a.rstd_o = rstd_o; a.M = M; a.H = H; a.eps = eps; a.thr = thr;
  ln_fwd_body<NCH>(a, blockIdx.x, gridDim.x);
}
#define LN_FWD_HOT(p) (p).x, (p).gamma, (p).beta, (p).y, (p).mean_o, (p).rstd_o, (p).M, (p).H, (p).eps, (p).thr,

// ------------------------------------------------------------------------------ LayerNorm bwd
// COMBINE: the four waves of a workgroup add their column partials through LDS and store ONE partial row set per
// workgroup ([3][gridDim.x][H], a quarter of the per-wave traffic for the kernel and for the finalize pass); without it
// (rows too long for 64 KB of LDS) every wave stores its own ([3][4 * gridDim.x][H]).
struct LnBwdP {
  const bf16_t* dy; const bf16_t* x; const float* mean; const float* rstd; const float* gamma; bf16_t* dx; bf16_t* dxl;
  float* partials; int M, H; uint32_t post_thr; float post_scale; uint32_t post_site; uint32_t lin_thr; float lin_scale;
  uint32_t lin_site; uint64_t seed;
  uint8_t* q_out; const float* q_scale; float* q_amax;       // optional e5m2 copy of the gradient the producing Linear's dgrad reads (dxl if given, else dx)
};
// X32: x (the saved pre-LayerNorm rows) is fp32 -- the fp32 residual stream (CrctLnBwdArgs.x_f32)
template <int NCH, bool COMBINE, bool X32>
__device__ __forceinline__ void ln_bwd_body(const LnBwdP& a, const int blk, const int nblk) {
  const bf16_t* __restrict__ dy_p = a.dy; const bf16_t* __restrict__ x_p = a.x;
  const float* __restrict__ mean_p = a.mean; const float* __restrict__ rstd_p = a.rstd; const float* __restrict__ gamma = a.gamma;
  bf16_t* __restrict__ dx_p = a.dx; bf16_t* __restrict__ dxl_p = a.dxl; float* __restrict__ partials = a.partials;
  const int M = a.M, H = a.H;
  const uint32_t post_thr = a.post_thr, post_site = a.post_site, lin_thr = a.lin_thr, lin_site = a.lin_site;
  const float post_scale = a.post_scale, lin_scale = a.lin_scale;
  const uint64_t seed = a.seed;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  Row<NCH> adg, adb, adl, g;
  row_zero(adg); row_zero(adb); row_zero(adl);
  row_load_f32(g, gamma, H, lane);
  const float qs = a.q_out ? a.q_scale[0] : 0.f;
  float q_amax = 0.f;
  // M / (4 x grid) is 1.6 rows per wave at the CRCT sizes: the second row's loads are issued before the first row is worked on
  const long stride = (long)nblk * ROWS_PER_BLOCK;
  long row = (long)blk * ROWS_PER_BLOCK + wave;
  RawRow<NCH> dy_n;
  typename std::conditional<X32, RawRowF<NCH>, RawRow<NCH>>::type x_n;
  const float* __restrict__ x32_p = reinterpret_cast<const float*>(a.x);
  float mean_n = 0.f, rstd_n = 0.f;
  if (row < M) {
    row_fetch_bf16(dy_n, dy_p + row * H, H, lane);
    if constexpr (X32) row_fetch_f32(x_n, x32_p + row * H, H, lane);
    else row_fetch_bf16(x_n, x_p + row * H, H, lane);
    mean_n = mean_p[row]; rstd_n = rstd_p[row];
  }
  for (; row < M; row += stride) {
    Row<NCH> dy, x;
    const float mean = mean_n, rstd = rstd_n;
    row_unpack_bf16(dy, dy_n, H, lane);
    if constexpr (X32) row_unpack_f32(x, x_n, H, lane);
    else row_unpack_bf16(x, x_n, H, lane);
    if (row + stride < M) {
      row_fetch_bf16(dy_n, dy_p + (row + stride) * H, H, lane);
      if constexpr (X32) row_fetch_f32(x_n, x32_p + (row + stride) * H, H, lane);
      else row_fetch_bf16(x_n, x_p + (row + stride) * H, H, lane);
      mean_n = mean_p[row + stride]; rstd_n = rstd_p[row + stride];
    }
    row_apply_dropmask(dy, H, lane, row, post_thr, post_scale, post_site, seed);
    row_ln_bwd(dy, x, g, H, lane, mean, rstd, adg, adb);
    row_store_bf16(dy, dx_p + row * H, H, lane);
    if (dxl_p) {
      row_apply_dropmask(dy, H, lane, row, lin_thr, lin_scale, lin_site, seed);
      row_store_bf16(dy, dxl_p + row * H, H, lane);
    }
    if (a.q_out) q_amax = fmaxf(q_amax, row_store_bf8(dy, a.q_out + row * H, H, lane, qs));
    row_acc(adl, dy);
  }
  if (a.q_out && a.q_amax) {
    q_amax = wave_max(q_amax);
    if (lane == 0) amax_update(a.q_amax, q_amax);
  }
  if constexpr (COMBINE) {
    extern __shared__ __attribute__((aligned(16))) float ln_lds[];      // [3][4][H]
    row_store_f32(adg, ln_lds + (0 * ROWS_PER_BLOCK + wave) * H, H, lane);
    row_store_f32(adb, ln_lds + (1 * ROWS_PER_BLOCK + wave) * H, H, lane);
    row_store_f32(adl, ln_lds + (2 * ROWS_PER_BLOCK + wave) * H, H, lane);
    __syncthreads();
    const int H4 = H >> 2;
    for (int i = threadIdx.x; i < 3 * H4; i += 256) {
      const int q = i / H4, c = (i - q * H4) << 2;
      const float* src = ln_lds + (long)q * ROWS_PER_BLOCK * H + c;
      const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + H);
      const float4 d = *reinterpret_cast<const float4*>(src + 2 * H), e = *reinterpret_cast<const float4*>(src + 3 * H);
      *reinterpret_cast<float4*>(partials + ((long)q * nblk + blk) * H + c) =
          make_float4((a.x + b.x) + (d.x + e.x), (a.y + b.y) + (d.y + e.y), (a.z + b.z) + (d.z + e.z), (a.w + b.w) + (d.w + e.w));
    }
  } else {
    const long nr = (long)nblk * ROWS_PER_BLOCK, pr = (long)blk * ROWS_PER_BLOCK + wave;
    row_store_f32(adg, partials + (0 * nr + pr) * H, H, lane);
    row_store_f32(adb, partials + (1 * nr + pr) * H, H, lane);
    row_store_f32(adl, partials + (2 * nr + pr) * H, H, lane);
  }
}

template <int NCH, bool COMBINE, bool X32 = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* dy, const bf16_t* x, const float* mean, const float* rstd, const float* gamma, int M, int H,
                                                       bf16_t* dx, bf16_t* dxl, const LnBwdP a_in) {
  LnBwdP a = a_in;
  a.dy = dy; a.x = x; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.M = M; a.H = H; a.dx = dx; a.dxl = dxl;
  ln_bwd_body<NCH, COMBINE, X32>(a, blockIdx.x, gridDim.x);
}
#define LN_BWD_HOT(p) (p).dy, (p).x, (p).mean, (p).rstd, (p).gamma, (p).M, (p).H, (p).dx, (p).dxl,

// out_q[c] (+)= sum_blk partials[q][blk][c]  for q < Q (NULL outputs skipped); out_q may have a leading
// dimension (ldo) > 1 column group: out index = c*stride_q
struct FinalizeArgs {
  float* out[10];
  int stride[10];
  int Q, nblk, H, accumulate;
  const float* partials;
};
// 32 columns (8 threads x float4) x 32 row-groups per 256-thread block: every thread sums nblk/32 partial rows with
// eight independent 16-byte loads in flight, then the 32 row-group sums of a column quad are added in a fixed order.
__global__ __launch_bounds__(256) void finalize_partials_kernel(const FinalizeArgs a) {
  __shared__ float4 red[32][9];
  const int q = blockIdx.y;
  float* o = a.out[q];
  if (!o) return;
  const int cl = threadIdx.x & 7, rg = threadIdx.x >> 3;
  const int c = blockIdx.x * 32 + cl * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < a.H) {
    const float* p = a.partials + ((long)q * a.nblk) * a.H + c;
    float4 acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = s;
    int b = rg;
    for (; b + 224 < a.nblk; b += 256) {            // eight independent 16-byte loads in flight per thread
      float4 x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const float4*>(p + (long)(b + 32 * u) * a.H);
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc[u].x += x[u].x; acc[u].y += x[u].y; acc[u].z += x[u].z; acc[u].w += x[u].w; }
    }
    for (; b + 96 < a.nblk; b += 128) {
      float4 x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const float4*>(p + (long)(b + 32 * u) * a.H);
#pragma unroll
      for (int u = 0; u < 4; ++u) { acc[u].x += x[u].x; acc[u].y += x[u].y; acc[u].z += x[u].z; acc[u].w += x[u].w; }
    }
    for (; b < a.nblk; b += 32) {
      const float4 x0 = *reinterpret_cast<const float4*>(p + (long)b * a.H);
      acc[0].x += x0.x; acc[0].y += x0.y; acc[0].z += x0.z; acc[0].w += x0.w;
    }
    s.x = ((acc[0].x + acc[1].x) + (acc[2].x + acc[3].x)) + ((acc[4].x + acc[5].x) + (acc[6].x + acc[7].x));
    s.y = ((acc[0].y + acc[1].y) + (acc[2].y + acc[3].y)) + ((acc[4].y + acc[5].y) + (acc[6].y + acc[7].y));
    s.z = ((acc[0].z + acc[1].z) + (acc[2].z + acc[3].z)) + ((acc[4].z + acc[5].z) + (acc[6].z + acc[7].z));
    s.w = ((acc[0].w + acc[1].w) + (acc[2].w + acc[3].w)) + ((acc[4].w + acc[5].w) + (acc[6].w + acc[7].w));
  }
  red[rg][cl] = s;
  __syncthreads();
  if (rg == 0 && c < a.H) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 32; ++k) { const float4 r = red[k][cl]; t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w; }
    const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (c + j < a.H) {
        const long oi = (long)(c + j) * a.stride[q];
        o[oi] = a.accumulate ? o[oi] + tv[j] : tv[j];
      }
    }
  }
}

// ------------------------------------------------------------------------------ column sum
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, long ld, float* __restrict__ partials, int M, int N) {
  // strip blockIdx.y owns rows y, y + nb, y + 2 nb, ...; a thread owns 4 columns (8-byte loads), 4 rows in flight
  const int nb = gridDim.y;
  const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (c >= N) return;
  float s[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int j = 0; j < 4; ++j) s[u][j] = 0.f;
  const bf16_t* p = x + c;
  int r = blockIdx.y;
  for (; r + 3 * nb < M; r += 4 * nb) {
    uint2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const uint2*>(p + (long)(r + u * nb) * ld);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s[u][0] += bf2f((bf16_t)(v[u].x & 0xffff)); s[u][1] += bf2f((bf16_t)(v[u].x >> 16));
      s[u][2] += bf2f((bf16_t)(v[u].y & 0xffff)); s[u][3] += bf2f((bf16_t)(v[u].y >> 16));
    }
  }
  for (; r < M; r += nb) {
    const uint2 v = *reinterpret_cast<const uint2*>(p + (long)r * ld);
    s[0][0] += bf2f((bf16_t)(v.x & 0xffff)); s[0][1] += bf2f((bf16_t)(v.x >> 16));
    s[0][2] += bf2f((bf16_t)(v.y & 0xffff)); s[0][3] += bf2f((bf16_t)(v.y >> 16));
  }
  float4 o;
  o.x = (s[0][0] + s[1][0]) + (s[2][0] + s[3][0]); o.y = (s[0][1] + s[1][1]) + (s[2][1] + s[3][1]);
  o.z = (s[0][2] + s[1][2]) + (s[2][2] + s[3][2]); o.w = (s[0][3] + s[1][3]) + (s[2][3] + s[3][3]);
  *reinterpret_cast<float4*>(partials + (long)blockIdx.y * N + c) = o;
}

// ------------------------------------------------------------------------------ row softmax f32 | bf16 -> bf16
// IN_BF16: the features crossed PCIe as bf16 (half the bytes of the step's only large host -> device copy)
template <bool IN_BF16>
__device__ __forceinline__ float4 softmax_load4(const void* xr, int c) {
  if constexpr (!IN_BF16) return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(xr) + c);
  else {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(xr) + c);
    return make_float4(bf2f((bf16_t)(u.x & 0xffff)), bf2f((bf16_t)(u.x >> 16)), bf2f((bf16_t)(u.y & 0xffff)), bf2f((bf16_t)(u.y >> 16)));
  }
}
template <bool IN_BF16>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const void* __restrict__ x, bf16_t* __restrict__ y, int M, int F) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long row = (long)blockIdx.x * ROWS_PER_BLOCK + wave; row < M; row += (long)gridDim.x * ROWS_PER_BLOCK) {
    const void* xr = IN_BF16 ? (const void*)(reinterpret_cast<const bf16_t*>(x) + row * F) : (const void*)(reinterpret_cast<const float*>(x) + row * F);
    float mx = -INFINITY;
    for (int c = lane * 4; c < F; c += 256) {
      const float4 v = softmax_load4<IN_BF16>(xr, c);
      mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    mx = wave_max(mx);
    float s = 0.f;
    for (int c = lane * 4; c < F; c += 256) {
      const float4 v = softmax_load4<IN_BF16>(xr, c);
      s += expf(v.x - mx) + expf(v.y - mx) + expf(v.z - mx) + expf(v.w - mx);
    }
    const float inv = 1.0f / wave_sum(s);
    for (int c = lane * 4; c < F; c += 256) {
      const float4 v = softmax_load4<IN_BF16>(xr, c);
      uint2 o = make_uint2(pack2bf(expf(v.x - mx) * inv, expf(v.y - mx) * inv), pack2bf(expf(v.z - mx) * inv, expf(v.w - mx) * inv));
      *reinterpret_cast<uint2*>(y + row * F + c) = o;
    }
  }
}

__global__ void cast_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  const long stride = (long)gridDim.x * blockDim.x * 8;
  for (; i + 8 <= n; i += stride) {
    const float4 a = *reinterpret_cast<const float4*>(x + i), b = *reinterpret_cast<const float4*>(x + i + 4);
    uint4 u = make_uint4(pack2bf(a.x, a.y), pack2bf(a.z, a.w), pack2bf(b.x, b.y), pack2bf(b.z, b.w));
    *reinterpret_cast<uint4*>(y + i) = u;
  }
  if (i < n && i + 8 > n)
    for (long j = i; j < n; ++j) y[j] = f2bf(x[j]);
}

// y[off[r] + i] = bf16(x[off[r] + i]) over the runs (off, len) of a chunk table (crct_adamw_plan over the run lengths): the gradients
// backward ACCUMULATES into (biases, LayerNorm, embeddings, heads) on their way into the bf16 exchange buffer, whose other
// elements -- the Linear weight gradients -- the weight-gradient GEMMs have written there themselves
__global__ __launch_bounds__(256) void cast_runs_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, const int64_t* __restrict__ run_off,
                                                        const int64_t* __restrict__ run_len, const int32_t* __restrict__ blk_seg,
                                                        const int64_t* __restrict__ blk_off, int n_blk) {
  for (int blk = blockIdx.x; blk < n_blk; blk += gridDim.x) {
    const int r = blk_seg[blk];
    const int64_t off = blk_off[blk], base = run_off[r] + off;
    int64_t n = run_len[r] - off;
    if (n > 4096) n = 4096;                      // = ADAMW_CHUNK (optim.hip)
    const int64_t head = (8 - (base & 7)) & 7;   // elements in front of the first 16-byte boundary of the bf16 side
    for (int64_t i = threadIdx.x; i < (head < n ? head : n); i += 256) y[base + i] = f2bf(x[base + i]);
    const int64_t nv = n > head ? (n - head) / 8 : 0;
    for (int64_t i = threadIdx.x; i < nv; i += 256) {
      const int64_t e = base + head + 8 * i;
      const float4 a = *reinterpret_cast<const float4*>(x + e), b = *reinterpret_cast<const float4*>(x + e + 4);
      *reinterpret_cast<uint4*>(y + e) = make_uint4(pack2bf(a.x, a.y), pack2bf(a.z, a.w), pack2bf(b.x, b.y), pack2bf(b.z, b.w));
    }
    for (int64_t i = head + nv * 8 + threadIdx.x; i < n; i += 256) y[base + i] = f2bf(x[base + i]);
  }
}

// the reverse over the same kind of table: y[off[r] + i] = float(x[off[r] + i]) (the all-reduced bf16 values of the gradients backward
// accumulates in fp32 -- biases, LayerNorm, embeddings, heads -- put back into the fp32 .grad views)
__global__ __launch_bounds__(256) void uncast_runs_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, const int64_t* __restrict__ run_off,
                                                          const int64_t* __restrict__ run_len, const int32_t* __restrict__ blk_seg,
                                                          const int64_t* __restrict__ blk_off, int n_blk) {
  for (int blk = blockIdx.x; blk < n_blk; blk += gridDim.x) {
    const int r = blk_seg[blk];
    const int64_t off = blk_off[blk], base = run_off[r] + off;
    int64_t n = run_len[r] - off;
    if (n > 4096) n = 4096;                      // = ADAMW_CHUNK (optim.hip)
    const int64_t head = (8 - (base & 7)) & 7;
    for (int64_t i = threadIdx.x; i < (head < n ? head : n); i += 256) y[base + i] = bf2f(x[base + i]);
    const int64_t nv = n > head ? (n - head) / 8 : 0;
    for (int64_t i = threadIdx.x; i < nv; i += 256) {
      const int64_t e = base + head + 8 * i;
      const uint4 u = *reinterpret_cast<const uint4*>(x + e);
      *reinterpret_cast<float4*>(y + e) = make_float4(bf2f((bf16_t)(u.x & 0xffff)), bf2f((bf16_t)(u.x >> 16)), bf2f((bf16_t)(u.y & 0xffff)), bf2f((bf16_t)(u.y >> 16)));
      *reinterpret_cast<float4*>(y + e + 4) = make_float4(bf2f((bf16_t)(u.z & 0xffff)), bf2f((bf16_t)(u.z >> 16)), bf2f((bf16_t)(u.w & 0xffff)), bf2f((bf16_t)(u.w >> 16)));
    }
    for (int64_t i = head + nv * 8 + threadIdx.x; i < n; i += 256) y[base + i] = bf2f(x[base + i]);
  }
}

__global__ void uncast_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, long n) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  const long stride = (long)gridDim.x * blockDim.x * 8;
  for (; i + 8 <= n; i += stride) {
    const uint4 u = *reinterpret_cast<const uint4*>(x + i);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
    float f[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[2 * j] = bf2f((bf16_t)(w[j] & 0xffff)); f[2 * j + 1] = bf2f((bf16_t)(w[j] >> 16)); }
    *reinterpret_cast<float4*>(y + i) = make_float4(f[0], f[1], f[2], f[3]);
    *reinterpret_cast<float4*>(y + i + 4) = make_float4(f[4], f[5], f[6], f[7]);
  }
  if (i < n && i + 8 > n)
    for (long j = i; j < n; ++j) y[j] = bf2f(x[j]);
}

// ------------------------------------------------------------------------------ text embedding
// first question/answer token of a batch row (segments -1 or 1), T if none: vilbert.py:327-332
__device__ __forceinline__ int first_qa_index(const int64_t* segs_row, int T, int lane) {
  int best = T;
  for (int t = lane; t < T; t += 64) {
    const long s = segs_row[t];
    if (s == -1 || s == 1) { best = t; break; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o, 64));
  return best;
}

template <int NCH>
__global__ __launch_bounds__(256) void embed_text_fwd_kernel(
    const int64_t* __restrict__ ids, const int64_t* __restrict__ segs, const float* __restrict__ loc,
    const float* __restrict__ word, const float* __restrict__ pos, const float* __restrict__ type,
    const float* __restrict__ w_loc, const float* __restrict__ b_loc, const float* __restrict__ gamma,
    const float* __restrict__ beta, bf16_t* __restrict__ sum_out, bf16_t* __restrict__ y,
    float* __restrict__ mean_o, float* __restrict__ rstd_o, int B, int T, int H, int n_pos, float eps,
    uint32_t thr, float scale, uint32_t site, uint64_t seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long M = (long)B * T;
  for (long row = (long)blockIdx.x * ROWS_PER_BLOCK + wave; row < M; row += (long)gridDim.x * ROWS_PER_BLOCK) {
    const int b = (int)(row / T), t = (int)(row % T);
    const long seg = segs[row];
    const bool qa = (seg == -1 || seg == 1);
    Row<NCH> r;
    row_load_f32(r, word + ids[row] * (long)H, H, lane);
    if (qa) {
      const int fq = first_qa_index(segs + (long)b * T, T, lane);
      int pid = t - fq;
      pid = pid < 0 ? 0 : (pid >= n_pos ? n_pos - 1 : pid);
      row_add_f32(r, pos + (long)pid * H, H, lane);
    }
    if (seg != 0) row_add_f32(r, type + (seg == -1 ? 0 : seg) * (long)H, H, lane);
    const float4 lv = *reinterpret_cast<const float4*>(loc + row * 4);
    const float l[4] = {lv.x, lv.y, lv.z, lv.w};
    if (fabsf(l[0]) + fabsf(l[1]) + fabsf(l[2]) + fabsf(l[3]) != 0.f) row_add_loc_linear(r, w_loc, b_loc, l, H, lane);
    // the saved pre-norm row is the bf16-rounded one, and the norm is taken over exactly that
    row_store_bf16(r, sum_out + row * H, H, lane);
    row_round_bf16(r);
    float mean, rstd;
    row_stats(r, H, lane, eps, mean, rstd);
    row_normalize(r, gamma, beta, H, lane, mean, rstd, row, thr, scale, site, seed);
    row_store_bf16(r, y + row * H, H, lane);
    if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
  }
}

template <int NCH>
__global__ __launch_bounds__(256) void embed_text_bwd_kernel(
    const bf16_t* __restrict__ dy_p, const bf16_t* __restrict__ sum_p, const float* __restrict__ mean_p,
    const float* __restrict__ rstd_p, const int64_t* __restrict__ ids, const int64_t* __restrict__ segs,
    const float* __restrict__ loc, const float* __restrict__ gamma, float* __restrict__ d_word,
    float* __restrict__ d_pos, float* __restrict__ d_type, float* __restrict__ partials, int B, int T, int H, int n_pos,
    uint32_t thr, float scale, uint32_t site, uint64_t seed, float* __restrict__ rows_scratch, int* __restrict__ idx_scratch,
    int type_partials, int* __restrict__ w_first, int* __restrict__ w_last) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long M = (long)B * T;
  Row<NCH> adg, adb, abl, aw0, aw1, aw2, aw3, at0, at1;
  row_zero(adg); row_zero(adb); row_zero(abl); row_zero(aw0); row_zero(aw1); row_zero(aw2); row_zero(aw3);
  row_zero(at0); row_zero(at1);
  for (long row = (long)blockIdx.x * ROWS_PER_BLOCK + wave; row < M; row += (long)gridDim.x * ROWS_PER_BLOCK) {
    const int b = (int)(row / T), t = (int)(row % T);
    const long seg = segs[row];
    const bool qa = (seg == -1 || seg == 1);
    Row<NCH> dy, x;
    row_load_bf16(dy, dy_p + row * H, H, lane);
    row_load_bf16(x, sum_p + row * H, H, lane);
    row_apply_dropmask(dy, H, lane, row, thr, scale, site, seed);
    row_ln_bwd(dy, x, gamma, H, lane, mean_p[row], rstd_p[row], adg, adb);
    if (!rows_scratch) row_atomic_add(dy, d_word + ids[row] * (long)H, H, lane);     // fall-back only: see word_scatter_kernel
    int pid = -1;
    if (qa) {
      const int fq = first_qa_index(segs + (long)b * T, T, lane);
      pid = t - fq;
      pid = pid < 0 ? 0 : (pid >= n_pos ? n_pos - 1 : pid);
    }
    int tyid = seg != 0 ? (int)(seg == -1 ? 0 : seg) : -1;
    if (type_partials && (tyid == 0 || tyid == 1)) {   // the two question / answer token types hold most tokens: their sums
      if (tyid == 0) row_acc(at0, dy);                 // ride along as partial rows 7 and 8 (one gather workgroup per type
      else row_acc(at1, dy);                           // row took 60-110 us at the tail of the step); other types: gather
      tyid = -1;
    }
    if (rows_scratch) {                                                 // position / type sums: gather_sum_kernel afterwards
      row_store_f32(dy, rows_scratch + row * H, H, lane);
      if (lane == 0) {
        idx_scratch[row] = pid; idx_scratch[M + row] = tyid;
        if (w_first) {                                   // the word scatter's index (word_scatter_indexed_body): first / last row of every id
          const long id = ids[row];
          atomicMax(w_first + id, (int)(M - row));
          atomicMax(w_last + id, (int)row + 1);
        }
      }
    } else {
      if (pid >= 0) row_atomic_add(dy, d_pos + (long)pid * H, H, lane);
      if (tyid >= 0) row_atomic_add(dy, d_type + (long)tyid * H, H, lane);
    }
    const float4 lv = *reinterpret_cast<const float4*>(loc + row * 4);
    if (fabsf(lv.x) + fabsf(lv.y) + fabsf(lv.z) + fabsf(lv.w) != 0.f) {
      row_acc(abl, dy);
      row_acc(aw0, dy, lv.x); row_acc(aw1, dy, lv.y); row_acc(aw2, dy, lv.z); row_acc(aw3, dy, lv.w);
    }
  }
  // every WAVE stores its own partial rows ([7 or 9][4 * gridDim.x][H]; the finalize pass sums them): no LDS tree and no
  // workgroup barrier in this kernel.  (The wrong lanes 48..63 once seen here beside the dgrad / wgrad GEMMs came from
  // packed-fp32 instructions, not from LDS: DESIGN.md section 8, tools/embed_stress.py; the build bans them.)
  const long nr = (long)gridDim.x * ROWS_PER_BLOCK, pr = (long)blockIdx.x * ROWS_PER_BLOCK + wave;
  row_store_f32(adg, partials + (0 * nr + pr) * H, H, lane);
  row_store_f32(adb, partials + (1 * nr + pr) * H, H, lane);
  row_store_f32(abl, partials + (2 * nr + pr) * H, H, lane);
  row_store_f32(aw0, partials + (3 * nr + pr) * H, H, lane);
  row_store_f32(aw1, partials + (4 * nr + pr) * H, H, lane);
  row_store_f32(aw2, partials + (5 * nr + pr) * H, H, lane);
  row_store_f32(aw3, partials + (6 * nr + pr) * H, H, lane);
  if (type_partials) {
    row_store_f32(at0, partials + (7 * nr + pr) * H, H, lane);
    row_store_f32(at1, partials + (8 * nr + pr) * H, H, lane);
  }
}

// ------------------------------------------------------------------------------ image embedding
template <int NCH>
__global__ __launch_bounds__(256) void embed_image_fwd_kernel(
    const bf16_t* __restrict__ img, const float* __restrict__ loc, const int64_t* __restrict__ target,
    const float* __restrict__ w_loc, const float* __restrict__ b_loc, const float* __restrict__ color,
    const float* __restrict__ gamma, const float* __restrict__ beta, bf16_t* __restrict__ sum_out,
    bf16_t* __restrict__ y, float* __restrict__ mean_o, float* __restrict__ rstd_o, int M, int H, float eps,
    uint32_t thr, float scale, uint32_t site, uint64_t seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long row = (long)blockIdx.x * ROWS_PER_BLOCK + wave; row < M; row += (long)gridDim.x * ROWS_PER_BLOCK) {
    Row<NCH> r;
    row_load_bf16(r, img + row * H, H, lane);
    const float4 lv = *reinterpret_cast<const float4*>(loc + row * 4);
    const float l[4] = {lv.x, lv.y, lv.z, lv.w};
    row_add_loc_linear(r, w_loc, b_loc, l, H, lane);
    row_add_f32(r, color + target[row] * (long)H, H, lane);
    row_store_bf16(r, sum_out + row * H, H, lane);
    row_round_bf16(r);
    float mean, rstd;
    row_stats(r, H, lane, eps, mean, rstd);
    row_normalize(r, gamma, beta, H, lane, mean, rstd, row, thr, scale, site, seed);
    row_store_bf16(r, y + row * H, H, lane);
    if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
  }
}

template <int NCH>
__global__ __launch_bounds__(256) void embed_image_bwd_kernel(
    const bf16_t* __restrict__ dy_p, const bf16_t* __restrict__ sum_p, const float* __restrict__ mean_p,
    const float* __restrict__ rstd_p, const float* __restrict__ loc, const int64_t* __restrict__ target,
    const float* __restrict__ gamma, bf16_t* __restrict__ dsum_p, float* __restrict__ d_color,
    float* __restrict__ partials, int M, int H, uint32_t thr, float scale, uint32_t site, uint64_t seed,
    float* __restrict__ rows_scratch, int* __restrict__ idx_scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  Row<NCH> adg, adb, abl, aw0, aw1, aw2, aw3;
  row_zero(adg); row_zero(adb); row_zero(abl); row_zero(aw0); row_zero(aw1); row_zero(aw2); row_zero(aw3);
  for (long row = (long)blockIdx.x * ROWS_PER_BLOCK + wave; row < M; row += (long)gridDim.x * ROWS_PER_BLOCK) {
    Row<NCH> dy, x;
    row_load_bf16(dy, dy_p + row * H, H, lane);
    row_load_bf16(x, sum_p + row * H, H, lane);
    row_apply_dropmask(dy, H, lane, row, thr, scale, site, seed);
    row_ln_bwd(dy, x, gamma, H, lane, mean_p[row], rstd_p[row], adg, adb);
    row_store_bf16(dy, dsum_p + row * H, H, lane);
    if (rows_scratch) {                                                 // colour sums: gather_sum_kernel afterwards
      row_store_f32(dy, rows_scratch + row * H, H, lane);
      if (lane == 0) idx_scratch[row] = (int)target[row];
    } else {
      row_atomic_add(dy, d_color + target[row] * (long)H, H, lane);
    }
    const float4 lv = *reinterpret_cast<const float4*>(loc + row * 4);
    row_acc(abl, dy);       // = d b_loc = d b_img (both are plain column sums of d_sum)
    row_acc(aw0, dy, lv.x); row_acc(aw1, dy, lv.y); row_acc(aw2, dy, lv.z); row_acc(aw3, dy, lv.w);
  }
  // every WAVE stores its own partial rows ([7][4 * gridDim.x][H]; the finalize pass sums them): no LDS tree and no
  // workgroup barrier in this kernel.  (The wrong lanes 48..63 once seen here beside the dgrad / wgrad GEMMs came from
  // packed-fp32 instructions, not from LDS: DESIGN.md section 8, tools/embed_stress.py; the build bans them.)
  const long nr = (long)gridDim.x * ROWS_PER_BLOCK, pr = (long)blockIdx.x * ROWS_PER_BLOCK + wave;
  row_store_f32(adg, partials + (0 * nr + pr) * H, H, lane);
  row_store_f32(adb, partials + (1 * nr + pr) * H, H, lane);
  row_store_f32(abl, partials + (2 * nr + pr) * H, H, lane);
  row_store_f32(aw0, partials + (3 * nr + pr) * H, H, lane);
  row_store_f32(aw1, partials + (4 * nr + pr) * H, H, lane);
  row_store_f32(aw2, partials + (5 * nr + pr) * H, H, lane);
  row_store_f32(aw3, partials + (6 * nr + pr) * H, H, lane);
}

inline int nch_for(int H) { return (H / 8 + 63) / 64; }
inline int row_grid(long M, int cap) {
  long g = (M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

#define DISPATCH_NCH(H, ...)                                           \
  switch (nch_for(H)) {                                                \
    case 1: { constexpr int NCH = 1; __VA_ARGS__; } break;             \
    case 2: { constexpr int NCH = 2; __VA_ARGS__; } break;             \
    case 3: { constexpr int NCH = 3; __VA_ARGS__; } break;             \
    case 4: { constexpr int NCH = 4; __VA_ARGS__; } break;             \
    default: crct_set_error("row width %d > 2048 unsupported", H); return 2; \
  }

int launch_finalize(const FinalizeArgs& fa, hipStream_t s) {
  crct_launch(finalize_partials_kernel, dim3((fa.H + 31) / 32, fa.Q), dim3(256), 0, s, fa);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

}  // namespace

extern "C" {

static LnFwdP ln_fwd_problem(const CrctLnFwdArgs& a) {
  return LnFwdP{(const bf16_t*)a.x, a.gamma, a.beta, (bf16_t*)a.y, a.mean, a.rstd, a.M, a.H, a.eps, a.drop_thr, a.drop_scale,
                a.drop_site, a.seed, (uint8_t*)a.q_out, a.q_scale, a.q_amax, a.x_f32, a.y_f32};
}
static int ln_fwd_check(const CrctLnFwdArgs& a) {
  CRCT_REQUIRE(a.H % 8 == 0 && a.H > 0, "layernorm: H=%d must be a positive multiple of 8", a.H);
  CRCT_REQUIRE(!a.q_out || a.q_scale, "layernorm_fwd: q_out needs q_scale");
  return 0;
}
static int ln_fwd_launch(const CrctLnFwdArgs& a, hipStream_t s) {
  if (int r = ln_fwd_check(a)) return r;
  if (a.M <= 0) return 0;
  const LnFwdP p = ln_fwd_problem(a);
  DISPATCH_NCH(a.H, crct_launch((ln_fwd_kernel<NCH>), dim3(row_grid(a.M, 2048)), dim3(256), 0, s, LN_FWD_HOT(p) p));
  CRCT_CHECK_HIP(hipGetLastError());
  return 0;
}

int crct_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                       int M, int H, float eps, uint32_t drop_thr, float drop_scale, uint32_t drop_site,
                       uint64_t seed, crct_stream_t stream) {
  const CrctLnFwdArgs a = {x, gamma, beta, y, mean, rstd, M, H, eps, drop_thr, drop_scale, drop_site, seed, nullptr, nullptr, nullptr, 0, nullptr};
  return ln_fwd_launch(a, (hipStream_t)stream);
}
int crct_layernorm_fwd_args(const CrctLnFwdArgs* a, crct_stream_t stream) {
  CRCT_REQUIRE(a, "layernorm_fwd_args: null argument");
  return ln_fwd_launch(*a, (hipStream_t)stream);
}

int crct_layernorm_fwd_q(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                         int M, int H, float eps, uint32_t drop_thr, float drop_scale, uint32_t drop_site,
                         uint64_t seed, void* q_out, const float* q_scale, float* q_amax, crct_stream_t stream) {
  CRCT_REQUIRE(q_out && q_scale, "layernorm_fwd_q: q_out and q_scale are required");
  const CrctLnFwdArgs a = {x, gamma, beta, y, mean, rstd, M, H, eps, drop_thr, drop_scale, drop_site, seed, q_out, q_scale, q_amax, 0, nullptr};
  return ln_fwd_launch(a, (hipStream_t)stream);
}

// the embedding backward kernels write 7 partial rows per wave: fewer, longer-running waves than the LayerNorm backward
static int embed_bwd_blocks(long M) {
  return row_grid(M, CRCT_LN_BWD_MAX_BLOCKS);
}
int crct_layernorm_bwd_blocks(int M) {
  return row_grid(M, CRCT_LN_BWD_MAX_BLOCKS);
}

// rows pass only: dx / dx_lin and the per-workgroup column partials [3][nblk][H]
// the LayerNorm backward combines its waves' partial rows in LDS when [3][4][H] fp32 fit the default 64 KB
static bool ln_bwd_combines(int H) {
  return (size_t)3 * ROWS_PER_BLOCK * H * 4 <= 64 * 1024;
}
static LnBwdP ln_bwd_problem(const CrctLnBwdArgs& a) {
  return LnBwdP{(const bf16_t*)a.dy, (const bf16_t*)a.x, a.mean, a.rstd, a.gamma, (bf16_t*)a.dx, (bf16_t*)a.dx_lin, a.partials,
                a.M, a.H, a.post_thr, a.post_scale, a.post_site, a.lin_thr, a.lin_scale, a.lin_site, a.seed,
                (uint8_t*)a.q_out, a.q_scale, a.q_amax};
}
static int ln_bwd_launch(const CrctLnBwdArgs& a, hipStream_t s) {
  CRCT_REQUIRE(a.H % 8 == 0 && a.H > 0, "layernorm_bwd: H=%d must be a positive multiple of 8", a.H);
  if (a.M <= 0) return 0;
  const int nb = crct_layernorm_bwd_blocks(a.M);
  const LnBwdP p = ln_bwd_problem(a);
  if (a.x_f32) {
    if (ln_bwd_combines(a.H)) {
      DISPATCH_NCH(a.H, crct_launch((ln_bwd_kernel<NCH, true, true>), dim3(nb), dim3(256), (size_t)3 * ROWS_PER_BLOCK * a.H * 4, s, LN_BWD_HOT(p) p));
    } else {
      DISPATCH_NCH(a.H, crct_launch((ln_bwd_kernel<NCH, false, true>), dim3(nb), dim3(256), 0, s, LN_BWD_HOT(p) p));
    }
  } else if (ln_bwd_combines(a.H)) {
    DISPATCH_NCH(a.H, crct_launch((ln_bwd_kernel<NCH, true>), dim3(nb), dim3(256), (size_t)3 * ROWS_PER_BLOCK * a.H * 4, s, LN_BWD_HOT(p) p));
  } else {
    DISPATCH_NCH(a.H, crct_launch((ln_bwd_kernel<NCH, false>), dim3(nb), dim3(256), 0, s, LN_BWD_HOT(p) p));
  }
  CRCT_CHECK_HIP(hipGetLastError());
  return 0;
}
int crct_layernorm_bwd_rows(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                            void* dx, void* dx_lin, float* partials, int M, int H, uint32_t post_thr, float post_scale,
                            uint32_t post_site, uint32_t lin_thr, float lin_scale, uint32_t lin_site, uint64_t seed,
                            crct_stream_t stream) {
  const CrctLnBwdArgs a = {dy, x, mean, rstd, gamma, dx, dx_lin, partials, M, H, post_thr, post_scale, post_site, lin_thr, lin_scale,
                           lin_site, seed, nullptr, nullptr, nullptr, 0};
  return ln_bwd_launch(a, (hipStream_t)stream);
}
int crct_layernorm_bwd_rows_args(const CrctLnBwdArgs* a, crct_stream_t stream) {
  CRCT_REQUIRE(a && (!a->q_out || a->q_scale), "layernorm_bwd_rows_args: bad arguments");
  return ln_bwd_launch(*a, (hipStream_t)stream);
}
// column pass: dgamma / dbeta / dbias_lin (+)= sum over the workgroup partials; may run on another stream
int crct_layernorm_bwd_finalize(const float* partials, float* dgamma, float* dbeta, float* dbias_lin, int M, int H,
                                int accumulate, crct_stream_t stream) {
  if (M <= 0) return 0;
  FinalizeArgs fa = {};
  fa.out[0] = dgamma; fa.out[1] = dbeta; fa.out[2] = dbias_lin;
  fa.stride[0] = fa.stride[1] = fa.stride[2] = 1;
  fa.Q = 3; fa.nblk = crct_layernorm_bwd_blocks(M) * (ln_bwd_combines(H) ? 1 : ROWS_PER_BLOCK);
  fa.H = H; fa.accumulate = accumulate; fa.partials = partials;
  return launch_finalize(fa, (hipStream_t)stream);
}

int crct_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                       void* dx, void* dx_lin, float* dgamma, float* dbeta, float* dbias_lin, float* partials,
                       int M, int H, int accumulate, uint32_t post_thr, float post_scale, uint32_t post_site,
                       uint32_t lin_thr, float lin_scale, uint32_t lin_site, uint64_t seed, crct_stream_t stream) {
  if (int r = crct_layernorm_bwd_rows(dy, x, mean, rstd, gamma, dx, dx_lin, partials, M, H, post_thr, post_scale, post_site,
                                      lin_thr, lin_scale, lin_site, seed, stream)) return r;
  return crct_layernorm_bwd_finalize(partials, dgamma, dbeta, dbias_lin, M, H, accumulate, stream);
}

int crct_colsum_blocks(int M) { int b = (M + 7) / 8; return b < 1 ? 1 : (b > 64 ? 64 : b); }

int crct_colsum_bf16(const void* x, int64_t ld, float* out, float* partials, int M, int N, int accumulate,
                     crct_stream_t stream) {
  CRCT_REQUIRE(N % 4 == 0 && ld % 4 == 0, "colsum: N=%d and ld must be multiples of 4", N);
  if (N <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int nb = crct_colsum_blocks(M);
  crct_launch(colsum_kernel, dim3((N / 4 + 255) / 256, nb), dim3(256), 0, s, (const bf16_t*)x, (long)ld, partials, M, N);
  CRCT_CHECK_HIP(hipGetLastError());
  FinalizeArgs fa = {};
  fa.out[0] = out; fa.stride[0] = 1; fa.Q = 1; fa.nblk = nb; fa.H = N; fa.accumulate = accumulate; fa.partials = partials;
  return launch_finalize(fa, s);
}

int crct_softmax_rows_f32_bf16(const float* x, void* y, int M, int F, crct_stream_t stream) {
  CRCT_REQUIRE(F % 4 == 0, "softmax_rows: F=%d must be a multiple of 4", F);
  if (M <= 0) return 0;
  crct_launch(softmax_rows_kernel<false>, dim3(row_grid(M, 4096)), dim3(256), 0, (hipStream_t)stream, (const void*)x, (bf16_t*)y, M, F);
  CRCT_CHECK_HIP(hipGetLastError());
  return 0;
}

int crct_softmax_rows_bf16_bf16(const void* x, void* y, int M, int F, crct_stream_t stream) {
  CRCT_REQUIRE(F % 4 == 0, "softmax_rows: F=%d must be a multiple of 4", F);
  if (M <= 0) return 0;
  crct_launch(softmax_rows_kernel<true>, dim3(row_grid(M, 4096)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, M, F);
  CRCT_CHECK_HIP(hipGetLastError());
  return 0;
}

// Key masks of the step from what the data loader ships (encoder_decorator.py:118-120, vilbert.py:1380-1396):
//   text key t of row b is attended iff t < sep_indices[b][hist_len[b]] + 1      (sequence_mask of the step adapter)
//   visual key v of row b is attended iff image_mask[b][v] != 0
// One launch instead of the gather / add / arange / compare / cast kernels the torch expression costs.
__global__ __launch_bounds__(256) void build_keymasks_kernel(const int64_t* __restrict__ sep_indices, const int64_t* __restrict__ hist_len,
                                                             int sep_stride, const int64_t* __restrict__ image_mask,
                                                             uint8_t* __restrict__ km_t, uint8_t* __restrict__ km_v, int B, int T, int V) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (km_t && i < B * T) {
    const int b = i / T, t = i % T;
    long h = hist_len[b];
    h = h < 0 ? 0 : (h >= sep_stride ? sep_stride - 1 : h);
    km_t[i] = (long)t < sep_indices[(long)b * sep_stride + h] + 1 ? 1 : 0;
  }
  if (km_v && i < B * V) km_v[i] = image_mask[i] != 0 ? 1 : 0;
}

extern "C" int crct_build_keymasks(const int64_t* sep_indices, const int64_t* hist_len, int sep_stride, const int64_t* image_mask,
                                   uint8_t* km_t, uint8_t* km_v, int B, int T, int V, crct_stream_t stream) {
  CRCT_REQUIRE(B > 0 && T > 0 && V > 0, "build_keymasks: bad sizes");
  CRCT_REQUIRE(!km_t || (sep_indices && hist_len && sep_stride > 0), "build_keymasks: sep_indices / hist_len are required for the text mask");
  CRCT_REQUIRE(!km_v || image_mask, "build_keymasks: image_mask is required for the visual mask");
  if (!km_t && !km_v) return 0;
  const int n = B * (T > V ? T : V);
  crct_launch(build_keymasks_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, sep_indices, hist_len, sep_stride,
                     image_mask, km_t, km_v, B, T, V);
  CRCT_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"

namespace {
// ---------------------------------------------------------------------------------- fp8 (OCP e4m3) quantisation passes
// bf16 tensor -> e4m3 with a given scale + amax (the embedding outputs, which no LayerNorm launch of ours re-reads)
__global__ __launch_bounds__(256) void fp8_quantize_bf16_kernel(const bf16_t* __restrict__ x, uint8_t* __restrict__ q,
                                                                const float* __restrict__ scale, float* __restrict__ amax_out, long n) {
  const float qs = scale[0];
  float amax = 0.f;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8; i + 8 <= n; i += (long)gridDim.x * 256 * 8) {
    const uint4 u = *reinterpret_cast<const uint4*>(x + i);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
    float v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[2 * j] = bf2f((bf16_t)(w[j] & 0xffff)); v[2 * j + 1] = bf2f((bf16_t)(w[j] >> 16)); }
#pragma unroll
    for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[j]));
    *reinterpret_cast<uint2*>(q + i) = pack8_fp8(v, qs);
  }
  if (amax_out) {
    amax = wave_max(amax);
    if ((threadIdx.x & 63) == 0) amax_update(amax_out, amax);
  }
}
// fp32 weights of the tensors listed by (seg_off, seg_len, seg_slot) over the chunk table of crct_adamw_plan:
// MODE 0: amax[slot] = max |w|;  MODE 1: q = e4m3(w * scale[slot]) at the same flat element offset
template <int MODE>
__global__ __launch_bounds__(256) void fp8_weights_kernel(const float* __restrict__ p, uint8_t* __restrict__ q,
                                                          const int64_t* __restrict__ seg_off, const int64_t* __restrict__ seg_len,
                                                          const int32_t* __restrict__ seg_slot, const int32_t* __restrict__ blk_seg,
                                                          const int64_t* __restrict__ blk_off, const float* __restrict__ scale,
                                                          float* __restrict__ amax, int n_blk) {
  for (int blk = blockIdx.x; blk < n_blk; blk += gridDim.x) {
    const int sgi = blk_seg[blk], slot = seg_slot[sgi];
    if (slot < 0) continue;
    const int64_t off = blk_off[blk], base = seg_off[sgi] + off;
    int64_t n = seg_len[sgi] - off;
    if (n > 4096) n = 4096;
    const float qs = MODE == 1 ? scale[slot] : 0.f;
    float am = 0.f;
    for (int64_t i = (int64_t)threadIdx.x * 4; i + 4 <= n; i += 1024) {
      const float4 v = *reinterpret_cast<const float4*>(p + base + i);
      if (MODE == 0) am = fmaxf(fmaxf(am, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
      else {
        uint32_t w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(f8_clamp(v.x * qs), f8_clamp(v.y * qs), w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(f8_clamp(v.z * qs), f8_clamp(v.w * qs), w, true);
        *reinterpret_cast<uint32_t*>(q + base + i) = w;
      }
    }
    if (MODE == 0) {
      am = wave_max(am);
      if ((threadIdx.x & 63) == 0) amax_update(amax + (long)slot * CRCT_FP8_AMAX_LANES, am);
    }
  }
}
// Transposed e4m3 weight shadow for the fp8 DATA-GRADIENT GEMMs: dx = dy W contracts over W's rows, and the fp8 kernel wants the
// contraction index contiguous in both operands, so every shadowed weight [out][in] is also kept as [in][out].  One workgroup per
// 64 x 64 byte tile through LDS (16-byte global accesses on both sides); run once per optimizer step behind the update.
__global__ __launch_bounds__(256) void fp8_transpose_kernel(const uint8_t* __restrict__ q, uint8_t* __restrict__ qt,
                                                            const int64_t* __restrict__ w_off, const int32_t* __restrict__ w_out,
                                                            const int32_t* __restrict__ w_in, const int64_t* __restrict__ tile_begin, int n_w,
                                                            long n_tiles) {
  __shared__ uint8_t t[64][80];
  // persistent grid: behind an overlapped AdamW the launch is capped (max_workgroups) so that it does not take the CUs from the
  // forward pass that is starting beside it -- uncapped, its 51k workgroups cost the step 0.33 ms (EXPERIMENTS.md, round 3)
  for (long bid = blockIdx.x; bid < n_tiles; bid += gridDim.x) {
  int lo = 0, hi = n_w - 1;                        // the weight this tile belongs to: last i with tile_begin[i] <= bid
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tile_begin[mid] <= bid) lo = mid; else hi = mid - 1;
  }
  const int wi = lo;
  const int out = w_out[wi], in = w_in[wi];
  const int tiles_in = (in + 63) / 64;
  const long tl = bid - tile_begin[wi];
  const int r0 = (int)(tl / tiles_in) * 64, c0 = (int)(tl % tiles_in) * 64;       // rows = out index, columns = in index
  const uint8_t* src = q + w_off[wi];
  uint8_t* dst = qt + w_off[wi];
  {
    const int r = threadIdx.x >> 2, ch = (threadIdx.x & 3) * 16;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r0 + r < out && c0 + ch < in) v = *reinterpret_cast<const uint4*>(src + (long)(r0 + r) * in + c0 + ch);
    *reinterpret_cast<uint4*>(&t[r][ch]) = v;
  }
  __syncthreads();
  {
    const int c = threadIdx.x >> 2, ch = (threadIdx.x & 3) * 16;       // output row = in index c0 + c, 16 consecutive out indices
    if (c0 + c < in && r0 + ch < out) {
      uint32_t w[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        w[k] = (uint32_t)t[ch + 4 * k][c] | ((uint32_t)t[ch + 4 * k + 1][c] << 8) | ((uint32_t)t[ch + 4 * k + 2][c] << 16) | ((uint32_t)t[ch + 4 * k + 3][c] << 24);
      *reinterpret_cast<uint4*>(dst + (long)(c0 + c) * out + r0 + ch) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
  __syncthreads();
  }
}

// delayed scaling: scale[i] = 448 / max(amax[i][0 .. LANES)) for the entries that saw data.  The maxima are RUNNING maxima
// (reset != 0 clears them: the caller does that every few hundred steps, the "max over a history window" of the usual fp8
// recipes): a wave only issues an atomic when it raises the word it reports into, so after the first steps of a window the
// kernels issue none at all -- resetting every step cost 2.4 ms per step in atomic storms (round-2 measurement).
__global__ void fp8_update_scales_kernel(float* __restrict__ scale, float* __restrict__ amax, int n, int reset, const float* __restrict__ skip_if, float fmax) {
  if (skip_if && skip_if[0] != 0.f) return;      // a skipped optimizer step (GradScaler found inf / nan) leaves the shadow, hence its scales, alone
  const int i = blockIdx.x * (blockDim.x / CRCT_FP8_AMAX_LANES) + threadIdx.x / CRCT_FP8_AMAX_LANES, l = threadIdx.x % CRCT_FP8_AMAX_LANES;
  if (i < n) {          // one wave (64 lanes = LANES) per entry
    float* w = amax + (long)i * CRCT_FP8_AMAX_LANES + l;
    const float a = wave_max(*w);
    if (reset) *w = 0.f;
    if (l == 0 && a > 0.f) scale[i] = fmax / a;
  }
}

}  // namespace

extern "C" {

int crct_fp8_quantize_bf16(const void* x, void* q, const float* scale, float* amax, int64_t n, crct_stream_t stream) {
  CRCT_REQUIRE(x && q && scale && n % 8 == 0, "fp8_quantize_bf16: bad arguments");
  if (n <= 0) return 0;
  long blocks = (n / 8 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  crct_launch(fp8_quantize_bf16_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (uint8_t*)q, scale, amax, (long)n);
  CRCT_CHECK_HIP(hipGetLastError());
  return 0;
}
int crct_fp8_update_scales(float* scale, float* amax, int n, int reset, const float* skip_if, float fmax, crct_stream_t stream) {
  CRCT_REQUIRE(scale && amax && n >= 0, "fp8_update_scales: bad arguments");
  if (n == 0) return 0;
  crct_launch(fp8_update_scales_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, scale, amax, n, reset, skip_if,
                     fmax > 0.f ? fmax : 448.0f);
  CRCT_CHECK_HIP(hipGetLastError());
  return 0;
}
int crct_fp8_quantize_weights(const float* p, void* q, const int64_t* seg_off, const int64_t* seg_len, const int32_t* seg_slot,
                              const int32_t* blk_seg, const int64_t* blk_off, int64_t n_blk, float* scale, float* amax, int n_slots,
                              crct_stream_t stream) {
  CRCT_REQUIRE(p && q && seg_off && seg_len && seg_slot && blk_seg && blk_off && scale && amax, "fp8_quantize_weights: null argument");
  if (n_blk <= 0 || n_slots <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int grid = n_blk > 2048 ? 2048 : (int)n_blk;
  CRCT_CHECK_HIP(hipMemsetAsync(amax, 0, (size_t)n_slots * CRCT_FP8_AMAX_LANES * 4, s));
  crct_launch(fp8_weights_kernel<0>, dim3(grid), dim3(256), 0, s, p, (uint8_t*)q, seg_off, seg_len, seg_slot, blk_seg, blk_off,
                     (const float*)scale, amax, (int)n_blk);
  crct_launch(fp8_update_scales_kernel, dim3((n_slots + 3) / 4), dim3(256), 0, s, scale, amax, n_slots, 0, (const float*)nullptr, 448.0f);
  crct_launch(fp8_weights_kernel<1>, dim3(grid), dim3(256), 0, s, p, (uint8_t*)q, seg_off, seg_len, seg_slot, blk_seg, blk_off,
                     (const float*)scale, amax, (int)n_blk);
  CRCT_CHECK_HIP(hipGetLastError());
  return 0;
}

int crct_fp8_transpose_weights(const void* q, void* qt, const int64_t* w_off, const int32_t* w_out, const int32_t* w_in,
                               const int64_t* tile_begin, int n_w, int64_t n_tiles, int max_workgroups, crct_stream_t stream) {
  CRCT_REQUIRE(q && qt && w_off && w_out && w_in && tile_begin, "fp8_transpose_weights: null argument");
  if (n_w <= 0 || n_tiles <= 0) return 0;
  long grid = n_tiles < 1048576 ? n_tiles : 1048576;
  if (max_workgroups > 0 && grid > max_workgroups) grid = max_workgroups;
  crct_launch(fp8_transpose_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)q, (uint8_t*)qt,
                     w_off, w_out, w_in, tile_begin, n_w, (long)n_tiles);
  CRCT_CHECK_HIP(hipGetLastError());
  return 0;
}

int crct_cast_f32_bf16(const float* x, void* y, int64_t n, crct_stream_t stream) {
  if (n <= 0) return 0;
  long blocks = (n / 8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  crct_launch(cast_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, (long)n);
  CRCT_CHECK_HIP(hipGetLastError());
  return 0;
}

int crct_cast_runs_f32_bf16(const float* x, void* y, const int64_t* off, const int64_t* len, const int32_t* blk_seg, const int64_t* blk_off,
                            int64_t n_blk, crct_stream_t stream) {
  CRCT_REQUIRE(x && y && off && len && blk_seg && blk_off, "cast_runs: null argument");
  CRCT_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "cast_runs: 16-byte aligned buffers");
  if (n_blk <= 0) return 0;
  const long grid = n_blk > 2048 ? 2048 : n_blk;
  crct_launch(cast_runs_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, off, len, blk_seg, blk_off, (int)n_blk);
  CRCT_CHECK_HIP(hipGetLastError());
  return 0;
}

int crct_cast_runs_bf16_f32(const void* x, float* y, const int64_t* off, const int64_t* len, const int32_t* blk_seg, const int64_t* blk_off,
                            int64_t n_blk, crct_stream_t stream) {
  CRCT_REQUIRE(x && y && off && len && blk_seg && blk_off, "cast_runs: null argument");
  CRCT_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "cast_runs: 16-byte aligned buffers");
  if (n_blk <= 0) return 0;
  const long grid = n_blk > 2048 ? 2048 : n_blk;
  crct_launch(uncast_runs_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, y, off, len, blk_seg, blk_off, (int)n_blk);
  CRCT_CHECK_HIP(hipGetLastError());
  return 0;
}

int crct_cast_bf16_f32(const void* x, float* y, int64_t n, crct_stream_t stream) {
  if (n <= 0) return 0;
  CRCT_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "cast_bf16_f32: 16-byte aligned buffers");
  long blocks = (n / 8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  crct_launch(uncast_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, y, (long)n);
  CRCT_CHECK_HIP(hipGetLastError());
  return 0;
}

int crct_embed_text_fwd(const int64_t* ids, const int64_t* segs, const float* loc, const float* word, const float* pos,
                        const float* type, const float* w_loc, const float* b_loc, const float* gamma,
                        const float* beta, void* sum_out, void* y, float* mean, float* rstd, int B, int T, int H,
                        int n_pos, float eps, uint32_t drop_thr, float drop_scale, uint32_t drop_site, uint64_t seed,
                        crct_stream_t stream) {
  CRCT_REQUIRE(H % 8 == 0 && H > 0, "embed_text: H=%d must be a positive multiple of 8", H);
  if ((long)B * T <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_NCH(H, crct_launch((embed_text_fwd_kernel<NCH>), dim3(row_grid((long)B * T, 2048)), dim3(256), 0, s, ids,
                                     segs, loc, word, pos, type, w_loc, b_loc, gamma, beta, (bf16_t*)sum_out, (bf16_t*)y,
                                     mean, rstd, B, T, H, n_pos, eps, drop_thr, drop_scale, drop_site, seed));
  CRCT_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"

static int g_embed_scatter_split = 0;
extern "C" void crct_embed_scatter_split(int on) { g_embed_scatter_split = on ? 1 : 0; }
static int g_embed_word_index = 1;      // crct_embed_word_index(0): crct_embed_text_bwd_indexed ignores its index (A/B timing, tests; same bits)
extern "C" void crct_embed_word_index(int on) { g_embed_word_index = on ? 1 : 0; }

extern "C" int crct_embed_text_bwd(const void* dy, const void* sum_saved, const float* mean, const float* rstd,
                                        const int64_t* ids, const int64_t* segs, const float* loc, const float* gamma,
                                        float* d_word, float* d_pos, float* d_type, float* d_wloc, float* d_bloc,
                                        float* d_gamma, float* d_beta, float* partials, int B, int T, int H, int n_pos,
                                        uint32_t drop_thr, float drop_scale, uint32_t drop_site, uint64_t seed,
                                        float* rows_scratch, int32_t* idx_scratch, int n_types, crct_stream_t stream) {
  return crct_embed_text_bwd_indexed(dy, sum_saved, mean, rstd, ids, segs, loc, gamma, d_word, d_pos, d_type, d_wloc, d_bloc, d_gamma, d_beta,
                                     partials, B, T, H, n_pos, drop_thr, drop_scale, drop_site, seed, rows_scratch, idx_scratch, n_types, nullptr, 0,
                                     stream);
}
extern "C" int crct_embed_text_bwd_indexed(const void* dy, const void* sum_saved, const float* mean, const float* rstd,
                                                const int64_t* ids, const int64_t* segs, const float* loc, const float* gamma,
                                                float* d_word, float* d_pos, float* d_type, float* d_wloc, float* d_bloc,
                                                float* d_gamma, float* d_beta, float* partials, int B, int T, int H, int n_pos,
                                                uint32_t drop_thr, float drop_scale, uint32_t drop_site, uint64_t seed,
                                                float* rows_scratch, int32_t* idx_scratch, int n_types, int32_t* word_index,
                                                int n_vocab, crct_stream_t stream) {
  CRCT_REQUIRE(H % 8 == 0 && H > 0, "embed_text_bwd: H=%d must be a positive multiple of 8", H);
  const long M = (long)B * T;
  if (M <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (!idx_scratch || M > GATHER_MAX_ROWS || n_types <= 0) rows_scratch = nullptr;      // atomics fall-back
  const int nb = embed_bwd_blocks(M);
  // word_index: int32 [2][n_vocab] (first / last row per id), zero on entry and on exit; without it (or in the two-launch test form) the scanning kernel
  int* w_first = (rows_scratch && word_index && n_vocab > 0 && !g_embed_scatter_split && g_embed_word_index) ? word_index : nullptr;
  int* w_last = w_first ? word_index + n_vocab : nullptr;
  const int type_partials = rows_scratch && n_types >= 2 && d_type;     // partials then hold 9 row sets
  DISPATCH_NCH(H, crct_launch((embed_text_bwd_kernel<NCH>), dim3(nb), dim3(256), 0, s, (const bf16_t*)dy,
                                     (const bf16_t*)sum_saved, mean, rstd, ids, segs, loc, gamma, d_word, d_pos, d_type,
                                     partials, B, T, H, n_pos, drop_thr, drop_scale, drop_site, seed, rows_scratch,
                                     rows_scratch ? idx_scratch : nullptr, type_partials, w_first, w_last));
  CRCT_CHECK_HIP(hipGetLastError());
  if (rows_scratch) {
    const int used_pos = n_pos < T ? n_pos : T;            // position ids are clamped to [0, n_pos) and never exceed T - 1
    const size_t word_lds = (size_t)(((w_first ? 1 : 2) * M + 3) & ~3L) * sizeof(int) + (size_t)ROWS_PER_BLOCK * H * sizeof(float);     // >= the gather's M ints
    CRCT_REQUIRE(word_lds <= 152 * 1024, "embed_text_bwd: B*T=%ld rows need %zu bytes of LDS for the word-gradient scan", M, word_lds);
    const int n_gather = used_pos + n_types;
    if (g_embed_scatter_split) {      // test hook: the two launches the merged kernel replaces (same bits)
      crct_launch(gather_sum_kernel, dim3(n_gather), dim3(256), (size_t)M * sizeof(int), s,
                         rows_scratch, idx_scratch, (int)M, H, d_pos, used_pos, idx_scratch + M, d_type);
      CRCT_CHECK_HIP(hipGetLastError());
      DISPATCH_NCH(H, {
        static bool big_lds_w = false;
        if (word_lds > 64 * 1024 && !big_lds_w) {
          CRCT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&word_scatter_kernel<NCH>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
          big_lds_w = true;
        }
        crct_launch((word_scatter_kernel<NCH>), dim3((int)((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), dim3(256), word_lds, s,
                           (const float*)rows_scratch, ids, (int)M, H, d_word);
      });
      CRCT_CHECK_HIP(hipGetLastError());
    } else
    DISPATCH_NCH(H, {
      static bool big_lds = false;                         // ids + match list of every row: above 64 KiB from ~6 600 rows on
      if (word_lds > 64 * 1024 && !big_lds) {
        CRCT_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&embed_scatter_kernel<NCH>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
        big_lds = true;
      }
      crct_launch((embed_scatter_kernel<NCH>), dim3(n_gather + (int)((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)), dim3(256), word_lds, s,
                         (const float*)rows_scratch, idx_scratch, (int)M, H, d_pos, used_pos, idx_scratch + M, d_type, n_gather, ids, d_word,
                         w_first, w_last);
    });
    CRCT_CHECK_HIP(hipGetLastError());
  }
  FinalizeArgs fa = {};
  fa.out[0] = d_gamma; fa.out[1] = d_beta; fa.out[2] = d_bloc;
  fa.stride[0] = fa.stride[1] = fa.stride[2] = 1;
  for (int k = 0; k < 4; ++k) { fa.out[3 + k] = d_wloc ? d_wloc + k : nullptr; fa.stride[3 + k] = 4; }
  fa.Q = 7;
  if (type_partials) {
    fa.out[7] = d_type; fa.out[8] = d_type + H; fa.stride[7] = fa.stride[8] = 1;
    fa.Q = 9;
  }
  fa.nblk = nb * ROWS_PER_BLOCK; fa.H = H; fa.accumulate = 1; fa.partials = partials;
  return launch_finalize(fa, s);
}

extern "C" int crct_embed_image_fwd(const void* img_lin, const float* loc, const int64_t* target, const float* w_loc,
                                    const float* b_loc, const float* color, const float* gamma, const float* beta,
                                    void* sum_out, void* y, float* mean, float* rstd, int M, int H, float eps,
                                    uint32_t drop_thr, float drop_scale, uint32_t drop_site, uint64_t seed,
                                    crct_stream_t stream) {
  CRCT_REQUIRE(H % 8 == 0 && H > 0, "embed_image: H=%d must be a positive multiple of 8", H);
  if (M <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_NCH(H, crct_launch((embed_image_fwd_kernel<NCH>), dim3(row_grid(M, 2048)), dim3(256), 0, s,
                                     (const bf16_t*)img_lin, loc, target, w_loc, b_loc, color, gamma, beta,
                                     (bf16_t*)sum_out, (bf16_t*)y, mean, rstd, M, H, eps, drop_thr, drop_scale, drop_site, seed));
  CRCT_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int crct_embed_image_bwd(const void* dy, const void* sum_saved, const float* mean, const float* rstd,
                                    const float* loc, const int64_t* target, const float* gamma, void* d_sum,
                                    float* d_color, float* d_wloc, float* d_bloc, float* d_bimg, float* d_gamma,
                                    float* d_beta, float* partials, int M, int H, uint32_t drop_thr, float drop_scale,
                                    uint32_t drop_site, uint64_t seed, float* rows_scratch, int32_t* idx_scratch, int n_color,
                                    crct_stream_t stream) {
  CRCT_REQUIRE(H % 8 == 0 && H > 0, "embed_image_bwd: H=%d must be a positive multiple of 8", H);
  if (M <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (!idx_scratch || M > GATHER_MAX_ROWS || n_color <= 0) rows_scratch = nullptr;      // atomics fall-back
  const int nb = embed_bwd_blocks(M);
  DISPATCH_NCH(H, crct_launch((embed_image_bwd_kernel<NCH>), dim3(nb), dim3(256), 0, s, (const bf16_t*)dy,
                                     (const bf16_t*)sum_saved, mean, rstd, loc, target, gamma, (bf16_t*)d_sum, d_color,
                                     partials, M, H, drop_thr, drop_scale, drop_site, seed, rows_scratch,
                                     rows_scratch ? idx_scratch : nullptr));
  CRCT_CHECK_HIP(hipGetLastError());
  if (rows_scratch) {
    crct_launch(gather_sum_kernel, dim3(n_color), dim3(256), (size_t)M * sizeof(int), s, rows_scratch, idx_scratch, M, H, d_color,
                       n_color, (const int*)nullptr, (float*)nullptr);
    CRCT_CHECK_HIP(hipGetLastError());
  }
  // two finalize passes share the column-sum partial (index 2): b_loc and b_img
  FinalizeArgs fa = {};
  fa.out[0] = d_gamma; fa.out[1] = d_beta; fa.out[2] = d_bloc;
  fa.stride[0] = fa.stride[1] = fa.stride[2] = 1;
  for (int k = 0; k < 4; ++k) { fa.out[3 + k] = d_wloc ? d_wloc + k : nullptr; fa.stride[3 + k] = 4; }
  fa.Q = 7; fa.nblk = nb * ROWS_PER_BLOCK; fa.H = H; fa.accumulate = 1; fa.partials = partials;
  if (launch_finalize(fa, s)) return 1;
  if (d_bimg) {
    FinalizeArgs fb = {};
    fb.out[0] = d_bimg; fb.stride[0] = 1; fb.Q = 1; fb.nblk = nb * ROWS_PER_BLOCK; fb.H = H; fb.accumulate = 1;
    fb.partials = partials + (size_t)2 * nb * ROWS_PER_BLOCK * H;
    return launch_finalize(fb, s);
  }
  return 0;
}
