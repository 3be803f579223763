// Classification head + regression tail + joint loss, and their gradient seeds, without host syncs.
// Reference: BertPreTrainingHeads.forward vilbert.py:1048-1062 (fusion 'mul'/'sum', dropout 0.1,
// Linear(1024, 2)); PlotQA_Regressor_v20 tail regressor.py:31-34,41 (Linear(256,1) + Tanh);
// regression bookkeeping + CrossEntropyLoss(ignore_index=-1) vilbert.py:1583-1657; loss
// combination encoder_decorator.py:144-153.  All B rows are regressed and masked by R[:,1]
// (the reference gathers those rows: same values, same gradients, static shapes).
#include "common.hip.h"
#include "crct_internal.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ bool head_keep(const CrctHeadArgs& a, int b, int c) {
  if (!a.drop_thr) return true;
  const uint64_t idx = (uint64_t)b * (uint64_t)a.Hb + (uint64_t)c;
  return (philox_keep8(a.seed, a.drop_site, idx >> 3, a.drop_thr) >> ((uint32_t)idx & 7u)) & 1u;
}

// scratch row layout (fp32 x 8): dlogit0, dlogit1, dz, nsp_loss_b, valid, ok5, okt, needs
__global__ __launch_bounds__(256) void head_rows_kernel(const CrctHeadArgs a, float* scratch) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const bf16_t* pt = reinterpret_cast<const bf16_t*>(a.pooled_t) + (long)b * a.Hb;
  const bf16_t* pv = reinterpret_cast<const bf16_t*>(a.pooled_v) + (long)b * a.Hb;
  const bf16_t* fh = reinterpret_cast<const bf16_t*>(a.fus_h) + (long)b * 256;
  const float dsc = a.drop_thr ? a.drop_scale : 1.0f;
  // ---- logits
  float l0 = 0.f, l1 = 0.f;
  for (int c = tid; c < a.Hb; c += 256) {
    const float t = bf2f(pt[c]), v = bf2f(pv[c]);
    float f = a.fusion_sum ? t + v : t * v;
    f = head_keep(a, b, c) ? f * dsc : 0.f;
    l0 += f * a.w_cls[c]; l1 += f * a.w_cls[a.Hb + c];
  }
  l0 = block_sum(l0, red) + a.b_cls[0];
  l1 = block_sum(l1, red) + a.b_cls[1];
  // ---- regression tail
  float z = 0.f;
  for (int c = tid; c < 256; c += 256) z += bf2f(fh[c]) * a.w_f6[c];
  z = block_sum(z, red) + a.b_f6[0];
  const float r = tanhf(z);
  // ---- labels: count of rows that enter the CE mean (ignore_index = -1)
  int n_valid = 0;
  long label = -1;
  if (a.labels) {
    for (int i = tid; i < a.B; i += 256) n_valid += (a.labels[i] != -1);
    n_valid = (int)(block_sum((float)n_valid, red) + 0.5f);
    label = a.labels[b];
  }
  // upstream gradients of the two differentiable outputs (nsp_loss scalar, reg_loss rows): either
  // device tensors handed over by autograd, or the fixed combination of encoder_decorator.py:144-153
  // g_loss_dev: upstream gradient of the COMBINED loss stats[0] (a device scalar: 1, 1 / batch_multiply, a GradScaler scale ...)
  const float g_loss = (a.g_loss_dev ? a.g_loss_dev[0] : 1.0f) * a.grad_scale;
  const float g_nsp = a.g_nsp_dev ? a.g_nsp_dev[0] * a.grad_scale : a.nsp_coeff * g_loss;
  const float g_reg = a.g_reg_dev ? a.g_reg_dev[b] * a.grad_scale : a.reg_coeff * g_loss / (float)a.B;
  // per-row scalars (every thread computes them identically)
  const float mx = fmaxf(l0, l1);
  const float lse = mx + logf(expf(l0 - mx) + expf(l1 - mx));
  float nsp_b = 0.f, dl0 = 0.f, dl1 = 0.f;
  const bool valid = a.labels && label != -1;
  if (valid) {
    nsp_b = lse - (label == 0 ? l0 : l1);
    const float w = g_nsp / (float)max(n_valid, 1);
    dl0 = (expf(l0 - lse) - (label == 0 ? 1.f : 0.f)) * w;
    dl1 = (expf(l1 - lse) - (label == 1 ? 1.f : 0.f)) * w;
  }
  const float* Rb = a.R + (long)b * 4;
  const bool needs = Rb[1] == 1.0f;
  const float target = Rb[0] / Rb[3];
  const float diff = r - target, l1v = fabsf(diff);
  float rl, drl;   // reg loss and d(reg loss)/dr
  if (a.use_l1) { rl = l1v; drl = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f); }
  else {           // SmoothL1, beta = 0.5
    if (l1v < 0.5f) { rl = diff * diff; drl = 2.0f * diff; }    // 0.5 d^2 / beta
    else { rl = l1v - 0.25f; drl = diff > 0.f ? 1.f : -1.f; }
  }
  const bool both0 = (r == 0.f) && (target == 0.f);
  float d5 = l1v / fabsf(target);
  if (target == 0.f) d5 = 1.f;
  if (both0) d5 = 0.f;
  const bool ok5 = ((d5 <= 0.05f) || both0) && needs;
  const bool okt = (l1v <= a.tol_margin) && needs;
  if (!a.kind_l1 && fabsf(target) > 1.f) { rl = 0.f; drl = 0.f; }
  if (!needs) { rl = 0.f; drl = 0.f; }
  const float dz = drl * g_reg * (1.f - r * r);
  if (tid == 0) {
    a.logits[b * 2] = l0; a.logits[b * 2 + 1] = l1;
    a.reg[0 * a.B + b] = needs ? r * Rb[3] : 0.f;
    a.reg[1 * a.B + b] = rl;
    a.reg[2 * a.B + b] = needs ? l1v : 0.f;
    a.reg[3 * a.B + b] = r;                       // raw tanh output (diagnostic)
    a.reg[4 * a.B + b] = needs ? d5 : 0.f;
    float* s = scratch + (long)b * 8;
    s[0] = dl0; s[1] = dl1; s[2] = dz; s[3] = nsp_b; s[4] = valid ? 1.f : 0.f;
    s[5] = ok5 ? 1.f : 0.f; s[6] = okt ? 1.f : 0.f; s[7] = needs ? 1.f : 0.f;
  }
  // ---- gradient seeds (w.r.t. the PRE-activations of the poolers / fusion.4)
  if (a.d_pooled_t) {
    bf16_t* dpt = reinterpret_cast<bf16_t*>(a.d_pooled_t) + (long)b * a.Hb;
    bf16_t* dpv = reinterpret_cast<bf16_t*>(a.d_pooled_v) + (long)b * a.Hb;
    for (int c = tid; c < a.Hb; c += 256) {
      const float t = bf2f(pt[c]), v = bf2f(pv[c]);
      float df = dl0 * a.w_cls[c] + dl1 * a.w_cls[a.Hb + c];
      df = head_keep(a, b, c) ? df * dsc : 0.f;
      const float gt = a.fusion_sum ? df : df * v, gv = a.fusion_sum ? df : df * t;
      dpt[c] = f2bf(t > 0.f ? gt : 0.f);        // relu'(pre) == (post > 0)
      dpv[c] = f2bf(v > 0.f ? gv : 0.f);
    }
    bf16_t* dfh = reinterpret_cast<bf16_t*>(a.d_fus_h) + (long)b * 256;
    for (int c = tid; c < 256; c += 256) {
      const float hval = bf2f(fh[c]);
      dfh[c] = f2bf(dz * a.w_f6[c] * (hval > 0.f ? 1.f : 0.01f));
    }
  }
}

// stats + parameter gradients of bi_seq_relationship and fusion.6 (sums over the batch rows)
__global__ __launch_bounds__(256) void head_reduce_kernel(const CrctHeadArgs a, const float* scratch) {
  const int tid = threadIdx.x;
  const float dsc = a.drop_thr ? a.drop_scale : 1.0f;
  // parameter gradients: 32 columns x 8 row groups per workgroup (a thread walks B / 8 rows, not all of them), the 8
  // partial sums are added in a fixed order.  Columns beyond Hb / 256 idle.
  __shared__ float part[3][8][33];
  const int cl = tid & 31, rg = tid >> 5;
  const int cc = blockIdx.x * 32 + cl;
  float g0 = 0.f, g1 = 0.f, g2 = 0.f;
  if (a.d_w_cls && cc < a.Hb) {
    const bf16_t* pt = reinterpret_cast<const bf16_t*>(a.pooled_t);
    const bf16_t* pv = reinterpret_cast<const bf16_t*>(a.pooled_v);
    for (int b = rg; b < a.B; b += 8) {
      const float t = bf2f(pt[(long)b * a.Hb + cc]), v = bf2f(pv[(long)b * a.Hb + cc]);
      float f = a.fusion_sum ? t + v : t * v;
      f = head_keep(a, b, cc) ? f * dsc : 0.f;
      g0 += scratch[b * 8] * f; g1 += scratch[b * 8 + 1] * f;
    }
  }
  if (a.d_w_f6 && cc < 256) {
    const bf16_t* fh = reinterpret_cast<const bf16_t*>(a.fus_h);
    for (int b = rg; b < a.B; b += 8) g2 += scratch[b * 8 + 2] * bf2f(fh[(long)b * 256 + cc]);
  }
  part[0][rg][cl] = g0; part[1][rg][cl] = g1; part[2][rg][cl] = g2;
  __syncthreads();
  if (rg == 0) {
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { t0 += part[0][k][cl]; t1 += part[1][k][cl]; t2 += part[2][k][cl]; }
    if (a.d_w_cls && cc < a.Hb) { a.d_w_cls[cc] += t0; a.d_w_cls[a.Hb + cc] += t1; }
    if (a.d_w_f6 && cc < 256) a.d_w_f6[cc] += t2;
  }
  if (blockIdx.x == 0 && tid < 64) {
    // one wave reduces the per-row records
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float rl = 0.f, d5 = 0.f;
    for (int b = tid; b < a.B; b += 64) {
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] += scratch[b * 8 + k];
      rl += a.reg[1 * a.B + b];
      d5 += a.reg[4 * a.B + b];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = wave_sum(s[k]);
    rl = wave_sum(rl);
    d5 = wave_sum(d5);
    if (tid == 0) {
      const float nsp = s[4] > 0.f ? s[3] / s[4] : 0.f;
      const float reg_mean = rl / (float)a.B;
      a.stats[0] = a.labels ? a.nsp_coeff * nsp + a.reg_coeff * reg_mean : 0.f;
      a.stats[1] = nsp; a.stats[2] = reg_mean; a.stats[3] = s[7]; a.stats[4] = s[5]; a.stats[5] = s[6];
      a.stats[6] = s[4]; a.stats[7] = 0.f;
      // stats[8..16]: the 9 floats train.py:181-183 shares between the ranks every iteration, ready for the all-reduce
      // (loss, lm_loss = 0, nsp_loss, mean reg_loss / mean reg_5_dist over the rows that need regression -- 0 when there
      // are none, as the reference's isnan test does --, legend_loss = 0, num_regs, reg_5_right, reg_t_right)
      a.stats[8] = a.stats[0]; a.stats[9] = 0.f; a.stats[10] = nsp;
      a.stats[11] = s[7] > 0.f ? rl / s[7] : 0.f; a.stats[12] = s[7] > 0.f ? d5 / s[7] : 0.f; a.stats[13] = 0.f;
      a.stats[14] = s[7]; a.stats[15] = s[5]; a.stats[16] = s[6];
      if (a.d_b_cls) { a.d_b_cls[0] += s[0]; a.d_b_cls[1] += s[1]; }
      if (a.d_b_f6) a.d_b_f6[0] += s[2];
    }
  }
}

// ------------------------------------------------------------------ evaluation: per-question answer selection
// One wave per question.  Question q owns the candidate rows [off_q, off_q + num_ans[q]) with off_q = sum of the
// earlier counts (recomputed by every wave: Q is a few hundred at most).  p0 = softmax(logits)[0] in fp32; the
// answer is the FIRST row with the largest p0 (torch.argmax), or forced[q] when given; the regressed value and its
// two error measures are gathered from the chosen row.
__global__ __launch_bounds__(64) void eval_select_kernel(const float* __restrict__ logits, const float* __restrict__ reg_out,
                                                         const float* __restrict__ reg_err, const float* __restrict__ reg_terr,
                                                         const int64_t* __restrict__ num_ans, const int64_t* __restrict__ forced,
                                                         int Q, long N, float* __restrict__ prob0, int64_t* __restrict__ answers,
                                                         float* __restrict__ sel_out, float* __restrict__ sel_err,
                                                         float* __restrict__ sel_terr) {
  const int q = blockIdx.x, lane = threadIdx.x;
  long off = 0;
  for (int i = lane; i < q; i += 64) off += num_ans[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) off += __shfl_xor(off, o, 64);
  const long n = num_ans[q];
  float best = -INFINITY;
  long best_i = 0x7fffffffffffffffL;
  for (long j = lane; j < n; j += 64) {
    const long r = off + j;
    float p = 0.f;
    if (r < N) {
      const float l0 = logits[2 * r], l1 = logits[2 * r + 1], m = fmaxf(l0, l1);
      const float e0 = expf(l0 - m), e1 = expf(l1 - m);
      p = e0 / (e0 + e1);
      if (prob0) prob0[r] = p;
    }
    if (p > best) { best = p; best_i = j; }       // strictly greater: the earliest row of this lane wins ties
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const long oi = __shfl_xor(best_i, o, 64);
    if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
  }
  if (lane == 0) {
    long a = forced ? forced[q] : (n > 0 ? best_i : 0);
    answers[q] = a;
    const long r = off + a;
    const bool ok = a >= 0 && a < n && r < N;
    sel_out[q] = ok ? reg_out[r] : 0.f;
    sel_err[q] = ok ? reg_err[r] : INFINITY;
    sel_terr[q] = ok ? reg_terr[r] : INFINITY;
  }
}

}  // namespace

extern "C" int crct_eval_select(const float* logits, const float* reg_out, const float* reg_err, const float* reg_terr,
                                const int64_t* num_ans, const int64_t* forced_answers, int Q, int64_t N, float* prob0,
                                int64_t* answers, float* sel_out, float* sel_err, float* sel_terr, crct_stream_t stream) {
  CRCT_REQUIRE(logits && reg_out && reg_err && reg_terr && num_ans && answers && sel_out && sel_err && sel_terr,
               "eval_select: null argument");
  CRCT_REQUIRE(Q >= 0 && N >= 0, "eval_select: bad sizes");
  if (Q == 0) return 0;
  crct_launch(eval_select_kernel, dim3(Q), dim3(64), 0, (hipStream_t)stream, logits, reg_out, reg_err, reg_terr, num_ans,
                     forced_answers, Q, (long)N, prob0, answers, sel_out, sel_err, sel_terr);
  CRCT_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int crct_head_loss(const CrctHeadArgs* args, crct_stream_t stream) {
  CRCT_REQUIRE(args && args->B > 0 && args->Hb > 0, "head_loss: bad sizes");
  CRCT_REQUIRE(args->scratch, "head_loss: scratch (fp32 [B][8]) is required");
  hipStream_t s = (hipStream_t)stream;
  crct_launch(head_rows_kernel, dim3(args->B), dim3(256), 0, s, *args, args->scratch);
  CRCT_CHECK_HIP(hipGetLastError());
  const int nb = ((args->Hb > 256 ? args->Hb : 256) + 31) / 32;
  crct_launch(head_reduce_kernel, dim3(nb), dim3(256), 0, s, *args, (const float*)args->scratch);
  CRCT_CHECK_HIP(hipGetLastError());
  return 0;
}
