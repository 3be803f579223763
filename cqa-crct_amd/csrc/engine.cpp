// Step engine: schedules the hand-written gfx950 kernels for one CRCT training step
// (forward + joint loss + backward) on one HIP stream, without host synchronisation.
//
// Reference path it replaces (levymsn/CQA-CRCT):
//   encoder_decorator.forward            CRCT/backbone/encoder_decorator.py:73-158
//   BertForMultiModalPreTraining.forward CRCT/backbone/vilbert.py:1540-1661
//   BertModel.forward / BertEncoder      vilbert.py:1348-1441 / :822-946 (layer order :852-939)
//   BertLayer / BertImageLayer           vilbert.py:361-485 / :488-616
//   BertConnectionLayer                  vilbert.py:619-788
//   poolers, heads, regressor            vilbert.py:949-976, :1048-1062; regressor.py:5-42
//   + torch autograd of all of it (train.py:208).
//
// Data layout in HBM: parameters live in ONE flat fp32 buffer (and a bf16 shadow with identical
// element offsets) ordered by first use, so the gradient buffer completes back-to-front during
// backward and contiguous ranges can be all-reduced while earlier layers are still computing.
// query/key/value weights of a layer are adjacent -> one [3H, H] fused-QKV GEMM operand.
// Activations kept for backward are bf16 in a caller-provided workspace.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <unordered_map>
#include <functional>
#include <vector>

#include "crct_internal.h"

typedef unsigned short bf16_t;
enum { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2, ACT_LEAKY = 3, ACT_TANH = 4 };

// ------------------------------------------------------------------------------------------ errors
static thread_local char g_err[1024] = "";
void crct_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* crct_last_error(void) { return g_err; }
extern "C" int crct_abi_version(void) { return 7; }

extern "C" int crct_gemm_bf16(const CrctGemmArgs* a, crct_stream_t stream) {
  CRCT_REQUIRE(a != nullptr, "gemm: null args");
  CRCT_REQUIRE(a->A && a->B && a->C, "gemm: null operand");
  CRCT_REQUIRE(a->N % 4 == 0, "gemm: N=%d must be a multiple of 4", a->N);
  CRCT_REQUIRE((a->ta && a->tb) || a->K % 8 == 0, "gemm: K=%d must be a multiple of 8 for a K-contiguous operand", a->K);
  CRCT_REQUIRE(a->lda % 8 == 0 && a->ldb % 8 == 0, "gemm: lda=%ld ldb=%ld must be multiples of 8", (long)a->lda, (long)a->ldb);
  CRCT_REQUIRE(a->ldc % 4 == 0, "gemm: ldc=%ld must be a multiple of 4", (long)a->ldc);
  CRCT_REQUIRE(!(a->ta && !a->tb), "gemm: (ta=1, tb=0) is not built (not used by the step)");
  CRCT_REQUIRE(!a->ta || a->M % 8 == 0, "gemm: transposed A needs M %% 8 == 0 (M=%d)", a->M);
  CRCT_REQUIRE(!a->tb || a->N % 8 == 0, "gemm: transposed B needs N %% 8 == 0 (N=%d)", a->N);
  CRCT_CHECK_HIP(crct_gemm_launch(*a, (hipStream_t)stream));
  return 0;
}

// target_wgs < 0: the library's default (crct_gemm_group_target_workgroups)
static int gemm_grouped_checked(const CrctGemmArgs* a, int n, crct_stream_t stream, int target_wgs) {
  CRCT_REQUIRE(a != nullptr && n >= 1, "gemm_grouped: bad arguments");
  for (int i = 0; i < n; ++i) {
    CRCT_REQUIRE(a[i].A && a[i].B && a[i].C, "gemm_grouped: null operand in problem %d", i);
    CRCT_REQUIRE(a[i].N % 4 == 0 && a[i].lda % 8 == 0 && a[i].ldb % 8 == 0 && a[i].ldc % 4 == 0, "gemm_grouped: alignment of problem %d", i);
    CRCT_REQUIRE((a[i].ta && a[i].tb) || a[i].K % 8 == 0, "gemm_grouped: K of problem %d", i);
    CRCT_REQUIRE(!(a[i].ta && !a[i].tb) && (!a[i].ta || a[i].M % 8 == 0) && (!a[i].tb || a[i].N % 8 == 0), "gemm_grouped: layout of problem %d", i);
  }
  if (target_wgs < 0) CRCT_CHECK_HIP(crct_gemm_launch_grouped(a, n, (hipStream_t)stream));
  else CRCT_CHECK_HIP(crct_gemm_launch_grouped_wgs(a, n, (hipStream_t)stream, target_wgs));
  return 0;
}
extern "C" int crct_gemm_bf16_grouped(const CrctGemmArgs* a, int n, crct_stream_t stream) { return gemm_grouped_checked(a, n, stream, -1); }

namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline uint32_t thr_of(float p) {
  double t = (double)p * 4294967296.0;
  if (t <= 0.0) return 0u;
  if (t >= 4294967295.0) return 4294967295u;
  return (uint32_t)t;
}

struct Arena {
  size_t top = 0;
  size_t take(size_t bytes) { size_t o = top; top = align_up(top + bytes, 256); return o; }
};

struct Drop { uint32_t thr = 0; float scale = 1.f; uint32_t site = 0; };

// ---- parameter offsets (elements into the flat buffers)
struct LinearP { int64_t w = -1, b = -1; int in = 0, out = 0; int site = 0; };
struct LnP { int64_t g = -1, b = -1; };
struct FfnP { LinearP up, down; LnP ln; };
struct ProjP { LinearP dense; LnP ln; };      // LN(dropout(dense(ctx)) + residual)
struct SelfLayerP { LinearP qkv; ProjP proj; FfnP ffn; int H, heads; float p_attn, p_hid; uint32_t site; };
struct ConnLayerP { LinearP qkv1, qkv2; ProjP proj_v, proj_t; FfnP ffn_v, ffn_t; uint32_t site; };

// ---- activation offsets (bytes into the workspace)
// s: the pre-LayerNorm sum, room for fp32 (the fp32 residual stream, CrctStepCfg.residual_fp32; bf16 in the first half otherwise);
// y32 / a32: the fp32 copy of the LayerNorm output that the NEXT block's epilogue adds as its residual
struct FfnA { size_t u, h, s, y, y32, mean, rstd, hq, yq; int site_h, site_y; int g_dl, g_du; };      // g_*: gradient scale sites (fp8 backward)      // hq / yq: e4m3 copies (fp8 forward), site_*: their scale slots
struct ProjA { size_t s, a, a32, mean, rstd, aq; int site_a; int g_dl; };
// ctxq / site_ctx: e4m3 copy of the attention context and its activation scale site; g_dqkv: gradient scale site of the fused dqkv buffer
// lse*: softmax row statistics [B][heads][Tq] fp32 the long-sequence attention forward leaves for its backward (CrctAttnQuant.row_lse)
struct SelfLayerA { size_t qkv, ctx, ctxq, lse; int site_ctx, g_dqkv; ProjA proj; FfnA ffn; };
struct ConnLayerA { size_t qkv1, qkv2, ctx1, ctx2, ctx1q, ctx2q, lse1, lse2; int site_ctx1, site_ctx2, g_dqkv1, g_dqkv2; ProjA proj_v, proj_t; FfnA ffn_v, ffn_t; };
struct StreamScratch { size_t dy[2], dres_a, dlin_a, dres_b, dlin_b, gc, du, dctx, dqkv, part_a, part_b, dlq_a, dlq_b, duq, dqkvq; };      // *q: e5m2 copies (fp8 backward)

struct Step { char kind; int idx; };
struct Tap { std::string name; size_t off; char stream; };

}  // namespace

struct crct_engine {
  CrctModelDims d;
  std::unordered_map<std::string, int64_t> off, size;
  int maxB, maxT, maxV;
  std::vector<Step> sched;
  std::vector<SelfLayerP> tl, vl;
  std::vector<ConnLayerP> cl;
  std::vector<SelfLayerA> tla, vla;
  std::vector<ConnLayerA> cla;
  struct { int64_t word, pos, type, wloc, bloc; LnP ln; } et;
  struct { LinearP img; int64_t color, wloc, bloc; LnP ln; } ev;
  struct { size_t sum, y, mean, rstd, yq; int site; } eta;
  struct { size_t soft, lin, sum, y, mean, rstd, yq; int site; } eva;
  LinearP t_pool, v_pool, cls, tp[4], vp[4], fu[4];
  struct { size_t pooled_t, pooled_v, t[3], v[3], cat, f[3], scratch, d_pt, d_pv, g[10]; } ha;   // g: one buffer per head gradient (see heads_bwd)
  StreamScratch st, sv;          // backward scratch per data stream (dy ping-pong lives in these)
  StreamScratch st2, sv2;        // second set: layers alternate sets so weight-gradient GEMMs may lag one layer behind
  size_t partials[2], colsum_part[4];   // per internal stream: [text, visual] / [text, visual, text-wgrad, visual-wgrad]
  size_t embed_rows[2], embed_idx[2];   // embedding backward: fp32 row gradients + table indices for the gather-sum pass
  size_t km_t = 0, km_v = 0;
  // per-site launch policy of the forward / data-gradient GEMMs (crct_engine_set_site_policy): [site][kind][phase]
  struct SitePolicy { int cfg = -1, split_k = 0; };
  SitePolicy policy[CRCT_SITE_COUNT][3][2];      // kind 2 (weight gradient): cfg only -- the layer's grouped launch takes the first problem's
  size_t sk_ws[2] = {0, 0}, sk_cnt[2] = {0, 0};   // split-K slab space / ticket words per data stream [text, visual]
  size_t sk_ws_elems[2] = {0, 0};
  int sk_tickets = 0;
  // fp8 forward (BASELINE configs[4]): scale slot of every Linear weight that has an e4m3 shadow, number of activation scale sites
  std::unordered_map<int64_t, int> wq_slot;
  std::unordered_map<size_t, size_t> res32;      // workspace offset of a LayerNorm's bf16 output -> offset of its fp32 copy (fp32 residual stream)
  int32_t* word_index = nullptr;      // device memory owned by the engine: crct_embed_text_bwd_indexed's first / last row per token id, zero between calls
  std::vector<std::pair<int64_t, int64_t>> wq_list;      // slot -> (flat offset, numel)
  int n_sites = 0;
  int n_gsites = 0;                    // gradient scale sites of the fp8 backward (CrctStepCfg.fp8_grad_scale / _amax)
  size_t ws_bytes = 0;
  // internal concurrency: the visual stream's layers and all weight-gradient GEMMs run on side HIP
  // streams, ordered against the caller's stream by events (fork / join inside every call)
  bool use_vis_stream = true, use_wgrad_stream = true, streams_forced = false;
  int wgrad_target = 96, wgrad_target_rows = 3000;      // crct_engine_set_wgrad_workgroups (Run::flush_wgrads)
  int wgrad_flush = 1;                 // crct_engine_set_wgrad_flush: extra flush points of a layer's queued weight gradients (Run::ffn_bwd).
                                       // 1 (round 4): FFN group 0.075 -> 0.081 of peak in the step, step -0.02 (bf16) / -0.06 (fp8) / -0.08 ms (long context)
  bool one_wgrad_stream = false;       // both data streams' weight gradients on ONE side stream (frees a hardware queue for the exchange)
  int first_conn = -1;                  // schedule index of the first co-attention layer
  hipStream_t side[3] = {nullptr, nullptr, nullptr};   // visual, text-wgrad, visual-wgrad
  // hardware-queue placement (streams.hip): the four streams below sit on queues other than the caller's stream's -- visual and
  // text-wgrad on queues of their own, the auxiliary stream (optimizer overlap during forward, gradient exchange during
  // backward: crct_engine_aux_stream) on the third, which the visual-wgrad stream shares (it is idle whenever the optimizer
  // runs; with one_wgrad_stream it is not used and the exchange has that queue to itself)
  hipStream_t aux = nullptr;
  hipStream_t placed_for = nullptr;
  bool placed = false;
  int queue_classes = 0;
  std::vector<hipEvent_t> evpool;
  size_t evnext = 0;
  std::vector<std::pair<int64_t, int64_t>> seg_range;
  // weight-gradient ownership (CrctStepCfg.wgrad_overwrite): the Linear weights whose gradient is produced by exactly ONE
  // weight-gradient GEMM per backward pass and by nothing else -- every Linear of the encoder layers, the image embedding,
  // the poolers and the regressor pipes.  Fixed by the schedule at crct_engine_create (offset -> numel); `wgrad_pass` counts
  // the productions of the running pass so that a violation is an error, never a silent overwrite.
  std::unordered_map<int64_t, int64_t> wgrad_owned;
  std::unordered_map<int64_t, int> wgrad_pass;
  void wgrad_pass_begin() { wgrad_pass.clear(); }
  std::vector<Tap> taps;
  size_t final_t = 0, final_v = 0;   // offsets of the last-layer outputs
  int cur_t = 0, cur_v = 0;          // ping-pong index of the running activation gradients
  bool bad = false;
  int64_t P(const std::string& k) {
    auto it = off.find(k);
    if (it == off.end()) { crct_set_error("engine: parameter '%s' missing from the layout", k.c_str()); bad = true; return 0; }
    return it->second;
  }
};

namespace {

LinearP linear_p(crct_engine* e, const std::string& name, int in, int out, int site = CRCT_SITE_HEAD) {
  LinearP l;
  l.w = e->P(name + ".weight"); l.b = e->P(name + ".bias"); l.in = in; l.out = out; l.site = site;
  return l;
}
LnP ln_p(crct_engine* e, const std::string& name) {
  LnP l; l.g = e->P(name + ".weight"); l.b = e->P(name + ".bias"); return l;
}
// three Linear(in, out) stored back to back -> one Linear(in, 3*out)
LinearP fused3(crct_engine* e, const std::string& a, const std::string& b, const std::string& c, int in, int out, int site) {
  LinearP l = linear_p(e, a, in, 3 * out, site);
  const int64_t wsz = (int64_t)in * out;
  if (e->P(b + ".weight") != l.w + wsz || e->P(c + ".weight") != l.w + 2 * wsz || e->P(b + ".bias") != l.b + out ||
      e->P(c + ".bias") != l.b + 2 * out) {
    crct_set_error("engine: %s / %s / %s must be adjacent in the flat layout (fused QKV)", a.c_str(), b.c_str(), c.c_str());
    e->bad = true;
  }
  return l;
}

FfnA ffn_a(Arena& ar, size_t M, int H, int I, int& sites, int& gsites) {
  FfnA a;
  a.g_dl = gsites++; a.g_du = gsites++;
  a.u = ar.take(M * I * 2); a.h = ar.take(M * I * 2); a.s = ar.take(M * H * 4); a.y = ar.take(M * H * 2); a.y32 = ar.take(M * H * 4);
  a.mean = ar.take(M * 4); a.rstd = ar.take(M * 4);
  a.hq = ar.take(M * I); a.yq = ar.take(M * H);
  a.site_h = sites++; a.site_y = sites++;
  return a;
}
ProjA proj_a(Arena& ar, size_t M, int H, int& sites, int& gsites) {
  ProjA a;
  a.g_dl = gsites++;
  a.s = ar.take(M * H * 4); a.a = ar.take(M * H * 2); a.a32 = ar.take(M * H * 4); a.mean = ar.take(M * 4); a.rstd = ar.take(M * 4);
  a.aq = ar.take(M * H);
  a.site_a = sites++;
  return a;
}
StreamScratch scratch_a(Arena& ar, size_t M, int H, int I, int Hb) {
  StreamScratch s;
  const int Hm = H > Hb ? H : Hb;
  s.dy[0] = ar.take(M * H * 2); s.dy[1] = ar.take(M * H * 2);
  s.dres_a = ar.take(M * H * 2); s.dlin_a = ar.take(M * H * 2); s.dres_b = ar.take(M * H * 2); s.dlin_b = ar.take(M * H * 2);
  s.gc = ar.take(M * H * 2); s.du = ar.take(M * (size_t)I * 2); s.dctx = ar.take(M * (size_t)Hm * 2);
  s.dqkv = ar.take(M * (size_t)3 * Hm * 2);
  s.part_a = ar.take((size_t)3 * 4 * CRCT_LN_BWD_MAX_BLOCKS * H * 4);   // [3][4 waves x blocks][H]      // LayerNorm-backward column partials of the layer's two norms
  s.part_b = ar.take((size_t)3 * 4 * CRCT_LN_BWD_MAX_BLOCKS * H * 4);
  s.dlq_a = ar.take(M * H); s.dlq_b = ar.take(M * H); s.duq = ar.take(M * (size_t)I);
  s.dqkvq = ar.take(M * (size_t)3 * Hm);
  return s;
}

// ================================================================================ per-call context
hipEvent_t ev_new(crct_engine* e) {
  if (e->evnext == e->evpool.size()) {
    hipEvent_t ev = nullptr;
    // ordering between streams of ONE device: no system-scope fence (host / peer visibility) at the record -- the hand-off
    // is 2.5-4 us shorter (tools/handoff_lab.cpp)
    const unsigned flags = hipEventDisableTiming | hipEventDisableSystemFence;
    if (hipEventCreateWithFlags(&ev, flags) != hipSuccess) return nullptr;
    e->evpool.push_back(ev);
  }
  return e->evpool[e->evnext++];
}
// everything enqueued on `from` so far happens before whatever is enqueued on `to` from now on
int order_streams(crct_engine* e, hipStream_t from, hipStream_t to) {
  if (from == to) return 0;
  hipEvent_t ev = ev_new(e);
  if (!ev || hipEventRecord(ev, from) != hipSuccess || hipStreamWaitEvent(to, ev, 0) != hipSuccess) {
    crct_set_error("engine: event ordering between internal streams failed");
    return 1;
  }
  return 0;
}

// fp8 copies of a weight gradient's operands (Run::lin_wgrad)
struct WgQ8 { size_t dyq = (size_t)-1; int g_dy = -1; size_t xq = (size_t)-1; int site_x = -1; };

struct Run {
  crct_engine* e;
  const float* p32; const bf16_t* p16; float* g32; char* ws; hipStream_t s;
  const CrctBatch* b; const CrctStepCfg* c;
  hipStream_t sw;                      // stream of this data stream's weight-gradient GEMMs (== s when disabled)
  size_t partials, colsum_part, colsum_part_w;
  int which = 0;                       // 0 = text stream, 1 = visual stream (owner of split-K workspace `which`)
  int phase = 1;                       // 0: text-only part of the schedule, 1: beside the visual stream (site policy)
  int rc = 0;
  bool sw_dirty = false;               // sw has work that s has not been ordered after yet (no empty forks / joins)
  // scratch double-buffering: layer n of this data stream uses scratch set (n & 1); before reusing a set the
  // data stream waits only for the weight-gradient work of the layer that used it LAST (two layers ago)
  const StreamScratch* sets[2] = {nullptr, nullptr};
  hipEvent_t set_free[2] = {nullptr, nullptr};
  int parity = 0;
  const StreamScratch& layer_begin() {
    if (!rc && sw != s && set_free[parity]) {
      if (hipStreamWaitEvent(s, set_free[parity], 0) != hipSuccess) { crct_set_error("engine: stream wait failed"); rc = 1; }
      set_free[parity] = nullptr;
    }
    return *sets[parity];
  }
  void layer_end() {
    flush_wgrads();
    if (!rc && sw != s && sw_dirty) {
      hipEvent_t ev = ev_new(e);
      if (!ev || hipEventRecord(ev, sw) != hipSuccess) { crct_set_error("engine: event record failed"); rc = 1; }
      set_free[parity] = ev;
    }
    parity ^= 1;
  }
  std::vector<CrctGemmArgs> pending;   // weight-gradient GEMMs of the current layer, launched as ONE grouped grid
  bool defer_wgrad = true;             // false: launch every weight gradient immediately on s (buffers are recycled)
  struct FinJob { const float* part; float* dg; float* db; float* dlb; int M, H; };
  std::vector<FinJob> pending_fin;     // LayerNorm column passes of the current layer (run on sw at the layer's flush)
  int tick = 0, ordered_tick = -1;     // launches enqueued on s / the tick sw was last ordered after (skip redundant events)
  void wgrad_after_main() {            // sw sees what s produced
    if (rc || sw == s) return;
    sw_dirty = true;
    if (ordered_tick == tick) return;  // nothing new on s since the last ordering: sw is already behind it
    fail(order_streams(e, s, sw));
    ordered_tick = tick;
  }
  void main_after_wgrad() {            // s may overwrite what sw read
    flush_wgrads();
    if (!rc && sw != s && sw_dirty) { fail(order_streams(e, sw, s)); sw_dirty = false; }
  }

  template <class T> T* W(size_t o) const { return reinterpret_cast<T*>(ws + o); }
  bf16_t* A(size_t o) const { return W<bf16_t>(o); }
  float* F(size_t o) const { return W<float>(o); }
  const float* P(int64_t o) const { return p32 + o; }
  const bf16_t* PB(int64_t o) const { return p16 + o; }
  float* G(int64_t o) const { return g32 + o; }
  Drop drop(float p, uint32_t site) const {
    Drop d; d.site = site;
    if (c->training && p > 0.f) { d.thr = thr_of(p); d.scale = 1.0f / (1.0f - p); }
    return d;
  }
  void fail(int r) { if (!rc && r) rc = r; }

  struct Opt {
    const float* bias = nullptr; void* preact = nullptr; const void* dact_src = nullptr; int dact = 0; int act = 0;
    const void* addend = nullptr; int64_t ld_aux = 0, ld_add = 0; Drop drop; bool f32 = false; bool acc = false;
    bool add_f32 = false, c_cached = false;      // the fp32 residual stream: fp32 addend; fp32 output that the next kernel reads
    int site = 0;
    void* q_out = nullptr; const float* q_scale = nullptr; float* q_amax = nullptr; int64_t ld_q = 0;      // fp8 copy of the result (calibration passes):
    bool q_e4m3 = false;                                                                                   // e5m2 (a gradient) unless q_e4m3 (an activation)
  };
  void gemm(const void* Ap, int64_t lda, bool ta, const void* Bp, int64_t ldb, bool tb, void* C, int64_t ldc, int M, int N,
            int K, const Opt& o, hipStream_t st = nullptr) {
    if (rc) return;
    if (!st) st = s;
    if (st == s) ++tick;
    CrctGemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = Ap; g.B = Bp; g.C = C; g.bias = o.bias; g.preact_out = o.preact; g.dact_src = o.dact_src; g.addend = o.addend;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ld_aux = o.ld_aux; g.ld_add = o.ld_add;
    g.M = M; g.N = N; g.K = K; g.ta = ta; g.tb = tb; g.act = o.act; g.dact = o.dact; g.c_is_f32 = o.f32; g.accumulate = o.acc;
    g.tile = -1; g.alpha = 1.0f; g.drop_thr = o.drop.thr; g.drop_scale = o.drop.scale; g.drop_site = o.drop.site; g.seed = c->seed;
    g.site = o.site; g.addend_f32 = o.add_f32; g.c_cached = o.c_cached;
    if (o.q_out) { g.q_out = o.q_out; g.q_scale = o.q_scale; g.q_amax = o.q_amax; g.ld_q = o.ld_q; g.fp8 = o.q_e4m3 ? 0 : 4; }      // bf16 GEMM + fp8 copy of its result
    if (!ta && st == s && o.site > 0 && o.site < CRCT_SITE_COUNT) {      // forward / data gradient on the data stream: the site's policy
      const crct_engine::SitePolicy& pol = e->policy[o.site][tb ? 1 : 0][phase];
      if (pol.cfg >= 0) g.tile = pol.cfg;
      if (pol.split_k > 1 && crct_gemm_splitk_ws_elems(M, N, pol.split_k) <= (int64_t)e->sk_ws_elems[which] &&
          crct_gemm_splitk_tickets(M, N) <= e->sk_tickets) {
        g.split_k = pol.split_k; g.splitk_ws = F(e->sk_ws[which]); g.splitk_cnt = W<uint32_t>(e->sk_cnt[which]);
      }
    }
    fail(crct_gemm_bf16(&g, st));
  }
  // both sides reach this point before either goes on: the two data streams are ordered against each other
  void cross_sync(Run& V) {
    if (!rc) fail(order_streams(e, V.s, s));
    if (!V.rc) V.fail(order_streams(e, s, V.s));
  }
  // y[M][out] = x W^T + b (+ epilogue)
  void lin_fwd(const void* x, int64_t ldx, const LinearP& l, int M, void* y, int64_t ldy, Opt o) {
    o.bias = P(l.b); o.site = l.site;
    gemm(x, ldx, false, PB(l.w), l.in, false, y, ldy, M, l.out, l.in, o);
  }
  // ---- fp8 forward (CrctStepCfg.fp8): the same Linear from the e4m3 copies of its input (scale site `site_in`) and of its
  // weight; optionally also emits the e4m3 copy of its own output (hq, site_out) for the next fp8 GEMM
  // CrctStepCfg.fp8 == 2 (calibration, the dry pass before the first fp8 forward): every producer writes its copy and collects its
  // maximum, the GEMMs themselves still read the bf16 operands -- the maxima are then those of the bf16 forward, not of a forward
  // whose GEMMs ran on unscaled (scale 1) e4m3 inputs
  int f8() const { return (c->fp8 && c->params_fp8 && c->fp8_w_scale && c->fp8_act_scale && c->fp8_act_amax) ? c->fp8 : 0; }
  bool f8_lin(const LinearP& l) const { return f8() && e->wq_slot.count(l.w) != 0; }
  // ---- fp8 backward (CrctStepCfg.fp8_bwd): data gradients dx = dy W of the FFN and attention-output Linears from the e5m2 copy
  // of dy (written by the producing LayerNorm-backward / GELU' epilogue, scale site g) and the TRANSPOSED e4m3 weight shadow.
  // fp8_bwd == 2 (calibration, the first backward pass): the producers collect the gradient maxima, the GEMMs still run in bf16.
  int f8b() const { return (c->fp8 && c->fp8_bwd && c->params_fp8_t && c->fp8_w_scale && c->fp8_grad_scale && c->fp8_grad_amax) ? c->fp8_bwd : 0; }
  bool f8b_lin(const LinearP& l) const { return f8b() && e->wq_slot.count(l.w) != 0; }
  const float* gscale(int g) const { return c->fp8_grad_scale + g; }
  float* gamax(int g) const { return c->fp8_grad_amax + (int64_t)g * CRCT_FP8_AMAX_LANES; }
  // dx[M][in] = dyq[M][out] Wt[in][out]^T (+ epilogue); q_out / g_out: optional e5m2 copy of the result for the next data gradient
  void lin_dgrad_f8(size_t dyq, int g_in, const void* dy_bf16, int64_t lddy, const LinearP& l, int M, void* dx, int64_t lddx, Opt o,
                    size_t q_out = (size_t)-1, int g_out = -1) {
    if (rc) return;
    if (f8b() != 1) {          // calibration pass: the bf16 GEMM, which still emits the e5m2 copy / amax of its result
      o.site = l.site;
      if (g_out >= 0) { o.q_out = W<uint8_t>(q_out); o.q_scale = gscale(g_out); o.q_amax = gamax(g_out); o.ld_q = l.in; }
      gemm(dy_bf16, lddy, false, PB(l.w), l.in, true, dx, lddx, M, l.in, l.out, o);
      return;
    }
    ++tick;
    CrctGemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = W<uint8_t>(dyq); g.B = reinterpret_cast<const uint8_t*>(c->params_fp8_t) + l.w; g.C = dx;
    g.preact_out = o.preact; g.dact_src = o.dact_src; g.addend = o.addend;
    g.lda = l.out; g.ldb = l.out; g.ldc = lddx; g.ld_aux = o.ld_aux; g.ld_add = o.ld_add;
    g.M = M; g.N = l.in; g.K = l.out; g.act = o.act; g.dact = o.dact; g.c_is_f32 = o.f32; g.accumulate = o.acc;
    g.tile = -1; g.alpha = 1.0f; g.seed = c->seed; g.site = l.site;
    g.fp8 = 1 | 2 | 8;                                   // A = e5m2 gradient, B = e4m3 weight; labelled as a data gradient
    if (l.site > 0 && l.site < CRCT_SITE_COUNT && e->policy[l.site][1][phase].cfg >= 0) g.tile = e->policy[l.site][1][phase].cfg;
    g.scale_a = gscale(g_in); g.scale_b = c->fp8_w_scale + e->wq_slot.at(l.w);
    if (g_out >= 0) { g.fp8 |= 4; g.q_out = W<uint8_t>(q_out); g.ld_q = l.in; g.q_scale = gscale(g_out); g.q_amax = gamax(g_out); }
    fail(crct_gemm_bf16(&g, s));
  }
  // x: the bf16 input (leading dimension l.in), read instead of xq by the calibration pass
  void lin_fwd_f8(const void* x, size_t xq, int site_in, const LinearP& l, int M, void* y, int64_t ldy, Opt o, size_t hq = (size_t)-1, int site_out = -1) {
    if (rc) return;
    if (f8() != 1) {           // calibration: the bf16 GEMM, which still emits the e4m3 copy / maximum of its result
      if (site_out >= 0) {
        o.q_out = W<uint8_t>(hq); o.ld_q = l.out; o.q_scale = c->fp8_act_scale + site_out;
        o.q_amax = c->fp8_act_amax + (int64_t)site_out * CRCT_FP8_AMAX_LANES; o.q_e4m3 = true;
      }
      lin_fwd(x, l.in, l, M, y, ldy, o);
      return;
    }
    ++tick;
    CrctGemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = W<uint8_t>(xq); g.B = reinterpret_cast<const uint8_t*>(c->params_fp8) + l.w; g.C = y; g.bias = P(l.b);
    g.preact_out = o.preact; g.dact_src = o.dact_src; g.addend = o.addend;
    g.lda = l.in; g.ldb = l.in; g.ldc = ldy; g.ld_aux = o.ld_aux; g.ld_add = o.ld_add;
    g.M = M; g.N = l.out; g.K = l.in; g.act = o.act; g.dact = o.dact; g.c_is_f32 = o.f32; g.accumulate = o.acc;
    g.tile = -1; g.alpha = 1.0f; g.drop_thr = o.drop.thr; g.drop_scale = o.drop.scale; g.drop_site = o.drop.site; g.seed = c->seed;
    g.addend_f32 = o.add_f32; g.c_cached = o.c_cached;
    g.fp8 = 1; g.scale_a = c->fp8_act_scale + site_in; g.scale_b = c->fp8_w_scale + e->wq_slot.at(l.w); g.site = l.site;
    if (l.site > 0 && l.site < CRCT_SITE_COUNT && e->policy[l.site][0][phase].cfg >= 0) g.tile = e->policy[l.site][0][phase].cfg;
    if (site_out >= 0) { g.q_out = W<uint8_t>(hq); g.ld_q = l.out; g.q_scale = c->fp8_act_scale + site_out; g.q_amax = c->fp8_act_amax + (int64_t)site_out * CRCT_FP8_AMAX_LANES; }
    fail(crct_gemm_bf16(&g, s));
  }
  // dW[out][in] += dy^T x
  // with_bias: also db[out] += column sums of dy.  When the contraction length qualifies for the LDS-DMA kernel the
  // sums come out of the weight-gradient kernel itself (CrctGemmArgs.rowsum_out); otherwise a column-sum launch.
  // q8: the fp8 copies of both operands where the other passes left them (fp8 backward, CrctStepCfg.fp8_wgrad): dy as OCP e5m2
  // with gradient scale site g_dy, x as e4m3 with activation scale site site_x -- the weight gradient then reads half the bytes
  // (gemm.hip, fp8 weight gradients); the bias gradient still sums the bf16 dy.
  void lin_wgrad(const void* dy, int64_t lddy, const void* x, int64_t ldx, const LinearP& l, int M, bool with_bias = false, WgQ8 q8 = WgQ8()) {
    if (rc) return;
    const bool f8w = f8b() == 1 && c->fp8_wgrad && defer_wgrad && q8.g_dy >= 0 && q8.site_x >= 0 && q8.dyq != (size_t)-1 && q8.xq != (size_t)-1 &&
                     l.in % 16 == 0 && l.out % 16 == 0 && lddy % 16 == 0 && ldx % 16 == 0;
    const bool fold = !f8w && with_bias && M % 64 == 0 && l.in % 8 == 0 && l.out % 8 == 0 && lddy % 8 == 0 && ldx % 8 == 0;
    if (with_bias && !fold) bias_grad(dy, lddy, l, M);
    if (rc) return;
    CrctGemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = dy; g.B = x; g.C = G(l.w); g.lda = lddy; g.ldb = ldx; g.ldc = l.in; g.M = l.out; g.N = l.in; g.K = M;
    g.ta = 1; g.tb = 1; g.c_is_f32 = 1; g.accumulate = 1; g.tile = -1; g.alpha = 1.0f; g.site = l.site;
    if (l.site > 0 && l.site < CRCT_SITE_COUNT && e->policy[l.site][2][phase].cfg >= 0) g.tile = e->policy[l.site][2][phase].cfg;
    if (c->wgrad_overwrite && e->wgrad_owned.count(l.w)) {
      if (++e->wgrad_pass[l.w] > 1) { rc = 1; crct_set_error("engine_backward: weight gradient at offset %lld is produced twice in one pass but is listed as owned", (long long)l.w); return; }
      g.accumulate = 0;               // the only producer of this gradient: write it, whatever the buffer held
      if (c->grads_bf16) {            // ... straight into the exchange's bf16 buffer (CrctStepCfg.grads_bf16): the caller packs none of the owned gradients
        if (l.in % 8 != 0 || l.w % 8 != 0) { rc = 1; crct_set_error("engine_backward: owned weight gradient at offset %lld (in = %d) cannot be written as bf16 rows", (long long)l.w, l.in); return; }
        g.C = reinterpret_cast<bf16_t*>(c->grads_bf16) + l.w; g.c_is_f32 = 0;
      }
    }
    if (fold) g.rowsum_out = G(l.b);
    if (f8w) {
      g.A = W<uint8_t>(q8.dyq); g.B = W<uint8_t>(q8.xq); g.fp8 = 1 | 2;
      if (g.tile != 36) g.tile = 37;                      // 2 stages (two workgroups per CU) unless the site policy asks for 3
      g.scale_a = gscale(q8.g_dy); g.scale_b = c->fp8_act_scale + q8.site_x;
      pending_f8.push_back(g);
      return;
    }
    if (!defer_wgrad) { ++tick; fail(crct_gemm_bf16(&g, s)); return; }     // head chain: in order, right now
    // queued also without a side stream (sw == s): the same groups, hence the same kernels and summation orders,
    // in every stream mode -- results stay bit-identical across modes
    pending.push_back(g);
    if (pending.size() == 8) flush_wgrads();
  }
  // dx[M][in] = dy W (+ epilogue)
  void lin_dgrad(const void* dy, int64_t lddy, const LinearP& l, int M, void* dx, int64_t lddx, Opt o) {
    o.site = l.site;
    gemm(dy, lddy, false, PB(l.w), l.in, true, dx, lddx, M, l.in, l.out, o);
  }
  // launch the queued weight-gradient GEMMs on the side stream, ordered after everything enqueued on s so far
  std::vector<CrctGemmArgs> pending_f8;      // the layer's fp8 weight gradients: one grouped launch of their own
  void flush_wgrads() {
    if (rc || (pending.empty() && pending_f8.empty() && pending_fin.empty() && pending_bias.empty())) return;
    if (sw == s) ++tick;
    wgrad_after_main();
    for (const BiasJob& j : pending_bias)
      if (!rc) fail(crct_colsum_bf16(j.dy, j.lddy, j.db, F(sw != s ? colsum_part_w : colsum_part), j.M, j.N, 1, sw));
    pending_bias.clear();
    for (const FinJob& f : pending_fin)
      if (!rc) fail(crct_layernorm_bwd_finalize(f.part, f.dg, f.db, f.dlb, f.M, f.H, 1, sw));
    pending_fin.clear();
    // a persistent grid for the group (gemm.hip, group_grid) where the throttled side stream stays off the critical path: a side
    // stream per data stream, and not the 2560 / 6400-row streams of the long-context configuration, whose data-gradient GEMMs fill the
    // chip themselves (measured there: 11.90 - 11.94 ms with the text stream's groups throttled, 11.96 - 12.04 with both, 11.80 - 11.86 without)
    // (not on the exchange mode's shared side stream, which would become the critical path: 8.0 -> 8.9 - 9.1 ms at 96 workgroups)
    const int target = (sw != s && !pending.empty() && pending[0].K <= e->wgrad_target_rows && !e->one_wgrad_stream) ? e->wgrad_target : 0;
    for (size_t i = 0; i < pending.size() && !rc; i += 8) {
      const int ng = (int)std::min<size_t>(8, pending.size() - i);
      fail(gemm_grouped_checked(pending.data() + i, ng, sw, target));
    }
    pending.clear();
    for (size_t i = 0; i < pending_f8.size() && !rc; i += 8)
      fail(crct_gemm_bf16_grouped(pending_f8.data() + i, (int)std::min<size_t>(8, pending_f8.size() - i), sw));
    pending_f8.clear();
  }
  // db[out] += column sums of dy: queued like the weight gradients (dy stays valid until the layer's flush), so the data
  // stream carries no ordering event per call -- the head chain alone had 13 of them between its 13 small data-gradient GEMMs
  struct BiasJob { const void* dy; int64_t lddy; float* db; int M, N; };
  std::vector<BiasJob> pending_bias;
  void bias_grad(const void* dy, int64_t lddy, const LinearP& l, int M) {
    if (rc) return;
    if (defer_wgrad) { pending_bias.push_back(BiasJob{dy, lddy, G(l.b), M, l.out}); return; }
    wgrad_after_main();
    if (rc) return;
    fail(crct_colsum_bf16(dy, lddy, G(l.b), F(sw != s ? colsum_part_w : colsum_part), M, l.out, 1, sw));
  }
  void ln_fwd(size_t x, const LnP& ln, size_t y, size_t mean, size_t rstd, int M, int H, size_t yq, int site) {
    if (rc) return;
    ++tick;
    CrctLnFwdArgs a = {A(x), P(ln.g), P(ln.b), A(y), F(mean), F(rstd), M, H, 1e-12f, 0, 1.f, 0, c->seed, nullptr, nullptr, nullptr, 0, nullptr};
    if (r32()) { a.x_f32 = 1; a.y_f32 = F(e->res32.at(y)); }
    if (f8()) { a.q_out = W<uint8_t>(yq); a.q_scale = c->fp8_act_scale + site; a.q_amax = c->fp8_act_amax + (int64_t)site * CRCT_FP8_AMAX_LANES; }
    fail(launch_ln_fwd(a, s));
  }
  static int launch_ln_fwd(const CrctLnFwdArgs& a, hipStream_t st) { return crct_layernorm_fwd_args(&a, st); }
  // The fp32 residual stream (CrctStepCfg.residual_fp32): the pre-LayerNorm sums are written and read as fp32, and every block adds
  // the fp32 copy of its input where the producing LayerNorm left one (the embeddings' outputs have none: bf16 there, one rounding)
  bool r32() const { return c->residual_fp32 != 0; }
  void residual(Opt& o, size_t x, int64_t ld) const {
    o.ld_add = ld;
    if (r32()) {
      o.f32 = true; o.c_cached = true;
      auto it = e->res32.find(x);
      if (it != e->res32.end()) { o.addend = F(it->second); o.add_f32 = true; return; }
    }
    o.addend = A(x);
  }
  static int launch_ln_bwd(const CrctLnBwdArgs& a, hipStream_t st) { return crct_layernorm_bwd_rows_args(&a, st); }
  // returns the buffer that holds the gradient of the producing Linear's output
  size_t ln_bwd(size_t dy, size_t x, size_t mean, size_t rstd, const LnP& ln, const LinearP& lin, size_t dres, size_t dlin,
                size_t part, int M, int H, const Drop& dr, size_t dlq = (size_t)-1, int g_site = -1) {
    if (rc) return dres;
    // rows pass on the data stream; the column pass (dgamma, dbeta, bias gradient of the producing Linear) joins the
    // weight-gradient work on the side stream -- `part` belongs to this layer's scratch set
    ++tick;
    CrctLnBwdArgs a = {A(dy), A(x), F(mean), F(rstd), P(ln.g), A(dres), dr.thr ? A(dlin) : nullptr, F(part), M, H,
                       0, 1.f, 0, dr.thr, dr.scale, dr.site, c->seed, nullptr, nullptr, nullptr, r32() ? 1 : 0};
    if (g_site >= 0 && f8b() && f8b_lin(lin)) { a.q_out = W<uint8_t>(dlq); a.q_scale = gscale(g_site); a.q_amax = gamax(g_site); }
    fail(launch_ln_bwd(a, s));
    // the column pass is queued like the weight gradients: ONE ordering event per layer covers all of them
    if (defer_wgrad) pending_fin.push_back(FinJob{F(part), G(ln.g), G(ln.b), G(lin.b), M, H});
    else { wgrad_after_main(); if (!rc) fail(crct_layernorm_bwd_finalize(F(part), G(ln.g), G(ln.b), G(lin.b), M, H, 1, sw)); }
    return dr.thr ? dlin : dres;
  }
  // ctxq / site_ctx: also the e4m3 copy of ctx (the fp8 forward GEMM and weight gradient of the attention-output projection read it)
  void attn_fwd(const bf16_t* q, int64_t ldq, const bf16_t* k, const bf16_t* v, int64_t ldk, const uint8_t* km, bf16_t* ctx,
                int64_t ldo, int B, int heads, int Tq, int Tk, int d, const Drop& dr, size_t ctxq = (size_t)-1, int site_ctx = -1,
                size_t lse = (size_t)-1) {
    if (rc) return;
    ++tick;
    const uint64_t seed = c->seed;
    CrctAttnQuant qz;
    memset(&qz, 0, sizeof(qz));
    if (lse != (size_t)-1) qz.row_lse = F(lse);
    if (ctxq != (size_t)-1 && site_ctx >= 0) {
      qz.ctx_q = W<uint8_t>(ctxq); qz.ctx_scale = c->fp8_act_scale + site_ctx; qz.ctx_amax = c->fp8_act_amax + (int64_t)site_ctx * CRCT_FP8_AMAX_LANES;
    }
    fail(crct_attention_fwd_q(q, k, v, km, ctx, B, heads, Tq, Tk, d, ldq, ldk, ldk, ldo, dr.thr, dr.scale, dr.site, seed, &qz, s));
  }
  // dqq / g_dq, dkq / dvq / g_dkv: also e5m2 copies of dq and of dk / dv (columns of fused dqkv buffers: one scale site per buffer)
  void attn_bwd(const bf16_t* q, int64_t ldq, const bf16_t* k, const bf16_t* v, int64_t ldk, const uint8_t* km,
                const bf16_t* dctx, int64_t ldo, bf16_t* dq, int64_t lddq, bf16_t* dk, bf16_t* dv, int64_t lddk, int B,
                int heads, int Tq, int Tk, int d, const Drop& dr, uint8_t* dqq = nullptr, int g_dq = -1, uint8_t* dkq = nullptr,
                uint8_t* dvq = nullptr, int g_dkv = -1, size_t lse = (size_t)-1, const bf16_t* ctx = nullptr, int64_t ldc = 0) {
    if (rc) return;
    ++tick;
    const uint64_t seed = c->seed;
    CrctAttnQuant qz;
    memset(&qz, 0, sizeof(qz));
    if (lse != (size_t)-1 && ctx) { qz.row_lse = F(lse); qz.ctx = ctx; qz.ld_ctx = ldc; }
    if (dqq && g_dq >= 0) { qz.dq_q = dqq; qz.dq_scale = gscale(g_dq); qz.dq_amax = gamax(g_dq); }
    if (dkq && dvq && g_dkv >= 0) { qz.dk_q = dkq; qz.dv_q = dvq; qz.dkv_scale = gscale(g_dkv); qz.dkv_amax = gamax(g_dkv); }
    fail(crct_attention_bwd_q(q, k, v, km, dctx, dq, dk, dv, B, heads, Tq, Tk, d, ldq, ldk, ldk, ldo, lddq, lddk, lddk, dr.thr, dr.scale,
                              dr.site, seed, &qz, s));
  }
  // the attention kernels that write fp8 copies cover this shape (MFMA kernels: include/crct_hip.h, CrctAttnQuant)
  static bool attn_q_ok(int Tq, int Tk, int d) { return crct_attention_quant_ok(Tq, Tk, d) != 0; }

  // ---------------------------------------------------------------- sub-blocks
  // a = LN(dropout(dense(ctx)) + x)          vilbert.py:424-428 / :555-559 / :749-756
  // ctxq / site_ctx: the e4m3 copy of ctx the attention kernel wrote (-1: none, the projection runs in bf16)
  void proj_fwd(const ProjP& p, const ProjA& a, size_t ctx, size_t x, int M, const Drop& dr, size_t ctxq = (size_t)-1, int site_ctx = -1) {
    Opt o; o.drop = dr; residual(o, x, p.dense.out);
    if (site_ctx >= 0 && f8_lin(p.dense)) lin_fwd_f8(A(ctx), ctxq, site_ctx, p.dense, M, A(a.s), p.dense.out, o);
    else lin_fwd(A(ctx), p.dense.in, p.dense, M, A(a.s), p.dense.out, o);
    ln_fwd(a.s, p.ln, a.a, a.mean, a.rstd, M, p.dense.out, a.aq, a.site_a);
  }
  // in: g = grad of a.  out: dres (residual gradient), dctx.  Parameter gradients accumulated.
  void proj_bwd(const ProjP& p, const ProjA& a, size_t ctx, size_t g, size_t dres, size_t dlin, size_t dctx, size_t part, int M, const Drop& dr,
                size_t dlq, size_t ctxq = (size_t)-1, int site_ctx = -1) {
    const size_t dl = ln_bwd(g, a.s, a.mean, a.rstd, p.ln, p.dense, dres, dlin, part, M, p.dense.out, dr, dlq, a.g_dl);
    WgQ8 w8;
    if (site_ctx >= 0 && f8b_lin(p.dense)) { w8.dyq = dlq; w8.g_dy = a.g_dl; w8.xq = ctxq; w8.site_x = site_ctx; }
    lin_wgrad(A(dl), p.dense.out, A(ctx), p.dense.in, p.dense, M, false, w8);
    if (f8b_lin(p.dense)) lin_dgrad_f8(dlq, a.g_dl, A(dl), p.dense.out, p.dense, M, A(dctx), p.dense.in, Opt());
    else lin_dgrad(A(dl), p.dense.out, p.dense, M, A(dctx), p.dense.in, Opt());
    if (e->wgrad_flush & 2) flush_wgrads();      // bit 1: the projection's weight gradient leaves behind its data gradient
  }
  // y = LN(dropout(down(gelu(up(x)))) + x)   vilbert.py:454-471 / :585-602 / :782-786
  // xq / site_x: the e4m3 copy of x and its scale site (the LayerNorm that produced x wrote both)
  void ffn_fwd(const FfnP& p, const FfnA& a, size_t x, size_t xq, int site_x, int M, const Drop& dr) {
    Opt o; o.preact = A(a.u); o.ld_aux = p.up.out; o.act = ACT_GELU;
    const bool q_up = f8_lin(p.up), q_dn = f8_lin(p.down);
    if (q_up) lin_fwd_f8(A(x), xq, site_x, p.up, M, A(a.h), p.up.out, o, a.hq, q_dn ? a.site_h : -1);
    else lin_fwd(A(x), p.up.in, p.up, M, A(a.h), p.up.out, o);
    Opt o2; o2.drop = dr; residual(o2, x, p.down.out);
    if (q_up && q_dn) lin_fwd_f8(A(a.h), a.hq, a.site_h, p.down, M, A(a.s), p.down.out, o2);
    else lin_fwd(A(a.h), p.down.in, p.down, M, A(a.s), p.down.out, o2);
    ln_fwd(a.s, p.ln, a.y, a.mean, a.rstd, M, p.down.out, a.yq, a.site_y);
  }
  // in: g = grad of a.y.  out: gx = grad of x.
  // xq / site_x: the e4m3 copy of x the forward pass read (fp8 weight gradient of the up projection)
  void ffn_bwd(const FfnP& p, const FfnA& a, size_t x, size_t xq, int site_x, size_t g, size_t gx, const StreamScratch& sc, int M, const Drop& dr) {
    const int H = p.down.out, I = p.up.out;
    const size_t dl = ln_bwd(g, a.s, a.mean, a.rstd, p.ln, p.down, sc.dres_a, sc.dlin_a, sc.part_a, M, H, dr, sc.dlq_a, a.g_dl);
    const bool q_dn = f8b_lin(p.down), q_up = f8b_lin(p.up);
    WgQ8 w_dn, w_up;
    if (q_dn && f8_lin(p.up) && f8_lin(p.down)) { w_dn.dyq = sc.dlq_a; w_dn.g_dy = a.g_dl; w_dn.xq = a.hq; w_dn.site_x = a.site_h; }      // hq exists when both forward GEMMs ran in fp8
    if (q_dn && q_up && f8_lin(p.up)) { w_up.dyq = sc.duq; w_up.g_dy = a.g_du; w_up.xq = xq; w_up.site_x = site_x; }
    lin_wgrad(A(dl), H, A(a.h), I, p.down, M, false, w_dn);
    Opt o; o.dact_src = A(a.u); o.dact = ACT_GELU; o.ld_aux = I;
    if (q_dn) lin_dgrad_f8(sc.dlq_a, a.g_dl, A(dl), H, p.down, M, A(sc.du), I, o, sc.duq, q_up ? a.g_du : -1);
    else lin_dgrad(A(dl), H, p.down, M, A(sc.du), I, o);
    lin_wgrad(A(sc.du), I, A(x), H, p.up, M, true, w_up);
    Opt o2; o2.addend = A(sc.dres_a); o2.ld_add = H;
    if (q_dn && q_up) lin_dgrad_f8(sc.duq, a.g_du, A(sc.du), I, p.up, M, A(gx), H, o2);
    else lin_dgrad(A(sc.du), I, p.up, M, A(gx), H, o2);
    // wgrad_flush & 1: the FFN block's two weight gradients (and the LayerNorm column pass) leave for the side stream HERE, behind
    // the FFN-up data gradient, instead of at the end of the layer with the projection's and the QKV's: the grouped launch then runs
    // beside this layer's LayerNorm backward / attention-output dgrad / attention backward / QKV dgrad -- kernels of <= 156
    // workgroups -- and not beside the next layer's FFN data gradients.  Same kernels per problem, same sums: bit-identical.
    if (e->wgrad_flush & 1) flush_wgrads();
  }

  // ---------------------------------------------------------------- self-attention layer
  void self_fwd(const SelfLayerP& p, const SelfLayerA& a, size_t x, size_t xq, int site_x, const uint8_t* km, int B, int T) {
    const int M = B * T, H = p.H, d = H / p.heads;
    if (f8_lin(p.qkv)) lin_fwd_f8(A(x), xq, site_x, p.qkv, M, A(a.qkv), 3 * H, Opt());
    else lin_fwd(A(x), p.qkv.in, p.qkv, M, A(a.qkv), 3 * H, Opt());
    const bool cq = f8_lin(p.proj.dense) && attn_q_ok(T, T, d);
    attn_fwd(A(a.qkv), 3 * H, A(a.qkv) + H, A(a.qkv) + 2 * H, 3 * H, km, A(a.ctx), H, B, p.heads, T, T, d, drop(p.p_attn, p.site),
             cq ? a.ctxq : (size_t)-1, cq ? a.site_ctx : -1, a.lse);
    proj_fwd(p.proj, a.proj, a.ctx, x, M, drop(p.p_hid, p.site + 1), a.ctxq, cq ? a.site_ctx : -1);
    ffn_fwd(p.ffn, a.ffn, a.proj.a, a.proj.aq, a.proj.site_a, M, drop(p.p_hid, p.site + 2));
  }
  // xq / site_x: the e4m3 copy of the layer input (fp8 weight gradient of the QKV projection)
  void self_bwd(const SelfLayerP& p, const SelfLayerA& a, size_t x, size_t xq, int site_x, size_t g, size_t gx, const uint8_t* km, int B, int T) {
    const int M = B * T, H = p.H, d = H / p.heads;
    const StreamScratch& sc = layer_begin();
    ffn_bwd(p.ffn, a.ffn, a.proj.a, a.proj.aq, a.proj.site_a, g, sc.gc, sc, M, drop(p.p_hid, p.site + 2));
    const bool aq = attn_q_ok(T, T, d);
    const bool cq = aq && f8_lin(p.proj.dense);                 // the forward pass wrote ctxq
    const bool gq = aq && f8b_lin(p.qkv) && f8_lin(p.qkv);      // e5m2 copy of dqkv: fp8 data and weight gradient of the QKV projection
    proj_bwd(p.proj, a.proj, a.ctx, sc.gc, sc.dres_b, sc.dlin_b, sc.dctx, sc.part_b, M, drop(p.p_hid, p.site + 1), sc.dlq_b, a.ctxq, cq ? a.site_ctx : -1);
    uint8_t* dq8 = gq ? W<uint8_t>(sc.dqkvq) : nullptr;
    attn_bwd(A(a.qkv), 3 * H, A(a.qkv) + H, A(a.qkv) + 2 * H, 3 * H, km, A(sc.dctx), H, A(sc.dqkv), 3 * H, A(sc.dqkv) + H,
             A(sc.dqkv) + 2 * H, 3 * H, B, p.heads, T, T, d, drop(p.p_attn, p.site), dq8, a.g_dqkv, gq ? dq8 + H : nullptr,
             gq ? dq8 + 2 * H : nullptr, a.g_dqkv, a.lse, A(a.ctx), H);
    WgQ8 w8;
    if (gq) { w8.dyq = sc.dqkvq; w8.g_dy = a.g_dqkv; w8.xq = xq; w8.site_x = site_x; }
    lin_wgrad(A(sc.dqkv), 3 * H, A(x), H, p.qkv, M, true, w8);
    Opt o; o.addend = A(sc.dres_b); o.ld_add = H;
    if (gq) lin_dgrad_f8(sc.dqkvq, a.g_dqkv, A(sc.dqkv), 3 * H, p.qkv, M, A(gx), H, o);
    else lin_dgrad(A(sc.dqkv), 3 * H, p.qkv, M, A(gx), H, o);
    layer_end();
  }

  // ---------------------------------------------------------------- connection layer (vilbert.py:774-788)
  // `this` drives the TEXT stream, `V` the VISUAL stream (they may share one HIP stream).
  void conn_fwd(Run& V, const ConnLayerP& p, const ConnLayerA& a, size_t xv, size_t xvq, int site_v, size_t xt, size_t xtq, int site_t) {
    const CrctModelDims& D = e->d;
    const int B = b->B, Mv = B * b->V, Mt = B * b->T, Hb = D.Hb, d = Hb / D.b_heads;
    if (V.f8_lin(p.qkv1)) V.lin_fwd_f8(V.A(xv), xvq, site_v, p.qkv1, Mv, V.A(a.qkv1), 3 * Hb, Opt());
    else V.lin_fwd(V.A(xv), p.qkv1.in, p.qkv1, Mv, V.A(a.qkv1), 3 * Hb, Opt());  // query1/key1/value1  :662-664
    if (f8_lin(p.qkv2)) lin_fwd_f8(A(xt), xtq, site_t, p.qkv2, Mt, A(a.qkv2), 3 * Hb, Opt());
    else lin_fwd(A(xt), p.qkv2.in, p.qkv2, Mt, A(a.qkv2), 3 * Hb, Opt());      // query2/key2/value2  :673-675
    cross_sync(V);                                    // text needs k1, v1; visual needs k2, v2
    // text queries over visual keys/values -> ctx1 [B,T,Hb]  :684-701 (dropout1 = v_attention prob)
    const bool aq = attn_q_ok(b->T, b->V, d) && attn_q_ok(b->V, b->T, d);
    const bool cq1 = aq && f8_lin(p.proj_t.dense), cq2 = aq && V.f8_lin(p.proj_v.dense);
    attn_fwd(A(a.qkv2), 3 * Hb, A(a.qkv1) + Hb, A(a.qkv1) + 2 * Hb, 3 * Hb, b->image_keymask, A(a.ctx1), Hb, B, D.b_heads,
             b->T, b->V, d, drop(D.p_v_attn, p.site), cq1 ? a.ctx1q : (size_t)-1, cq1 ? a.site_ctx1 : -1, a.lse1);
    // visual queries over text keys/values -> ctx2 [B,V,Hb]  :704-723
    V.attn_fwd(A(a.qkv1), 3 * Hb, A(a.qkv2) + Hb, A(a.qkv2) + 2 * Hb, 3 * Hb, b->text_keymask, A(a.ctx2), Hb, B, D.b_heads,
               b->V, b->T, d, drop(D.p_attn, p.site + 1), cq2 ? a.ctx2q : (size_t)-1, cq2 ? a.site_ctx2 : -1, a.lse2);
    // cross wiring :780 -- visual stream takes ctx2, text stream takes ctx1
    V.proj_fwd(p.proj_v, a.proj_v, a.ctx2, xv, Mv, drop(D.p_v_hidden, p.site + 2), a.ctx2q, cq2 ? a.site_ctx2 : -1);
    proj_fwd(p.proj_t, a.proj_t, a.ctx1, xt, Mt, drop(D.p_hidden, p.site + 3), a.ctx1q, cq1 ? a.site_ctx1 : -1);
    V.ffn_fwd(p.ffn_v, a.ffn_v, a.proj_v.a, a.proj_v.aq, a.proj_v.site_a, Mv, drop(D.p_v_hidden, p.site + 4));
    ffn_fwd(p.ffn_t, a.ffn_t, a.proj_t.a, a.proj_t.aq, a.proj_t.site_a, Mt, drop(D.p_hidden, p.site + 5));
  }
  // x*q / site_*: the e4m3 copies of the two layer inputs (fp8 weight gradients of the QKV projections)
  void conn_bwd(Run& V, const ConnLayerP& p, const ConnLayerA& a, size_t xv, size_t xvq, int site_v, size_t xt, size_t xtq, int site_t, size_t gv,
                size_t gt, size_t gxv, size_t gxt) {
    const CrctModelDims& D = e->d;
    hipEvent_t free_v = V.set_free[V.parity], free_t = set_free[parity];      // "the last readers of this scratch set are done"
    const StreamScratch& sv = V.layer_begin(); const StreamScratch& st = layer_begin();
    const int B = b->B, Mv = B * b->V, Mt = B * b->T, Hb = D.Hb, d = Hb / D.b_heads;
    V.ffn_bwd(p.ffn_v, a.ffn_v, a.proj_v.a, a.proj_v.aq, a.proj_v.site_a, gv, sv.gc, sv, Mv, drop(D.p_v_hidden, p.site + 4));
    ffn_bwd(p.ffn_t, a.ffn_t, a.proj_t.a, a.proj_t.aq, a.proj_t.site_a, gt, st.gc, st, Mt, drop(D.p_hidden, p.site + 5));
    const bool aq = attn_q_ok(b->T, b->V, d) && attn_q_ok(b->V, b->T, d);
    const bool cq1 = aq && f8_lin(p.proj_t.dense), cq2 = aq && V.f8_lin(p.proj_v.dense);
    // e5m2 copies of the two fused dqkv buffers: only when BOTH QKV projections run their gradients in fp8 (each buffer is written by
    // both attention kernels)
    const bool gq = aq && f8b_lin(p.qkv1) && f8b_lin(p.qkv2) && f8_lin(p.qkv1) && f8_lin(p.qkv2);
    V.proj_bwd(p.proj_v, a.proj_v, a.ctx2, sv.gc, sv.dres_b, sv.dlin_b, sv.dctx, sv.part_b, Mv, drop(D.p_v_hidden, p.site + 2), sv.dlq_b, a.ctx2q,
               cq2 ? a.site_ctx2 : -1);   // dctx2 [Mv,Hb]
    proj_bwd(p.proj_t, a.proj_t, a.ctx1, st.gc, st.dres_b, st.dlin_b, st.dctx, st.part_b, Mt, drop(D.p_hidden, p.site + 3), st.dlq_b, a.ctx1q,
             cq1 ? a.site_ctx1 : -1);       // dctx1 [Mt,Hb]
    // each attention backward also writes into the OTHER stream's dqkv scratch, which the layer that used this scratch set
    // last (its dgrad, and its weight-gradient GEMMs on the side stream) may still be reading.  With side streams that
    // layer's end is marked by the set's free event (recorded on the side stream behind everything the layer enqueued):
    // each data stream also waits for the OTHER side's event -- long signalled, so the wait is free, where a fresh
    // two-way hand-off costs 10+ us on the critical stream (tools/handoff_lab.cpp).  Without side streams: a full ordering.
    if (sw != s && V.sw != V.s) {
      if (free_v && !rc && hipStreamWaitEvent(s, free_v, 0) != hipSuccess) { crct_set_error("engine: stream wait failed"); rc = 1; }
      if (free_t && !V.rc && hipStreamWaitEvent(V.s, free_t, 0) != hipSuccess) { crct_set_error("engine: stream wait failed"); V.rc = 1; }
    } else cross_sync(V);
    // ctx1 = attn(q2, k1, v1): dq2 -> dqkv2[:, 0:Hb], dk1/dv1 -> dqkv1[:, Hb:3Hb]            (text stream)
    uint8_t* tq8 = gq ? W<uint8_t>(st.dqkvq) : nullptr;      // text buffer (site g_dqkv2), visual buffer (site g_dqkv1)
    uint8_t* vq8 = gq ? W<uint8_t>(sv.dqkvq) : nullptr;
    attn_bwd(A(a.qkv2), 3 * Hb, A(a.qkv1) + Hb, A(a.qkv1) + 2 * Hb, 3 * Hb, b->image_keymask, A(st.dctx), Hb, A(st.dqkv),
             3 * Hb, A(sv.dqkv) + Hb, A(sv.dqkv) + 2 * Hb, 3 * Hb, B, D.b_heads, b->T, b->V, d, drop(D.p_v_attn, p.site),
             tq8, a.g_dqkv2, gq ? vq8 + Hb : nullptr, gq ? vq8 + 2 * Hb : nullptr, a.g_dqkv1, a.lse1, A(a.ctx1), Hb);
    // ctx2 = attn(q1, k2, v2): dq1 -> dqkv1[:, 0:Hb], dk2/dv2 -> dqkv2[:, Hb:3Hb]            (visual stream)
    V.attn_bwd(A(a.qkv1), 3 * Hb, A(a.qkv2) + Hb, A(a.qkv2) + 2 * Hb, 3 * Hb, b->text_keymask, A(sv.dctx), Hb, A(sv.dqkv),
               3 * Hb, A(st.dqkv) + Hb, A(st.dqkv) + 2 * Hb, 3 * Hb, B, D.b_heads, b->V, b->T, d, drop(D.p_attn, p.site + 1),
               vq8, a.g_dqkv1, gq ? tq8 + Hb : nullptr, gq ? tq8 + 2 * Hb : nullptr, a.g_dqkv2, a.lse2, A(a.ctx2), Hb);
    // each stream's dqkv buffer has been written by BOTH attention backward kernels
    cross_sync(V);
    WgQ8 wv, wt;
    if (gq) { wv.dyq = sv.dqkvq; wv.g_dy = a.g_dqkv1; wv.xq = xvq; wv.site_x = site_v; wt.dyq = st.dqkvq; wt.g_dy = a.g_dqkv2; wt.xq = xtq; wt.site_x = site_t; }
    V.lin_wgrad(A(sv.dqkv), 3 * Hb, A(xv), D.Hv, p.qkv1, Mv, true, wv);
    Opt ov; ov.addend = A(sv.dres_b); ov.ld_add = D.Hv;
    if (gq) V.lin_dgrad_f8(sv.dqkvq, a.g_dqkv1, A(sv.dqkv), 3 * Hb, p.qkv1, Mv, A(gxv), D.Hv, ov);
    else V.lin_dgrad(A(sv.dqkv), 3 * Hb, p.qkv1, Mv, A(gxv), D.Hv, ov);
    lin_wgrad(A(st.dqkv), 3 * Hb, A(xt), D.H, p.qkv2, Mt, true, wt);
    Opt ot; ot.addend = A(st.dres_b); ot.ld_add = D.H;
    if (gq) lin_dgrad_f8(st.dqkvq, a.g_dqkv2, A(st.dqkv), 3 * Hb, p.qkv2, Mt, A(gxt), D.H, ot);
    else lin_dgrad(A(st.dqkv), 3 * Hb, p.qkv2, Mt, A(gxt), D.H, ot);
    V.layer_end();
    layer_end();
  }

  // ---------------------------------------------------------------- embeddings
  void embed_text_fwd() {
    const CrctModelDims& D = e->d;
    const Drop dt = drop(D.p_hidden, 1);       // hidden_dropout_prob (vilbert.py:315)
    if (!rc) fail(crct_embed_text_fwd(b->tokens, b->segments, b->loc, P(e->et.word), P(e->et.pos), P(e->et.type), P(e->et.wloc),
                                      P(e->et.bloc), P(e->et.ln.g), P(e->et.ln.b), A(e->eta.sum), A(e->eta.y), F(e->eta.mean),
                                      F(e->eta.rstd), b->B, b->T, D.H, D.n_pos, 1e-12f, dt.thr, dt.scale, dt.site, c->seed, s));
    if (!rc && f8()) fail(crct_fp8_quantize_bf16(A(e->eta.y), W<uint8_t>(e->eta.yq), c->fp8_act_scale + e->eta.site,
                                                 c->fp8_act_amax + (int64_t)e->eta.site * CRCT_FP8_AMAX_LANES, (int64_t)b->B * b->T * D.H, s));
  }
  void embed_image_fwd() {
    const CrctModelDims& D = e->d;
    const int Mv = b->B * b->V;
    const Drop dv = drop(D.p_hidden, 2);       // also the TEXT probability (vilbert.py:1470)
    if (!rc) fail(b->image_feat_bf16 ? crct_softmax_rows_bf16_bf16(b->image_feat, A(e->eva.soft), Mv, D.Fv, s)
                                     : crct_softmax_rows_f32_bf16((const float*)b->image_feat, A(e->eva.soft), Mv, D.Fv, s));
    lin_fwd(A(e->eva.soft), D.Fv, e->ev.img, Mv, A(e->eva.lin), D.Hv, Opt());
    if (!rc) fail(crct_embed_image_fwd(A(e->eva.lin), b->image_loc, b->image_target, P(e->ev.wloc), P(e->ev.bloc), P(e->ev.color),
                                       P(e->ev.ln.g), P(e->ev.ln.b), A(e->eva.sum), A(e->eva.y), F(e->eva.mean), F(e->eva.rstd),
                                       Mv, D.Hv, 1e-12f, dv.thr, dv.scale, dv.site, c->seed, s));
    if (!rc && f8()) fail(crct_fp8_quantize_bf16(A(e->eva.y), W<uint8_t>(e->eva.yq), c->fp8_act_scale + e->eva.site,
                                                 c->fp8_act_amax + (int64_t)e->eva.site * CRCT_FP8_AMAX_LANES, (int64_t)Mv * D.Hv, s));
  }
  void embed_text_bwd(size_t gt) {
    const CrctModelDims& D = e->d;
    const Drop dt = drop(D.p_hidden, 1);
    layer_begin();
    ++tick;
    // the word-table index (first / last row per token id) is the ENGINE's memory: it must be zero whenever the call starts -- allocated and
    // zeroed once, kept zero by the kernels themselves (include/crct_hip.h) -- which a caller-provided workspace cannot promise
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (!rc && !e->word_index && hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) {      // never allocate inside a capture
      if (hipMalloc((void**)&e->word_index, (size_t)2 * D.vocab * sizeof(int32_t)) != hipSuccess) { e->word_index = nullptr; (void)hipGetLastError(); }
      else if (hipMemsetAsync(e->word_index, 0, (size_t)2 * D.vocab * sizeof(int32_t), s) != hipSuccess) { crct_set_error("engine: memset of the word index failed"); rc = 1; }
    }
    if (!rc) fail(crct_embed_text_bwd_indexed(A(gt), A(e->eta.sum), F(e->eta.mean), F(e->eta.rstd), b->tokens, b->segments, b->loc,
                                              P(e->et.ln.g), G(e->et.word), G(e->et.pos), G(e->et.type), G(e->et.wloc), G(e->et.bloc),
                                              G(e->et.ln.g), G(e->et.ln.b), F(partials), b->B, b->T, D.H, D.n_pos, dt.thr, dt.scale,
                                              dt.site, c->seed, F(e->embed_rows[0]), W<int32_t>(e->embed_idx[0]), D.n_types, e->word_index,
                                              e->word_index ? D.vocab : 0, s));
  }
  void embed_image_bwd(size_t gv) {
    const CrctModelDims& D = e->d;
    const int Mv = b->B * b->V;
    const Drop dv = drop(D.p_hidden, 2);
    const StreamScratch& sc = layer_begin();
    ++tick;
    if (!rc) fail(crct_embed_image_bwd(A(gv), A(e->eva.sum), F(e->eva.mean), F(e->eva.rstd), b->image_loc, b->image_target,
                                       P(e->ev.ln.g), A(sc.gc), G(e->ev.color), G(e->ev.wloc), G(e->ev.bloc), G(e->ev.img.b),
                                       G(e->ev.ln.g), G(e->ev.ln.b), F(partials), Mv, D.Hv, dv.thr, dv.scale, dv.site,
                                       c->seed, F(e->embed_rows[1]), W<int32_t>(e->embed_idx[1]), D.n_color, s));
    lin_wgrad(A(sc.gc), D.Hv, A(e->eva.soft), D.Fv, e->ev.img, Mv);   // no dgrad: features are inputs
  }

  // ---------------------------------------------------------------- heads
  void pipe_fwd(const LinearP* l, const bf16_t* x0, int64_t ldx0, const size_t* acts, bf16_t* out_last, int64_t ld_last, int B) {
    // Linear+LeakyReLU x3, then a plain Linear into `out_last` (regressor.py:8-28)
    Opt o; o.act = ACT_LEAKY;
    lin_fwd(x0, ldx0, l[0], B, A(acts[0]), l[0].out, o);
    lin_fwd(A(acts[0]), l[0].out, l[1], B, A(acts[1]), l[1].out, o);
    lin_fwd(A(acts[1]), l[1].out, l[2], B, A(acts[2]), l[2].out, o);
    lin_fwd(A(acts[2]), l[2].out, l[3], B, out_last, ld_last, Opt());
  }
  // backward of a pipe: du3 = grad of the last Linear's output (ld = ld3); writes dx0 (+= if acc).  gb = three gradient
  // buffers of this pipe alone (nothing is recycled: the weight-gradient GEMMs that read them run later, on the side stream)
  void pipe_bwd(const LinearP* l, const bf16_t* x0, int64_t ldx0, const size_t* acts, const bf16_t* du3, int64_t ld3,
                bf16_t* dx0, int64_t lddx0, bool acc0, int B, const size_t* gb) {
    bias_grad(du3, ld3, l[3], B);
    lin_wgrad(du3, ld3, A(acts[2]), l[2].out, l[3], B);
    Opt o; o.dact = ACT_LEAKY; o.dact_src = A(acts[2]); o.ld_aux = l[2].out;
    lin_dgrad(du3, ld3, l[3], B, A(gb[0]), l[2].out, o);                      // du2
    bias_grad(A(gb[0]), l[2].out, l[2], B);
    lin_wgrad(A(gb[0]), l[2].out, A(acts[1]), l[1].out, l[2], B);
    o.dact_src = A(acts[1]); o.ld_aux = l[1].out;
    lin_dgrad(A(gb[0]), l[2].out, l[2], B, A(gb[1]), l[1].out, o);            // du1
    bias_grad(A(gb[1]), l[1].out, l[1], B);
    lin_wgrad(A(gb[1]), l[1].out, A(acts[0]), l[0].out, l[1], B);
    o.dact_src = A(acts[0]); o.ld_aux = l[0].out;
    lin_dgrad(A(gb[1]), l[1].out, l[1], B, A(gb[2]), l[0].out, o);            // du0
    bias_grad(A(gb[2]), l[0].out, l[0], B);
    lin_wgrad(A(gb[2]), l[0].out, x0, ldx0, l[0], B);
    if (dx0) { Opt od; od.acc = acc0; lin_dgrad(A(gb[2]), l[0].out, l[0], B, dx0, lddx0, od); }
  }

  // Heads, forward.  The pooler and the regressor pipe of a stream only need that stream's last hidden states, so each data
  // stream runs its own branch (this = text or visual Run) before the two join for the fusion MLP and the loss kernel.
  void heads_branch_fwd(bool visual, size_t seq) {
    const CrctModelDims& D = e->d;
    const int B = b->B;
    const int64_t ld = visual ? (int64_t)b->V * D.Hv : (int64_t)b->T * D.H;  // CLS / IMG rows: hidden_states[:, 0]
    Opt orelu; orelu.act = ACT_RELU;
    if (visual) {
      lin_fwd(A(seq), ld, e->v_pool, B, A(e->ha.pooled_v), D.Hb, orelu);     // vilbert.py:970-976
      pipe_fwd(e->vp, A(seq), ld, e->ha.v, A(e->ha.cat), 512, B);            // regressor on the raw IMG state; cat = (hv, hw): regressor.py:39-41
    } else {
      lin_fwd(A(seq), ld, e->t_pool, B, A(e->ha.pooled_t), D.Hb, orelu);     // vilbert.py:955-961
      pipe_fwd(e->tp, A(seq), ld, e->ha.t, A(e->ha.cat) + 256, 512, B);
    }
  }
  void fill_head_args(CrctHeadArgs& h, float* logits, float* reg, float* stats, bool with_grad) {
    const CrctModelDims& D = e->d;
    memset(&h, 0, sizeof(h));
    h.pooled_t = A(e->ha.pooled_t); h.pooled_v = A(e->ha.pooled_v); h.fus_h = A(e->ha.f[2]);
    h.w_cls = P(e->cls.w); h.b_cls = P(e->cls.b); h.w_f6 = P(e->fu[3].w); h.b_f6 = P(e->fu[3].b);
    h.R = b->R; h.labels = b->labels; h.logits = logits; h.reg = reg; h.stats = stats; h.scratch = F(e->ha.scratch);
    if (with_grad && g32) {
      h.d_pooled_t = A(e->ha.d_pt); h.d_pooled_v = A(e->ha.d_pv); h.d_fus_h = A(e->ha.g[0]);
      h.d_w_cls = G(e->cls.w); h.d_b_cls = G(e->cls.b); h.d_w_f6 = G(e->fu[3].w); h.d_b_f6 = G(e->fu[3].b);
    }
    h.g_nsp_dev = c->g_nsp_dev; h.g_reg_dev = c->g_reg_dev; h.g_loss_dev = c->g_loss_dev;
    h.B = b->B; h.Hb = D.Hb; h.fusion_sum = D.fusion_sum; h.use_l1 = c->use_l1; h.kind_l1 = c->kind_l1;
    h.tol_margin = c->tol_margin; h.nsp_coeff = c->nsp_coeff; h.reg_coeff = c->reg_coeff; h.grad_scale = c->grad_scale;
    const Drop dc = drop(D.p_cls, 3);
    h.drop_thr = dc.thr; h.drop_scale = dc.scale; h.drop_site = dc.site; h.seed = c->seed;
  }
  void heads_tail_fwd(float* logits, float* reg, float* stats) {
    const int B = b->B;
    Opt o; o.act = ACT_LEAKY;
    lin_fwd(A(e->ha.cat), 512, e->fu[0], B, A(e->ha.f[0]), 512, o);
    lin_fwd(A(e->ha.f[0]), 512, e->fu[1], B, A(e->ha.f[1]), 256, o);
    lin_fwd(A(e->ha.f[1]), 256, e->fu[2], B, A(e->ha.f[2]), 256, o);
    if (rc) return;
    CrctHeadArgs h;
    fill_head_args(h, logits, reg, stats, false);
    fail(crct_head_loss(&h, s));
  }
  // Heads, backward (this = text stream, V = visual stream): fills the CLS / IMG rows of the running activation gradients
  // (other rows zero).  On the critical chain are only the loss kernel and the 13 small data-gradient GEMMs -- the visual
  // pooler / pipe on the visual stream beside the text ones; the 13 weight-gradient GEMMs and bias column sums are queued
  // for the side streams like those of every encoder layer (every gradient buffer below has ONE producer and is not
  // recycled within the call).
  void heads_bwd(Run& V, size_t seq_t, size_t seq_v, size_t gt, size_t gv, float* logits, float* reg, float* stats) {
    const CrctModelDims& D = e->d;
    const int B = b->B;
    const int64_t ldt = (int64_t)b->T * D.H, ldv = (int64_t)b->V * D.Hv;
    const size_t* g = e->ha.g;
    // the loss kernel is re-run with gradient outputs enabled (cheap: B rows) so that forward can be called alone for evaluation
    {
      CrctHeadArgs h;
      fill_head_args(h, logits, reg, stats, true);
      if (!rc) fail(crct_head_loss(&h, s));
    }
    if (rc) return;
    if (!V.rc) V.fail(order_streams(e, s, V.s));                              // d_pooled_v is ready
    if (hipMemsetAsync(A(gt), 0, (size_t)B * b->T * D.H * 2, s) != hipSuccess ||
        hipMemsetAsync(V.A(gv), 0, (size_t)B * b->V * D.Hv * 2, V.s) != hipSuccess) { crct_set_error("engine: memset failed"); rc = 1; return; }
    ++tick; ++V.tick;
    // poolers (gradients already w.r.t. the pre-activations)
    bias_grad(A(e->ha.d_pt), D.Hb, e->t_pool, B);
    lin_wgrad(A(e->ha.d_pt), D.Hb, A(seq_t), ldt, e->t_pool, B);
    lin_dgrad(A(e->ha.d_pt), D.Hb, e->t_pool, B, A(gt), ldt, Opt());
    V.bias_grad(V.A(e->ha.d_pv), D.Hb, e->v_pool, B);
    V.lin_wgrad(V.A(e->ha.d_pv), D.Hb, V.A(seq_v), ldv, e->v_pool, B);
    V.lin_dgrad(V.A(e->ha.d_pv), D.Hb, e->v_pool, B, V.A(gv), ldv, Opt());
    // fusion MLP: g[0] = grad of fusion.4's pre-activation (from the loss kernel)
    bias_grad(A(g[0]), 256, e->fu[2], B);
    lin_wgrad(A(g[0]), 256, A(e->ha.f[1]), 256, e->fu[2], B);
    Opt o; o.dact = ACT_LEAKY; o.dact_src = A(e->ha.f[1]); o.ld_aux = 256;
    lin_dgrad(A(g[0]), 256, e->fu[2], B, A(g[1]), 256, o);                     // d fusion.2 pre-act
    bias_grad(A(g[1]), 256, e->fu[1], B);
    lin_wgrad(A(g[1]), 256, A(e->ha.f[0]), 512, e->fu[1], B);
    o.dact_src = A(e->ha.f[0]); o.ld_aux = 512;
    lin_dgrad(A(g[1]), 256, e->fu[1], B, A(g[2]), 512, o);                     // d fusion.0 pre-act [B,512]
    bias_grad(A(g[2]), 512, e->fu[0], B);
    lin_wgrad(A(g[2]), 512, A(e->ha.cat), 512, e->fu[0], B);
    lin_dgrad(A(g[2]), 512, e->fu[0], B, A(g[3]), 512, Opt());                 // d cat [B,512] = (d hv, d hw)
    if (!rc && !V.rc) V.fail(order_streams(e, s, V.s));                        // d cat is ready
    // pipes; their input gradients accumulate onto the pooler's rows
    V.pipe_bwd(e->vp, V.A(seq_v), ldv, e->ha.v, V.A(g[3]), 512, V.A(gv), ldv, true, B, g + 4);
    pipe_bwd(e->tp, A(seq_t), ldt, e->ha.t, A(g[3]) + 256, 512, A(gt), ldt, true, B, g + 7);
    flush_wgrads();
    V.flush_wgrads();
  }
};

int check_batch(const crct_engine* e, const CrctBatch* b) {
  CRCT_REQUIRE(b && b->B >= 1 && b->T >= 1 && b->V >= 1, "engine: bad batch sizes");
  CRCT_REQUIRE(b->B <= e->maxB && b->T <= e->maxT && b->V <= e->maxV, "engine: batch (B=%d,T=%d,V=%d) exceeds the engine maximum (%d,%d,%d)",
               b->B, b->T, b->V, e->maxB, e->maxT, e->maxV);
  CRCT_REQUIRE(b->tokens && b->segments && b->loc && b->image_feat && b->image_loc && b->image_target && b->R, "engine: null batch pointer");
  CRCT_REQUIRE(b->text_keymask || (b->sep_indices && b->hist_len && b->sep_stride > 0), "engine: text_keymask, or sep_indices + hist_len, is required");
  CRCT_REQUIRE(b->image_keymask || b->image_mask, "engine: image_keymask or image_mask is required");
  return 0;
}

}  // namespace

// =================================================================================== C ABI
extern "C" crct_engine_t* crct_engine_create(const CrctModelDims* dims, const char* names, const int64_t* offsets,
                                             const int64_t* sizes, int n_params, int max_B, int max_T, int max_V) {
  if (!dims || !names || !offsets || !sizes) { crct_set_error("engine_create: null argument"); return nullptr; }
  crct_engine* e = new crct_engine();
  e->d = *dims; e->maxB = max_B; e->maxT = max_T; e->maxV = max_V;
  const CrctModelDims& D = e->d;
  {
    const char* p = names;
    for (int i = 0; i < n_params; ++i) {
      const char* q = strchr(p, '\n');
      std::string k = q ? std::string(p, q - p) : std::string(p);
      e->off[k] = offsets[i]; e->size[k] = sizes[i];
      if (!q) break;
      p = q + 1;
    }
  }
  auto fail = [&](const char* msg) -> crct_engine_t* { if (msg) crct_set_error("%s", msg); delete e; return nullptr; };
  if (D.H % D.heads || D.Hv % D.v_heads || D.Hb % D.b_heads) return fail("engine_create: hidden size not a multiple of heads");
  if (D.H % 8 || D.Hv % 8 || D.Hb % 8 || D.I % 8 || D.Iv % 8 || D.Fv % 8) return fail("engine_create: sizes must be multiples of 8");
  if (D.n_conn > 32) return fail("engine_create: more than 32 connection layers");
  // ---- schedule (vilbert.py:852-939)
  {
    int vs = 0, ts = 0;
    for (int c = 0; c < D.n_conn; ++c) {
      for (int i = vs; i < D.v_biatt[c]; ++i) e->sched.push_back({'v', i});
      for (int i = ts; i < D.t_biatt[c]; ++i) e->sched.push_back({'t', i});
      if (D.with_coattention) e->sched.push_back({'c', c});
      vs = D.v_biatt[c]; ts = D.t_biatt[c];
    }
    for (int i = vs; i < D.Lv; ++i) e->sched.push_back({'v', i});
    for (int i = ts; i < D.L; ++i) e->sched.push_back({'t', i});
    for (size_t i = 0; i < e->sched.size() && e->first_conn < 0; ++i)
      if (e->sched[i].kind == 'c') e->first_conn = (int)i;
  }
  // ---- parameters
  char buf[256];
  uint32_t site = 16;
  for (int i = 0; i < D.L; ++i) {
    snprintf(buf, sizeof(buf), "bert.encoder.layer.%d.", i);
    std::string p(buf);
    SelfLayerP l;
    l.H = D.H; l.heads = D.heads; l.p_attn = D.p_attn; l.p_hid = D.p_hidden; l.site = site; site += 4;
    l.qkv = fused3(e, p + "attention.self.query", p + "attention.self.key", p + "attention.self.value", D.H, D.H, CRCT_SITE_T_QKV);
    l.proj.dense = linear_p(e, p + "attention.output.dense", D.H, D.H, CRCT_SITE_T_OUT);
    l.proj.ln = ln_p(e, p + "attention.output.LayerNorm");
    l.ffn.up = linear_p(e, p + "intermediate.dense", D.H, D.I, CRCT_SITE_T_FFN_UP);
    l.ffn.down = linear_p(e, p + "output.dense", D.I, D.H, CRCT_SITE_T_FFN_DN);
    l.ffn.ln = ln_p(e, p + "output.LayerNorm");
    e->tl.push_back(l);
  }
  for (int i = 0; i < D.Lv; ++i) {
    snprintf(buf, sizeof(buf), "bert.encoder.v_layer.%d.", i);
    std::string p(buf);
    SelfLayerP l;
    l.H = D.Hv; l.heads = D.v_heads; l.p_attn = D.p_v_attn; l.p_hid = D.p_v_hidden; l.site = site; site += 4;
    l.qkv = fused3(e, p + "attention.self.query", p + "attention.self.key", p + "attention.self.value", D.Hv, D.Hv, CRCT_SITE_V_QKV);
    l.proj.dense = linear_p(e, p + "attention.output.dense", D.Hv, D.Hv, CRCT_SITE_V_OUT);
    l.proj.ln = ln_p(e, p + "attention.output.LayerNorm");
    l.ffn.up = linear_p(e, p + "intermediate.dense", D.Hv, D.Iv, CRCT_SITE_V_FFN_UP);
    l.ffn.down = linear_p(e, p + "output.dense", D.Iv, D.Hv, CRCT_SITE_V_FFN_DN);
    l.ffn.ln = ln_p(e, p + "output.LayerNorm");
    e->vl.push_back(l);
  }
  for (int i = 0; i < D.n_conn; ++i) {
    snprintf(buf, sizeof(buf), "bert.encoder.c_layer.%d.", i);
    std::string p(buf);
    ConnLayerP l;
    l.site = site; site += 8;
    l.qkv1 = fused3(e, p + "biattention.query1", p + "biattention.key1", p + "biattention.value1", D.Hv, D.Hb, CRCT_SITE_C_QKV_V);
    l.qkv2 = fused3(e, p + "biattention.query2", p + "biattention.key2", p + "biattention.value2", D.H, D.Hb, CRCT_SITE_C_QKV_T);
    l.proj_v.dense = linear_p(e, p + "biOutput.dense1", D.Hb, D.Hv, CRCT_SITE_C_OUT_V); l.proj_v.ln = ln_p(e, p + "biOutput.LayerNorm1");
    l.proj_t.dense = linear_p(e, p + "biOutput.dense2", D.Hb, D.H, CRCT_SITE_C_OUT_T); l.proj_t.ln = ln_p(e, p + "biOutput.LayerNorm2");
    l.ffn_v.up = linear_p(e, p + "v_intermediate.dense", D.Hv, D.Iv, CRCT_SITE_V_FFN_UP);
    l.ffn_v.down = linear_p(e, p + "v_output.dense", D.Iv, D.Hv, CRCT_SITE_V_FFN_DN); l.ffn_v.ln = ln_p(e, p + "v_output.LayerNorm");
    l.ffn_t.up = linear_p(e, p + "t_intermediate.dense", D.H, D.I, CRCT_SITE_T_FFN_UP);
    l.ffn_t.down = linear_p(e, p + "t_output.dense", D.I, D.H, CRCT_SITE_T_FFN_DN); l.ffn_t.ln = ln_p(e, p + "t_output.LayerNorm");
    e->cl.push_back(l);
  }
  e->et.word = e->P("bert.embeddings.word_embeddings.weight");
  e->et.pos = e->P("bert.embeddings.position_embeddings.weight");
  e->et.type = e->P("bert.embeddings.plotqa_type_embeddings.weight");
  e->et.wloc = e->P("bert.embeddings.txt_location_embeddings.weight");
  e->et.bloc = e->P("bert.embeddings.txt_location_embeddings.bias");
  e->et.ln = ln_p(e, "bert.embeddings.LayerNorm");
  e->ev.img = linear_p(e, "bert.v_embeddings.new_image_embeddings", D.Fv, D.Hv, CRCT_SITE_IMG_EMB);
  e->ev.color = e->P("bert.v_embeddings.color_emb.weight");
  e->ev.wloc = e->P("bert.v_embeddings.new_loc_emb.weight");
  e->ev.bloc = e->P("bert.v_embeddings.new_loc_emb.bias");
  e->ev.ln = ln_p(e, "bert.v_embeddings.LayerNorm");
  e->t_pool = linear_p(e, "bert.t_pooler.dense", D.H, D.Hb);
  e->v_pool = linear_p(e, "bert.v_pooler.dense", D.Hv, D.Hb);
  e->cls = linear_p(e, "cls.bi_seq_relationship", D.Hb, 2);
  {
    const int tw[5] = {D.H, D.H, 512, 256, 256}, vw[5] = {D.Hv, D.Hv, 512, 256, 256}, fw[5] = {512, 512, 256, 256, 1};
    for (int j = 0; j < 4; ++j) {
      snprintf(buf, sizeof(buf), "regressor.txt_pipe.%d", 2 * j); e->tp[j] = linear_p(e, buf, tw[j], tw[j + 1]);
      snprintf(buf, sizeof(buf), "regressor.vis_pipe.%d", 2 * j); e->vp[j] = linear_p(e, buf, vw[j], vw[j + 1]);
      snprintf(buf, sizeof(buf), "regressor.fusion.%d", 2 * j); e->fu[j] = linear_p(e, buf, fw[j], fw[j + 1]);
    }
  }
  if (e->bad) return fail(nullptr);
  {
    auto own = [&](const LinearP& l) { e->wgrad_owned[l.w] = (int64_t)l.in * l.out; };
    for (const SelfLayerP& l : e->tl) { own(l.qkv); own(l.proj.dense); own(l.ffn.up); own(l.ffn.down); }
    for (const SelfLayerP& l : e->vl) { own(l.qkv); own(l.proj.dense); own(l.ffn.up); own(l.ffn.down); }
    for (const ConnLayerP& l : e->cl) {
      own(l.qkv1); own(l.qkv2); own(l.proj_v.dense); own(l.proj_t.dense);
      own(l.ffn_v.up); own(l.ffn_v.down); own(l.ffn_t.up); own(l.ffn_t.down);
    }
    own(e->ev.img); own(e->t_pool); own(e->v_pool);
    for (int j = 0; j < 4; ++j) { own(e->tp[j]); own(e->vp[j]); }
    for (int j = 0; j < 3; ++j) own(e->fu[j]);     // fusion.6 (fu[3]) and bi_seq_relationship are produced by the head kernel: accumulate-only
  }
  {
    // fp8: every Linear of the encoder whose two dimensions are whole numbers of 128-deep fp8 K tiles gets an e4m3 weight shadow
    // and a scale slot
    auto slot = [&](const LinearP& l) {
      if (l.in % 128 != 0 || l.out % 128 != 0) return;      // forward contracts over `in`, the data gradient over `out`: whole fp8 K tiles both ways
      e->wq_slot[l.w] = (int)e->wq_list.size();
      e->wq_list.push_back({l.w, (int64_t)l.in * l.out});
    };
    for (const SelfLayerP& l : e->tl) { slot(l.qkv); slot(l.ffn.up); slot(l.ffn.down); slot(l.proj.dense); }
    for (const SelfLayerP& l : e->vl) { slot(l.qkv); slot(l.ffn.up); slot(l.ffn.down); slot(l.proj.dense); }
    for (const ConnLayerP& l : e->cl) {
      slot(l.qkv1); slot(l.qkv2); slot(l.ffn_v.up); slot(l.ffn_v.down); slot(l.ffn_t.up); slot(l.ffn_t.down);
      slot(l.proj_v.dense); slot(l.proj_t.dense);
    }
  }

  // ---- workspace
  Arena ar;
  const size_t Mt = (size_t)max_B * max_T, Mv = (size_t)max_B * max_V, B = max_B;
  e->eta.sum = ar.take(Mt * D.H * 2); e->eta.y = ar.take(Mt * D.H * 2); e->eta.mean = ar.take(Mt * 4); e->eta.rstd = ar.take(Mt * 4);
  e->eva.soft = ar.take(Mv * D.Fv * 2); e->eva.lin = ar.take(Mv * D.Hv * 2); e->eva.sum = ar.take(Mv * D.Hv * 2);
  e->eva.y = ar.take(Mv * D.Hv * 2); e->eva.mean = ar.take(Mv * 4); e->eva.rstd = ar.take(Mv * 4);
  e->eta.yq = ar.take(Mt * D.H); e->eva.yq = ar.take(Mv * D.Hv);
  e->eta.site = e->n_sites++; e->eva.site = e->n_sites++;
  e->taps.push_back({"emb.t", e->eta.y, 't'});
  e->taps.push_back({"emb.v", e->eva.y, 'v'});
  e->tla.resize(D.L); e->vla.resize(D.Lv); e->cla.resize(D.n_conn);
  for (int i = 0; i < D.L; ++i) {
    SelfLayerA& a = e->tla[i];
    a.qkv = ar.take(Mt * 3 * D.H * 2); a.ctx = ar.take(Mt * D.H * 2); a.ctxq = ar.take(Mt * D.H); a.lse = ar.take(Mt * D.heads * 4); a.site_ctx = e->n_sites++; a.g_dqkv = e->n_gsites++; a.proj = proj_a(ar, Mt, D.H, e->n_sites, e->n_gsites); a.ffn = ffn_a(ar, Mt, D.H, D.I, e->n_sites, e->n_gsites);
  }
  for (int i = 0; i < D.Lv; ++i) {
    SelfLayerA& a = e->vla[i];
    a.qkv = ar.take(Mv * 3 * D.Hv * 2); a.ctx = ar.take(Mv * D.Hv * 2); a.ctxq = ar.take(Mv * D.Hv); a.lse = ar.take(Mv * D.v_heads * 4); a.site_ctx = e->n_sites++; a.g_dqkv = e->n_gsites++; a.proj = proj_a(ar, Mv, D.Hv, e->n_sites, e->n_gsites); a.ffn = ffn_a(ar, Mv, D.Hv, D.Iv, e->n_sites, e->n_gsites);
  }
  for (int i = 0; i < D.n_conn; ++i) {
    ConnLayerA& a = e->cla[i];
    a.qkv1 = ar.take(Mv * 3 * D.Hb * 2); a.qkv2 = ar.take(Mt * 3 * D.Hb * 2);
    a.ctx1 = ar.take(Mt * D.Hb * 2); a.ctx2 = ar.take(Mv * D.Hb * 2);
    a.ctx1q = ar.take(Mt * D.Hb); a.ctx2q = ar.take(Mv * D.Hb);
    a.lse1 = ar.take(Mt * D.b_heads * 4); a.lse2 = ar.take(Mv * D.b_heads * 4);
    a.site_ctx1 = e->n_sites++; a.site_ctx2 = e->n_sites++; a.g_dqkv1 = e->n_gsites++; a.g_dqkv2 = e->n_gsites++;
    a.proj_v = proj_a(ar, Mv, D.Hv, e->n_sites, e->n_gsites); a.proj_t = proj_a(ar, Mt, D.H, e->n_sites, e->n_gsites);
    a.ffn_v = ffn_a(ar, Mv, D.Hv, D.Iv, e->n_sites, e->n_gsites); a.ffn_t = ffn_a(ar, Mt, D.H, D.I, e->n_sites, e->n_gsites);
  }
  e->ha.pooled_t = ar.take(B * D.Hb * 2); e->ha.pooled_v = ar.take(B * D.Hb * 2);
  e->ha.t[0] = ar.take(B * D.H * 2); e->ha.t[1] = ar.take(B * 512 * 2); e->ha.t[2] = ar.take(B * 256 * 2);
  e->ha.v[0] = ar.take(B * D.Hv * 2); e->ha.v[1] = ar.take(B * 512 * 2); e->ha.v[2] = ar.take(B * 256 * 2);
  e->ha.cat = ar.take(B * 512 * 2);
  e->ha.f[0] = ar.take(B * 512 * 2); e->ha.f[1] = ar.take(B * 256 * 2); e->ha.f[2] = ar.take(B * 256 * 2);
  e->ha.scratch = ar.take(B * 8 * 4);
  e->ha.d_pt = ar.take(B * D.Hb * 2); e->ha.d_pv = ar.take(B * D.Hb * 2);
  {
    size_t w = 512;
    if ((size_t)D.H > w) w = D.H;
    if ((size_t)D.Hv > w) w = D.Hv;
    for (int k = 0; k < 10; ++k) e->ha.g[k] = ar.take(B * w * 2);
  }
  e->st = scratch_a(ar, Mt, D.H, D.I, D.Hb);
  e->sv = scratch_a(ar, Mv, D.Hv, D.Iv, D.Hb);
  e->st2 = scratch_a(ar, Mt, D.H, D.I, D.Hb);
  e->sv2 = scratch_a(ar, Mv, D.Hv, D.Iv, D.Hb);
  {
    size_t wmax = D.H > D.Hv ? D.H : D.Hv;
    for (int k = 0; k < 2; ++k) {
      e->partials[k] = ar.take((size_t)10 * 4 * CRCT_LN_BWD_MAX_BLOCKS * wmax * 4);   // [<= 9][4 waves x blocks][H]
    }
    e->embed_rows[0] = ar.take(Mt * D.H * 4);  e->embed_idx[0] = ar.take(2 * Mt * 4);
    e->embed_rows[1] = ar.take(Mv * D.Hv * 4); e->embed_idx[1] = ar.take(Mv * 4);
    size_t nmax = 3 * (size_t)D.Hb;
    if ((size_t)D.I > nmax) nmax = D.I;
    if ((size_t)D.Iv > nmax) nmax = D.Iv;
    if (3 * (size_t)D.H > nmax) nmax = 3 * (size_t)D.H;
    if (3 * (size_t)D.Hv > nmax) nmax = 3 * (size_t)D.Hv;
    if (nmax < 1024) nmax = 1024;
    for (int k = 0; k < 4; ++k) e->colsum_part[k] = ar.take((size_t)64 * nmax * 4);
  }
  e->km_t = ar.take(Mt); e->km_v = ar.take(Mv);      // uint8 key masks built from sep_indices / hist_len / image_mask (CrctBatch)
  {
    // split-K slab space per data stream: the narrow outputs (N <= the widest hidden size) with up to 4 slices; wider outputs
    // have enough tiles and are never split.  Ticket words: zeroed at the start of every engine call.
    const int Hmax = std::max(std::max(D.H, D.Hv), D.Hb);
    e->sk_ws_elems[0] = (size_t)crct_gemm_splitk_ws_elems((int)Mt, Hmax, 4);
    e->sk_ws_elems[1] = (size_t)crct_gemm_splitk_ws_elems((int)Mv, Hmax, 4);
    e->sk_tickets = std::max(crct_gemm_splitk_tickets((int)Mt, Hmax), crct_gemm_splitk_tickets((int)Mv, Hmax));
    e->sk_tickets = (e->sk_tickets + 3) / 4 * 4;
    for (int k = 0; k < 2; ++k) e->sk_ws[k] = ar.take(e->sk_ws_elems[k] * 4);
    e->sk_cnt[0] = ar.take((size_t)2 * e->sk_tickets * 4);      // [text | visual] in one block: one memset per call
    e->sk_cnt[1] = e->sk_cnt[0] + (size_t)e->sk_tickets * 4;
  }
  {
    auto reg_ffn = [&](const FfnA& a) { e->res32[a.y] = a.y32; };
    auto reg_proj = [&](const ProjA& a) { e->res32[a.a] = a.a32; };
    for (const SelfLayerA& a : e->tla) { reg_proj(a.proj); reg_ffn(a.ffn); }
    for (const SelfLayerA& a : e->vla) { reg_proj(a.proj); reg_ffn(a.ffn); }
    for (const ConnLayerA& a : e->cla) { reg_proj(a.proj_v); reg_proj(a.proj_t); reg_ffn(a.ffn_v); reg_ffn(a.ffn_t); }
  }
  e->ws_bytes = ar.top;

  // ---- taps + final outputs, following the schedule
  {
    size_t xt = e->eta.y, xv = e->eva.y;
    for (const Step& st : e->sched) {
      if (st.kind == 't') xt = e->tla[st.idx].ffn.y;
      else if (st.kind == 'v') xv = e->vla[st.idx].ffn.y;
      else { xv = e->cla[st.idx].ffn_v.y; xt = e->cla[st.idx].ffn_t.y; }
      snprintf(buf, sizeof(buf), "%c%d.t", st.kind, st.idx); e->taps.push_back({buf, xt, 't'});
      snprintf(buf, sizeof(buf), "%c%d.v", st.kind, st.idx); e->taps.push_back({buf, xv, 'v'});
    }
    e->final_t = xt; e->final_v = xv;
    e->taps.push_back({"seq_t", xt, 't'});
    e->taps.push_back({"seq_v", xv, 'v'});
  }

  // ---- gradient segments in backward order: heads, schedule reversed, embeddings
  auto range_of = [&](std::vector<std::string> prefixes) {
    int64_t lo = INT64_MAX, hi = -1;
    for (auto& kv : e->off)
      for (auto& pf : prefixes)
        if (kv.first.compare(0, pf.size(), pf) == 0 && e->size[kv.first] > 0) {    // size 0 = never receives a gradient
          if (kv.second < lo) lo = kv.second;
          const int64_t end = kv.second + e->size[kv.first];
          if (end > hi) hi = end;
        }
    if (hi < 0) { lo = 0; hi = 0; }
    return std::make_pair(lo, hi);
  };
  e->seg_range.push_back(range_of({"bert.t_pooler.", "bert.v_pooler.", "cls.bi_seq_relationship.", "regressor."}));
  for (int i = (int)e->sched.size() - 1; i >= 0; --i) {
    const Step& st = e->sched[i];
    snprintf(buf, sizeof(buf), "bert.encoder.%s.%d.", st.kind == 't' ? "layer" : (st.kind == 'v' ? "v_layer" : "c_layer"), st.idx);
    e->seg_range.push_back(range_of({std::string(buf)}));
  }
  e->seg_range.push_back(range_of({"bert.embeddings.", "bert.v_embeddings."}));
  return e;
}

extern "C" void crct_engine_destroy(crct_engine_t* e) {
  if (!e) return;
  for (auto ev : e->evpool) (void)hipEventDestroy(ev);
  for (auto st : e->side) if (st) (void)hipStreamDestroy(st);
  if (e->aux) (void)hipStreamDestroy(e->aux);
  if (e->word_index) (void)hipFree(e->word_index);
  delete e;
}
extern "C" size_t crct_engine_workspace_bytes(const crct_engine_t* e) { return e ? e->ws_bytes : 0; }
extern "C" int crct_engine_num_segments(const crct_engine_t* e) { return e ? (int)e->seg_range.size() : 0; }
extern "C" int crct_engine_segment_range(const crct_engine_t* e, int seg, int64_t* lo, int64_t* hi) {
  CRCT_REQUIRE(e && seg >= 0 && seg < (int)e->seg_range.size(), "segment_range: bad segment %d", seg);
  *lo = e->seg_range[seg].first; *hi = e->seg_range[seg].second;
  return 0;
}

namespace {

int ensure_streams(crct_engine* e, hipStream_t main) {
  // all internal streams share the caller's (default) priority: giving the weight-gradient streams the lowest or the
  // visual stream the highest priority (hipStreamCreateWithPriority) was measured to DOUBLE the step time on MI355X
  // (10.7 -> 21.9 ms, round 1) -- cross-priority event waits are slow -- so there is no priority knob
  if (!e->placed) {          // once per engine: streams on hardware queues that do not collide with the caller's or each other
    e->placed = true;
    e->placed_for = main;
    hipStream_t out[4];
    if (int r = crct_streams_place(main, out, &e->queue_classes)) return r;
    e->side[0] = out[0]; e->side[1] = out[1]; e->aux = out[2]; e->side[2] = out[3];
  }
  for (int k = 0; k < 3; ++k) {
    const bool need = k == 0 ? e->use_vis_stream : (e->use_wgrad_stream && !(k == 2 && e->one_wgrad_stream));
    if (need && !e->side[k] && hipStreamCreateWithFlags(&e->side[k], hipStreamNonBlocking) != hipSuccess) {
      crct_set_error("engine: cannot create an internal HIP stream");
      return 1;
    }
  }
  return 0;
}

// two drivers over one workspace: Rt = text stream on the caller's stream, Rv = visual stream
void make_runs(crct_engine* e, const float* p32, const void* p16, float* g32, void* ws, hipStream_t main, const CrctBatch* batch,
               const CrctStepCfg* cfg, Run& Rt, Run& Rv) {
  hipStream_t vis = e->use_vis_stream ? e->side[0] : main;
  Rt = Run{e, p32, (const bf16_t*)p16, g32, (char*)ws, main, batch, cfg, e->use_wgrad_stream ? e->side[1] : main,
           e->partials[0], e->colsum_part[0], e->colsum_part[2]};
  Rv = Run{e, p32, (const bf16_t*)p16, g32, (char*)ws, vis, batch, cfg, e->use_wgrad_stream ? e->side[e->one_wgrad_stream ? 1 : 2] : vis,
           e->partials[1], e->colsum_part[1], e->colsum_part[3]};
  Rt.sets[0] = &e->st; Rt.sets[1] = &e->st2;
  Rv.sets[0] = &e->sv; Rv.sets[1] = &e->sv2;
  Rt.which = 0; Rv.which = 1;
}

// the split-K ticket words of both data streams start every engine call at zero (an aborted launch must not poison the next)
int reset_tickets(crct_engine* e, void* ws, hipStream_t s) {
  bool any = false;
  for (int a = 1; a < CRCT_SITE_COUNT && !any; ++a)
    for (int k = 0; k < 2; ++k)
      for (int ph = 0; ph < 2; ++ph) any = any || e->policy[a][k][ph].split_k > 1;      // weight gradients are never split
  if (!any) return 0;
  CRCT_CHECK_HIP(hipMemsetAsync((char*)ws + e->sk_cnt[0], 0, (size_t)2 * e->sk_tickets * 4, s));
  return 0;
}

}  // namespace

static int engine_forward_impl(crct_engine_t* e, const float* params_f32, const void* params_bf16, const CrctBatch* batch,
                               const CrctStepCfg* cfg, void* workspace, float* logits, float* reg, float* stats,
                               crct_stream_t stream) {
  CRCT_REQUIRE(e && params_f32 && params_bf16 && cfg && workspace && logits && reg && stats, "engine_forward: null argument");
  if (int r = check_batch(e, batch)) return r;
  if (int r = ensure_streams(e, (hipStream_t)stream)) return r;
  e->evnext = 0;
  // key masks the caller did not supply are built here (one launch) and kept in the workspace for the backward pass
  CrctBatch bl = *batch;
  if (!bl.text_keymask || !bl.image_keymask) {
    uint8_t* kt = bl.text_keymask ? nullptr : (uint8_t*)workspace + e->km_t;
    uint8_t* kv = bl.image_keymask ? nullptr : (uint8_t*)workspace + e->km_v;
    if (int r = crct_build_keymasks(bl.sep_indices, bl.hist_len, bl.sep_stride, bl.image_mask, kt, kv, bl.B, bl.T, bl.V, stream)) return r;
    if (kt) bl.text_keymask = kt;
    if (kv) bl.image_keymask = kv;
  }
  batch = &bl;
  Run Rt, Rv;
  make_runs(e, params_f32, params_bf16, nullptr, workspace, (hipStream_t)stream, batch, cfg, Rt, Rv);
  if (int r = reset_tickets(e, workspace, (hipStream_t)stream)) return r;
  Rv.fail(order_streams(e, Rt.s, Rv.s));                 // fork: the visual stream starts after the caller's prior work
  // parameters of backward-segment `seg` may still be in the hands of an optimizer update running on another
  // stream (crct.optim overlap mode): wait for its event right before the first kernel that reads them
  const int nseg = (int)e->seg_range.size();
  auto wait_params = [&](Run& R, int seg) {
    if (!cfg->seg_ready_events || R.rc) return;
#ifdef CRCT_GEMM_LAB   // timing only: the forward does not wait for the overlapped optimizer update (reads parameters mid-update)
    static const bool lab_no_wait = getenv("CRCT_LAB_NO_PARAM_WAIT") != nullptr;
    if (lab_no_wait) return;
#endif
    hipEvent_t ev = (hipEvent_t)cfg->seg_ready_events[seg];
    if (ev && hipStreamWaitEvent(R.s, ev, 0) != hipSuccess) { crct_set_error("engine: wait on a parameter-ready event failed"); R.rc = 1; }
  };
  wait_params(Rt, nseg - 1);
  wait_params(Rv, nseg - 1);
  Rt.embed_text_fwd();
  Rv.embed_image_fwd();
  size_t xt = e->eta.y, xv = e->eva.y;
  size_t xtq = e->eta.yq, xvq = e->eva.yq;               // e4m3 copies of the running hidden states and their scale sites (fp8 forward)
  int site_t = e->eta.site, site_v = e->eva.site;
  int step_i = 0;
  for (const Step& st : e->sched) {
    const int seg = (int)e->sched.size() - step_i;       // backward segment of this schedule step
    Rt.phase = (e->first_conn < 0 || step_i < e->first_conn) ? 0 : 1;      // text-only prefix: nothing else on the data path
    ++step_i;
    if (st.kind == 't') wait_params(Rt, seg);
    else if (st.kind == 'v') wait_params(Rv, seg);
    else { wait_params(Rt, seg); wait_params(Rv, seg); }
    if (st.kind == 't') {
      const SelfLayerA& a = e->tla[st.idx];
      Rt.self_fwd(e->tl[st.idx], a, xt, xtq, site_t, batch->text_keymask, batch->B, batch->T);
      xt = a.ffn.y; xtq = a.ffn.yq; site_t = a.ffn.site_y;
    } else if (st.kind == 'v') {
      const SelfLayerA& a = e->vla[st.idx];
      Rv.self_fwd(e->vl[st.idx], a, xv, xvq, site_v, batch->image_keymask, batch->B, batch->V);
      xv = a.ffn.y; xvq = a.ffn.yq; site_v = a.ffn.site_y;
    } else {
      const ConnLayerA& a = e->cla[st.idx];
      Rt.conn_fwd(Rv, e->cl[st.idx], a, xv, xvq, site_v, xt, xtq, site_t);
      xv = a.ffn_v.y; xvq = a.ffn_v.yq; site_v = a.ffn_v.site_y;
      xt = a.ffn_t.y; xtq = a.ffn_t.yq; site_t = a.ffn_t.site_y;
    }
  }
  Rt.phase = 1;
  wait_params(Rt, 0);
  wait_params(Rv, 0);
  Rt.heads_branch_fwd(false, xt);
  Rv.heads_branch_fwd(true, xv);
  Rt.fail(order_streams(e, Rv.s, Rt.s));                 // join
  Rt.heads_tail_fwd(logits, reg, stats);
  return Rt.rc ? Rt.rc : Rv.rc;
}

static int engine_backward_impl(crct_engine_t* e, const float* params_f32, const void* params_bf16, const CrctBatch* batch,
                                const CrctStepCfg* cfg, void* workspace, float* grads_f32, float* logits, float* reg,
                                float* stats, int seg, crct_stream_t stream) {
  CRCT_REQUIRE(e && params_f32 && params_bf16 && cfg && workspace && grads_f32 && logits && reg && stats, "engine_backward: null argument");
  CRCT_REQUIRE(batch && batch->labels, "engine_backward: labels are required (training step)");
  if (int r = check_batch(e, batch)) return r;
  if (int r = ensure_streams(e, (hipStream_t)stream)) return r;
  e->evnext = 0;
  CrctBatch bl = *batch;                                 // masks built by the forward pass of this batch live in the workspace
  if (!bl.text_keymask) bl.text_keymask = (const uint8_t*)workspace + e->km_t;
  if (!bl.image_keymask) bl.image_keymask = (const uint8_t*)workspace + e->km_v;
  batch = &bl;
  Run Rt, Rv;
  make_runs(e, params_f32, params_bf16, grads_f32, workspace, (hipStream_t)stream, batch, cfg, Rt, Rv);
  const int nseg = (int)e->seg_range.size();
  const int s0 = seg < 0 ? 0 : seg, s1 = seg < 0 ? nseg : seg + 1;
  CRCT_REQUIRE(s1 <= nseg, "engine_backward: bad segment %d", seg);
  if (s0 == 0) e->wgrad_pass_begin();
  // inputs of every schedule step (outputs of the previous step of that stream)
  // ... with their e4m3 copies and activation scale sites (fp8 weight gradients of the QKV projections)
  std::vector<size_t> in_t(e->sched.size()), in_v(e->sched.size()), inq_t(e->sched.size()), inq_v(e->sched.size());
  std::vector<int> ins_t(e->sched.size()), ins_v(e->sched.size());
  {
    size_t xt = e->eta.y, xv = e->eva.y, xtq = e->eta.yq, xvq = e->eva.yq;
    int site_t = e->eta.site, site_v = e->eva.site;
    for (size_t i = 0; i < e->sched.size(); ++i) {
      in_t[i] = xt; in_v[i] = xv; inq_t[i] = xtq; inq_v[i] = xvq; ins_t[i] = site_t; ins_v[i] = site_v;
      const Step& st = e->sched[i];
      if (st.kind == 't') { const FfnA& f = e->tla[st.idx].ffn; xt = f.y; xtq = f.yq; site_t = f.site_y; }
      else if (st.kind == 'v') { const FfnA& f = e->vla[st.idx].ffn; xv = f.y; xvq = f.yq; site_v = f.site_y; }
      else {
        const FfnA& fv = e->cla[st.idx].ffn_v; const FfnA& ft = e->cla[st.idx].ffn_t;
        xv = fv.y; xvq = fv.yq; site_v = fv.site_y; xt = ft.y; xtq = ft.yq; site_t = ft.site_y;
      }
    }
  }
  if (int r = reset_tickets(e, workspace, (hipStream_t)stream)) return r;
  // fork: every internal stream starts after the caller's prior work (previous segment, optimizer, ...)
  Rv.fail(order_streams(e, Rt.s, Rv.s));
  int ev_from = s0;                                      // segments enqueued completely but not yet marked for the data-parallel caller
  for (int sgi = s0; sgi < s1 && !Rt.rc && !Rv.rc; ++sgi) {
    const bool in_sched = sgi != 0 && sgi != nseg - 1;
    const size_t si = in_sched ? e->sched.size() - (size_t)sgi : 0;
    Rt.phase = (in_sched && (e->first_conn < 0 || (int)si < e->first_conn)) ? 0 : 1;      // backward tail through the text-only layers
    if (sgi == 0) {
      e->cur_t = 0; e->cur_v = 0;
      Rt.heads_bwd(Rv, e->final_t, e->final_v, e->st.dy[0], e->sv.dy[0], logits, reg, stats);
    } else if (sgi == nseg - 1) {
      Rt.embed_text_bwd(e->st.dy[e->cur_t]);
      Rv.embed_image_bwd(e->sv.dy[e->cur_v]);
    } else {
      const size_t i = e->sched.size() - (size_t)sgi;
      const Step& st = e->sched[i];
      if (st.kind == 't') {
        Rt.self_bwd(e->tl[st.idx], e->tla[st.idx], in_t[i], inq_t[i], ins_t[i], e->st.dy[e->cur_t], e->st.dy[e->cur_t ^ 1], batch->text_keymask, batch->B, batch->T);
        e->cur_t ^= 1;
      } else if (st.kind == 'v') {
        Rv.self_bwd(e->vl[st.idx], e->vla[st.idx], in_v[i], inq_v[i], ins_v[i], e->sv.dy[e->cur_v], e->sv.dy[e->cur_v ^ 1], batch->image_keymask, batch->B, batch->V);
        e->cur_v ^= 1;
      } else {
        Rt.conn_bwd(Rv, e->cl[st.idx], e->cla[st.idx], in_v[i], inq_v[i], ins_v[i], in_t[i], inq_t[i], ins_t[i], e->sv.dy[e->cur_v], e->st.dy[e->cur_t], e->sv.dy[e->cur_v ^ 1], e->st.dy[e->cur_t ^ 1]);
        e->cur_t ^= 1; e->cur_v ^= 1;
      }
    }
    if (seg < 0 && cfg->seg_done_events && !Rt.rc && !Rv.rc) {
      // segments ev_from .. sgi are completely enqueued: mark that point on every internal stream for the data-parallel caller
      Rt.flush_wgrads();
      Rv.flush_wgrads();
      hipStream_t ss[4] = {Rt.s, Rt.sw, Rv.s, Rv.sw};
      for (int sg = ev_from; sg <= sgi; ++sg) {
        if (cfg->seg_done_mask && !cfg->seg_done_mask[sg]) continue;
        for (int k = 0; k < 4; ++k) {
          hipEvent_t ev = (hipEvent_t)cfg->seg_done_events[4 * sg + k];
          if (ev && hipEventRecord(ev, ss[k]) != hipSuccess) { crct_set_error("engine_backward: cannot record a segment event"); Rt.rc = 1; }
        }
        // the data-parallel caller launches the bucket this segment completes NOW, while the rest of backward is still being enqueued
        if (cfg->seg_enqueued && !Rt.rc) cfg->seg_enqueued(sg, cfg->seg_enqueued_user);
      }
    }
    ev_from = sgi + 1;
  }
  // join: everything this call enqueued anywhere is ordered before later work on the caller's stream
  Rt.main_after_wgrad();
  Rv.main_after_wgrad();
  Rt.fail(order_streams(e, Rv.s, Rt.s));
  return Rt.rc ? Rt.rc : Rv.rc;
}

extern "C" int crct_engine_forward(crct_engine_t* e, const float* params_f32, const void* params_bf16, const CrctBatch* batch,
                                   const CrctStepCfg* cfg, void* workspace, float* logits, float* reg, float* stats,
                                   crct_stream_t stream) {
  return engine_forward_impl(e, params_f32, params_bf16, batch, cfg, workspace, logits, reg, stats, stream);
}

extern "C" int crct_engine_backward(crct_engine_t* e, const float* params_f32, const void* params_bf16, const CrctBatch* batch,
                                    const CrctStepCfg* cfg, void* workspace, float* grads_f32, float* logits, float* reg,
                                    float* stats, int seg, crct_stream_t stream) {
  return engine_backward_impl(e, params_f32, params_bf16, batch, cfg, workspace, grads_f32, logits, reg, stats, seg, stream);
}

extern "C" int crct_engine_wgrad_owned(crct_engine_t* e, int64_t* offsets, int64_t* numels, int cap) {
  if (!e) return -1;
  std::vector<std::pair<int64_t, int64_t>> v(e->wgrad_owned.begin(), e->wgrad_owned.end());
  std::sort(v.begin(), v.end());
  int n = 0;
  for (const auto& kv : v) {
    if (offsets && numels && n < cap) { offsets[n] = kv.first; numels[n] = kv.second; }
    ++n;
  }
  return n;
}

extern "C" int crct_engine_fp8_sites(const crct_engine_t* e) { return e ? e->n_sites : 0; }
extern "C" int crct_engine_fp8_grad_sites(const crct_engine_t* e) { return e ? e->n_gsites : 0; }
extern "C" int crct_engine_fp8_weights(const crct_engine_t* e, int64_t* offsets, int64_t* numels, int cap) {
  if (!e) return -1;
  const int n = (int)e->wq_list.size();
  for (int i = 0; i < n && i < cap && offsets && numels; ++i) { offsets[i] = e->wq_list[i].first; numels[i] = e->wq_list[i].second; }
  return n;
}

extern "C" int crct_engine_set_streams(crct_engine_t* e, int use_visual_stream, int use_wgrad_streams) {
  if (!e) return 1;
  e->use_vis_stream = use_visual_stream != 0;
  e->use_wgrad_stream = use_wgrad_streams != 0;
  e->one_wgrad_stream = use_wgrad_streams == 2;
  e->streams_forced = true;
  return 0;
}

// ---- device-scope ordering events for the host-side glue (optimizer overlap, data-parallel buckets): hipEventDisableTiming |
// hipEventDisableSystemFence, like the engine's internal ones.  A stock torch.cuda.Event carries a system-scope fence (host /
// peer visibility) in every record, which these same-device stream orderings do not need.
extern "C" void* crct_event_create(void) {
  hipEvent_t ev = nullptr;
  const unsigned flags = hipEventDisableTiming | hipEventDisableSystemFence;
  if (hipEventCreateWithFlags(&ev, flags) != hipSuccess) { crct_set_error("event_create: hipEventCreateWithFlags failed"); return nullptr; }
  return ev;
}
extern "C" void crct_event_destroy(void* ev) { if (ev) (void)hipEventDestroy((hipEvent_t)ev); }
extern "C" int crct_event_record(void* ev, crct_stream_t stream) {
  CRCT_REQUIRE(ev, "event_record: null event");
  CRCT_CHECK_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
  return 0;
}
extern "C" int crct_stream_wait_event(crct_stream_t stream, void* ev) {
  CRCT_REQUIRE(ev, "stream_wait_event: null event");
  CRCT_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0));
  return 0;
}
extern "C" int crct_event_query(void* ev) {          // 1: everything before the last record has finished, 0: not yet, < 0: error
  if (!ev) return -1;
  const hipError_t r = hipEventQuery((hipEvent_t)ev);
  if (r == hipSuccess) return 1;
  if (r == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
  crct_set_error("event_query: %s", hipGetErrorString(r));
  return -1;
}
extern "C" int crct_event_synchronize(void* ev) {
  CRCT_REQUIRE(ev, "event_synchronize: null event");
  CRCT_CHECK_HIP(hipEventSynchronize((hipEvent_t)ev));
  return 0;
}

extern "C" crct_stream_t crct_engine_aux_stream(crct_engine_t* e, crct_stream_t main_stream, int* queue_classes) {
  if (!e) return nullptr;
  if (ensure_streams(e, (hipStream_t)main_stream)) return nullptr;
  if (queue_classes) *queue_classes = e->queue_classes;
  return e->aux;
}

extern "C" int crct_engine_streams(crct_engine_t* e, crct_stream_t out[4]) {
  if (!e || !out) return 1;
  out[0] = e->side[0]; out[1] = e->side[1]; out[2] = e->side[2]; out[3] = e->aux;
  return 0;
}

extern "C" int crct_engine_set_wgrad_workgroups(crct_engine_t* e, int target_wgs, int max_rows) {
  if (!e) return 1;
  e->wgrad_target = target_wgs > 0 ? target_wgs : 0;
  e->wgrad_target_rows = max_rows;
  return 0;
}
extern "C" int crct_engine_set_wgrad_flush(crct_engine_t* e, int mode) {
  if (!e) return 1;
  e->wgrad_flush = mode;
  return 0;
}

extern "C" int crct_engine_set_site_policy(crct_engine_t* e, int site, int kind, int phase, int cfg, int split_k) {
  CRCT_REQUIRE(e && site > 0 && site < CRCT_SITE_COUNT && kind >= CRCT_KIND_FWD && kind <= CRCT_KIND_WGRAD && phase <= 1,
               "set_site_policy: bad site / kind / phase (%d, %d, %d)", site, kind, phase);
  CRCT_REQUIRE(cfg >= -1 && cfg <= 71 && split_k >= 0 && split_k <= 4, "set_site_policy: cfg %d / split_k %d out of range", cfg, split_k);
  for (int ph = 0; ph < 2; ++ph)
    if (phase < 0 || phase == ph) { e->policy[site][kind][ph].cfg = cfg; e->policy[site][kind][ph].split_k = split_k; }
  return 0;
}

extern "C" int64_t crct_engine_tap(crct_engine_t* e, const void* workspace, const char* name, int B, int T, int V, void* out,
                                   int64_t cap, crct_stream_t stream) {
  if (!e || !name) return -1;
  for (const Tap& t : e->taps)
    if (t.name == name) {
      const int64_t n = t.stream == 't' ? (int64_t)B * T * e->d.H : (int64_t)B * V * e->d.Hv;
      if (n > cap) { crct_set_error("tap: buffer too small"); return -1; }
      if (hipMemcpyAsync(out, (const char*)workspace + t.off, (size_t)n * 2, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return -1;
      return n;
    }
  crct_set_error("tap: unknown activation '%s'", name);
  return -1;
}
