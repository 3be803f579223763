// Placement of the step's HIP streams on the GPU's hardware queues.
//
// MI355X / ROCm 7.2 multiplexes all HIP streams of a process onto GPU_MAX_HW_QUEUES (= 4) hardware queues; a stream is bound
// to a queue when it is created (the queue with the fewest users) and two streams that share a queue run their kernels
// strictly one after the other.  Which streams share is an accident of creation order: in a process that has set up RCCL
// first, the engine's text and visual streams landed on ONE queue and a training step took 12.1 instead of 7.6 ms
// (profiles/r3_ddp_stream_placement.txt).  There is no API to ask for a queue, but whether two streams share one can be
// MEASURED: two short spin kernels, one per stream, take twice as long when they are serialised.  crct_streams_place creates
// candidate streams, sorts them into queue classes against the caller's stream by that probe, and hands back one stream per
// foreign class (+ a second one of the last class): the engine's visual / weight-gradient streams and the auxiliary stream the
// host-side glue (optimizer overlap, gradient exchange) runs on.  One-time cost: a few milliseconds per engine.
#include <atomic>
#include <mutex>
#include <vector>

#include "common.hip.h"
#include "crct_internal.h"

namespace {

__global__ void spin_kernel(long long ticks) {          // s_memrealtime: 100 MHz wall clock
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// Footprint of a ring all-reduce on THIS GPU without peers (crct_ghost_collective): `channels` workgroups -- RCCL runs one per channel --
// stream the payload through HBM (read + write back the same bytes: every element belongs to one thread, nothing changes), `passes` times
// (a ring all-reduce over N ranks reads and writes ~2 (N - 1) / N of the buffer), and pace themselves so that the whole kernel takes
// `ticks` of the 100 MHz wall clock -- the time the real collective would hold its channels while the bytes cross xGMI.
typedef unsigned ghost_u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void ghost_collective_kernel(ghost_u4* __restrict__ buf, long n16, int passes, long long ticks) {
  const long long t0 = wall_clock64();
  const long per = (n16 + gridDim.x - 1) / gridDim.x, lo = (long)blockIdx.x * per, hi = lo + per < n16 ? lo + per : n16;
  constexpr int SLICES = 16;
  const long span = hi > lo ? hi - lo : 0, sl = (span + SLICES - 1) / SLICES;
  for (int p = 0; p < passes; ++p)
    for (int s = 0; s < SLICES; ++s) {
      const long a = lo + (long)s * sl, b = a + sl < hi ? a + sl : hi;
      for (long i = a + threadIdx.x; i < b; i += 256) {
        const ghost_u4 v = __builtin_nontemporal_load(buf + i);
        __builtin_nontemporal_store(v, buf + i);
      }
      const long long due = ticks * (long long)(p * SLICES + s + 1) / (long long)(passes * SLICES);
      while (wall_clock64() - t0 < due) __builtin_amdgcn_s_sleep(16);
    }
}

// do kernels on streams a and b serialise?  (both streams idle on entry)
int conflict(hipStream_t a, hipStream_t b, hipEvent_t e0, hipEvent_t e1, bool* out) {
  constexpr long long SPIN_US = 120;
  CRCT_CHECK_HIP(hipEventRecord(e0, a));
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, SPIN_US * 100);
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, SPIN_US * 100);
  CRCT_CHECK_HIP(hipEventRecord(e1, b));
  CRCT_CHECK_HIP(hipEventSynchronize(e1));
  CRCT_CHECK_HIP(hipStreamSynchronize(a));
  float ms = 0.f;
  CRCT_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
  *out = ms * 1e3f > 1.6f * SPIN_US;                     // concurrent: ~1.0-1.2 x, serialised: >= 2 x
  return 0;
}

}  // namespace

// Stand-in for a collective that cannot run on a one-GPU box (bench.py --ghost-ranks N): occupies `channels` workgroups on `stream` for
// `microseconds`, streaming `bytes` at `ptr` (16-byte aligned; left unchanged) through HBM `passes` times.  The data-parallel exchange
// launches it per bucket where ncclAllReduce would run, so that what RCCL will take from the step -- CUs, HBM bandwidth, a hardware
// queue, for as long as xGMI needs -- is in the measured step today.  Not a collective: nothing is reduced.
extern "C" int crct_ghost_collective(void* ptr, int64_t bytes, int channels, int passes, double microseconds, crct_stream_t stream) {
  CRCT_REQUIRE(ptr && bytes >= 16 && channels >= 1 && passes >= 1 && microseconds >= 0, "ghost_collective: bad arguments");
  crct_launch(ghost_collective_kernel, dim3((unsigned)channels), dim3(256), 0, (hipStream_t)stream, (ghost_u4*)ptr, (long)(bytes / 16), passes,
              (long long)(microseconds * 100.0));
  CRCT_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------- kernel stamps
namespace {
struct Stamp { hipEvent_t a, b; hipStream_t s; bool own; };
std::vector<Stamp> g_stamps;          // slots [0, g_used) belong to the running collection; events of `own` slots are reused after a reset
size_t g_used = 0;
std::atomic<bool> g_stamp_on{false};
std::mutex g_stamp_mu;                // forward and backward are enqueued by different host threads (autograd's worker): one store, one lock
}  // namespace
void crct_stamp_enable(int on) { g_stamp_on.store(on != 0); }
void crct_stamp_reset(void) {
  std::lock_guard<std::mutex> lk(g_stamp_mu);
  // adopted pairs belong to the GEMM profile (gemm.hip): drop their slots, keep our own events for reuse
  std::vector<Stamp> keep;
  for (const Stamp& st : g_stamps) if (st.own) keep.push_back(st);
  g_stamps.swap(keep);
  g_used = 0;
}
bool crct_stamp_begin(hipStream_t s, hipEvent_t* start, hipEvent_t* stop) {
  if (!g_stamp_on.load(std::memory_order_relaxed)) return false;
  std::lock_guard<std::mutex> lk(g_stamp_mu);
  while (g_used < g_stamps.size() && !g_stamps[g_used].own) ++g_used;          // (adopted slots sit where they were appended)
  if (g_used == g_stamps.size()) {
    Stamp st; st.own = true; st.s = s;
    if (hipEventCreate(&st.a) != hipSuccess || hipEventCreate(&st.b) != hipSuccess) return false;
    g_stamps.push_back(st);
  }
  Stamp& st = g_stamps[g_used++];
  st.s = s;
  *start = st.a; *stop = st.b;
  return true;
}
void crct_stamp_adopt(hipStream_t s, hipEvent_t start, hipEvent_t stop) {
  if (!g_stamp_on.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lk(g_stamp_mu);
  Stamp st; st.a = start; st.b = stop; st.s = s; st.own = false;
  g_stamps.insert(g_stamps.begin() + (long)g_used, st);
  ++g_used;
}
// Number of stamped launches since the last reset / the i-th one: its stream and its begin / end in milliseconds after the FIRST stamped
// launch began (synchronises on the events).
extern "C" int crct_prof_stamp_count(void) { std::lock_guard<std::mutex> lk(g_stamp_mu); return (int)g_used; }
extern "C" int crct_prof_stamp_read(int i, void** stream, double* t0_ms, double* t1_ms) {
  std::lock_guard<std::mutex> lk(g_stamp_mu);
  if (i < 0 || (size_t)i >= g_used || !stream || !t0_ms || !t1_ms) return 1;
  const Stamp st = g_stamps[(size_t)i];
  CRCT_CHECK_HIP(hipEventSynchronize(st.b));
  float a = 0.f, d = 0.f;
  if (i > 0) {
    CRCT_CHECK_HIP(hipEventSynchronize(g_stamps[0].a));
    CRCT_CHECK_HIP(hipEventElapsedTime(&a, g_stamps[0].a, st.a));
  }
  CRCT_CHECK_HIP(hipEventElapsedTime(&d, st.a, st.b));
  *stream = (void*)st.s; *t0_ms = (double)a; *t1_ms = (double)a + (double)d;
  return 0;
}

// out[0..2]: streams of three queue classes other than main's (nullptr where fewer classes exist), out[3]: a second stream of
// out[2]'s class (or nullptr); n_classes = queue classes seen (main's included).  Streams not handed out are destroyed.
int crct_streams_place(hipStream_t main, hipStream_t out[4], int* n_classes) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  CRCT_CHECK_HIP(hipEventCreate(&e0));
  CRCT_CHECK_HIP(hipEventCreate(&e1));
  CRCT_CHECK_HIP(hipStreamSynchronize(main));
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, main, 100);      // code object load / first-launch cost out of the probes
  CRCT_CHECK_HIP(hipStreamSynchronize(main));
  std::vector<std::vector<hipStream_t>> cls(1);
  cls[0].push_back(main);
  std::vector<hipStream_t> mine;
  int rc = 0;
  for (int i = 0; i < 16 && !rc; ++i) {
    bool done = cls.size() >= 4;
    for (size_t c = 1; c < cls.size() && done; ++c) done = cls[c].size() >= (c + 1 == cls.size() ? 2u : 1u);
    if (done) break;
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { crct_set_error("streams: cannot create a HIP stream"); rc = 1; break; }
    mine.push_back(s);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, 100);
    (void)hipStreamSynchronize(s);
    size_t home = cls.size();
    for (size_t c = 0; c < cls.size() && !rc; ++c) {
      bool same = false;
      rc = conflict(cls[c][0], s, e0, e1, &same);
      if (same) { home = c; break; }
    }
    if (rc) break;
    if (home == cls.size()) cls.emplace_back();
    cls[home].push_back(s);
    if (i == 5 && cls.size() == 1) break;      // six streams in a row serialise with the caller's: a profiler (rocprofv3 --pmc) or a
  }                                            // one-queue configuration serialises everything -- placement is moot, take any streams
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  for (int k = 0; k < 4; ++k) out[k] = nullptr;
  if (!rc) {
    for (size_t c = 1; c < cls.size() && c <= 3; ++c) out[c - 1] = cls[c][0];
    const size_t last = cls.size() > 3 ? 3 : cls.size() - 1;
    if (last >= 1 && cls[last].size() > 1) out[3] = cls[last][1];
    if (n_classes) *n_classes = (int)cls.size();
  }
  for (hipStream_t s : mine) {
    bool kept = false;
    for (int k = 0; k < 4; ++k) kept = kept || out[k] == s;
    if (!kept) (void)hipStreamDestroy(s);
  }
  for (int k = 0; k < 3 && !rc; ++k)           // fewer than four queue classes: plain streams, wherever they land
    if (!out[k] && hipStreamCreateWithFlags(&out[k], hipStreamNonBlocking) != hipSuccess) { crct_set_error("streams: cannot create a HIP stream"); rc = 1; }
  return rc;
}

#ifdef CRCT_GEMM_LAB
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
namespace {
struct LabSkip {
  std::mutex mu;
  std::vector<std::pair<std::string, int>> rules;       // (substring, stream ordinal or -1)
  std::vector<hipStream_t> order;
  std::map<std::pair<int, std::string>, std::pair<long, long>> seen;   // (ordinal, name) -> (launched, skipped)
  bool report = false;
  LabSkip() {
    const char* e = getenv("CRCT_LAB_SKIP");
    report = getenv("CRCT_LAB_REPORT") != nullptr;
    if (!e) return;
    std::string all(e);
    size_t i = 0;
    while (i <= all.size()) {
      size_t j = all.find(',', i);
      if (j == std::string::npos) j = all.size();
      std::string r = all.substr(i, j - i);
      if (!r.empty()) {
        int ord = -1;
        const size_t at = r.find('@');
        if (at != std::string::npos) { ord = atoi(r.c_str() + at + 1); r = r.substr(0, at); }
        rules.emplace_back(r, ord);
      }
      i = j + 1;
    }
  }
  ~LabSkip() {
    if (!report) return;
    for (const auto& kv : seen)
      fprintf(stderr, "[crct lab] stream %d %-60.60s launched %ld skipped %ld\n", kv.first.first, kv.first.second.c_str(), kv.second.first, kv.second.second);
  }
};
LabSkip g_lab_skip;
}  // namespace
bool crct_lab_skip(const void* kern, hipStream_t s) {
  LabSkip& L = g_lab_skip;
  if (L.rules.empty() && !L.report) return false;
  std::lock_guard<std::mutex> lk(L.mu);
  int ord = -1;
  for (size_t i = 0; i < L.order.size(); ++i) if (L.order[i] == s) ord = (int)i;
  if (ord < 0) { ord = (int)L.order.size(); L.order.push_back(s); }
  const char* nm = hipKernelNameRefByPtr(kern, s);
  const std::string name(nm ? nm : "?");
  bool skip = false;
  for (const auto& r : L.rules)
    if ((r.second < 0 || r.second == ord) && (r.first == "*" || name.find(r.first) != std::string::npos)) skip = true;
  if (L.report) { auto& c = L.seen[{ord, name}]; (skip ? c.second : c.first) += 1; }
  return skip;
}
#endif
