// What does an event record cost the stream it is recorded on?  A GPU-bound dependent chain of ~8 us kernels: plain; with hipEventRecord
// behind every kernel (a marker packet between two kernels of the chain); with the event attached to the kernel itself
// (hipExtLaunchKernelGGL stop event: the kernel's own completion signal, no extra packet); and with a second stream waiting for each
// event (hipStreamWaitEvent + a small kernel), as the step's weight-gradient streams do.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 -o event_probe.bin event_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(256) void k(float* p, const float* q, int n, float s) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = q[i] * s + 1.0f;
  for (int t = 0; t < 10; ++t) __builtin_amdgcn_s_sleep(32);
}
__global__ __launch_bounds__(256) void side(float* p, int n) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] += 1.0f; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipSetDevice(0);
  const int n = 256 * 256, N = 3000;
  float *p, *q, *r; hipMalloc(&p, n * 4); hipMalloc(&q, n * 4); hipMalloc(&r, n * 4); hipMemset(q, 0, n * 4); hipMemset(r, 0, n * 4);
  hipStream_t s, w; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipStreamCreateWithFlags(&w, hipStreamNonBlocking);
  std::vector<hipEvent_t> ev(N);
  // the step engine's events: no timing, no system-scope fence at the record (ordering between streams of one device)
  const unsigned flags = getenv("PROBE_SYSTEM_FENCE") ? hipEventDisableTiming : (hipEventDisableTiming | hipEventDisableSystemFence);
  printf("events: %s\n", getenv("PROBE_SYSTEM_FENCE") ? "hipEventDisableTiming" : "hipEventDisableTiming | hipEventDisableSystemFence");
  for (auto& e : ev) hipEventCreateWithFlags(&e, flags);
  for (int rep = 0; rep < 3; ++rep) {
    double t[6];
    hipDeviceSynchronize(); t[0] = now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, s, (i & 1) ? p : q, (const float*)((i & 1) ? q : p), n, 0.5f);
    hipDeviceSynchronize(); t[1] = now();
    for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, s, (i & 1) ? p : q, (const float*)((i & 1) ? q : p), n, 0.5f); hipEventRecord(ev[i], s); }
    hipDeviceSynchronize(); t[2] = now();
    for (int i = 0; i < N; ++i) hipExtLaunchKernelGGL(k, dim3(256), dim3(256), 0, s, nullptr, ev[i], 0, (i & 1) ? p : q, (const float*)((i & 1) ? q : p), n, 0.5f);
    hipDeviceSynchronize(); t[3] = now();
    for (int i = 0; i < N; ++i) {
      hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, s, (i & 1) ? p : q, (const float*)((i & 1) ? q : p), n, 0.5f);
      hipEventRecord(ev[i], s); hipStreamWaitEvent(w, ev[i], 0); hipLaunchKernelGGL(side, dim3(64), dim3(256), 0, w, r, n);
    }
    hipDeviceSynchronize(); t[4] = now();
    for (int i = 0; i < N; ++i) {
      hipExtLaunchKernelGGL(k, dim3(256), dim3(256), 0, s, nullptr, ev[i], 0, (i & 1) ? p : q, (const float*)((i & 1) ? q : p), n, 0.5f);
      hipStreamWaitEvent(w, ev[i], 0); hipLaunchKernelGGL(side, dim3(64), dim3(256), 0, w, r, n);
    }
    hipDeviceSynchronize(); t[5] = now();
    printf("rep %d, us per link of the chain: plain %.2f | + event record %.2f | event on the kernel (hipExtLaunch stop event) %.2f | record + a waiting stream %.2f | stop event + a waiting stream %.2f\n",
           rep, (t[1] - t[0]) / N * 1e6, (t[2] - t[1]) / N * 1e6, (t[3] - t[2]) / N * 1e6, (t[4] - t[3]) / N * 1e6, (t[5] - t[4]) / N * 1e6);
  }
  // host cost: an empty kernel, 20000 launches, the queue kept short by a sync every 2000
  {
    hipEvent_t ring[64];
    for (auto& e : ring) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    double h0 = 0, h1 = 0;
    for (int b = 0; b < 10; ++b) {
      hipDeviceSynchronize(); double a = now();
      for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(side, dim3(1), dim3(64), 0, s, r, 0);
      double c = now(); hipDeviceSynchronize(); double d = now();
      for (int i = 0; i < 2000; ++i) hipExtLaunchKernelGGL(side, dim3(1), dim3(64), 0, s, nullptr, ring[i & 63], 0, r, 0);
      double f = now();
      h0 += c - a; h1 += f - d;
    }
    hipDeviceSynchronize();
    printf("host cost per launch: hipLaunchKernelGGL %.2f us | hipExtLaunchKernelGGL with a stop event (ring of 64, timing disabled) %.2f us\n", h0 / 20000 * 1e6, h1 / 20000 * 1e6);
  }
  return 0;
}
