// Can two host threads enqueue faster than one?  N small kernels to each of two streams: one thread alternating between the streams
// against two threads, one per stream (with and without an event hand-off every 8 launches).  Host time per launch, GPU idle otherwise.
//   hipcc --offload-arch=gfx950 -O2 -o enqueue_probe.bin enqueue_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
struct Args { float* p; int n; float a, b, c, d; long e, f; };
__global__ void tiny(Args a) { if (a.n < 0) a.p[threadIdx.x] = a.a; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipSetDevice(0);
  float* buf; hipMalloc(&buf, 1024);
  hipStream_t s[2]; hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking); hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking);
  const int N = 4000;
  Args a{buf, 0, 1, 2, 3, 4, 5, 6};
  std::vector<hipEvent_t> ev(N);
  for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  for (int rep = 0; rep < 3; ++rep) {
    hipDeviceSynchronize();
    double t0 = now();
    for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s[0], a); hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s[1], a); }
    double t1 = now(); hipDeviceSynchronize(); double t1b = now();
    auto worker = [&](int k) { hipSetDevice(0); for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s[k], a); };
    double t2 = now();
    { std::thread th(worker, 1); worker(0); th.join(); }
    double t3 = now(); hipDeviceSynchronize(); double t3b = now();
    // with hand-offs: every 8 launches stream 0 records an event that stream 1 waits for (one thread)
    double t4 = now();
    for (int i = 0; i < N; ++i) {
      hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s[0], a);
      if (i % 8 == 7) { hipEventRecord(ev[i], s[0]); hipStreamWaitEvent(s[1], ev[i], 0); }
      hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s[1], a);
    }
    double t5 = now(); hipDeviceSynchronize(); double t5b = now();
    // two threads, the waiter spins on a host ticket until the record has been issued
    std::atomic<int> ticket{-1};
    auto prod = [&]() { for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s[0], a); if (i % 8 == 7) { hipEventRecord(ev[i], s[0]); ticket.store(i, std::memory_order_release); } } };
    auto cons = [&]() { hipSetDevice(0); for (int i = 0; i < N; ++i) { if (i % 8 == 7) { while (ticket.load(std::memory_order_acquire) < i) {} hipStreamWaitEvent(s[1], ev[i], 0); } hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s[1], a); } };
    double t6 = now();
    { std::thread th(cons); prod(); th.join(); }
    double t7 = now(); hipDeviceSynchronize(); double t7b = now();
    printf("rep %d: per launch, host (until drained): one thread %.2f us (%.2f) | two threads %.2f us (%.2f) | one thread + hand-offs %.2f (%.2f) | two threads + ticketed hand-offs %.2f (%.2f)\n", rep,
           (t1 - t0) / (2 * N) * 1e6, (t1b - t0) / (2 * N) * 1e6, (t3 - t2) / (2 * N) * 1e6, (t3b - t2) / (2 * N) * 1e6,
           (t5 - t4) / (2 * N) * 1e6, (t5b - t4) / (2 * N) * 1e6, (t7 - t6) / (2 * N) * 1e6, (t7b - t6) / (2 * N) * 1e6);
  }
  return 0;
}
