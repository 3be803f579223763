// Developer microbenchmark: cost of a cross-stream dependency (event record on one stream, wait on the other) on this stack.
//   hipcc -O2 --offload-arch=gfx950 tools/handoff_lab.cpp -o tools/handoff_lab.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__global__ void spin(long cycles, int* sink) {
  const long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  if (sink && threadIdx.x == 9999) *sink = 1;
}
int main(int argc, char** argv) {
  hipStream_t s0, s1; hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  const int N = 400;
  std::vector<hipEvent_t> ev(2 * N);
  const unsigned flags = argc > 1 ? (unsigned)strtoul(argv[1], nullptr, 0) : (unsigned)hipEventDisableTiming;
  printf("event flags 0x%x (DisableTiming 0x%x, DisableSystemFence 0x%x, ReleaseToDevice 0x%x)\n", flags, hipEventDisableTiming,
         hipEventDisableSystemFence, hipEventReleaseToDevice);
  for (auto& e : ev) if (hipEventCreateWithFlags(&e, flags) != hipSuccess) { printf("event flags rejected\n"); return 1; }
  hipEvent_t t0, t1; hipEventCreate(&t0); hipEventCreate(&t1);
  for (long cyc : {2000L, 20000L}) {          // ~1 us and ~10 us kernels (100 MHz counter? measured below)
    float ms_chain, ms_pp, ms_pp2;
    // (a) 2N kernels back to back on one stream
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(t0, s0);
      for (int i = 0; i < 2 * N; ++i) hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s0, cyc, (int*)nullptr);
      hipEventRecord(t1, s0); hipEventSynchronize(t1); hipEventElapsedTime(&ms_chain, t0, t1);
    }
    // (b) ping-pong: kernel on s0 -> event -> kernel on s1 -> event -> s0 ...
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(t0, s0);
      for (int i = 0; i < N; ++i) {
        hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s0, cyc, (int*)nullptr);
        hipEventRecord(ev[2 * i], s0); hipStreamWaitEvent(s1, ev[2 * i], 0);
        hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s1, cyc, (int*)nullptr);
        hipEventRecord(ev[2 * i + 1], s1); hipStreamWaitEvent(s0, ev[2 * i + 1], 0);
      }
      hipEventRecord(t1, s0); hipEventSynchronize(t1); hipEventElapsedTime(&ms_pp, t0, t1);
    }
    // (c) both streams run their own chain and exchange events both ways after every kernel (the co-attention pattern)
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(t0, s0);
      hipEventRecord(ev[0], s0); hipStreamWaitEvent(s1, ev[0], 0);
      for (int i = 0; i < N - 1; ++i) {
        hipLaunchKernelGGL(spin, dim3(128), dim3(64), 0, s0, cyc, (int*)nullptr);
        hipLaunchKernelGGL(spin, dim3(128), dim3(64), 0, s1, cyc, (int*)nullptr);
        hipEventRecord(ev[2 * i + 1], s0); hipEventRecord(ev[2 * i + 2], s1);
        hipStreamWaitEvent(s1, ev[2 * i + 1], 0); hipStreamWaitEvent(s0, ev[2 * i + 2], 0);
      }
      hipEventRecord(t1, s0); hipEventSynchronize(t1); hipEventElapsedTime(&ms_pp2, t0, t1);
    }
    // (d) one chain, every kernel preceded by a wait on an event of the OTHER stream that has long been signalled
    float ms_old;
    hipEventRecord(ev[0], s1); hipStreamSynchronize(s1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(t0, s0);
      for (int i = 0; i < 2 * N; ++i) { hipStreamWaitEvent(s0, ev[0], 0); hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s0, cyc, (int*)nullptr); }
      hipEventRecord(t1, s0); hipEventSynchronize(t1); hipEventElapsedTime(&ms_old, t0, t1);
    }
    // (f) one chain with an event RECORD after every kernel (nobody waits): what a record costs the recording stream;
    // (g) the same with the event attached to the kernel's own dispatch packet (hipExtLaunchKernelGGL stop event)
    float ms_rec, ms_ext;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(t0, s0);
      for (int i = 0; i < 2 * N; ++i) { hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s0, cyc, (int*)nullptr); hipEventRecord(ev[i], s0); }
      hipEventRecord(t1, s0); hipEventSynchronize(t1); hipEventElapsedTime(&ms_rec, t0, t1);
    }
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(t0, s0);
      for (int i = 0; i < 2 * N; ++i) hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s0, nullptr, ev[i], 0, cyc, (int*)nullptr);
      hipEventRecord(t1, s0); hipEventSynchronize(t1); hipEventElapsedTime(&ms_ext, t0, t1);
    }
    printf("          event record after every kernel: %.2f us/kernel (+%.2f us) | as the kernel's stop event: %.2f us/kernel (+%.2f us)\n",
           ms_rec * 1e3 / (2 * N), (ms_rec - ms_chain) * 1e3 / (2 * N), ms_ext * 1e3 / (2 * N), (ms_ext - ms_chain) * 1e3 / (2 * N));
    // (e) ping-pong through stream memory operations (write a counter after the kernel, the other stream waits for it)
    float ms_val = -1.f;
    {
      uint32_t* flag = nullptr;
      if (hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory) == hipSuccess && hipMemset(flag, 0, 8) == hipSuccess) {
        hipDeviceSynchronize();
        uint32_t v = 0;
        bool ok = true;
        for (int rep = 0; rep < 2 && ok; ++rep) {
          hipEventRecord(t0, s0);
          for (int i = 0; i < N && ok; ++i) {
            hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s0, cyc, (int*)nullptr);
            ok = ok && hipStreamWriteValue32(s0, flag, ++v, 0) == hipSuccess;
            ok = ok && hipStreamWaitValue32(s1, flag, v, hipStreamWaitValueGte, 0xffffffffu) == hipSuccess;
            hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, s1, cyc, (int*)nullptr);
            ok = ok && hipStreamWriteValue32(s1, flag + 1, v, 0) == hipSuccess;
            ok = ok && hipStreamWaitValue32(s0, flag + 1, v, hipStreamWaitValueGte, 0xffffffffu) == hipSuccess;
          }
          hipEventRecord(t1, s0); hipEventSynchronize(t1); hipEventElapsedTime(&ms_val, t0, t1);
        }
        if (!ok) { ms_val = -1.f; printf("stream memory operations failed: %s\n", hipGetErrorString(hipGetLastError())); }
      } else printf("no signal memory: %s\n", hipGetErrorString(hipGetLastError()));
    }
    printf("          wait on a long-signalled event before every kernel: %.2f us/kernel (+%.2f us) | ping-pong via stream write/wait value: %.2f us/kernel (handoff +%.2f us)\n",
           ms_old * 1e3 / (2 * N), (ms_old - ms_chain) * 1e3 / (2 * N), ms_val * 1e3 / (2 * N), (ms_val - ms_chain) * 1e3 / (2 * N));
    printf("spin %6ld: one stream %.2f us/kernel | ping-pong %.2f us/kernel (handoff +%.2f us) | two chains with a 2-way sync per kernel %.2f us/step (+%.2f us)\n",
           cyc, ms_chain * 1e3 / (2 * N), ms_pp * 1e3 / (2 * N), (ms_pp - ms_chain) * 1e3 / (2 * N), ms_pp2 * 1e3 / (N - 1),
           ms_pp2 * 1e3 / (N - 1) - ms_chain * 1e3 / (2 * N));
  }
  return 0;
}
