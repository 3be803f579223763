// Does kernel-argument PRELOAD (gfx950: the first <= 16 argument dwords arrive in SGPRs with the wave, -mllvm -amdgpu-kernarg-preload-count=16;
// only for scalar / pointer arguments, not for structs passed by value) shorten a dependent chain of small kernels?  Same kernel body with
// its arguments as one struct (no preload) and as leading scalars (preload), 256 workgroups, 4000 launches back to back on one stream.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 -o kernarg_probe.bin kernarg_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Args { float* p; const float* q; int n; float a; long pad[40]; };
__global__ __launch_bounds__(256) void k_struct(const Args a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < a.n) a.p[i] = a.q[i] * a.a + 1.0f;
  for (int k = 0; k < 10; ++k) __builtin_amdgcn_s_sleep(32);       // ~8 us: the chain is GPU-bound, not enqueue-bound
}
__global__ __launch_bounds__(256) void k_scalar(float* p, const float* q, int n, float s, const Args rest) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = q[i] * s + 1.0f;
  for (int k = 0; k < 10; ++k) __builtin_amdgcn_s_sleep(32);
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipSetDevice(0);
  const int n = 256 * 256;
  float *p, *q; hipMalloc(&p, n * 4); hipMalloc(&q, n * 4); hipMemset(q, 0, n * 4);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  Args a{}; a.p = p; a.q = q; a.n = n; a.a = 0.5f;
  const int N = 4000;
  for (int rep = 0; rep < 3; ++rep) {
    hipStreamSynchronize(s);
    double t0 = now();
    for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(k_struct, dim3(256), dim3(256), 0, s, a); Args b = a; b.p = (i & 1) ? p : q; b.q = (i & 1) ? q : p; a = b; }
    hipStreamSynchronize(s);
    double t1 = now();
    for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(k_scalar, dim3(256), dim3(256), 0, s, (i & 1) ? p : q, (const float*)((i & 1) ? q : p), n, 0.5f, a); }
    hipStreamSynchronize(s);
    double t2 = now();
    printf("rep %d: dependent chain, per launch: arguments as a struct %.2f us | leading scalars (preloaded) %.2f us\n", rep, (t1 - t0) / N * 1e6, (t2 - t1) / N * 1e6);
  }
  return 0;
}
