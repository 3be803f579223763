// Internal declarations shared by the kernel translation units and the step engine.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "../../include/crct_hip.h"

void crct_set_error(const char* fmt, ...);
#define CRCT_CHECK_HIP(expr)                                                          \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) {                                                           \
      crct_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)
#define CRCT_REQUIRE(cond, ...)                 \
  do {                                          \
    if (!(cond)) {                              \
      crct_set_error(__VA_ARGS__);              \
      return 2;                                 \
    }                                           \
  } while (0)

hipError_t crct_gemm_launch(const CrctGemmArgs& g, hipStream_t s);
hipError_t crct_gemm_launch_grouped(const CrctGemmArgs* gs, int n, hipStream_t s);
// target_wgs > 0: a grouped bf16 weight-gradient launch runs as a persistent grid of about that many workgroups (gemm.hip, group_grid)
hipError_t crct_gemm_launch_grouped_wgs(const CrctGemmArgs* gs, int n, hipStream_t s, int target_wgs);
// streams.hip: three streams on hardware queues other than main's (+ a second one on the last queue), found by probing
int crct_streams_place(hipStream_t main, hipStream_t out[4], int* n_classes);

// upper bound of the workgroup count of the LayerNorm-backward style kernels (4 rows per workgroup and pass):
// sizes the column-partials scratch [partials][blocks][H]
#define CRCT_LN_BWD_MAX_BLOCKS 256

// ---- live stamps of EVERY kernel the library launches (crct_prof_enable(2); bench.py config.critical_path): while armed, a launch is
// dispatched with a start / stop event pair (hipExtLaunchKernelGGL: the begin / end stamps of that kernel, what rocprofv3's kernel
// trace reports) and remembered with its stream.  Off (the product): one predictable branch per launch.
bool crct_stamp_begin(hipStream_t s, hipEvent_t* start, hipEvent_t* stop);      // streams.hip; false = stamping is off
void crct_stamp_adopt(hipStream_t s, hipEvent_t start, hipEvent_t stop);        // a GEMM launch that carries its own event pair (gemm.hip)
void crct_stamp_enable(int on);
void crct_stamp_reset(void);
#ifdef CRCT_GEMM_LAB
// LAB build only (make lab -> tools/lab/libcrct_lab.so, never the package's library): launches whose kernel name contains one of the
// comma-separated entries of $CRCT_LAB_SKIP ("name" or "name@k", k = ordinal of the stream in order of first use) are LEFT OUT --
// wrong results, timing only: what the step gains when a class of kernels costs nothing (tools/lab/step_sensitivity.sh).
bool crct_lab_skip(const void* kern, hipStream_t s);
#endif
template <class K, class... A>
inline void crct_launch(K kern, dim3 grid, dim3 block, size_t lds, hipStream_t s, A... a) {
#ifdef CRCT_GEMM_LAB
  if (crct_lab_skip((const void*)kern, s)) return;
#endif
  hipEvent_t e0, e1;
  if (crct_stamp_begin(s, &e0, &e1)) hipExtLaunchKernelGGL(kern, grid, block, (uint32_t)lds, s, e0, e1, 0u, a...);
  else hipLaunchKernelGGL(kern, grid, block, lds, s, a...);
}
