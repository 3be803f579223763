// Internal declarations shared by the kernel translation units and the step engine.
#pragma once
#include <hip/hip_runtime.h>
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
// streams.hip: three streams on hardware queues other than main's (+ a second one on the last queue), found by probing
int crct_streams_place(hipStream_t main, hipStream_t out[4], int* n_classes);

// upper bound of the workgroup count of the LayerNorm-backward style kernels (4 rows per workgroup and pass):
// sizes the column-partials scratch [partials][blocks][H]
#define CRCT_LN_BWD_MAX_BLOCKS 256
