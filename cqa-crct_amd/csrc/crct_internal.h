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
// target_wgs > 0: a grouped bf16 weight-gradient launch runs as a persistent grid of about that many workgroups (gemm.hip, group_grid)
hipError_t crct_gemm_launch_grouped_wgs(const CrctGemmArgs* gs, int n, hipStream_t s, int target_wgs);
// streams.hip: three streams on hardware queues other than main's (+ a second one on the last queue), found by probing
int crct_streams_place(hipStream_t main, hipStream_t out[4], int* n_classes);

// Lab hook (crct_lab_xcd_band): > 0 = rows per XCD band.  The GEMM tile maps then give XCD x (= block % 8)
// the row tiles of band x and ALL column tiles, and the LayerNorm forward maps workgroup b to rows of band b % 8 -- an
// activation row is produced and consumed on one XCD (its private L2) through a LayerNorm -> GEMM -> GEMM chain.  0 (the product):
// the rectangle maps of make_tile_map / rows dealt round-robin.  Placement only ever affects speed.
extern int g_crct_lab_band_rows;

// upper bound of the workgroup count of the LayerNorm-backward style kernels (4 rows per workgroup and pass):
// sizes the column-partials scratch [partials][blocks][H]
#define CRCT_LN_BWD_MAX_BLOCKS 256
