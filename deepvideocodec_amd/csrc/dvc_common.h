// Shared helpers for the libdvc_hip.so translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/dvc_hip.h"

namespace dvc {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return DVC_EHIP;
  }
  return DVC_OK;
}

#define DVC_REQUIRE(cond, ...)       \
  do {                               \
    if (!(cond)) {                   \
      dvc::set_error(__VA_ARGS__);   \
      return DVC_EINVAL;             \
    }                                \
  } while (0)

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

// Grid size for a grid-stride elementwise launch: enough blocks to fill the
// 256 CUs several times over without launching millions of tiny blocks.
inline int stream_grid(int64_t work_items, int block) {
  int64_t g = (work_items + block - 1) / block;
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace dvc
