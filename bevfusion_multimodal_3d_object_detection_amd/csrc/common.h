// Shared helpers for the gfx950 kernels of libbevf_hip.so (internal header).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/bevf.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void bevf_set_error(const char* fmt, ...);

#define BEVF_REQUIRE(cond, ...)              \
  do {                                       \
    if (!(cond)) {                           \
      bevf_set_error(__VA_ARGS__);           \
      return BEVF_ERR_ARG;                   \
    }                                        \
  } while (0)

static inline int bevf_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    bevf_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return BEVF_ERR_LAUNCH;
  }
  return BEVF_OK;
}

static inline bool bevf_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Blocks b and b+8 share an XCD (one L2 each): give every XCD a contiguous range of tiles so
// that neighbouring tiles -- which share input rows and the weight panel -- hit the same L2.
// Bijective for any grid size (cdna_hip_programming.md, 256^2 template, "XCD swizzle").
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}
