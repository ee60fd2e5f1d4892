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

// ---- 16-byte channel vectors of either storage type (fp32: 4 channels, bf16: 8 channels) ----------------------
template <typename T> struct vec16 { static constexpr int N = 16 / (int)sizeof(T); };
template <typename T>
__device__ __forceinline__ void load16(const T* p, float (&f)[vec16<T>::N]) {
  if constexpr (sizeof(T) == 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = v[j];
  } else {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (float)v[j];
  }
}
template <typename T>
__device__ __forceinline__ void store16(T* p, const float (&f)[vec16<T>::N]) {
  if constexpr (sizeof(T) == 4) {
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = f[j];
    *reinterpret_cast<f32x4*>(p) = v;
  } else {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    bf16x8_t v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)f[j];
    *reinterpret_cast<bf16x8_t*>(p) = v;
  }
}
