// Pieces shared by the implicit-GEMM convolution kernels (conv_igemm.hip: exact fp32 / bf16; conv_split.hip: fp32
// through three bf16 planes): launch arguments, buffer-load helper, multiply-high division, and the epilogue.
#pragma once
#include "common.h"

#include <type_traits>

namespace {

// Element type T of activations / weights: float (v_mfma_f32_32x32x2_f32, exact fp32) or __bf16
// (v_mfma_f32_32x32x16_bf16, fp32 accumulate).  Both share the byte geometry: a K step is 8 chunks of 16 B per
// row (32 floats or 64 bf16), so staging, LDS layout, swizzle and fragment reads are identical; for bf16 the
// 16-B fragment a lane reads (k = 8h..8h+7 of a 16-deep group) is exactly the MFMA's operand layout.
struct ConvArgs {
  const void* x;
  const void* w;
  const float* scale;
  const float* shift;
  const void* res;
  void* y;
  uint32_t* colmax;
  int N, H, W, Cin, x_cs;
  int Ho, Wo, Cout, y_cs, res_cs;
  int KH, KW, stride, pad;
  int relu, rows_per_group;
  int M, K, tilesM, tilesN;
  int m_split, nbig, tilesN_big;   // hybrid launch: blocks [0,nbig) = big tiles over rows [0,m_split)
  unsigned div_hw_mul, div_hw_sh, div_w_mul, div_w_sh;   // m / (Ho*Wo) and r / Wo as multiply-high + shift
};

constexpr int BK = 32;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr unsigned kOob = 0x80000000u;   // >= num_records of every descriptor below: the load returns 0

__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
  return __builtin_bit_cast(f32x4, v);
}

// Division of a dividend < 2^31 by a launch constant: q = umulhi(n, mul) >> sh with mul = ceil(2^(31+s)/d),
// s = ceil(log2 d), sh = s-1 (exact for every n < 2^31 because the rounding error e < d <= 2^s); mul = 0 means d = 1.
__device__ __forceinline__ int fastdiv(int n, unsigned mul, unsigned sh) {
  return mul ? (int)(__umulhi((unsigned)n, mul) >> sh) : n;
}
static void fastdiv_make(int d, unsigned* mul, unsigned* sh) {
  if (d <= 1) { *mul = 0; *sh = 0; return; }
  int s = 0;
  while ((1ll << s) < d) ++s;
  *mul = (unsigned)(((1ull << (31 + s)) + (unsigned long long)d - 1) / (unsigned long long)d);
  *sh = (unsigned)(s - 1);
}

// Epilogue of one workgroup tile.  acc[mi][ni] is the 32x32 MFMA tile (mi, ni) of this wave: C[i][j] with
// j = lane&31 (channel) and i = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel).  T = storage type of y / res.
template <typename T, int BM, int BN, int WM, int WN>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x16 (&acc)[WM / 32][WN / 32], const int m0,
                                              const int n0, const int m_hi, const int wm, const int wn, const int lane,
                                              char* const scratch = nullptr) {
  constexpr int ES = (int)sizeof(T);
  constexpr bool kF32 = std::is_same<T, float>::value;
  constexpr int MI = WM / 32, NI = WN / 32;
  const int h = lane >> 5, l31 = lane & 31;
  // ---- epilogue: C[i][j], j = lane&31 (channel), i = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel) ----
  // Branch-free per element: residual loads of a 32x32 tile are issued as one batch (rows past
  // the end are clamped for the load and masked at the store).
  const int m_base = m0 + wm * WM, n_base = n0 + wn * WN;
  if (!p.colmax && m0 + BM <= m_hi && n0 + BN <= p.Cout) {
    // full tile: buffer stores whose row offset is the instruction's SCALAR offset (SALU arithmetic) and whose
    // lane offset is computed once; residual / ReLU chosen once -- 2-3 VALU instructions per output element
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        static_cast<T*>(p.y) + (size_t)m_base * p.y_cs + n_base, 0, (int)kOob, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(static_cast<const T*>(p.res)) + (size_t)m_base * p.res_cs + n_base, 0, (int)kOob, 0x00020000);
    const unsigned y_lane = (unsigned)((4 * h * p.y_cs + l31) * ES);
    const unsigned r_lane = (unsigned)((4 * h * p.res_cs + l31) * ES);
    auto run = [&](auto has_res, auto do_relu) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const float sc = p.scale ? p.scale[n_base + ni * 32 + l31] : 1.f;
        const float sh = p.shift ? p.shift[n_base + ni * 32 + l31] : 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          float rv[16];
          if constexpr (decltype(has_res)::value) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const unsigned so = (unsigned)(((mi * 32 + (r & 3) + 8 * (r >> 2)) * p.res_cs + ni * 32) * ES);
              if constexpr (kF32) {
                rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, r_lane, so, 0));
              } else {
                const unsigned short u = __builtin_amdgcn_raw_buffer_load_b16(rr, r_lane, so, 0);
                rv[r] = __builtin_bit_cast(float, (unsigned)u << 16);
              }
            }
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float t = fmaf(acc[mi][ni][r], sc, sh);
            if constexpr (decltype(has_res)::value) t += rv[r];
            if constexpr (decltype(do_relu)::value) t = fmaxf(t, 0.f);
            const unsigned so = (unsigned)(((mi * 32 + (r & 3) + 8 * (r >> 2)) * p.y_cs + ni * 32) * ES);
            if constexpr (kF32)
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, t), ry, y_lane, so, 0);
            else
              __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (__bf16)t), ry, y_lane, so, 0);
          }
        }
      }
    };
    using TT = std::true_type;
    using FF = std::false_type;
    if constexpr (!kF32) {
      // bf16 output without a residual: the accumulator layout puts ONE channel of 16 pixels in a lane -- 2-byte stores, 16 per 32x32
      // tile, 64-byte runs (the access SHAPE that bounded conv3x3_bf16's short-K layers and 40 % of the bf16 stem).  Through a wave-local
      // LDS transpose (scratch: the operand tiles are dead after the K loop's last barrier) a lane stores 16 bytes = 8 channels of a pixel
      // and a wave row is WN * 2 contiguous bytes (a whole 128-byte line at WN = 64).  Same values, same rounding.
      if (scratch && !p.res && (p.y_cs * ES) % 16 == 0 && (reinterpret_cast<uintptr_t>(p.y) & 15u) == 0) {
        constexpr int PITCH = NI * 64 + 16;                           // bytes per pixel row of the wave's tile (pad: the two lane halves on different banks)
        constexpr int PPP = NI * 4;                                   // 16-byte pieces per pixel
        char* const ws = scratch + (wm * (BN / WN) + wn) * (32 * PITCH);
        float scv[NI], shv[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          scv[ni] = p.scale ? p.scale[n_base + ni * 32 + l31] : 1.f;
          shv[ni] = p.shift ? p.shift[n_base + ni * 32 + l31] : 0.f;
        }
        const bool relu = p.relu != 0;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float t = fmaf(acc[mi][ni][r], scv[ni], shv[ni]);
              t = relu ? fmaxf(t, 0.f) : t;
              *reinterpret_cast<__bf16*>(ws + ((r & 3) + 8 * (r >> 2) + 4 * h) * PITCH + (ni * 32 + l31) * 2) = (__bf16)t;
            }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int k = 0; k < NI * 2; ++k) {
            const int idx = k * 64 + lane, px = idx / PPP, piece = idx - px * PPP;
            const u32x4 q = *reinterpret_cast<const u32x4*>(ws + px * PITCH + piece * 16);
            __builtin_amdgcn_raw_buffer_store_b128(q, ry, (unsigned)(((mi * 32 + px) * p.y_cs) * ES + piece * 16), 0, 0);
          }
          asm volatile("" ::: "memory");
        }
        return;
      }
    }
    if (p.res) { if (p.relu) run(TT{}, TT{}); else run(TT{}, FF{}); }
    else       { if (p.relu) run(FF{}, TT{}); else run(FF{}, FF{}); }
    return;
  }
  bool cm_fast = false;
  int cm_group = 0;
  if (p.colmax) {
    cm_group = m0 / p.rows_per_group;
    cm_fast = (m0 + BM <= m_hi) && ((m0 + BM - 1) / p.rows_per_group == cm_group);
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n_base + ni * 32 + l31;
    const bool nok = n < p.Cout;
    const int nc = nok ? n : p.Cout - 1;
    const float sc = p.scale ? p.scale[nc] : 1.f;
    const float sh = p.shift ? p.shift[nc] : 0.f;
    float vmax = 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int mrow = m_base + mi * 32 + 4 * h;
      float rv[16];
      if (p.res) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int m = mrow + (r & 3) + 8 * (r >> 2);
          m = m < m_hi ? m : m_hi - 1;
          rv[r] = (float)static_cast<const T*>(p.res)[(size_t)m * p.res_cs + nc];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = 0.f;
      }
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float t = fmaf(acc[mi][ni][r], sc, sh) + rv[r];
        v[r] = p.relu ? fmaxf(t, 0.f) : t;
      }
      if (p.y) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mrow + (r & 3) + 8 * (r >> 2);
          if (m < m_hi && nok) static_cast<T*>(p.y)[(size_t)m * p.y_cs + n] = (T)v[r];
        }
      }
      if (p.colmax) {
        if (cm_fast) {
#pragma unroll
          for (int r = 0; r < 16; ++r) vmax = fmaxf(vmax, v[r]);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = mrow + (r & 3) + 8 * (r >> 2);
            if (m < m_hi && nok)
              atomicMax(&p.colmax[(size_t)(m / p.rows_per_group) * p.Cout + n], __float_as_uint(fmaxf(v[r], 0.f)));
          }
        }
      }
    }
    if (p.colmax && cm_fast) {
      vmax = fmaxf(vmax, __shfl_xor(vmax, 32));
      if (h == 0 && nok) atomicMax(&p.colmax[(size_t)cm_group * p.Cout + n], __float_as_uint(fmaxf(vmax, 0.f)));
    }
  }
}

}  // namespace
