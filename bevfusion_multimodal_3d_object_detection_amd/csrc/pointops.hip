// Point-cloud side kernels: PointNet first layer (K = 4|5), fused radar MLP + max, and the
// weight-streaming dense layers (lidar_init, radar_proj, fusion_fc).
#include "common.h"

namespace {

// ---- y[m][n] = act((sum_k x[m][k] w[n][k]) * scale[n] + shift[n]),  K <= 16 -------------------
// PointNet conv1 + bn1 + relu (ref src/encoders.py:289).  Thread = (row, 4 channels).
template <typename TO>
__global__ __launch_bounds__(256) void pointwise_smallk(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift, TO* __restrict__ y,
                                                         int M, int K, int Cout, int relu) {
  extern __shared__ float wl[];  // [Cout][K]
  for (int i = threadIdx.x; i < Cout * K; i += 256) wl[i] = w[i];
  __syncthreads();
  const int c4 = Cout >> 2;
  const long long total = (long long)M * c4;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int m = (int)(i / c4), n = (int)(i % c4) * 4;
    float xv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) xv[k] = k < K ? x[(size_t)m * K + k] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float acc = 0.f;
      for (int k = 0; k < K; ++k) acc = fmaf(xv[k], wl[(n + j) * K + k], acc);
      float v = fmaf(acc, scale ? scale[n + j] : 1.f, shift ? shift[n + j] : 0.f);
      y[(size_t)m * Cout + n + j] = (TO)((relu && !(v > 0.f)) ? 0.f : v);
    }
  }
}

// ---- radar: 4 x (pointwise linear + BN + ReLU) + max over points --------------------------------------------
// ref src/encoders.py:549-555.  One workgroup per (radar, batch element, chunk of 32 points); activations
// ping-pong between two LDS buffers.  Thread = (output channel, point lane): the k loop is outermost so a
// weight is read from global memory once per chunk (coalesced over channels) and reused from a register for the
// thread's points, whose inputs are LDS broadcasts.  The last layer's per-chunk maximum goes out with an
// integer atomicMax on the non-negative post-ReLU bit pattern (out zero-filled by the caller): deterministic.
struct RadarArgs {
  const float* x;
  const float* w[4];      // k-major: [c_{i-1}][c_i]
  const float* scale[4];
  const float* shift[4];
  float* out;
  int R, B, P, Cin, nchunks;
  int c[4];
};
constexpr int RCH = 32;   // points per chunk

__global__ __launch_bounds__(256) void radar_mlp_max(const RadarArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int cmax_a = a.c[0] > a.c[2] ? a.c[0] : a.c[2];
  float* bufA = sm;                                               // [RCH][max(Cin, c0, c2)]
  float* bufB = sm + RCH * (cmax_a > a.Cin ? cmax_a : a.Cin);     // [RCH][max(Cin, c1)]
  const int chunk = blockIdx.x % a.nchunks, rb = blockIdx.x / a.nchunks;
  const int r = rb / a.B, b = rb % a.B;
  const int p0 = chunk * RCH, np = a.P - p0 < RCH ? a.P - p0 : RCH;
  const float* xp = a.x + (((size_t)r * a.B + b) * a.P + p0) * a.Cin;
  const int tid = threadIdx.x;
  for (int i = tid; i < np * a.Cin; i += 256) bufB[i] = xp[i];
  __syncthreads();
  const float* in = bufB;
  float* outb = bufA;
  int cin = a.Cin;
  for (int l = 0; l < 4; ++l) {
    const int co = a.c[l];
    for (int ch = tid % (co < 256 ? co : 256); ch < co; ch += 256) {     // co > 256: a thread takes several channels
      const int lanes = co < 256 ? 256 / co : 1, pl = co < 256 ? tid / co : 0;
      if (pl >= lanes) break;
      float acc[RCH];
#pragma unroll
      for (int q = 0; q < RCH; ++q) acc[q] = 0.f;
      // eight k at a time: the eight weight loads are independent (one k per iteration left every iteration waiting out an L2 round
      // trip: 272 us for 55 MFLOP at B = 8) and a point's eight inputs are two 16-byte LDS reads; each accumulator still sees its
      // products in ascending k -> the same bits as the one-k loop
      int k = 0;
      if ((cin & 3) == 0) {
        for (; k + 8 <= cin; k += 8) {
          float wv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) wv[j] = a.w[l][(size_t)(k + j) * co + ch];
#pragma unroll
          for (int q = 0; q < RCH; ++q) {
            const int pt = pl + q * lanes;
            if (q * lanes < RCH && pt < np) {
              const f32x4 i0 = *reinterpret_cast<const f32x4*>(in + pt * cin + k), i1 = *reinterpret_cast<const f32x4*>(in + pt * cin + k + 4);
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[q] = fmaf(i0[j], wv[j], acc[q]);
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[q] = fmaf(i1[j], wv[4 + j], acc[q]);
            }
          }
        }
      }
      for (; k < cin; ++k) {
        const float wv = a.w[l][(size_t)k * co + ch];
#pragma unroll
        for (int q = 0; q < RCH; ++q) {
          const int pt = pl + q * lanes;
          if (q * lanes < RCH && pt < np) acc[q] = fmaf(in[pt * cin + k], wv, acc[q]);
        }
      }
      const float sc = a.scale[l][ch], sh = a.shift[l][ch];
      if (l < 3) {
#pragma unroll
        for (int q = 0; q < RCH; ++q) {
          const int pt = pl + q * lanes;
          if (q * lanes < RCH && pt < np) outb[pt * co + ch] = fmaxf(fmaf(acc[q], sc, sh), 0.f);
        }
      } else {
        float m = 0.f;                                              // max(relu(v)) == max(0, v)
#pragma unroll
        for (int q = 0; q < RCH; ++q) {
          const int pt = pl + q * lanes;
          if (q * lanes < RCH && pt < np) m = fmaxf(m, fmaf(acc[q], sc, sh));
        }
        atomicMax(reinterpret_cast<unsigned*>(a.out) + ((size_t)b * a.R + r) * co + ch, __float_as_uint(m));
      }
    }
    __syncthreads();
    in = outb;
    outb = (outb == bufA) ? bufB : bufA;
    cin = co;
  }
}

// ---- dense layer, small batch: a wave owns R consecutive output rows at a time, weights streamed once -----------------
// The loads of the R rows are issued together (R x 16 B per lane in flight: one row at a time left the wave waiting out a full
// HBM round trip per 2 KB of weights); every row keeps its own accumulators, lane partition, FMA order and shuffle tree, so the
// values do not depend on R.
template <int NB, int R, typename TW, typename TY>      // x is always fp32 (a small B x K matrix); weights / outputs fp32 or bf16
__global__ __launch_bounds__(256) void linear_gemv(const float* __restrict__ x, const TW* __restrict__ w,
                                                    const float* __restrict__ bias, TY* __restrict__ y, int K,
                                                    int O, int relu, int perm_inner, int perm_outer) {
  constexpr int V = vec16<TW>::N;
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const int kv = K / V;
  for (int o0 = wave_global * R; o0 < O; o0 += nwaves * R) {
    float acc[R][NB];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[r][b] = 0.f;
    for (int i = lane; i < kv; i += 64) {
      float wv[R][V];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int o = o0 + r < O ? o0 + r : O - 1;                  // past the end: re-read the last row, never stored
        load16(w + (size_t)o * K + i * V, wv[r]);
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float* xb = x + (size_t)b * K + i * V;
        float xv[V];
#pragma unroll
        for (int j = 0; j < V; ++j) xv[j] = xb[j];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int j = 0; j < V; ++j) acc[r][b] = fmaf(wv[r][j], xv[j], acc[r][b]);
      }
    }
    // Per row: the NB accumulators summed over the 64 lanes with the xor-butterfly 32, 16, .., 1 -- TRANSPOSING while more than one
    // value is left: at mask m a lane keeps the half of its values its bit m selects and adds the partner's copy of that half
    // (NB - 1 + 6 - log2 NB shuffles a row instead of 6 NB; the same pairs are added at every level, so the same bits).
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int shift = 6;
#pragma unroll
      for (int m = 32, cur = NB; m >= 1; m >>= 1) {
        if (R > 1 && cur > 1) {                     // one row per wave (small layers: latency-bound) keeps the plain butterfly
          const bool up = (lane & m) != 0;
#pragma unroll
          for (int i = 0; i < cur / 2; ++i) {
            const float send = up ? acc[r][i] : acc[r][i + cur / 2];
            const float keep = up ? acc[r][i + cur / 2] : acc[r][i];
            acc[r][i] = keep + __shfl_xor(send, m);
          }
          cur >>= 1;
          --shift;
        } else {
          acc[r][0] += __shfl_xor(acc[r][0], m);
        }
      }
      const int o = o0 + r;
      if (R > 1) {
        if ((lane & ((1 << shift) - 1)) == 0 && o < O) {          // one writer per batch row b
          const int b = (lane >> shift) & (NB - 1);
          const int oo = perm_inner > 0 ? (o % perm_inner) * perm_outer + o / perm_inner : o;
          const float v = acc[r][0] + (bias ? bias[o] : 0.f);
          y[(size_t)b * O + oo] = (TY)((relu && !(v > 0.f)) ? 0.f : v);
        }
      } else {
#pragma unroll
        for (int b = 1; b < NB; ++b) {                             // plain butterfly for the remaining accumulators
#pragma unroll
          for (int m = 32; m >= 1; m >>= 1) acc[r][b] += __shfl_xor(acc[r][b], m);
        }
        if (lane == 0 && o < O) {
          const int oo = perm_inner > 0 ? (o % perm_inner) * perm_outer + o / perm_inner : o;
          const float bv = bias ? bias[o] : 0.f;
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const float v = acc[r][b] + bv;
            y[(size_t)b * O + oo] = (TY)((relu && !(v > 0.f)) ? 0.f : v);
          }
        }
      }
    }
  }
}

// ---- y[g][c] = max_p x[g][p][c]   (VFE "max over the points of a voxel", ref src/encoders.py:452) ----
__global__ __launch_bounds__(256) void group_max(const float* __restrict__ x, float* __restrict__ y, int P, int C,
                                                  long long total) {
  const int c4 = C >> 2;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long g = i / c4;
    const int c = (int)(i - g * c4) * 4;
    const float* src = x + (size_t)g * P * C + c;
    f32x4 m = *reinterpret_cast<const f32x4*>(src);
    for (int p = 1; p < P; ++p) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)p * C);
      m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    }
    *reinterpret_cast<f32x4*>(y + (size_t)g * C + c) = m;
  }
}

// ---- VFELayer in one pass (K <= 16): y[g][n] = max_p relu((sum_k x[g][p][k] w[n][k]) * scale[n] + shift[n]) -------------------
// ref src/encoders.py:431-455 (Linear -> BN1d -> ReLU -> max over the P points of a voxel; zero-padded points take part, as there).
// A wave owns a voxel, a lane a channel (its K weights and scale / shift live in registers); the voxel's P x K inputs are wave-wide
// broadcast loads.  Same fma chain, same relu and the same fmaxf as pointwise_smallk + group_max -> the same bits, without the
// [G][P][Cout] intermediate (164 MB written and read back at B = 8, 2500 pillars x 32 points x 64 channels).
template <int K>
__global__ __launch_bounds__(256) void vfe_smallk_max(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, float* __restrict__ y, int G, int P, int Cout) {
  constexpr int PC = 64 / K;                                      // points per coalesced 64-lane load
  const int lane = threadIdx.x & 63;
  const long long wave = (blockIdx.x * 256ll + threadIdx.x) >> 6, nwaves = (long long)gridDim.x * 4;
  for (int c0 = 0; c0 < Cout; c0 += 64) {
    const int n = c0 + lane;
    const bool live = n < Cout;
    float wv[K];
#pragma unroll
    for (int k = 0; k < K; ++k) wv[k] = live ? w[(size_t)n * K + k] : 0.f;
    const float sc = (live && scale) ? scale[n] : 1.f, sh = (live && shift) ? shift[n] : 0.f;
    for (long long g = wave; g < G; g += nwaves) {
      const float* src = x + (size_t)g * P * K;
      float m = 0.f;
      for (int p0 = 0; p0 < P; p0 += PC) {                        // lane l holds element l of this run of points; a point's inputs
        const int np = P - p0 < PC ? P - p0 : PC;                 // reach every lane as scalar operands (v_readlane, uniform index)
        const int xv = lane < np * K ? __float_as_int(src[p0 * K + lane]) : 0;
        for (int pp = 0; pp < np; ++pp) {
          float acc = 0.f;
#pragma unroll
          for (int k = 0; k < K; ++k) acc = fmaf(__int_as_float(__builtin_amdgcn_readlane(xv, pp * K + k)), wv[k], acc);
          float v = fmaf(acc, sc, sh);
          v = !(v > 0.f) ? 0.f : v;
          m = (p0 + pp) == 0 ? v : fmaxf(m, v);
        }
      }
      if (live) y[(size_t)g * Cout + n] = m;
    }
  }
}

}  // namespace

extern "C" int bevf_vfe_smallk_max_f32(const float* x, const float* w, const float* scale, const float* shift, float* y, int G, int P,
                                       int K, int Cout, void* stream) {
  BEVF_REQUIRE(x && w && y, "vfe: null pointer");
  BEVF_REQUIRE(G > 0 && P > 0 && K > 0 && K <= 16 && Cout > 0, "vfe: need G, P, Cout > 0 and 0 < K <= 16 (K=%d)", K);
  const long long blocks = ((long long)G + 3) / 4;
  const dim3 grid((unsigned)(blocks > 8192 ? 8192 : blocks));
  hipStream_t st = static_cast<hipStream_t>(stream);
#define BEVF_VFE(Kv) case Kv: hipLaunchKernelGGL(vfe_smallk_max<Kv>, grid, dim3(256), 0, st, x, w, scale, shift, y, G, P, Cout); break;
  switch (K) {
    BEVF_VFE(1) BEVF_VFE(2) BEVF_VFE(3) BEVF_VFE(4) BEVF_VFE(5) BEVF_VFE(6) BEVF_VFE(7) BEVF_VFE(8)
    BEVF_VFE(9) BEVF_VFE(10) BEVF_VFE(11) BEVF_VFE(12) BEVF_VFE(13) BEVF_VFE(14) BEVF_VFE(15) BEVF_VFE(16)
  }
#undef BEVF_VFE
  return bevf_check_launch("bevf_vfe_smallk_max_f32");
}

extern "C" int bevf_group_max_f32(const float* x, float* y, int G, int P, int C, void* stream) {
  BEVF_REQUIRE(x && y && G > 0 && P > 0 && C > 0 && C % 4 == 0, "group_max: bad arguments (C=%d)", C);
  BEVF_REQUIRE(bevf_aligned16(x) && bevf_aligned16(y), "group_max: unaligned");
  const long long total = (long long)G * (C / 4);
  const unsigned grid = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(group_max, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, P, C, total);
  return bevf_check_launch("bevf_group_max_f32");
}

template <typename TO>
static int smallk_entry(const float* x, const float* w, const float* scale, const float* shift, void* y, int M, int K,
                        int Cout, int relu, void* stream) {
  BEVF_REQUIRE(x && w && y, "pointwise: null pointer");
  BEVF_REQUIRE(M > 0 && K > 0 && K <= 16 && Cout > 0 && Cout % 4 == 0, "pointwise: need 0<K<=16, Cout%%4==0 (K=%d Cout=%d)", K, Cout);
  const long long total = (long long)M * (Cout / 4);
  const unsigned grid = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(pointwise_smallk<TO>, dim3(grid), dim3(256), (size_t)Cout * K * sizeof(float),
                     static_cast<hipStream_t>(stream), x, w, scale, shift, static_cast<TO*>(y), M, K, Cout, relu);
  return bevf_check_launch("bevf_pointwise_smallk");
}
extern "C" int bevf_pointwise_smallk_f32(const float* x, const float* w, const float* scale, const float* shift,
                                         float* y, int M, int K, int Cout, int relu, void* stream) {
  return smallk_entry<float>(x, w, scale, shift, y, M, K, Cout, relu, stream);
}
extern "C" int bevf_pointwise_smallk_bf16out(const float* x, const float* w, const float* scale, const float* shift,
                                             void* y, int M, int K, int Cout, int relu, void* stream) {
  return smallk_entry<__bf16>(x, w, scale, shift, y, M, K, Cout, relu, stream);
}

extern "C" int bevf_radar_mlp_max_f32(const bevf_radar_desc* d, void* stream) {
  BEVF_REQUIRE(d && d->x && d->out, "radar: null pointer");
  BEVF_REQUIRE(d->R > 0 && d->B > 0 && d->P > 0 && d->Cin > 0, "radar: empty shape");
  RadarArgs a;
  a.x = d->x; a.out = d->out; a.R = d->R; a.B = d->B; a.P = d->P; a.Cin = d->Cin;
  int cmax_a = d->Cin;
  for (int i = 0; i < 4; ++i) {
    BEVF_REQUIRE(d->w[i] && d->scale[i] && d->shift[i] && d->c[i] > 0, "radar: layer %d incomplete", i);
    a.w[i] = d->w[i]; a.scale[i] = d->scale[i]; a.shift[i] = d->shift[i]; a.c[i] = d->c[i];
  }
  BEVF_REQUIRE(d->c[3] <= 1024, "radar: last width %d > 1024", d->c[3]);
  if (d->c[0] > cmax_a) cmax_a = d->c[0];
  if (d->c[2] > cmax_a) cmax_a = d->c[2];
  const int cmax_b = d->c[1] > d->Cin ? d->c[1] : d->Cin;
  const size_t lds = (size_t)RCH * (cmax_a + cmax_b) * sizeof(float);
  BEVF_REQUIRE(lds <= 64 * 1024, "radar: layer widths need %zu B of LDS", lds);
  a.nchunks = (d->P + RCH - 1) / RCH;
  hipLaunchKernelGGL(radar_mlp_max, dim3(d->R * d->B * a.nchunks), dim3(256), lds, static_cast<hipStream_t>(stream), a);
  return bevf_check_launch("bevf_radar_mlp_max_f32");
}

template <typename TW, typename TY>
static int linear_entry(const float* x, const void* w, const float* bias, void* y, int B, int K, int O, int relu,
                        int perm_inner, int perm_outer, void* stream) {
  BEVF_REQUIRE(x && w && y, "linear: null pointer");
  BEVF_REQUIRE(B > 0 && K > 0 && O > 0 && K % vec16<TW>::N == 0, "linear: K=%d must be a positive multiple of %d", K, vec16<TW>::N);
  BEVF_REQUIRE(bevf_aligned16(x) && bevf_aligned16(w), "linear: x/w unaligned");
  BEVF_REQUIRE(perm_inner <= 0 || (long long)perm_inner * perm_outer == O, "linear: perm_inner*perm_outer != O");
  hipStream_t st = static_cast<hipStream_t>(stream);
  // R rows in flight per wave (4 at NB = 8, else 8) when that still leaves every CU several workgroups; small layers keep one
  // row per wave (more waves, shorter chains).  One step per wave where that fits: workgroups are handed out as others
  // finish, which balances better than a fixed stride over a ragged number of steps.
  const bool wide = O >= 16384;
  auto grid_for = [&](int R) { const int g = (O + 4 * R - 1) / (4 * R); return dim3((unsigned)(g > 16384 ? 16384 : g)); };
#define BEVF_GEMV(NBv, Rv) \
  hipLaunchKernelGGL((linear_gemv<NBv, Rv, TW, TY>), grid_for(Rv), dim3(256), 0, st, xb, wt, bias, yb, K, O, relu, perm_inner, perm_outer)
  const TW* wt = static_cast<const TW*>(w);
  for (int b0 = 0; b0 < B;) {
    const int rem = B - b0;
    const float* xb = x + (size_t)b0 * K;
    TY* yb = static_cast<TY*>(y) + (size_t)b0 * O;
    if (rem >= 8) {
      if (wide) BEVF_GEMV(8, 4); else BEVF_GEMV(8, 1);
      b0 += 8;
    } else if (rem >= 4) {
      if (wide) BEVF_GEMV(4, 8); else BEVF_GEMV(4, 1);
      b0 += 4;
    } else if (rem >= 2) {
      if (wide) BEVF_GEMV(2, 8); else BEVF_GEMV(2, 1);
      b0 += 2;
    } else {
      if (wide) BEVF_GEMV(1, 8); else BEVF_GEMV(1, 1);
      b0 += 1;
    }
  }
#undef BEVF_GEMV
  return bevf_check_launch("bevf_linear");
}
extern "C" int bevf_linear_f32(const float* x, const float* w, const float* bias, float* y, int B, int K, int O,
                               int relu, int perm_inner, int perm_outer, void* stream) {
  return linear_entry<float, float>(x, w, bias, y, B, K, O, relu, perm_inner, perm_outer, stream);
}
// bf16 weights (half the bytes of the weight stream), fp32 activations in; output fp32 (y_bf16 == 0) or bf16
extern "C" int bevf_linear_bf16w(const float* x, const void* w, const float* bias, void* y, int y_bf16, int B, int K,
                                 int O, int relu, int perm_inner, int perm_outer, void* stream) {
  return y_bf16 ? linear_entry<__bf16, __bf16>(x, w, bias, y, B, K, O, relu, perm_inner, perm_outer, stream)
                : linear_entry<__bf16, float>(x, w, bias, y, B, K, O, relu, perm_inner, perm_outer, stream);
}
