// HBM-bound glue of the BEV path: camera mean ("BEV pooling" part 1), bilinear resample
// (part 2, also nn.Upsample), radar broadcast, CenterNet head tail, layout changes, fill.
// All streaming kernels move 16 B per lane, channel-contiguous (NHWC).
#include "common.h"

namespace {

constexpr unsigned MAX_GRID = 256 * 8 * 4;

static inline unsigned stream_grid(long long work_items) {
  long long g = (work_items + 255) / 256;
  return (unsigned)(g > MAX_GRID ? MAX_GRID : (g < 1 ? 1 : g));
}

// y[b][p][c] = (x[b][0][p][c] + ... + x[b][n-1][p][c]) / n      ref src/fusion.py:233-234
template <typename T>
__global__ __launch_bounds__(256) void cam_mean(const T* __restrict__ x, T* __restrict__ y, int ncam, long long pcv,
                                                 long long total) {
  constexpr int V = vec16<T>::N;
  const float div = (float)ncam;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long b = i / pcv, r = i - b * pcv;
    const T* src = x + (b * ncam * pcv + r) * V;
    float s[V], v[V];
    load16(src, s);
    for (int n = 1; n < ncam; ++n) {
      load16(src + (long long)n * pcv * V, v);
#pragma unroll
      for (int j = 0; j < V; ++j) s[j] += v[j];
    }
#pragma unroll
    for (int j = 0; j < V; ++j) s[j] /= div;
    store16(y + i * V, s);
  }
}

// torch upsample_bilinear2d, align_corners=False: src = scale*(dst+0.5)-0.5 clamped at 0,
// i0 = floor, i1 = i0 + (i0 < in-1), lambda = src - i0.     ref src/fusion.py:242-247, :156
__device__ __forceinline__ void lin_coord(int o, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
  float s = scale * ((float)o + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = fminf(fmaxf(s - (float)i0, 0.f), 1.f);
  l0 = 1.f - l1;
}

template <typename T>
__global__ __launch_bounds__(256) void bilinear_nhwc(const T* __restrict__ x, T* __restrict__ y, int Hi, int Wi, int C,
                                                      int x_cs, int Ho, int Wo, int y_cs, float sh, float sw,
                                                      long long total) {
  constexpr int V = vec16<T>::N;
  const int cv = C / V;
  const bool small = total < (1ll << 31);                        // 32-bit index arithmetic (three 64-bit divisions were most of the kernel's VALU)
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    int c, ow, oh, b;
    long long pix;
    if (small) {
      const unsigned u = (unsigned)i, up = u / (unsigned)cv, ut = up / (unsigned)Wo;
      c = (int)(u - up * (unsigned)cv) * V;
      ow = (int)(up - ut * (unsigned)Wo);
      b = (int)(ut / (unsigned)Ho);
      oh = (int)(ut - (unsigned)b * (unsigned)Ho);
      pix = up;
    } else {
      c = (int)(i % cv) * V;
      pix = i / cv;
      ow = (int)(pix % Wo);
      const long long t = pix / Wo;
      oh = (int)(t % Ho);
      b = (int)(t / Ho);
    }
    int h0, h1, w0, w1;
    float lh0, lh1, lw0, lw1;
    lin_coord(oh, sh, Hi, h0, h1, lh0, lh1);
    lin_coord(ow, sw, Wi, w0, w1, lw0, lw1);
    const T* base = x + (size_t)b * Hi * Wi * x_cs + c;
    float p00[V], p01[V], p10[V], p11[V], o[V];
    load16(base + ((size_t)h0 * Wi + w0) * x_cs, p00);
    load16(base + ((size_t)h0 * Wi + w1) * x_cs, p01);
    load16(base + ((size_t)h1 * Wi + w0) * x_cs, p10);
    load16(base + ((size_t)h1 * Wi + w1) * x_cs, p11);
#pragma unroll
    for (int j = 0; j < V; ++j)
      o[j] = lh0 * (lw0 * p00[j] + lw1 * p01[j]) + lh1 * (lw0 * p10[j] + lw1 * p11[j]);
    store16(y + (size_t)pix * y_cs + c, o);
  }
}

template <typename T>     // v is always fp32 (a B x C vector)
__global__ __launch_bounds__(256) void broadcast_nhwc(const float* __restrict__ v, T* __restrict__ y, int P, int C,
                                                       int y_cs, long long total) {
  constexpr int V = vec16<T>::N;
  const int cv = C / V;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % cv) * V;
    const long long pix = i / cv;
    const int b = (int)(pix / P);
    float f[V];
#pragma unroll
    for (int j = 0; j < V; ++j) f[j] = v[(size_t)b * C + c + j];
    store16(y + (size_t)pix * y_cs + c, f);
  }
}

// A 3x3/pad-1 conv stack applied to a spatially CONSTANT image (the radar branch, ref src/fusion.py:277-281)
// produces at most 5x5 distinct pixel values per channel after two layers: a pixel's value depends only on
// which of its taps fall outside the image, i.e. on its border class {0, 1, interior, S-2, S-1} per axis.
// The convs therefore run on a 5x5 image (identical FMA chains, bit-identical values) and this kernel
// expands the 5x5 classes to the S_h x S_w map, written straight into the concat slice.
__device__ __forceinline__ int border_class(int i, int S) { return i < 2 ? i : (i >= S - 2 ? 4 - (S - 1 - i) : 2); }

template <typename T>
__global__ __launch_bounds__(256) void expand_border_classes(const T* __restrict__ small, T* __restrict__ y, int Sh,
                                                              int Sw, int C, int y_cs, long long total) {
  constexpr int V = vec16<T>::N;
  const int cv = C / V;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % cv) * V;
    const long long pix = i / cv;
    const int w = (int)(pix % Sw);
    const long long t = pix / Sw;
    const int hh = (int)(t % Sh), b = (int)(t / Sh);
    const f32x4 v = *reinterpret_cast<const f32x4*>(
        small + ((size_t)(b * 5 + border_class(hh, Sh)) * 5 + border_class(w, Sw)) * C + c);
    *reinterpret_cast<f32x4*>(y + (size_t)pix * y_cs + c) = v;      // 16 raw bytes either way
  }
}

// CenterNet head tail (ref src/fusion.py:869-884): per pixel, five 1x1 convs on the five
// hc-wide slices of the hidden map, sigmoid on the first n_sigmoid outputs, NCHW stores.
struct HeadArgs {
  const void* hid;
  const float* w;
  const float* bias;
  float* out[5];
  int B, P, hc;
  int c[5];
  int n_sigmoid;
};

template <typename T>
__global__ __launch_bounds__(256) void head_tail(const HeadArgs a) {
  constexpr int V = vec16<T>::N;
  extern __shared__ float wl[];   // [ctot][hc] then bias[ctot]
  const int ctot = a.c[0] + a.c[1] + a.c[2] + a.c[3] + a.c[4];
  for (int i = threadIdx.x; i < ctot * a.hc; i += 256) wl[i] = a.w[i];
  for (int i = threadIdx.x; i < ctot; i += 256) wl[ctot * a.hc + i] = a.bias[i];
  __syncthreads();
  const float* bl = wl + ctot * a.hc;
  const long long total = (long long)a.B * a.P;
  for (long long pix = blockIdx.x * 256ll + threadIdx.x; pix < total; pix += (long long)gridDim.x * 256) {
    const int b = (int)(pix / a.P), p = (int)(pix - (long long)b * a.P);
    const T* hp = static_cast<const T*>(a.hid) + (size_t)pix * 5 * a.hc;
    int oc = 0;
    for (int k = 0; k < 5; ++k) {
      for (int c = 0; c < a.c[k]; ++c, ++oc) {
        float acc = 0.f;
        for (int j = 0; j < a.hc; j += V) {
          float hv[V];
          load16(hp + k * a.hc + j, hv);
          const float* wr = wl + oc * a.hc + j;
#pragma unroll
          for (int q = 0; q < V; ++q) acc = fmaf(hv[q], wr[q], acc);
        }
        float v = acc + bl[oc];
        if (oc < a.n_sigmoid) v = 1.f / (1.f + expf(-v));
        a.out[k][((size_t)b * a.c[k] + c) * a.P + p] = v;
      }
    }
  }
}

// Coalesced variant for hc/V a power of two <= 64 (hc = 64: the reference's head): the CH = hc/V lanes of a pixel read
// consecutive 16-byte chunks of one branch's hidden vector (a wave covers 64/CH pixels per pass, one branch per
// iteration), accumulate their slice of the branch's outputs and combine with a butterfly over the CH lanes.  The
// thread-per-pixel form above strides 1280 B between lanes and thrashes L1 (1 TB/s); this one streams.
template <typename T, int CH>
__global__ __launch_bounds__(256) void head_tail_coalesced(const HeadArgs a) {
  constexpr int V = vec16<T>::N, PPW = 64 / CH;
  extern __shared__ float wl[];   // [ctot][hc] then bias[ctot]
  const int ctot = a.c[0] + a.c[1] + a.c[2] + a.c[3] + a.c[4];
  for (int i = threadIdx.x; i < ctot * a.hc; i += 256) wl[i] = a.w[i];
  for (int i = threadIdx.x; i < ctot; i += 256) wl[ctot * a.hc + i] = a.bias[i];
  __syncthreads();
  const float* bl = wl + ctot * a.hc;
  const int lane = threadIdx.x & 63, sub = lane % CH, pw = lane / CH;
  const long long total = (long long)a.B * a.P;
  const long long wave_global = (blockIdx.x * 256ll + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * 256) >> 6;
  for (long long base = wave_global * PPW; base < total; base += nwaves * PPW) {
    const long long pix = base + pw;
    const bool ok = pix < total;
    const long long pc = ok ? pix : total - 1;
    const int b = (int)(pc / a.P), p = (int)(pc - (long long)b * a.P);
    const T* hp = static_cast<const T*>(a.hid) + (size_t)pc * 5 * a.hc + sub * V;
    int oc = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      float hv[V];
      load16(hp + k * a.hc, hv);
      for (int c = 0; c < a.c[k]; ++c, ++oc) {
        const float* wr = wl + oc * a.hc + sub * V;
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < V; ++q) acc = fmaf(hv[q], wr[q], acc);
#pragma unroll
        for (int s = 1; s < CH; s <<= 1) acc += __shfl_xor(acc, s);
        if (sub == 0 && ok) {
          float v = acc + bl[oc];
          if (oc < a.n_sigmoid) v = 1.f / (1.f + expf(-v));
          a.out[k][((size_t)b * a.c[k] + c) * a.P + p] = v;
        }
      }
    }
  }
}

// [N][C][P] -> [N][P][y_cs] and back, 32x32 tiles through LDS (both sides coalesced)
__global__ __launch_bounds__(256) void nchw_to_nhwc(const float* __restrict__ x, float* __restrict__ y, int C, int P,
                                                     int y_cs) {
  __shared__ float t[32][33];
  const int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, p = p0 + tx;
    t[r][tx] = (c < C && p < P) ? x[((size_t)n * C + c) * P + p] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int p = p0 + r, c = c0 + tx;
    if (p < P && c < C) y[((size_t)n * P + p) * y_cs + c] = t[tx][r];
  }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw(const float* __restrict__ x, float* __restrict__ y, int C, int P,
                                                     int x_cs) {
  __shared__ float t[32][33];
  const int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int p = p0 + r, c = c0 + tx;
    t[r][tx] = (c < C && p < P) ? x[((size_t)n * P + p) * x_cs + c] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, p = p0 + tx;
    if (p < P && c < C) y[((size_t)n * C + c) * P + p] = t[tx][r];
  }
}

__global__ __launch_bounds__(256) void fill_f32(float* __restrict__ y, float v, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = v;
}

}  // namespace

template <typename T>
static int cam_mean_entry(const void* x, void* y, int B, int ncam, int P, int C, void* stream) {
  constexpr int V = vec16<T>::N;
  BEVF_REQUIRE(x && y, "cam_mean: null pointer");
  BEVF_REQUIRE(B > 0 && ncam > 0 && P > 0 && C > 0 && C % V == 0, "cam_mean: C=%d must be a positive multiple of %d", C, V);
  BEVF_REQUIRE(bevf_aligned16(x) && bevf_aligned16(y), "cam_mean: unaligned");
  const long long pcv = (long long)P * C / V, total = pcv * B;
  hipLaunchKernelGGL(cam_mean<T>, dim3(stream_grid(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const T*>(x), static_cast<T*>(y), ncam, pcv, total);
  return bevf_check_launch("bevf_cam_mean");
}
extern "C" int bevf_cam_mean_f32(const float* x, float* y, int B, int ncam, int P, int C, void* stream) {
  return cam_mean_entry<float>(x, y, B, ncam, P, C, stream);
}
extern "C" int bevf_cam_mean_bf16(const void* x, void* y, int B, int ncam, int P, int C, void* stream) {
  return cam_mean_entry<__bf16>(x, y, B, ncam, P, C, stream);
}

template <typename T>
static int bilinear_entry(const void* x, void* y, int B, int Hi, int Wi, int C, int x_cs, int Ho, int Wo, int y_cs,
                          void* stream) {
  constexpr int V = vec16<T>::N;
  BEVF_REQUIRE(x && y, "bilinear: null pointer");
  BEVF_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && C % V == 0, "bilinear: bad shape (C=%d)", C);
  BEVF_REQUIRE(x_cs >= C && y_cs >= C && x_cs % V == 0 && y_cs % V == 0, "bilinear: channel strides must be >= C and 16-byte multiples");
  BEVF_REQUIRE(bevf_aligned16(x) && bevf_aligned16(y), "bilinear: unaligned");
  const long long total = (long long)B * Ho * Wo * (C / V);
  hipLaunchKernelGGL(bilinear_nhwc<T>, dim3(stream_grid(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const T*>(x), static_cast<T*>(y), Hi, Wi, C, x_cs, Ho, Wo, y_cs, (float)Hi / (float)Ho,
                     (float)Wi / (float)Wo, total);
  return bevf_check_launch("bevf_bilinear_nhwc");
}
extern "C" int bevf_bilinear_nhwc_f32(const float* x, float* y, int B, int Hi, int Wi, int C, int x_cs, int Ho,
                                      int Wo, int y_cs, void* stream) {
  return bilinear_entry<float>(x, y, B, Hi, Wi, C, x_cs, Ho, Wo, y_cs, stream);
}
extern "C" int bevf_bilinear_nhwc_bf16(const void* x, void* y, int B, int Hi, int Wi, int C, int x_cs, int Ho, int Wo,
                                       int y_cs, void* stream) {
  return bilinear_entry<__bf16>(x, y, B, Hi, Wi, C, x_cs, Ho, Wo, y_cs, stream);
}

template <typename T>
static int broadcast_entry(const float* v, void* y, int B, int P, int C, int y_cs, void* stream) {
  constexpr int V = vec16<T>::N;
  BEVF_REQUIRE(v && y, "broadcast: null pointer");
  BEVF_REQUIRE(B > 0 && P > 0 && C > 0 && C % V == 0 && y_cs >= C && y_cs % V == 0, "broadcast: bad shape");
  BEVF_REQUIRE(bevf_aligned16(y), "broadcast: unaligned");
  const long long total = (long long)B * P * (C / V);
  hipLaunchKernelGGL(broadcast_nhwc<T>, dim3(stream_grid(total)), dim3(256), 0, static_cast<hipStream_t>(stream), v,
                     static_cast<T*>(y), P, C, y_cs, total);
  return bevf_check_launch("bevf_broadcast_nhwc");
}
extern "C" int bevf_broadcast_nhwc_f32(const float* v, float* y, int B, int P, int C, int y_cs, void* stream) {
  return broadcast_entry<float>(v, y, B, P, C, y_cs, stream);
}
extern "C" int bevf_broadcast_nhwc_bf16(const float* v, void* y, int B, int P, int C, int y_cs, void* stream) {
  return broadcast_entry<__bf16>(v, y, B, P, C, y_cs, stream);
}

template <typename T>
static int expand_entry(const void* small, void* y, int B, int Sh, int Sw, int C, int y_cs, void* stream) {
  constexpr int V = vec16<T>::N;
  BEVF_REQUIRE(small && y, "expand: null pointer");
  BEVF_REQUIRE(B > 0 && Sh >= 5 && Sw >= 5 && C > 0 && C % V == 0 && y_cs >= C && y_cs % V == 0,
               "expand: needs S >= 5 and 16-byte channel groups (Sh=%d Sw=%d C=%d)", Sh, Sw, C);
  BEVF_REQUIRE(bevf_aligned16(small) && bevf_aligned16(y), "expand: unaligned");
  const long long total = (long long)B * Sh * Sw * (C / V);
  hipLaunchKernelGGL(expand_border_classes<T>, dim3(stream_grid(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const T*>(small), static_cast<T*>(y), Sh, Sw, C, y_cs, total);
  return bevf_check_launch("bevf_expand_border_classes");
}
extern "C" int bevf_expand_border_classes_f32(const float* small, float* y, int B, int Sh, int Sw, int C, int y_cs,
                                             void* stream) {
  return expand_entry<float>(small, y, B, Sh, Sw, C, y_cs, stream);
}
extern "C" int bevf_expand_border_classes_bf16(const void* small, void* y, int B, int Sh, int Sw, int C, int y_cs,
                                              void* stream) {
  return expand_entry<__bf16>(small, y, B, Sh, Sw, C, y_cs, stream);
}

template <typename T>
static int head_tail_entry(const bevf_head_desc* d, void* stream) {
  BEVF_REQUIRE(d && d->hid && d->w && d->bias, "head_tail: null pointer");
  BEVF_REQUIRE(d->B > 0 && d->P > 0 && d->hc > 0 && d->hc % vec16<T>::N == 0, "head_tail: hc=%d must be a positive multiple of %d", d->hc, vec16<T>::N);
  BEVF_REQUIRE(bevf_aligned16(d->hid), "head_tail: hid unaligned");
  HeadArgs a;
  a.hid = d->hid; a.w = d->w; a.bias = d->bias; a.B = d->B; a.P = d->P; a.hc = d->hc; a.n_sigmoid = d->n_sigmoid;
  int ctot = 0;
  for (int k = 0; k < 5; ++k) {
    BEVF_REQUIRE(d->c[k] >= 0 && (d->c[k] == 0 || d->out[k]), "head_tail: branch %d has no output buffer", k);
    a.out[k] = d->out[k]; a.c[k] = d->c[k]; ctot += d->c[k];
  }
  const size_t lds = (size_t)ctot * (d->hc + 1) * sizeof(float);
  BEVF_REQUIRE(lds <= 64 * 1024, "head_tail: weights need %zu B of LDS", lds);
  constexpr int V = vec16<T>::N;
  if (d->hc == 16 * V) {                                  // hc = 64 (fp32) / 128 (bf16): 16 lanes per pixel
    const long long waves = ((long long)d->B * d->P + 3) / 4;
    hipLaunchKernelGGL((head_tail_coalesced<T, 16>), dim3(stream_grid(waves * 64)), dim3(256), lds,
                       static_cast<hipStream_t>(stream), a);
  } else if (d->hc == 8 * V) {                            // hc = 64 in bf16: 8 lanes per pixel
    const long long waves = ((long long)d->B * d->P + 7) / 8;
    hipLaunchKernelGGL((head_tail_coalesced<T, 8>), dim3(stream_grid(waves * 64)), dim3(256), lds,
                       static_cast<hipStream_t>(stream), a);
  } else {
    hipLaunchKernelGGL(head_tail<T>, dim3(stream_grid((long long)d->B * d->P)), dim3(256), lds,
                       static_cast<hipStream_t>(stream), a);
  }
  return bevf_check_launch("bevf_head_tail");
}
extern "C" int bevf_head_tail_f32(const bevf_head_desc* d, void* stream) { return head_tail_entry<float>(d, stream); }
extern "C" int bevf_head_tail_bf16(const bevf_head_desc* d, void* stream) { return head_tail_entry<__bf16>(d, stream); }

extern "C" int bevf_nchw_to_nhwc_f32(const float* x, float* y, int N, int C, int P, int y_cs, void* stream) {
  BEVF_REQUIRE(x && y && N > 0 && C > 0 && P > 0 && y_cs >= C, "nchw_to_nhwc: bad arguments");
  BEVF_REQUIRE(N <= 65535 && (C + 31) / 32 <= 65535, "nchw_to_nhwc: grid too large");
  hipLaunchKernelGGL(nchw_to_nhwc, dim3((P + 31) / 32, (C + 31) / 32, N), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, y, C, P, y_cs);
  return bevf_check_launch("bevf_nchw_to_nhwc_f32");
}

extern "C" int bevf_nhwc_to_nchw_f32(const float* x, float* y, int N, int C, int P, int x_cs, void* stream) {
  BEVF_REQUIRE(x && y && N > 0 && C > 0 && P > 0 && x_cs >= C, "nhwc_to_nchw: bad arguments");
  BEVF_REQUIRE(N <= 65535 && (C + 31) / 32 <= 65535, "nhwc_to_nchw: grid too large");
  hipLaunchKernelGGL(nhwc_to_nchw, dim3((P + 31) / 32, (C + 31) / 32, N), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, y, C, P, x_cs);
  return bevf_check_launch("bevf_nhwc_to_nchw_f32");
}

extern "C" int bevf_fill_f32(float* y, float v, size_t n, void* stream) {
  BEVF_REQUIRE(y || n == 0, "fill: null pointer");
  if (n == 0) return BEVF_OK;
  hipLaunchKernelGGL(fill_f32, dim3(stream_grid((long long)n)), dim3(256), 0, static_cast<hipStream_t>(stream), y, v, n);
  return bevf_check_launch("bevf_fill_f32");
}
