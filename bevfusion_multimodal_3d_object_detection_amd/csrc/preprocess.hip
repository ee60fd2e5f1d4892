// Input pipeline on the device (ref src/train_detect.py:123-189).
//
// resize_normalize_u8: Pillow's antialiased bilinear resize (what torchvision's T.Resize does to a PIL image) in
//   Pillow's own 22-bit fixed-point arithmetic -- horizontal pass rounded and clamped to uint8, then the vertical
//   pass -- followed by ToTensor (/255) and Normalize ((t-mean)/std) in fp32, written planar (NCHW) for the stem.
//   The per-axis tables (first source index, count, integer weights) come from the host
//   (preprocess.resample_tables); the resized uint8 value is bit-identical to Pillow's.
// lidar_filter_pad: range filter with strict inequalities, order-preserving compaction (wave ballot + block scan),
//   zero padding / index gather to exactly max_points rows.
#include "common.h"

namespace {

constexpr int kPrec = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int v) {
  v >>= kPrec;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ __launch_bounds__(256) void resize_normalize_u8(const unsigned char* __restrict__ x, float* __restrict__ out,
                                                            int n, int H, int W, int Ho, int Wo,
                                                            const int* __restrict__ bh, const int* __restrict__ kh, int ksh,
                                                            const int* __restrict__ bv, const int* __restrict__ kv, int ksv,
                                                            float m0, float m1, float m2, float s0, float s1, float s2) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  const long long total = (long long)n * Ho * Wo;
  if (i >= total) return;
  const int ox = (int)(i % Wo);
  const int oy = (int)((i / Wo) % Ho);
  const int img = (int)(i / ((long long)Wo * Ho));
  const int x0 = bh[2 * ox], nx = bh[2 * ox + 1], y0 = bv[2 * oy], ny = bv[2 * oy + 1];
  const int* const kx = kh + (size_t)ox * ksh;
  const int* const ky = kv + (size_t)oy * ksv;
  const unsigned char* const base = x + (size_t)img * H * W * 3;
  int v0 = 1 << (kPrec - 1), v1 = v0, v2 = v0;
  for (int j = 0; j < ny; ++j) {
    const unsigned char* row = base + ((size_t)(y0 + j) * W + x0) * 3;
    int h0 = 1 << (kPrec - 1), h1 = h0, h2 = h0;
    for (int t = 0; t < nx; ++t) {
      const int k = kx[t];
      h0 += (int)row[3 * t] * k;
      h1 += (int)row[3 * t + 1] * k;
      h2 += (int)row[3 * t + 2] * k;
    }
    const int k = ky[j];                              // the horizontal pass leaves a uint8 image
    v0 += clip8(h0) * k;
    v1 += clip8(h1) * k;
    v2 += clip8(h2) * k;
  }
  const size_t plane = (size_t)Ho * Wo, o = (size_t)img * 3 * plane + (size_t)oy * Wo + ox;
  out[o] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8(v0), 255.f), m0), s0);
  out[o + plane] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8(v1), 255.f), m1), s1);
  out[o + 2 * plane] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8(v2), 255.f), m2), s2);
}

// ---- the same resize in two passes through LDS (round 3) ----------------------------------------------------------------------
// The one-thread-per-pixel form above redoes the horizontal pass for every output row that touches an input row (each input row
// feeds ~2.5 output rows at a 2x down-scale) with 3 byte loads per tap: 75 uncoalesced byte loads and 75 integer MACs per output
// pixel, 1.0 TB/s.  Here a workgroup owns RT output rows x 64 output columns: pass 1 runs Pillow's horizontal pass ONCE per
// needed input row into an LDS tile of uint8 (exactly the intermediate image Pillow itself makes), pass 2 the vertical pass and
// the normalisation from LDS.  Same integer arithmetic, same rounding, same clamp -> the same bits.
constexpr int RT = 16, CT_ = 64, RMAX = 48;                          // output rows / columns per workgroup, input rows the tile may need

__global__ __launch_bounds__(256) void resize_normalize_u8_tiled(const unsigned char* __restrict__ x, float* __restrict__ out,
                                                                  int n, int H, int W, int Ho, int Wo,
                                                                  const int* __restrict__ bh, const int* __restrict__ kh, int ksh,
                                                                  const int* __restrict__ bv, const int* __restrict__ kv, int ksv,
                                                                  float m0, float m1, float m2, float s0, float s1, float s2, int tilesX,
                                                                  int tilesY) {
  __shared__ unsigned char hbuf[RMAX][CT_][4];                       // horizontal-pass result, 3 channels (+1 pad) per pixel
  const int tid = threadIdx.x;
  const int tx = blockIdx.x % tilesX, ty = (blockIdx.x / tilesX) % tilesY, img = blockIdx.x / (tilesX * tilesY);
  const int ox0 = tx * CT_, oy0 = ty * RT;
  const int oy1 = (oy0 + RT < Ho ? oy0 + RT : Ho) - 1;
  const int r0 = bv[2 * oy0], r1 = bv[2 * oy1] + bv[2 * oy1 + 1];  // input rows [r0, r1) feed this tile (bounds are monotonic)
  const int nr = r1 - r0;                                            // <= RMAX (checked on the host)
  const unsigned char* const base = x + (size_t)img * H * W * 3;
  {                                                                  // pass 1: thread = (output column, channel-free), rows strided
    const int c = tid & (CT_ - 1), rq = tid >> 6;                    // 4 row phases
    const int ox = ox0 + c;
    if (ox < Wo) {
      const int x0 = bh[2 * ox], nx = bh[2 * ox + 1];
      const int* const kx = kh + (size_t)ox * ksh;
      for (int r = rq; r < nr; r += 4) {
        const unsigned char* row = base + ((size_t)(r0 + r) * W + x0) * 3;
        int h0 = 1 << (kPrec - 1), h1 = h0, h2 = h0;
        for (int t = 0; t < nx; ++t) {
          const int k = kx[t];
          h0 += (int)row[3 * t] * k;
          h1 += (int)row[3 * t + 1] * k;
          h2 += (int)row[3 * t + 2] * k;
        }
        hbuf[r][c][0] = (unsigned char)clip8(h0);
        hbuf[r][c][1] = (unsigned char)clip8(h1);
        hbuf[r][c][2] = (unsigned char)clip8(h2);
      }
    }
  }
  __syncthreads();
  const size_t plane = (size_t)Ho * Wo;
  for (int e = tid; e < RT * CT_; e += 256) {                        // pass 2: thread = output pixel
    const int c = e & (CT_ - 1), ry = e >> 6;
    const int ox = ox0 + c, oy = oy0 + ry;
    if (ox >= Wo || oy >= Ho) continue;
    const int y0 = bv[2 * oy] - r0, ny = bv[2 * oy + 1];
    const int* const ky = kv + (size_t)oy * ksv;
    int v0 = 1 << (kPrec - 1), v1 = v0, v2 = v0;
    for (int j = 0; j < ny; ++j) {
      const int k = ky[j];
      v0 += (int)hbuf[y0 + j][c][0] * k;
      v1 += (int)hbuf[y0 + j][c][1] * k;
      v2 += (int)hbuf[y0 + j][c][2] * k;
    }
    const size_t o = (size_t)img * 3 * plane + (size_t)oy * Wo + ox;
    out[o] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8(v0), 255.f), m0), s0);
    out[o + plane] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8(v1), 255.f), m1), s1);
    out[o + 2 * plane] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8(v2), 255.f), m2), s2);
  }
}

// Three small launches over many workgroups (round 3: the one-workgroup sweep took 190 us for 120 k points): a tile of 1024 points counts its
// survivors; every tile adds the counts in front of it (<= a few hundred integers), compacts its survivors in point order into `work` and
// the last one leaves the total; the output rows (survivors then zeros, or the caller's random choice among them) are an elementwise pass.
constexpr int LFT = 1024;
__device__ __forceinline__ bool lfp_keep(const float* __restrict__ pts, int i, int N, int C, float x0, float y0, float z0, float x1,
                                         float y1, float z1) {
  if (i >= N) return false;
  const float px = pts[(size_t)i * C], py = pts[(size_t)i * C + 1], pz = pts[(size_t)i * C + 2];
  return px > x0 && px < x1 && py > y0 && py < y1 && pz > z0 && pz < z1;         // NaN fails every test, like numpy
}
__global__ __launch_bounds__(LFT) void lfp_count(const float* __restrict__ pts, int* __restrict__ tcount, int N, int C, float x0, float y0,
                                                 float z0, float x1, float y1, float z1) {
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long bal = __ballot(lfp_keep(pts, blockIdx.x * LFT + tid, N, C, x0, y0, z0, x1, y1, z1));
  if (lane == 0) wsum[wave] = __popcll(bal);
  __syncthreads();
  if (tid == 0) {
    int t = 0;
    for (int w = 0; w < 16; ++w) t += wsum[w];
    tcount[blockIdx.x] = t;
  }
}
__global__ __launch_bounds__(LFT) void lfp_compact(const float* __restrict__ pts, float* __restrict__ work, const int* __restrict__ tcount,
                                                   int* __restrict__ count, int N, int C, float x0, float y0, float z0, float x1,
                                                   float y1, float z1) {
  __shared__ int wsum[16], wpre[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, t = blockIdx.x;
  int part = 0;                                                      // survivors in the tiles in front of this one
  for (int j = tid; j < t; j += LFT) part += tcount[j];
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
  if (lane == 0) wpre[wave] = part;
  const int i = t * LFT + tid;
  const bool keep = lfp_keep(pts, i, N, C, x0, y0, z0, x1, y1, z1);
  const unsigned long long bal = __ballot(keep);
  const int before = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) wsum[wave] = __popcll(bal);
  __syncthreads();
  int off = 0, woff = 0, tot = 0;
  for (int w = 0; w < 16; ++w) {
    off += wpre[w];
    if (w < wave) woff += wsum[w];
    tot += wsum[w];
  }
  if (keep) {
    float* dst = work + (size_t)(off + woff + before) * C;
    for (int c = 0; c < C; ++c) dst[c] = pts[(size_t)i * C + c];
  }
  if (tid == 0 && t == (int)gridDim.x - 1) *count = off + tot;
}
__global__ __launch_bounds__(256) void lfp_output(const float* __restrict__ work, float* __restrict__ out, const int* __restrict__ count,
                                                  const long long* __restrict__ choice, int C, int max_points) {
  const int total = *count;
  const bool gather = choice != nullptr && total >= max_points;
  const long long n = (long long)max_points * C;
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const int r = (int)(e / C), c = (int)(e - (long long)r * C);
    float v = 0.f;
    if (gather) {
      const long long s = choice[r];
      if (s >= 0 && s < total) v = work[(size_t)s * C + c];
    } else if (r < total) {
      v = work[(size_t)r * C + c];
    }
    out[e] = v;
  }
}

}  // namespace

extern "C" int bevf_resize_normalize_u8(const unsigned char* x, float* out, int n, int H, int W, int Ho, int Wo,
                                        const int32_t* bounds_h, const int32_t* coef_h, int ksize_h,
                                        const int32_t* bounds_v, const int32_t* coef_v, int ksize_v, const float* mean3,
                                        const float* std3, void* stream) {
  BEVF_REQUIRE(x && out && bounds_h && coef_h && bounds_v && coef_v && mean3 && std3, "resize_normalize: null pointer");
  BEVF_REQUIRE(n > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && ksize_h > 0 && ksize_v > 0, "resize_normalize: bad shape");
  BEVF_REQUIRE(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, "resize_normalize: zero std");
  const long long total = (long long)n * Ho * Wo;
  BEVF_REQUIRE((total + 255) / 256 < (1ll << 31), "resize_normalize: grid too large");
  // two-pass tiled form when a tile of 16 output rows never needs more than RMAX input rows (down-scales up to ~2.7x; the row
  // support is ksize_v, consecutive windows advance by the scale): RT * scale + ksize_v <= RMAX
  const double scale_v = (double)H / Ho;
  if (scale_v >= 1.0 && (RT - 1) * scale_v + ksize_v + 2 <= RMAX) {
    const int tilesX = (Wo + CT_ - 1) / CT_, tilesY = (Ho + RT - 1) / RT;
    hipLaunchKernelGGL(resize_normalize_u8_tiled, dim3((unsigned)((long long)n * tilesX * tilesY)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, out, n, H, W, Ho, Wo, bounds_h, coef_h, ksize_h, bounds_v, coef_v, ksize_v,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], tilesX, tilesY);
    return bevf_check_launch("bevf_resize_normalize_u8");
  }
  hipLaunchKernelGGL(resize_normalize_u8, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, out, n, H, W, Ho, Wo, bounds_h, coef_h, ksize_h, bounds_v,
                     coef_v, ksize_v, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  return bevf_check_launch("bevf_resize_normalize_u8");
}

extern "C" int bevf_lidar_filter_pad_f32(const float* points, float* out, int32_t* count, float* work,
                                         const int64_t* choice, int N, int C, int max_points, const float* pc_range6,
                                         void* stream) {
  BEVF_REQUIRE((points || N == 0) && out && count && work && pc_range6, "lidar_filter_pad: null pointer");
  BEVF_REQUIRE(N >= 0 && C >= 3 && max_points > 0, "lidar_filter_pad: need N >= 0, C >= 3, max_points > 0");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int tiles = (N + LFT - 1) / LFT;
  int* const tcount = reinterpret_cast<int*>(work + (size_t)N * C);   // the tile counts live behind the compacted rows
  const float x0 = pc_range6[0], y0 = pc_range6[1], z0 = pc_range6[2], x1 = pc_range6[3], y1 = pc_range6[4], z1 = pc_range6[5];
  if (tiles > 0) {
    hipLaunchKernelGGL(lfp_count, dim3(tiles), dim3(LFT), 0, st, points, tcount, N, C, x0, y0, z0, x1, y1, z1);
    hipLaunchKernelGGL(lfp_compact, dim3(tiles), dim3(LFT), 0, st, points, work, tcount, count, N, C, x0, y0, z0, x1, y1, z1);
  } else {
    (void)hipMemsetAsync(count, 0, sizeof(int32_t), st);
  }
  const long long n = (long long)max_points * C;
  hipLaunchKernelGGL(lfp_output, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, st, work, out, count,
                     reinterpret_cast<const long long*>(choice), C, max_points);
  return bevf_check_launch("bevf_lidar_filter_pad_f32");
}
