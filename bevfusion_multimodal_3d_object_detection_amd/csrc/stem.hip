// ResNet stem: 7x7 stride-2 pad-3 convolution of a planar 3-channel image, fused BN + ReLU,
// on v_mfma_f32_32x32x2_f32.  ref src/encoders.py:154-156.
//
// One workgroup = one output row segment of 128 pixels x all 64 channels.  The 3 x 7 input
// rows it needs (261 columns) are staged once in LDS straight from the NCHW image (coalesced
// along W, zero-filled outside the image); the whole filter bank sits beside it as
// [k][channel] (packed once by the host) so the B-operand reads are conflict-free.  The A operand is read
// element-wise out of the patch: A[pixel i][k=(c,kh,kw)] = patch[c*7+kh][2*i + kw].
// K = 147 is padded to 148 and split in two halves of 74: lane half h walks k = p + 74*h.
#include "common.h"

namespace {

constexpr int TP = 128;              // output pixels per workgroup (along W)
constexpr int PW = 2 * TP + 8;       // patch row pitch (261 used)
constexpr int PROWS = 22;            // 21 patch rows + 1 zero row read by the padded k = 147
constexpr int KPAD = 148, KHALF = 74;

__global__ __launch_bounds__(256) void stem_conv7x7_f32(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float* __restrict__ y,
                                                         int H, int W, int Ho, int Wo, int tilesW) {
  __shared__ __attribute__((aligned(16))) float patch[PROWS * PW];
  __shared__ __attribute__((aligned(16))) float wl[KPAD * 64];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tw = blockIdx.x % tilesW;
  const int oh = (blockIdx.x / tilesW) % Ho;
  const int n = blockIdx.x / (tilesW * Ho);
  const int ow0 = tw * TP;

  // filter bank, pre-packed by the host as [k = c*49+kh*7+kw][co] with a zero row k = 147
  for (int i = tid; i < KPAD * 16; i += 256)
    reinterpret_cast<f32x4*>(wl)[i] = reinterpret_cast<const f32x4*>(w)[i];
  // patch rows r = c*7 + kh  <->  image row ih = 2*oh - 3 + kh of plane c; columns iw = 2*ow0 - 3 + col
  const float* img = x + (size_t)n * 3 * H * W;
  for (int i = tid; i < PROWS * PW; i += 256) {
    const int r = i / PW, col = i - r * PW;
    float v = 0.f;
    if (r < 21) {
      const int c = r / 7, kh = r - c * 7;
      const int ih = 2 * oh - 3 + kh, iw = 2 * ow0 - 3 + col;
      if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) v = img[((size_t)c * H + ih) * W + iw];
    }
    patch[i] = v;
  }
  __syncthreads();

  const int h = lane >> 5, l31 = lane & 31;
  const int pix = wave * 32 + l31;             // pixel within the tile
  int k = KHALF * h;                           // this lane half's first k
  int kw = k % 7;
  int offA = (k / 7) * PW + kw + 2 * pix;
  int offB = k * 64 + l31;
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  for (int p = 0; p < KHALF; ++p) {
    const float a = patch[offA];
    const float b0 = wl[offB], b1 = wl[offB + 32];
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
    if (++kw == 7) { kw = 0; offA += PW - 6; } else { offA += 1; }
    offB += 64;
  }

  // epilogue: column j = lane&31 -> channel, row i -> pixel wave*32 + i
  const float sc0 = scale[l31], sh0 = shift[l31], sc1 = scale[l31 + 32], sh1 = shift[l31 + 32];
  float* yrow = y + ((size_t)(n * Ho + oh) * Wo + ow0) * 64;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (ow0 + i < Wo) {
      const float v0 = fmaf(acc0[r], sc0, sh0), v1 = fmaf(acc1[r], sc1, sh1);
      yrow[(size_t)i * 64 + l31] = v0 > 0.f ? v0 : 0.f;
      yrow[(size_t)i * 64 + l31 + 32] = v1 > 0.f ? v1 : 0.f;
    }
  }
}

// 3x3 stride-2 pad-1 max-pool on NHWC, 4 channels per thread.  ref src/encoders.py:157.
__global__ __launch_bounds__(256) void maxpool3x3s2_nhwc(const float* __restrict__ x, float* __restrict__ y, int N,
                                                          int H, int W, int C, int Ho, int Wo) {
  const int c4 = C >> 2;
  const long long total = (long long)N * Ho * Wo * c4;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % c4);
    long long pix = i / c4;
    const int ow = (int)(pix % Wo);
    pix /= Wo;
    const int oh = (int)(pix % Ho), n = (int)(pix / Ho);
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
      const int ih = 2 * oh - 1 + dh;
      if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        const int iw = 2 * ow - 1 + dw;
        if ((unsigned)iw >= (unsigned)W) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((size_t)(n * H + ih) * W + iw) * C + c * 4);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    *reinterpret_cast<f32x4*>(y + (size_t)i * 4) = m;
  }
}

}  // namespace

extern "C" int bevf_stem_conv7x7_f32(const float* x, const float* w, const float* scale, const float* shift,
                                     float* y, int N, int H, int W, void* stream) {
  BEVF_REQUIRE(x && w && scale && shift && y, "stem: null pointer");
  BEVF_REQUIRE(bevf_aligned16(w), "stem: packed filter bank must be 16-byte aligned");
  BEVF_REQUIRE(N > 0 && H >= 1 && W >= 1, "stem: empty shape");
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  const int tilesW = (Wo + TP - 1) / TP;
  const long long grid = (long long)N * Ho * tilesW;
  BEVF_REQUIRE(grid < (1ll << 31), "stem: grid too large");
  hipLaunchKernelGGL(stem_conv7x7_f32, dim3((unsigned)grid), dim3(256), 0, static_cast<hipStream_t>(stream), x, w,
                     scale, shift, y, H, W, Ho, Wo, tilesW);
  return bevf_check_launch("bevf_stem_conv7x7_f32");
}

extern "C" int bevf_maxpool3x3s2_nhwc_f32(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  BEVF_REQUIRE(x && y, "maxpool: null pointer");
  BEVF_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool: C=%d must be a positive multiple of 4", C);
  BEVF_REQUIRE(bevf_aligned16(x) && bevf_aligned16(y), "maxpool: unaligned");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long long total = (long long)N * Ho * Wo * (C / 4);
  const unsigned grid = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(maxpool3x3s2_nhwc, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, N, H, W, C,
                     Ho, Wo);
  return bevf_check_launch("bevf_maxpool3x3s2_nhwc_f32");
}
