// Weight gradient of the NHWC convolution on v_mfma_f32_32x32x2_f32 (training).
//
//   dW[co][tap][ci] = sum over output pixels m of  dY[m][co] * X[pixel(m) + tap][ci]
//
// GEMM view C[i = co][j = (tap,ci)] with the reduction running over pixels: both operands are stored
// pixel-major (channel contiguous), so a 32-pixel step stages [32][BI] of dY and [32][BJ] of (shifted) X
// in LDS and the MFMA fragments are plain ds_read_b32 with immediate offsets (lane = channel, the two
// lane halves take the two pixels of a step).  The pixel range is split over workgroups (the output is
// tiny, the reduction huge) and partial tiles are accumulated with fp32 atomics into a zeroed dW.
// Pixel -> input address decoding comes from a per-shape table of byte offsets, one per (output pixel, filter tap),
// so the loop has NO per-step address arithmetic on the vector ALU (which v_mfma_f32_32x32x2_f32 shares): the table
// and the dY rows are read through buffer descriptors whose base the scalar unit advances each step, the per-lane
// offsets are loop constants, and the only VALU work per staged row is one add (tap offset + channel offset).
#include "common.h"

#include <type_traits>

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned kOob = 0x80000000u;
constexpr int PS = 32;   // pixels per step

__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
}

// taptab[m][t] = byte offset of x[n][ih][iw][0] for output pixel m under filter tap t, or kOob when the tap falls into
// the padding (the buffer load then returns zeros); rows m in [M, Mpad) are all kOob (Mpad = M rounded up to PS)
__global__ __launch_bounds__(256) void build_taptab(int* __restrict__ tab, int M, int Mpad, int H, int W, int Ho, int Wo,
                                                     int KH, int KW, int stride, int pad, int x_cs) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  const int T = KH * KW;
  if (i >= (long long)Mpad * T) return;
  const int m = (int)(i / T), t = (int)(i - (long long)m * T);
  unsigned v = kOob;
  if (m < M) {
    const int n = m / (Ho * Wo), r = m - n * Ho * Wo, oh = r / Wo, ow = r - oh * Wo;
    const int ih = oh * stride - pad + t / KW, iw = ow * stride - pad + t % KW;
    if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
      v = (unsigned)((((long long)n * H + ih) * W + iw) * x_cs * 4);
  }
  tab[i] = (int)v;
}

struct WgradArgs {
  const float* x;
  const float* dy;
  float* dw;
  const int* taptab;
  int H, W, Cin, x_cs, Cout, dy_cs, KW, M, K, T;
  int tilesI, tilesJ, steps_per_split;
};

template <int BI, int BJ, int WI, int WJ>
__global__ __launch_bounds__(256) void conv_wgrad_f32(const WgradArgs p) {
  constexpr int MI = WI / 32, NI = WJ / 32, WAVES_J = BJ / WJ;
  static_assert((BI / WI) * (BJ / WJ) == 4, "4 waves");
  constexpr int A_BYTES = PS * BI * 4, B_BYTES = PS * BJ * 4;
  constexpr int ACH = BI / 4, BCH = BJ / 4;            // 16-B chunks per staged row
  constexpr int AROWS = 256 / ACH, BROWS = 256 / BCH;  // rows covered per pass
  constexpr int AP = PS / AROWS, BP = PS / BROWS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const ldsb = reinterpret_cast<char*>(lds);     // [A0][A1][B0][B1]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave / WAVES_J, wj = wave % WAVES_J;
  const int split = blockIdx.x / (p.tilesI * p.tilesJ);
  const int t = blockIdx.x - split * (p.tilesI * p.tilesJ);
  const int i0 = (t / p.tilesJ) * BI, j0 = (t % p.tilesJ) * BJ;
  const int step0 = split * p.steps_per_split;
  int nsteps = (p.M + PS - 1) / PS - step0;
  if (nsteps > p.steps_per_split) nsteps = p.steps_per_split;
  if (nsteps <= 0) return;

  // ---- staging roles ---------------------------------------------------------------------------------------
  const int a_chunk = tid % ACH, a_row = tid / ACH;
  const int b_chunk = tid % BCH, b_row = tid / BCH;
  const bool a_ok = i0 + a_chunk * 4 < p.Cout;
  const int jj = j0 + b_chunk * 4;                     // this thread's column of C: fixed (tap, ci)
  const bool b_ok = jj < p.K;
  const int tap = b_ok ? jj / p.Cin : 0, ci = b_ok ? jj - tap * p.Cin : 0;
  // Per-step descriptors (scalar unit): dY rows of this step with the bound at the end of the tensor, so rows past M
  // read as zeros; the table rows of this step.  Per-lane offsets below never change.
  unsigned a_voff[AP], t_voff[BP];
#pragma unroll
  for (int q = 0; q < AP; ++q)
    a_voff[q] = a_ok ? (unsigned)(((a_row + q * AROWS) * p.dy_cs + i0 + a_chunk * 4) * 4) : kOob;
#pragma unroll
  for (int q = 0; q < BP; ++q) t_voff[q] = b_ok ? (unsigned)(((b_row + q * BROWS) * p.T + tap) * 4) : kOob;
  const unsigned ci_bytes = (unsigned)(ci * 4);
  const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)kOob, 0x00020000);

  f32x4 ra[AP], rb[BP];
  unsigned tabv[BP];
  auto load_tab = [&](int step) {
    const __amdgpu_buffer_rsrc_t rsrcT = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int*>(p.taptab) + (size_t)(step0 + step) * PS * p.T, 0, (int)kOob, 0x00020000);
#pragma unroll
    for (int q = 0; q < BP; ++q) tabv[q] = __builtin_amdgcn_raw_buffer_load_b32(rsrcT, t_voff[q], 0, 0);
  };
  auto load_tiles = [&](int step) {
    const int row0 = (step0 + step) * PS;
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.dy) + (size_t)row0 * p.dy_cs, 0, (int)(((size_t)(p.M - row0 - 1) * p.dy_cs + p.Cout) * 4),
        0x00020000);
#pragma unroll
    for (int q = 0; q < AP; ++q) ra[q] = buf_load16(rsrcA, a_voff[q], 0);
#pragma unroll
    for (int q = 0; q < BP; ++q) rb[q] = buf_load16(rsrcB, tabv[q] + ci_bytes, 0);   // kOob + ci_bytes stays out of range
  };
  const int a_wr = (a_row * BI + a_chunk * 4) * 4, b_wr = 2 * A_BYTES + (b_row * BJ + b_chunk * 4) * 4;
  auto store_tiles = [&](auto bufc) {
    constexpr int buf = decltype(bufc)::value;
#pragma unroll
    for (int q = 0; q < AP; ++q) *reinterpret_cast<f32x4*>(ldsb + a_wr + buf * A_BYTES + q * AROWS * BI * 4) = ra[q];
#pragma unroll
    for (int q = 0; q < BP; ++q) *reinterpret_cast<f32x4*>(ldsb + b_wr + buf * B_BYTES + q * BROWS * BJ * 4) = rb[q];
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // MFMA tile mi of a wave covers the channels 2*l+mi (l = lane&31) of its 64-channel slab, not 32 consecutive ones:
  // a lane's MI (NI) operands are then adjacent in the pixel-major LDS rows and come in ONE ds_read_b64 whose offset
  // is an instruction immediate -- no address arithmetic in the loop.  The epilogue undoes the interleave.
  const int h = lane >> 5, l31 = lane & 31;
  const int a_rd = (h * BI + wi * WI + MI * l31) * 4, b_rd = 2 * A_BYTES + (h * BJ + wj * WJ + NI * l31) * 4;
  typedef float fragA __attribute__((ext_vector_type(MI)));
  typedef float fragB __attribute__((ext_vector_type(NI)));
  auto compute = [&](auto bufc) {
    constexpr int buf = decltype(bufc)::value;
    auto rd_a = [&](int st) { return *reinterpret_cast<const fragA*>(ldsb + a_rd + buf * A_BYTES + st * 2 * BI * 4); };
    auto rd_b = [&](int st) { return *reinterpret_cast<const fragB*>(ldsb + b_rd + buf * B_BYTES + st * 2 * BJ * 4); };
    fragA a = rd_a(0);
    fragB b = rd_b(0);
#pragma unroll
    for (int st = 0; st < PS / 2; ++st) {
      fragA an = a;
      fragB bn = b;
      if (st + 1 < PS / 2) { an = rd_a(st + 1); bn = rd_b(st + 1); }       // next fragments in flight under the MFMAs
      __builtin_amdgcn_sched_barrier(0);                                    // (keep the reads ahead of them)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
      a = an;
      b = bn;
    }
  };

  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  load_tab(0);
  load_tiles(0);
  if (nsteps > 1) load_tab(1);
  store_tiles(B0{});
  __syncthreads();
  int s = 0;
  for (; s + 2 <= nsteps; s += 2) {
    load_tiles(s + 1);
    if (s + 2 < nsteps) load_tab(s + 2);
    compute(B0{});
    store_tiles(B1{});
    __syncthreads();
    const bool more = s + 2 < nsteps;
    if (more) {
      load_tiles(s + 2);
      if (s + 3 < nsteps) load_tab(s + 3);
    }
    compute(B1{});
    if (more) store_tiles(B0{});
    __syncthreads();
  }
  if (s < nsteps) {
    compute(B0{});
    __syncthreads();
  }

  // ---- accumulate the partial tile: rows i = co, columns j = (tap,ci), both de-interleaved (see above) ----
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int j = j0 + wj * WJ + NI * l31 + ni;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = i0 + wi * WI + MI * ((r & 3) + 8 * (r >> 2) + 4 * h) + mi;
        if (co < p.Cout && j < p.K) atomicAdd(&p.dw[(size_t)co * p.K + j], acc[mi][ni][r]);
      }
    }
}

template <int BI, int BJ, int WI, int WJ>
int launch_wgrad(WgradArgs a, hipStream_t st) {
  a.tilesI = (a.Cout + BI - 1) / BI;
  a.tilesJ = (a.K + BJ - 1) / BJ;
  const int steps = (a.M + PS - 1) / PS;
  constexpr size_t lds_bytes = size_t(2) * PS * (BI + BJ) * sizeof(float);
  // One round of the chip: as many workgroups as are resident at once (LDS-bound: 2 or 3 per CU).  Measured with
  // in-kernel stamps: the K loop runs at 92-100 % of the MFMA rate, but 1020 equal workgroups on 768 slots left the
  // second round a third full (-25 %); fewer, longer workgroups also halve the atomics' share.
  const int resident = 256 * (int)((160 * 1024) / lds_bytes);
  int splits = resident / (a.tilesI * a.tilesJ);
  if (splits < 1) splits = 1;
  if (splits > steps) splits = steps;
  a.steps_per_split = (steps + splits - 1) / splits;
  splits = (steps + a.steps_per_split - 1) / a.steps_per_split;
  hipLaunchKernelGGL((conv_wgrad_f32<BI, BJ, WI, WJ>), dim3(a.tilesI * a.tilesJ * splits), dim3(256), lds_bytes, st, a);
  return bevf_check_launch("bevf_conv2d_wgrad_f32");
}

}  // namespace

extern "C" size_t bevf_conv_pixtab_bytes(int N, int H, int W, int KH, int KW, int stride, int pad) {
  if (N <= 0 || H <= 0 || W <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0) return 0;
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return 0;
  const long long M = (long long)N * Ho * Wo, Mpad = (M + PS - 1) / PS * PS;
  return (size_t)Mpad * KH * KW * sizeof(int32_t);
}

extern "C" int bevf_conv_pixtab(int32_t* tab, int N, int H, int W, int KH, int KW, int stride, int pad, int x_cs,
                                void* stream) {
  BEVF_REQUIRE(tab && N > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0 && x_cs > 0,
               "pixtab: bad arguments");
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  const long long M = (long long)N * Ho * Wo, Mpad = (M + PS - 1) / PS * PS;
  BEVF_REQUIRE(Ho > 0 && Wo > 0 && M < (1ll << 31), "pixtab: bad pixel count");
  BEVF_REQUIRE((long long)N * H * W * x_cs * 4 < (1ll << 31), "pixtab: x must stay below 2 GiB (32-bit buffer offsets)");
  BEVF_REQUIRE(Mpad * KH * KW * 4 < (1ll << 31), "pixtab: table must stay below 2 GiB");
  const long long items = Mpad * KH * KW;
  hipLaunchKernelGGL(build_taptab, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     tab, (int)M, (int)Mpad, H, W, Ho, Wo, KH, KW, stride, pad, x_cs);
  return bevf_check_launch("bevf_conv_pixtab");
}

extern "C" int bevf_conv2d_wgrad_f32(const bevf_wgrad_desc* d, void* stream) {
  BEVF_REQUIRE(d && d->x && d->dy && d->dw && d->pixtab, "wgrad: null pointer");
  BEVF_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cout > 0 && d->Cin > 0 && d->Cin % 4 == 0,
               "wgrad: Cin=%d must be a positive multiple of 4", d->Cin);
  BEVF_REQUIRE(d->x_cs >= d->Cin && d->x_cs % 4 == 0 && d->dy_cs >= d->Cout && d->dy_cs % 4 == 0 && d->Cout % 4 == 0,
               "wgrad: channel strides / Cout must be multiples of 4");
  BEVF_REQUIRE(bevf_aligned16(d->x) && bevf_aligned16(d->dy), "wgrad: unaligned");
  const int Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1, Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
  const long long M = (long long)d->N * Ho * Wo;
  BEVF_REQUIRE(M > 0 && M < (1ll << 31), "wgrad: bad pixel count");
  BEVF_REQUIRE((long long)d->N * d->H * d->W * d->x_cs * 4 < (1ll << 31) && M * d->dy_cs * 4 < (1ll << 31),
               "wgrad: x / dy buffers must stay below 2 GiB (32-bit buffer offsets)");
  WgradArgs a;
  a.x = d->x; a.dy = d->dy; a.dw = d->dw; a.taptab = d->pixtab;
  a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.x_cs = d->x_cs; a.Cout = d->Cout; a.dy_cs = d->dy_cs; a.KW = d->KW;
  a.M = (int)M; a.K = d->KH * d->KW * d->Cin; a.T = d->KH * d->KW;
  a.tilesI = a.tilesJ = a.steps_per_split = 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (d->Cout <= 64) return launch_wgrad<64, 128, 64, 32>(a, st);
  return launch_wgrad<128, 128, 64, 64>(a, st);
}
