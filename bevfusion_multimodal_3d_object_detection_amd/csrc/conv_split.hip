// fp32 implicit-GEMM convolution through three bf16 planes ("f32x3"), gfx950.
//
// Same GEMM view, tiling, prologue and epilogue as conv_igemm.hip, but the multiply runs on
// v_mfma_f32_32x32x16_bf16 (16x the rate of v_mfma_f32_32x32x2_f32): every fp32 operand is split exactly into
//     x = hi + mid + lo,   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)      (|residual| <~ 2^-25 |x|)
// and a product a*b is accumulated in fp32 as the six partial products that matter,
//     hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi          (dropped: mid*lo, lo*mid, lo*lo ~ 2^-24 |a*b|)
// so the result carries fp32-level error (measured against fp64 next to the exact-fp32 kernel in
// tests/test_gpu_split.py) without being bit-identical to an fp32 FMA chain.  It is an OPT-IN mode
// (engine.set_conv_mode("f32x3")); the default path stays exact fp32.
// Measured (MI355X, random data): 195-210 TF algorithmic on the 3x3 layers vs 135-146 for the exact kernel.  Halving the
// MFMA count in a what-if build showed the six products alone take ~2/3 of the time at ~1.77 PFLOP/s bf16 -- the chip
// holds ~1.7 GHz under this load -- so the scheme's ceiling is ~295 TF; the rest is staging (the split costs ~12 %).
//
// Activations stay fp32 in HBM and are split while they are staged into LDS (about 26 vector-ALU instructions per
// 4 elements; the bf16 MFMA leaves most of the vector issue slots free, unlike the fp32 MFMA).  Weights are split
// once by bevf_split_weights_f32x3 into [3][Cout][K] bf16 planes.
// LDS (single buffer, next K step prefetched in registers): per plane a row of a K step is 32 bf16 = 64 B = 4 chunks
// of 16 B; chunk c of row r sits at r*64 + ((c ^ ((r>>1)&3)) << 4), which makes the ds_read_b128 fragment reads
// (32 rows x one chunk per half-wave) bank-conflict free.  Lane half h of k-group g reads chunk 2g+h = k 16g+8h..+7,
// exactly the operand layout of v_mfma_f32_32x32x16_bf16.
#include "conv_common.h"

#include <cmath>

namespace {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split3(const f32x4 v, u32x2& hi, u32x2& mid, u32x2& lo) {
  const bf16x4 h = __builtin_convertvector(v, bf16x4);
  const f32x4 r1 = v - __builtin_convertvector(h, f32x4);
  const bf16x4 m = __builtin_convertvector(r1, bf16x4);
  const f32x4 r2 = r1 - __builtin_convertvector(m, f32x4);
  const bf16x4 l = __builtin_convertvector(r2, bf16x4);
  hi = __builtin_bit_cast(u32x2, h);
  mid = __builtin_bit_cast(u32x2, m);
  lo = __builtin_bit_cast(u32x2, l);
}

__global__ __launch_bounds__(256) void split_weights(const float* __restrict__ w, __bf16* __restrict__ out, long long n) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= n) return;
  const float x = w[i];
  const __bf16 h = (__bf16)x;
  const float r1 = x - (float)h;
  const __bf16 m = (__bf16)r1;
  const float r2 = r1 - (float)m;
  out[i] = h;
  out[n + i] = m;
  out[2 * n + i] = (__bf16)r2;
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_split(const ConvArgs p) {
  constexpr int WAVES_N = BN / WN;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int AP = BM / 32, BPJ = BN / 64;
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
  constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64, B_BASE = 3 * A_PLANE;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  char* const ldsb = reinterpret_cast<char*>(lds);

  const int sid = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = sid / p.tilesN, tn = sid % p.tilesN;
  const int m_hi = p.M;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- A staging: thread owns fp32 chunk `chunk` (k = 4*chunk..+3) of rows srow + 32*j ----------------------
  const int chunk = tid & 7, srow = tid >> 3;
  unsigned a_voff[AP], a_eff[AP];
  int a_ih0[AP], a_iw0[AP];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int j = 0; j < AP; ++j) {
    const int m = m0 + srow + 32 * j;
    if (m < m_hi) {
      const int n = fastdiv(m, p.div_hw_mul, p.div_hw_sh), r = m - n * HoWo;
      const int oh = fastdiv(r, p.div_w_mul, p.div_w_sh), ow = r - oh * p.Wo;
      a_voff[j] = (unsigned)((((n * p.H + oh * p.stride) * p.W + ow * p.stride) * p.x_cs + chunk * 4) * 4);
      a_ih0[j] = oh * p.stride - p.pad;
      a_iw0[j] = ow * p.stride - p.pad;
    } else {
      a_voff[j] = kOob;
      a_ih0[j] = -(1 << 24);
      a_iw0[j] = -(1 << 24);
    }
  }
  // ---- B staging: thread owns 16-B chunk q (8 bf16) of rows brow + 64*j of each plane -----------------------
  const int bq = tid & 3, brow = tid >> 2;
  unsigned b_voff[BPJ];
#pragma unroll
  for (int j = 0; j < BPJ; ++j) {
    const int n = n0 + brow + 64 * j;
    b_voff[j] = n < p.Cout ? (unsigned)(((size_t)n * p.K) * 2 + bq * 16) : kOob;
  }
  const size_t plane_elems = (size_t)p.Cout * p.K;
  const __bf16* const wb = static_cast<const __bf16*>(p.w);
  const __amdgpu_buffer_rsrc_t rsrcB0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(wb), 0, (int)kOob, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcB1 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(wb + plane_elems), 0, (int)kOob, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcB2 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(wb + 2 * plane_elems), 0, (int)kOob, 0x00020000);

  int kh = 0, kw = 0, c0 = 0;
  auto set_tap = [&]() {
#pragma unroll
    for (int j = 0; j < AP; ++j) {
      const int ih = a_ih0[j] + kh, iw = a_iw0[j] + kw;
      a_eff[j] = ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) ? a_voff[j] : kOob;
    }
  };
  f32x4 ra[AP];
  u32x4 rb[3][BPJ];
  auto load_tile = [&](int kstep) {
    const char* base = static_cast<const char*>(p.x) + ((long)(kh - p.pad) * p.W + (kw - p.pad)) * p.x_cs * 4;
    const __amdgpu_buffer_rsrc_t rsrcA =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)kOob, 0x00020000);
#pragma unroll
    for (int j = 0; j < AP; ++j) ra[j] = buf_load16(rsrcA, a_eff[j], (unsigned)(c0 * 4));
#pragma unroll
    for (int j = 0; j < BPJ; ++j) {
      rb[0][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrcB0, b_voff[j], (unsigned)kstep * 64u, 0);
      rb[1][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrcB1, b_voff[j], (unsigned)kstep * 64u, 0);
      rb[2][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrcB2, b_voff[j], (unsigned)kstep * 64u, 0);
    }
  };
  auto advance = [&]() {
    c0 += BK;
    if (c0 == p.Cin) {
      c0 = 0;
      if (++kw == p.KW) { kw = 0; ++kh; }
      set_tap();
    }
  };

  // LDS addresses
  const int a_wr = srow * 64 + (((chunk >> 1) ^ ((srow >> 1) & 3)) << 4) + (chunk & 1) * 8;
  const int b_wr = B_BASE + brow * 64 + ((bq ^ ((brow >> 1) & 3)) << 4);
  const int h = lane >> 5, l31 = lane & 31;
  int a_rd[2], b_rd[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    a_rd[g] = (wm * WM + l31) * 64 + (((2 * g + h) ^ ((l31 >> 1) & 3)) << 4);
    b_rd[g] = B_BASE + (wn * WN + l31) * 64 + (((2 * g + h) ^ ((l31 >> 1) & 3)) << 4);
  }
  auto store_tile = [&]() {
#pragma unroll
    for (int j = 0; j < AP; ++j) {
      u32x2 hi, mid, lo;
      split3(ra[j], hi, mid, lo);
      *reinterpret_cast<u32x2*>(ldsb + a_wr + j * 2048) = hi;
      *reinterpret_cast<u32x2*>(ldsb + a_wr + j * 2048 + A_PLANE) = mid;
      *reinterpret_cast<u32x2*>(ldsb + a_wr + j * 2048 + 2 * A_PLANE) = lo;
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int j = 0; j < BPJ; ++j) *reinterpret_cast<u32x4*>(ldsb + b_wr + pl * B_PLANE + j * 4096) = rb[pl][j];
  };

  f32x16 acc[MI][NI];
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto compute = [&](auto firstc) {
    constexpr bool first = decltype(firstc)::value;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      bf16x8 a[3][MI], b[3][NI];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          a[pl][mi] = *reinterpret_cast<const bf16x8*>(ldsb + a_rd[g] + pl * A_PLANE + mi * 2048);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          b[pl][ni] = *reinterpret_cast<const bf16x8*>(ldsb + b_rd[g] + pl * B_PLANE + ni * 2048);
      }
      // small partial products first: (hi,lo) (lo,hi) (mid,mid) (hi,mid) (mid,hi) (hi,hi)
      constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t]][mi], b[PB[t]][ni],
                                                                  first && g == 0 && t == 0 ? zero : acc[mi][ni], 0, 0, 0);
    }
  };

  // ---- K loop: single LDS buffer, the next step's operands are in flight in registers during the MFMAs ----
  const int KT = p.K / BK;
  set_tap();
  load_tile(0);
  store_tile();
  __syncthreads();
  if (KT > 1) {
    advance();
    load_tile(1);
  }
  __builtin_amdgcn_sched_barrier(0);
  compute(std::true_type{});
  for (int kt = 1; kt < KT; ++kt) {
    __syncthreads();                                // every wave has read step kt-1
    store_tile();
    __syncthreads();
    if (kt + 1 < KT) {
      advance();
      load_tile(kt + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    compute(std::false_type{});
  }
  conv_epilogue<float, BM, BN, WM, WN>(p, acc, m0, n0, m_hi, wm, wn, lane);
}

template <int BM, int BN, int WM, int WN>
int launch_split(ConvArgs a, hipStream_t st) {
  a.tilesM = (a.M + BM - 1) / BM;
  a.tilesN = (a.Cout + BN - 1) / BN;
  constexpr size_t lds_bytes = size_t(3) * (BM + BN) * 64;
  hipLaunchKernelGGL((conv_split<BM, BN, WM, WN>), dim3(a.tilesM * a.tilesN), dim3(256), lds_bytes, st, a);
  return bevf_check_launch("bevf_conv2d_nhwc_f32x3");
}

}  // namespace

extern "C" int bevf_split_weights_f32x3(const float* w, void* planes, size_t n, void* stream) {
  BEVF_REQUIRE(w && planes && n > 0, "split_weights: bad arguments");
  hipLaunchKernelGGL(split_weights, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), w,
                     static_cast<__bf16*>(planes), (long long)n);
  return bevf_check_launch("bevf_split_weights_f32x3");
}

extern "C" int bevf_conv2d_nhwc_f32x3(const bevf_conv_desc* d, void* stream) {
  BEVF_REQUIRE(d && d->x && d->w, "conv f32x3: null x/w");
  BEVF_REQUIRE(d->y || d->colmax, "conv f32x3: needs y or colmax");
  BEVF_REQUIRE(!d->colmax || (d->rows_per_group > 0 && d->relu), "conv f32x3: colmax needs rows_per_group > 0 and relu (the max is taken over max(v, 0))");
  BEVF_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0 && d->Cout > 0, "conv f32x3: empty shape");
  BEVF_REQUIRE(d->Cin > 0 && d->Cin % BK == 0, "conv f32x3: Cin=%d must be a positive multiple of %d", d->Cin, BK);
  BEVF_REQUIRE(d->x_cs >= d->Cin && d->x_cs % 4 == 0, "conv f32x3: x_cs=%d must be >= Cin and a multiple of 4", d->x_cs);
  BEVF_REQUIRE(bevf_aligned16(d->x) && bevf_aligned16(d->w), "conv f32x3: x/w must be 16-byte aligned");
  BEVF_REQUIRE(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->pad >= 0, "conv f32x3: bad kernel geometry");
  BEVF_REQUIRE((d->H + 2 * d->pad - d->KH) / d->stride + 1 == d->Ho && (d->W + 2 * d->pad - d->KW) / d->stride + 1 == d->Wo,
               "conv f32x3: Ho/Wo (%d,%d) inconsistent with H,W,k,stride,pad", d->Ho, d->Wo);
  BEVF_REQUIRE(d->y_cs >= d->Cout, "conv f32x3: y_cs=%d < Cout=%d", d->y_cs, d->Cout);
  BEVF_REQUIRE(!d->res || d->res_cs >= d->Cout, "conv f32x3: res_cs < Cout");
  BEVF_REQUIRE((long long)d->N * d->H * d->W * d->x_cs * 4 < (1ll << 31) &&
                   (long long)d->Cout * d->KH * d->KW * d->Cin * 2 < (1ll << 31),
               "conv f32x3: input / weight buffers must stay below 2 GiB (32-bit buffer offsets)");
  BEVF_REQUIRE(((size_t)d->Cout * d->KH * d->KW * d->Cin * 2) % 16 == 0, "conv f32x3: plane size must be a multiple of 16 bytes");
  const long long M = (long long)d->N * d->Ho * d->Wo;
  BEVF_REQUIRE(M < (1ll << 31), "conv f32x3: pixel count overflows int32");
  ConvArgs a;
  a.x = d->x; a.w = d->w; a.scale = d->scale; a.shift = d->shift; a.res = d->res; a.y = d->y; a.colmax = d->colmax;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.x_cs = d->x_cs;
  a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout; a.y_cs = d->y_cs; a.res_cs = d->res_cs;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
  a.relu = d->relu; a.rows_per_group = d->rows_per_group;
  a.M = (int)M; a.K = d->KH * d->KW * d->Cin; a.tilesM = a.tilesN = 0;
  a.m_split = 0; a.nbig = 0; a.tilesN_big = 0;
  fastdiv_make(d->Ho * d->Wo, &a.div_hw_mul, &a.div_hw_sh);
  fastdiv_make(d->Wo, &a.div_w_mul, &a.div_w_sh);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (d->tile) {
    case 0: break;
    case 1: return launch_split<128, 128, 64, 64>(a, st);
    case 3: return launch_split<128, 64, 64, 32>(a, st);
    case 4: return launch_split<64, 64, 32, 32>(a, st);
    default: bevf_set_error("conv f32x3: unknown tile variant %d", d->tile); return BEVF_ERR_ARG;
  }
  const double wg128 = std::ceil(M / 128.0) * ((d->Cout + 127) / 128);
  if (d->Cout > 64 && wg128 >= 768) return launch_split<128, 128, 64, 64>(a, st);
  if (std::ceil(M / 128.0) * ((d->Cout + 63) / 64) >= 1024) return launch_split<128, 64, 64, 32>(a, st);
  return launch_split<64, 64, 32, 32>(a, st);
}
