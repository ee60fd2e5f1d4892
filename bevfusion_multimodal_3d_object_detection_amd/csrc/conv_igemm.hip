// Implicit-GEMM NHWC convolution on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain), gfx950.
//
//   GEMM view:  C[M = N*Ho*Wo pixels][Cout] = A[M][K = KH*KW*Cin] * W^T[K][Cout]
//   A is never materialised: a BK = 32 slice of K lies inside one filter tap (Cin % 32 == 0),
//   so a row of the A tile is 128 contiguous bytes of one input pixel (or zeros for padding).
//
// Workgroup = 256 threads = 4 waves, tile BM x BN, wave tile WM x WN built from 32x32 MFMA
// tiles.  Global -> registers -> LDS staging, double-buffered in LDS with the loads for K-step
// t+1 issued before the MFMAs of step t and written to the other buffer after them (one
// barrier per K-step).  LDS rows are 128 B; the 16-B chunk index is XOR-swizzled with
// (row>>1)&7 so that the ds_read_b128 fragment reads (32 rows x one chunk per half-wave) are
// bank-conflict free.  Each lane reads 4 consecutive k of its row: lane half h of k-group g
// owns k = 8g+4h..8g+4h+3, and MFMA step s pairs (k=8g+s | k=8g+4+s) -- a fixed permutation
// of the K order that A and B share.
//
// Wave quantisation: a layer whose big-tile count is not a multiple of the 512 resident
// workgroups (2 per CU) would pay a whole extra round for a few leftover tiles.  The hybrid
// kernel therefore covers rows [0, m_split) with big tiles (whole rounds) and the remaining
// rows with 64x64 tiles in the SAME grid: the small workgroups are dispatched last and fill
// CUs as the big ones drain.
#include "common.h"

namespace {

struct ConvArgs {
  const float* x;
  const float* w;
  const float* scale;
  const float* shift;
  const float* res;
  float* y;
  uint32_t* colmax;
  int N, H, W, Cin, x_cs;
  int Ho, Wo, Cout, y_cs, res_cs;
  int KH, KW, stride, pad;
  int relu, rows_per_group;
  int M, K, tilesM, tilesN;
  int m_split, nbig, tilesN_big;   // hybrid launch: blocks [0,nbig) = big tiles over rows [0,m_split)
};

constexpr int BK = 32;

template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void conv_tile(const ConvArgs& p, float* lds, const int m_lo, const int m_hi,
                                          const int tm, const int tn) {
  constexpr int WAVES_N = BN / WN;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int AP = BM / 32, BP = BN / 32;
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");

  float* As = lds;                 // [2][BM][BK]
  float* Bs = lds + 2 * BM * BK;   // [2][BN][BK]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int m0 = m_lo + tm * BM, n0 = tn * BN;

  // ---- staging role: thread owns 16-B chunk `chunk` of rows srow + 32*j -------------------
  const int chunk = tid & 7, srow = tid >> 3;
  int a_pix[AP], a_ih0[AP], a_iw0[AP];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int j = 0; j < AP; ++j) {
    const int m = m0 + srow + 32 * j;
    if (m < m_hi) {
      const int n = m / HoWo, r = m - n * HoWo;
      const int oh = r / p.Wo, ow = r - oh * p.Wo;
      a_pix[j] = n * p.H * p.W;
      a_ih0[j] = oh * p.stride - p.pad;
      a_iw0[j] = ow * p.stride - p.pad;
    } else {
      a_pix[j] = 0;
      a_ih0[j] = -(1 << 24);
      a_iw0[j] = -(1 << 24);
    }
  }
  const float* wrow[BP];
  bool wok[BP];
#pragma unroll
  for (int j = 0; j < BP; ++j) {
    const int n = n0 + srow + 32 * j;
    wok[j] = n < p.Cout;
    wrow[j] = p.w + (size_t)(wok[j] ? n : 0) * p.K + chunk * 4;
  }

  f32x4 ra[AP], rb[BP];
  auto load_tile = [&](int kh, int kw, int c0, int kofs) {
#pragma unroll
    for (int j = 0; j < AP; ++j) {
      const int ih = a_ih0[j] + kh, iw = a_iw0[j] + kw;
      const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      const float* src = p.x + (size_t)(a_pix[j] + ih * p.W + iw) * p.x_cs + c0 + chunk * 4;
      ra[j] = ok ? *reinterpret_cast<const f32x4*>(src) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < BP; ++j)
      rb[j] = wok[j] ? *reinterpret_cast<const f32x4*>(wrow[j] + kofs) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int j = 0; j < AP; ++j) {
      const int row = srow + 32 * j;
      *reinterpret_cast<f32x4*>(&As[(buf * BM + row) * BK + ((chunk ^ ((row >> 1) & 7)) << 2)]) = ra[j];
    }
#pragma unroll
    for (int j = 0; j < BP; ++j) {
      const int row = srow + 32 * j;
      *reinterpret_cast<f32x4*>(&Bs[(buf * BN + row) * BK + ((chunk ^ ((row >> 1) & 7)) << 2)]) = rb[j];
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int h = lane >> 5, l31 = lane & 31;
  auto compute = [&](int buf) {
    const float* Ab = As + buf * BM * BK;
    const float* Bb = Bs + buf * BN * BK;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 a[MI], b[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int row = wm * WM + mi * 32 + l31;
        a[mi] = *reinterpret_cast<const f32x4*>(&Ab[row * BK + (((2 * g + h) ^ ((row >> 1) & 7)) << 2)]);
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int row = wn * WN + ni * 32 + l31;
        b[ni] = *reinterpret_cast<const f32x4*>(&Bb[row * BK + (((2 * g + h) ^ ((row >> 1) & 7)) << 2)]);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][s], b[ni][s], acc[mi][ni], 0, 0, 0);
    }
  };

  // ---- K loop --------------------------------------------------------------------------------
  const int KT = p.K / BK;
  int kh = 0, kw = 0, c0 = 0;
  load_tile(0, 0, 0, 0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < KT;
    if (more) {
      c0 += BK;
      if (c0 == p.Cin) {
        c0 = 0;
        if (++kw == p.KW) { kw = 0; ++kh; }
      }
      load_tile(kh, kw, c0, (kt + 1) * BK);
    }
    compute(cur);
    if (more) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C[i][j], j = lane&31 (channel), i = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel) ----
  // Branch-free per element: residual loads of a 32x32 tile are issued as one batch (rows past
  // the end are clamped for the load and masked at the store).
  const int m_base = m0 + wm * WM, n_base = n0 + wn * WN;
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
          rv[r] = p.res[(size_t)m * p.res_cs + nc];
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
          if (m < m_hi && nok) p.y[(size_t)m * p.y_cs + n] = v[r];
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

// single tile shape over all rows
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_f32(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int sid = xcd_remap(blockIdx.x, gridDim.x);
  conv_tile<BM, BN, WM, WN>(p, lds, 0, p.M, sid / p.tilesN, sid % p.tilesN);
}

// big tiles over rows [0, m_split) + 64x64 tiles over rows [m_split, M) in one grid
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_f32_hybrid(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if ((int)blockIdx.x < p.nbig) {
    const int sid = xcd_remap(blockIdx.x, p.nbig);
    conv_tile<BM, BN, WM, WN>(p, lds, 0, p.m_split, sid / p.tilesN_big, sid % p.tilesN_big);
  } else {
    const int sid = xcd_remap(blockIdx.x - p.nbig, gridDim.x - p.nbig);
    conv_tile<64, 64, 32, 32>(p, lds, p.m_split, p.M, sid / p.tilesN, sid % p.tilesN);
  }
}

template <int BM, int BN, int WM, int WN>
int launch(const ConvArgs& a0, hipStream_t st) {
  ConvArgs a = a0;
  a.tilesM = (a.M + BM - 1) / BM;
  a.tilesN = (a.Cout + BN - 1) / BN;
  constexpr size_t lds_bytes = size_t(2) * (BM + BN) * BK * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    if (lds_bytes > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_f32<BM, BN, WM, WN>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_igemm_f32<BM, BN, WM, WN>), dim3(a.tilesM * a.tilesN), dim3(256), lds_bytes, st, a);
  return bevf_check_launch("bevf_conv2d_nhwc_f32");
}

template <int BM, int BN, int WM, int WN>
int launch_hybrid(const ConvArgs& a0, int big_mtiles, hipStream_t st) {
  ConvArgs a = a0;
  a.tilesN_big = (a.Cout + BN - 1) / BN;
  a.nbig = big_mtiles * a.tilesN_big;
  a.m_split = big_mtiles * BM;
  a.tilesN = (a.Cout + 63) / 64;                               // small-tile geometry
  a.tilesM = (a.M - a.m_split + 63) / 64;
  const int nsmall = a.tilesM * a.tilesN;
  constexpr size_t lds_bytes = size_t(2) * (BM + BN) * BK * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    if (lds_bytes > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_f32_hybrid<BM, BN, WM, WN>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_igemm_f32_hybrid<BM, BN, WM, WN>), dim3(a.nbig + nsmall), dim3(256), lds_bytes, st, a);
  return bevf_check_launch("bevf_conv2d_nhwc_f32");
}

constexpr int kResidentBig = 512;     // 256 CUs x 2 workgroups (64-80 KB of LDS each)

// Rows that big tiles should cover so that they fill whole rounds of the chip; the rest goes to
// 64x64 tiles.  Returns the number of big M-tiles (0 = all small, tilesM = all big).
static int split_big_mtiles(long long M, int BM, int tilesN_big) {
  const int tilesM = (int)((M + BM - 1) / BM);
  const long long nb = (long long)tilesM * tilesN_big;
  const long long rem = nb % kResidentBig;
  if (rem == 0 || rem >= (kResidentBig * 3) / 4) return tilesM;          // last round is (nearly) full anyway
  const long long full = nb - rem;
  return (int)(full / tilesN_big);                                       // whole rounds only (may be 0)
}

}  // namespace

extern "C" int bevf_conv2d_nhwc_f32(const bevf_conv_desc* d, void* stream) {
  BEVF_REQUIRE(d && d->x && d->w, "conv: null x/w");
  BEVF_REQUIRE(d->y || d->colmax, "conv: neither y nor colmax given");
  BEVF_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0 && d->Cout > 0, "conv: empty shape");
  BEVF_REQUIRE(d->Cin > 0 && d->Cin % 32 == 0, "conv: Cin=%d must be a positive multiple of 32", d->Cin);
  BEVF_REQUIRE(d->x_cs >= d->Cin && d->x_cs % 4 == 0, "conv: x_cs=%d must be >= Cin and a multiple of 4", d->x_cs);
  BEVF_REQUIRE(bevf_aligned16(d->x) && bevf_aligned16(d->w), "conv: x/w must be 16-byte aligned");
  BEVF_REQUIRE(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->pad >= 0, "conv: bad kernel geometry");
  BEVF_REQUIRE((d->H + 2 * d->pad - d->KH) / d->stride + 1 == d->Ho && (d->W + 2 * d->pad - d->KW) / d->stride + 1 == d->Wo,
               "conv: Ho/Wo (%d,%d) inconsistent with H,W,k,stride,pad", d->Ho, d->Wo);
  BEVF_REQUIRE(!d->y || d->y_cs >= d->Cout, "conv: y_cs=%d < Cout=%d", d->y_cs, d->Cout);
  BEVF_REQUIRE(!d->res || d->res_cs >= d->Cout, "conv: res_cs < Cout");
  BEVF_REQUIRE(!d->colmax || (d->rows_per_group > 0 && d->relu), "conv: colmax needs rows_per_group > 0 and relu");
  const long long M = (long long)d->N * d->Ho * d->Wo;
  BEVF_REQUIRE(M < (1ll << 31) && (long long)d->N * d->H * d->W < (1ll << 31), "conv: pixel count overflows int32");

  ConvArgs a;
  a.x = d->x; a.w = d->w; a.scale = d->scale; a.shift = d->shift; a.res = d->res; a.y = d->y; a.colmax = d->colmax;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.x_cs = d->x_cs;
  a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout; a.y_cs = d->y_cs; a.res_cs = d->res_cs;
  a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
  a.relu = d->relu; a.rows_per_group = d->rows_per_group;
  a.M = (int)M; a.K = d->KH * d->KW * d->Cin; a.tilesM = a.tilesN = 0;
  a.m_split = 0; a.nbig = 0; a.tilesN_big = 0;
  hipStream_t st = static_cast<hipStream_t>(stream);

  switch (d->tile) {
    case 0: break;
    case 1: return launch<128, 128, 64, 64>(a, st);
    case 2: return launch<256, 64, 64, 64>(a, st);
    case 3: return launch<128, 64, 64, 32>(a, st);
    case 4: return launch<64, 64, 32, 32>(a, st);
    case 5: return launch_hybrid<128, 128, 64, 64>(a, (int)(M / 128) / 2, st);     // tests: forced mid split
    case 6: return launch_hybrid<256, 64, 64, 64>(a, (int)(M / 256) / 2, st);
    default: bevf_set_error("conv: unknown tile variant %d", d->tile); return BEVF_ERR_ARG;
  }
  // auto: big tiles for whole rounds of the chip, 64x64 tiles for the remaining rows
  if (d->Cout <= 64) {
    const int big = split_big_mtiles(M, 256, 1);
    if (big == 0) return launch<64, 64, 32, 32>(a, st);
    if (big == (int)((M + 255) / 256)) return launch<256, 64, 64, 64>(a, st);
    return launch_hybrid<256, 64, 64, 64>(a, big, st);
  }
  const int tn = (d->Cout + 127) / 128;
  const int big = split_big_mtiles(M, 128, tn);
  if (big == 0) return launch<64, 64, 32, 32>(a, st);
  if (big == (int)((M + 127) / 128)) return launch<128, 128, 64, 64>(a, st);
  return launch_hybrid<128, 128, 64, 64>(a, big, st);
}
