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
#include "conv_common.h"

#include <cmath>
#include <type_traits>

namespace {

// Why prologue and epilogue are written for a minimal VALU count: v_mfma_f32_32x32x2_f32 occupies the SIMD's
// vector pipe for 64 cycles, and a VALU instruction of ANY wave on that SIMD waits for the MFMA in flight --
// measured with s_memrealtime stamps: ~300 VALU instructions of tile set-up took 8 us (64 cycles each) next to two
// other workgroups' MFMA streams, and so did the epilogue; together a third of a layer1 workgroup's lifetime, during
// which it contributes no MFMA work.  Hence: multiply-high instead of integer division, scalar (wave-uniform) row
// addresses in the epilogue, compile-time residual / ReLU variants, no per-element bounds logic on full tiles.
//
// K-loop design notes (what the ISA must look like): v_mfma_f32_32x32x2_f32 runs at the fp32
// VALU rate, so every VALU instruction in the loop costs MFMA issue time.  The loop therefore
// has no per-step address arithmetic: global operands come through buffer loads whose per-lane
// byte offset is fixed for the whole tile (invalid = padding taps / rows past the end use an
// out-of-range offset, which the hardware answers with zeros), the filter tap moves the
// descriptor's scalar base, the channel chunk is the scalar offset, LDS addresses are
// precomputed and everything that varies per step is an instruction immediate.
template <typename T, int BM, int BN, int WM, int WN>
__device__ __forceinline__ void conv_tile(const ConvArgs& p, float* lds, const int m_lo, const int m_hi,
                                          const int tm, const int tn) {
  constexpr int ES = (int)sizeof(T), EPC = 16 / ES, BKE = 8 * EPC;   // element size, elements per chunk / K step
  constexpr bool kF32 = std::is_same<T, float>::value;
  constexpr int WAVES_N = BN / WN;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int AP = BM / 32, BP = BN / 32;
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
  constexpr int A_BYTES = BM * BK * 4, B_BYTES = BN * BK * 4;   // one buffer each
  char* const ldsb = reinterpret_cast<char*>(lds);              // [A0][A1][B0][B1]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);           // scalar: wave-uniform addresses stay in SGPRs
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int m0 = m_lo + tm * BM, n0 = tn * BN;

  // ---- staging role: thread owns 16-B chunk `chunk` of rows srow + 32*j -------------------
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
      // byte offset of the pixel under filter tap (pad,pad); the tap itself moves the scalar base
      a_voff[j] = (unsigned)((((n * p.H + oh * p.stride) * p.W + ow * p.stride) * p.x_cs + chunk * EPC) * ES);
      a_ih0[j] = oh * p.stride - p.pad;
      a_iw0[j] = ow * p.stride - p.pad;
    } else {
      a_voff[j] = kOob;
      a_ih0[j] = -(1 << 24);
      a_iw0[j] = -(1 << 24);
    }
  }
  unsigned b_voff[BP];
#pragma unroll
  for (int j = 0; j < BP; ++j) {
    const int n = n0 + srow + 32 * j;
    b_voff[j] = n < p.Cout ? (unsigned)(((size_t)n * p.K + chunk * EPC) * ES) : kOob;
  }
  const __amdgpu_buffer_rsrc_t rsrcB =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)kOob, 0x00020000);

  int kh = 0, kw = 0, c0 = 0;                       // scalar state of the NEXT tile to load
  auto set_tap = [&]() {                            // VALU work only when the filter tap changes
#pragma unroll
    for (int j = 0; j < AP; ++j) {
      const int ih = a_ih0[j] + kh, iw = a_iw0[j] + kw;
      a_eff[j] = ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) ? a_voff[j] : kOob;
    }
  };
  f32x4 ra[AP], rb[BP];
  auto load_tile = [&](int kstep) {
    const char* base = static_cast<const char*>(p.x) + ((long)(kh - p.pad) * p.W + (kw - p.pad)) * p.x_cs * ES;
    const __amdgpu_buffer_rsrc_t rsrcA =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)kOob, 0x00020000);
#pragma unroll
    for (int j = 0; j < AP; ++j) ra[j] = buf_load16(rsrcA, a_eff[j], (unsigned)(c0 * ES));
#pragma unroll
    for (int j = 0; j < BP; ++j) rb[j] = buf_load16(rsrcB, b_voff[j], (unsigned)kstep * 128u);
  };
  auto advance = [&]() {
    c0 += BKE;
    if (c0 == p.Cin) {
      c0 = 0;
      if (++kw == p.KW) { kw = 0; ++kh; }
      set_tap();
    }
  };

  // LDS addresses (bytes).  Row r, 16-B chunk c lives at r*128 + ((c ^ ((r>>1)&7)) << 4); rows r and
  // r+32 share the swizzle, so the j / mi / ni / buffer strides are immediates.
  const int wr_off = srow * 128 + ((chunk ^ ((srow >> 1) & 7)) << 4);
  const int h = lane >> 5, l31 = lane & 31;
  int a_rd[4], b_rd[4];
  {
    const int ra_row = wm * WM + l31, rb_row = wn * WN + l31;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      a_rd[g] = ra_row * 128 + (((2 * g + h) ^ ((ra_row >> 1) & 7)) << 4);
      b_rd[g] = 2 * A_BYTES + rb_row * 128 + (((2 * g + h) ^ ((rb_row >> 1) & 7)) << 4);
    }
  }
  auto store_tile = [&](auto bufc) {
    constexpr int buf = decltype(bufc)::value;
#pragma unroll
    for (int j = 0; j < AP; ++j)
      *reinterpret_cast<f32x4*>(ldsb + wr_off + buf * A_BYTES + j * 4096) = ra[j];
#pragma unroll
    for (int j = 0; j < BP; ++j)
      *reinterpret_cast<f32x4*>(ldsb + wr_off + 2 * A_BYTES + buf * B_BYTES + j * 4096) = rb[j];
  };

  f32x16 acc[MI][NI];                                 // no zero fill: the very first MFMA of a tile takes C = 0
  auto compute = [&](auto bufc, auto firstc) {
    constexpr int buf = decltype(bufc)::value;
    constexpr bool first = decltype(firstc)::value;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x4 a[MI], b[NI], an[MI], bn[NI];
    auto rd = [&](int g, f32x4 (&fa)[MI], f32x4 (&fb)[NI]) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        fa[mi] = *reinterpret_cast<const f32x4*>(ldsb + a_rd[g] + buf * A_BYTES + mi * 4096);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        fb[ni] = *reinterpret_cast<const f32x4*>(ldsb + b_rd[g] + buf * B_BYTES + ni * 4096);
    };
    rd(0, a, b);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g < 3) rd(g + 1, an, bn);                 // fragments of the next k-group in flight under this group's MFMAs
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (kF32) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][s], b[ni][s],
                                                                 first && g == 0 && s == 0 ? zero : acc[mi][ni], 0, 0, 0);
      } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[mi]),
                                                                  __builtin_bit_cast(bf16x8, b[ni]),
                                                                  first && g == 0 ? zero : acc[mi][ni], 0, 0, 0);
      }
      if (g < 3) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[mi] = an[mi];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[ni] = bn[ni];
      }
    }
  };

  // ---- K loop: loads for step t+1 are issued before the MFMAs of step t, written to the other
  //      LDS buffer after them; one barrier per step; unrolled by two so the buffer is static ----
  const int KT = p.K / BKE;
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  set_tap();
  load_tile(0);
  store_tile(B0{});
  __syncthreads();
  using First = std::true_type;
  using Later = std::false_type;
  int kt = 0;
  if (KT >= 2) {                                    // peeled first pair: the first MFMA starts the accumulators
    advance();
    load_tile(1);
    __builtin_amdgcn_sched_barrier(0);              // keep the global loads AHEAD of the MFMAs that hide them
    compute(B0{}, First{});
    __builtin_amdgcn_sched_barrier(0);
    store_tile(B1{});
    __syncthreads();
    const bool more = 2 < KT;
    if (more) {
      advance();
      load_tile(2);
    }
    __builtin_amdgcn_sched_barrier(0);
    compute(B1{}, Later{});
    __builtin_amdgcn_sched_barrier(0);
    if (more) store_tile(B0{});
    __syncthreads();
    kt = 2;
  } else {
    compute(B0{}, First{});                         // KT == 1
    __syncthreads();
    kt = 1;
  }
  for (; kt + 2 <= KT; kt += 2) {
    advance();
    load_tile(kt + 1);
    __builtin_amdgcn_sched_barrier(0);
    compute(B0{}, Later{});
    __builtin_amdgcn_sched_barrier(0);
    store_tile(B1{});
    __syncthreads();
    const bool more = kt + 2 < KT;
    if (more) {
      advance();
      load_tile(kt + 2);
    }
    __builtin_amdgcn_sched_barrier(0);
    compute(B1{}, Later{});
    __builtin_amdgcn_sched_barrier(0);
    if (more) store_tile(B0{});
    __syncthreads();
  }
  if (kt < KT) {                                    // odd number of K steps: the last one sits in buffer 0
    compute(B0{}, Later{});
    __syncthreads();
  }

  conv_epilogue<T, BM, BN, WM, WN>(p, acc, m0, n0, m_hi, wm, wn, lane, reinterpret_cast<char*>(lds));
}

// single tile shape over all rows
template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int sid = xcd_remap(blockIdx.x, gridDim.x);
  conv_tile<T, BM, BN, WM, WN>(p, lds, 0, p.M, sid / p.tilesN, sid % p.tilesN);
}

// big tiles over rows [0, m_split) + 64x64 tiles over rows [m_split, M) in one grid
template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_hybrid(const ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if ((int)blockIdx.x < p.nbig) {
    const int sid = xcd_remap(blockIdx.x, p.nbig);
    conv_tile<T, BM, BN, WM, WN>(p, lds, 0, p.m_split, sid / p.tilesN_big, sid % p.tilesN_big);
  } else {
    const int sid = xcd_remap(blockIdx.x - p.nbig, gridDim.x - p.nbig);
    conv_tile<T, 64, 64, 32, 32>(p, lds, p.m_split, p.M, sid / p.tilesN, sid % p.tilesN);
  }
}

template <typename T, int BM, int BN, int WM, int WN>
int launch(const ConvArgs& a0, hipStream_t st) {
  ConvArgs a = a0;
  a.tilesM = (a.M + BM - 1) / BM;
  a.tilesN = (a.Cout + BN - 1) / BN;
  constexpr size_t lds_bytes = size_t(2) * (BM + BN) * BK * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    if (lds_bytes > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm<T, BM, BN, WM, WN>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_igemm<T, BM, BN, WM, WN>), dim3(a.tilesM * a.tilesN), dim3(256), lds_bytes, st, a);
  return bevf_check_launch("bevf_conv2d_nhwc_f32");
}

template <typename T, int BM, int BN, int WM, int WN>
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
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_hybrid<T, BM, BN, WM, WN>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_igemm_hybrid<T, BM, BN, WM, WN>), dim3(a.nbig + nsmall), dim3(256), lds_bytes, st, a);
  return bevf_check_launch("bevf_conv2d_nhwc_f32");
}

constexpr int kResidentBig = 512;     // 256 CUs x 2 workgroups (64-80 KB of LDS each)

// Rows that big tiles should cover so that they fill whole rounds of the chip (`resident` workgroups at once); the
// rest goes to 64x64 tiles.  Returns the number of big M-tiles (0 = all small, tilesM = all big).
static int split_big_mtiles(long long M, int BM, int tilesN_big, int resident = kResidentBig) {
  const int tilesM = (int)((M + BM - 1) / BM);
  const long long nb = (long long)tilesM * tilesN_big;
  const long long rem = nb % resident;
  if (rem == 0 || rem >= (resident * 3) / 4) return tilesM;              // last round is (nearly) full anyway
  const long long full = nb - rem;
  return (int)(full / tilesN_big);                                       // whole rounds only (may be 0)
}

}  // namespace

template <typename T>
int conv_entry(const bevf_conv_desc* d, void* stream) {
  constexpr int ES = (int)sizeof(T), BKE = 128 / ES;
  BEVF_REQUIRE(d && d->x && d->w, "conv: null x/w");
  BEVF_REQUIRE(d->y || d->colmax, "conv: neither y nor colmax given");
  BEVF_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0 && d->Cout > 0, "conv: empty shape");
  BEVF_REQUIRE(d->Cin > 0 && d->Cin % BKE == 0, "conv: Cin=%d must be a positive multiple of %d", d->Cin, BKE);
  BEVF_REQUIRE(d->x_cs >= d->Cin && d->x_cs % (16 / ES) == 0, "conv: x_cs=%d must be >= Cin and a multiple of %d", d->x_cs, 16 / ES);
  BEVF_REQUIRE(bevf_aligned16(d->x) && bevf_aligned16(d->w), "conv: x/w must be 16-byte aligned");
  BEVF_REQUIRE(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->pad >= 0, "conv: bad kernel geometry");
  BEVF_REQUIRE((d->H + 2 * d->pad - d->KH) / d->stride + 1 == d->Ho && (d->W + 2 * d->pad - d->KW) / d->stride + 1 == d->Wo,
               "conv: Ho/Wo (%d,%d) inconsistent with H,W,k,stride,pad", d->Ho, d->Wo);
  BEVF_REQUIRE(!d->y || d->y_cs >= d->Cout, "conv: y_cs=%d < Cout=%d", d->y_cs, d->Cout);
  BEVF_REQUIRE(!d->res || d->res_cs >= d->Cout, "conv: res_cs < Cout");
  BEVF_REQUIRE(!d->colmax || (d->rows_per_group > 0 && d->relu), "conv: colmax needs rows_per_group > 0 and relu");
  BEVF_REQUIRE((long long)d->N * d->H * d->W * d->x_cs * ES < (1ll << 31) &&
                   (long long)d->Cout * d->KH * d->KW * d->Cin * ES < (1ll << 31),
               "conv: input / weight buffers must stay below 2 GiB (32-bit buffer offsets)");
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
  fastdiv_make(d->Ho * d->Wo, &a.div_hw_mul, &a.div_hw_sh);
  fastdiv_make(d->Wo, &a.div_w_mul, &a.div_w_sh);
  hipStream_t st = static_cast<hipStream_t>(stream);

  switch (d->tile) {
    case 0: break;
    case 1: return launch<T, 128, 128, 64, 64>(a, st);
    case 2: return launch<T, 256, 64, 64, 64>(a, st);
    case 3: return launch<T, 128, 64, 64, 32>(a, st);
    case 4: return launch<T, 64, 64, 32, 32>(a, st);
    case 5: return launch_hybrid<T, 128, 128, 64, 64>(a, (int)(M / 128) / 2, st);     // tests: forced mid split
    case 6: return launch_hybrid<T, 256, 64, 64, 64>(a, (int)(M / 256) / 2, st);
    case 7: return launch_hybrid<T, 128, 64, 64, 32>(a, (int)(M / 128) / 2, st);
    default: bevf_set_error("conv: unknown tile variant %d", d->tile); return BEVF_ERR_ARG;
  }
  // auto: a small cost model over the tile shapes.  A CU runs `per_cu` workgroups of a shape at once (LDS-bound),
  // they share its MFMA pipes, so a round of the chip costs area * per_cu / eff and a launch costs
  // ceil(workgroups / (256 * per_cu)) rounds (measured with tools/conv_bench.py: 128x64 and 64x64 tiles at 3 and
  // 5 workgroups per CU beat 128x128 at 2 per CU on everything but the longest-K layers).
  auto wgs = [&](int bm, int bn) { return (double)((M + bm - 1) / bm) * ((d->Cout + bn - 1) / bn); };
  auto rounds = [](double n, int per_cu) { return std::ceil(n / (256.0 * per_cu)); };
  const double c64 = rounds(wgs(64, 64), 5) * 64 * 64 * 5 / 0.95;
  const double c128x64 = rounds(wgs(128, 64), 3) * 128 * 64 * 3 / 0.98;
  double c128 = 1e300;
  int big = 0;
  const int tn = (d->Cout + 127) / 128;
  if (d->Cout > 64) {
    big = split_big_mtiles(M, 128, tn);
    const long long tail_rows = M - (long long)big * 128;
    const double tail = tail_rows > 0 ? rounds((double)((tail_rows + 63) / 64) * ((d->Cout + 63) / 64), 5) * 64 * 64 * 5 / 0.95 : 0.0;
    c128 = big > 0 ? rounds((double)big * tn, 2) * 128 * 128 * 2 + tail : 1e300;
  }
  // bf16: the K loop is bound by LDS reads, not by the MFMA pipe, so the shapes with 64x64 wave tiles (twice the MFMAs per
  // fragment read) win by more than their occupancy costs them (tools/conv_bench.py ... bf16: 128x128 +4..15 % on the
  // Cout >= 128 layers, 256x64 +25 % on Cout = 64)
  if (sizeof(T) == 2) {
    c128 *= 0.85;
    if (d->Cout <= 64 && rounds(wgs(256, 64), 2) * 256 * 64 * 2 * 0.8 <= (c128x64 < c64 ? c128x64 : c64))
      return launch<T, 256, 64, 64, 64>(a, st);
  }
  if (c128 <= c128x64 && c128 <= c64) {
    if (big == (int)((M + 127) / 128)) return launch<T, 128, 128, 64, 64>(a, st);
    return launch_hybrid<T, 128, 128, 64, 64>(a, big, st);
  }
  // same wave-quantisation fix for the 128x64 shape (3 workgroups per CU = 768 at once): matters for small M
  // (B = 1: layer 1 is 1055 tiles = 1.37 rounds; the tail as 64x64 tiles costs half a round instead of a whole one)
  const int tn64 = (d->Cout + 63) / 64;
  const int big64 = split_big_mtiles(M, 128, tn64, 768);
  double hyb64 = 1e300;
  if (big64 > 0 && big64 < (int)((M + 127) / 128)) {
    const long long tail_rows = M - (long long)big64 * 128;
    hyb64 = rounds((double)big64 * tn64, 3) * 128 * 64 * 3 / 0.98 +
            rounds((double)((tail_rows + 63) / 64) * tn64, 3) * 64 * 64 * 3 / 0.95;
  }
  if (hyb64 < c128x64 && hyb64 < c64) return launch_hybrid<T, 128, 64, 64, 32>(a, big64, st);
  if (c128x64 <= c64) return launch<T, 128, 64, 64, 32>(a, st);
  return launch<T, 64, 64, 32, 32>(a, st);
}

extern "C" int bevf_conv2d_nhwc_f32(const bevf_conv_desc* d, void* stream) { return conv_entry<float>(d, stream); }
extern "C" int bevf_conv2d_nhwc_bf16(const bevf_conv_desc* d, void* stream) { return conv_entry<__bf16>(d, stream); }
