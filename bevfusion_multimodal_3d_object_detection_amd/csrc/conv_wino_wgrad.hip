// Weight gradient of the 3x3 / stride 1 / pad 1 NHWC convolution in the Winograd F(2x2,3x3) domain, fp32 on
// v_mfma_f32_16x16x4_f32 (training; the forward is conv_wino.hip).
//
//   dW = G^T [ sum over 2x2 output tiles of (A dY A^T) .* (B^T X B) ] G          (the transpose of the forward algorithm)
//
// per (output channel, input channel): the 4x4 input patch of a tile and its 2x2 block of dY are transformed to 16
// "frequencies", the 16 products are summed over all tiles of the batch, and the sum goes back to the 9 taps.  16
// multiplies per tile instead of 36: 2.25x fewer MFMA FLOPs than the pixel-GEMM of conv_wgrad.hip, which the fp32
// matrix pipe bounds.  The products are fp32 FMAs with fp32 accumulation like the direct kernel; the summation
// structure differs, so results agree with it to a few 1e-7 relative of the accumulated magnitude, not bit for bit.
//
// Workgroup = 4 waves = one (64 output channels x 64 input channels) block of dW over a contiguous range of tiles.
// Wave w owns frequency row fr = w (4 frequencies) of the whole 64x64 block: 4 x 16 accumulator tiles of 16x16 = all
// 256 AGPRs (one wave per SIMD), and NOTHING is shared between the waves: no LDS, no barrier.  The GEMM reduction
// index is the tile: one MFMA k-step = 4 tiles (lane group k = lane>>4), and lane (k, q = lane&15) loads, straight from
// HBM/L2 into registers, the 16 bytes (channels 4q..4q+3 of the block) of each pixel it needs of tile k:
//   X : rows rA, rB of the 4x4 patch (frequency row fr of B^T X combines exactly two rows) x 4 columns   8 x b128
//   dY: both rows x 2 columns (a row the frequency row does not use is answered with zeros by the buffer unit)  4 x b128
// Out-of-image pixels (the pad ring, the odd last row / column, tiles past the end) are out-of-range offsets of the
// buffer descriptor and read as 0.  Component e of a lane's 16 bytes is channel 4q + e: MFMA block e of the operand
// therefore holds channels {4q + e}, a permutation that the store of the partial sums undoes for free.
// Per k-step a wave issues 64 MFMAs (2048 cycles of the matrix pipe) against 12 loads and ~100 VALU instructions, all
// pinned one small piece between two MFMAs (sched_barrier) as in conv_wino.hip; operands are computed one step ahead,
// loads run two steps ahead.
// The tile range is split over 256 / (block pairs) workgroups; each writes its partial [16 f][64][64] block to a
// workspace and wino_wgrad_reduce sums the splits in a fixed order (deterministic, unlike the atomics of
// conv_wgrad.hip), applies G^T . G and writes (or accumulates into) dW [Cout][3][3][Cin].
#include "conv_common.h"

#include <type_traits>

namespace {

#define MFMA(acc_, a_, b_) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc_) : "v"(a_), "v"(b_))

struct WwArgs {
  const float* x;      // NHWC, channel stride x_cs
  const float* dy;     // NHWC, channel stride dy_cs
  float* part;         // [splits][16 f][Cout][Cin]
  int H, W, x_cs, dy_cs, Cin, Cout;
  int tilesX, tilesY, T;            // tiles per row, rows of tiles per image, tiles in the batch
  unsigned mX, mY;                  // floor(2^32 / tilesX) + 1, floor(2^32 / tilesY) + 1: exact umulhi division below 2^32 / d
  int steps, steps_per_split;       // k-steps of 4 tiles
  int nci, nco;                     // 64-channel blocks
};

template <int C> using ic = std::integral_constant<int, C>;

__global__ __launch_bounds__(256, 1) void wino_wgrad_f32(const WwArgs p) {
  const int lane = threadIdx.x & 63, k = lane >> 4, q = lane & 15;
  const int fr = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int npairs = p.nci * p.nco;
  const int split = blockIdx.x / npairs, pair = blockIdx.x - split * npairs;
  const int co0 = (pair / p.nci) * 64, ci0 = (pair % p.nci) * 64;
  const int s_begin = split * p.steps_per_split;
  const int s_end = s_begin + p.steps_per_split < p.steps ? s_begin + p.steps_per_split : p.steps;
  // frequency row fr of B^T X = x[rA] + sB * x[rB];  of A dY = cA * dy[0] + cB * dy[1]
  const int rA = fr == 0 ? 0 : (fr == 2 ? 2 : 1);
  const int rB = fr == 2 ? 1 : (fr == 3 ? 3 : 2);
  const float sB = fr == 1 ? 1.f : -1.f;
  const float cA = fr == 3 ? 0.f : 1.f, cB = fr == 0 ? 0.f : (fr == 1 ? 1.f : -1.f);
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + ci0), 0, (int)kOob, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy + co0), 0, (int)kOob, 0x00020000);
  const int xs4 = p.x_cs * 4, ys4 = p.dy_cs * 4, q16 = q * 16;

  f32x4 acc[4][4][4];                                 // [fc][co block][ci block]
  f32x4 rx[2][2][4], ry[2][2][2];                     // raw loads [set][row][col]
  f32x4 ao[2][4], bo[2][4];                           // MFMA operands [set][fc], component = channel block
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[a][b][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- address generation of one k-step: byte offsets of the lane's tile, kOob where the pixel does not exist -----
  unsigned t_, row_;
  int tx_, ty_, n_, xb_, yb_;
  bool okA_, okB_, ok0_, ok1_;
  // (PIN: an empty asm that "rewrites" a value -- hipcc cannot move the arithmetic consuming it above this point, so
  //  each piece stays in the MFMA gap it was written in; sched_barrier alone only binds the machine scheduler)
#define PIN(v_) asm volatile("" : "+v"(v_))
  auto addr_a = [&](int s) {
    t_ = 4u * (unsigned)s + (unsigned)k;
    PIN(t_);
    row_ = __umulhi(t_, p.mX);                        // n * tilesY + ty
    tx_ = (int)(t_ - row_ * (unsigned)p.tilesX);
    PIN(tx_);
  };
  auto addr_b = [&]() {
    PIN(row_);
    n_ = (int)__umulhi(row_, p.mY);
    ty_ = (int)row_ - n_ * p.tilesY;
    PIN(ty_);
    PIN(n_);
  };
  auto addr_c = [&](int s) {
    PIN(ty_);
    const bool live = s < s_end && t_ < (unsigned)p.T;
    const int y0 = 2 * ty_ - 1;
    okA_ = live && (unsigned)(y0 + rA) < (unsigned)p.H;
    okB_ = live && (unsigned)(y0 + rB) < (unsigned)p.H;
    ok0_ = live && fr != 3;
    ok1_ = live && fr != 0 && 2 * ty_ + 1 < p.H;
  };
  auto addr_d = [&]() {
    PIN(tx_);
    const int y0 = 2 * ty_ - 1, x0 = 2 * tx_ - 1;
    xb_ = ((n_ * p.H + y0) * p.W + x0) * xs4 + q16;
    yb_ = ((n_ * p.H + 2 * ty_) * p.W + 2 * tx_) * ys4 + q16;
    PIN(xb_);
    PIN(yb_);
  };
  auto load_x = [&](auto setc, auto rc, auto cc) {
    constexpr int set = decltype(setc)::value, r = decltype(rc)::value, c = decltype(cc)::value;
    const int rr = r ? rB : rA;
    const bool ok = (r ? okB_ : okA_) && (unsigned)(2 * tx_ - 1 + c) < (unsigned)p.W;
    rx[set][r][c] = buf_load16(rsx, ok ? (unsigned)(xb_ + (rr * p.W + c) * xs4) : kOob, 0);
  };
  auto load_y = [&](auto setc, auto rc, auto cc) {
    constexpr int set = decltype(setc)::value, r = decltype(rc)::value, c = decltype(cc)::value;
    const bool ok = (r ? ok1_ : ok0_) && (c == 0 || 2 * tx_ + 1 < p.W);
    ry[set][r][c] = buf_load16(rsy, ok ? (unsigned)(yb_ + (r * p.W + c) * ys4) : kOob, 0);
  };
  // ---- transforms: raw set -> operand set ---------------------------------------------------------------------------
  f32x4 v_[4], r_[2];
  auto xf_row = [&](auto setc, auto cc) {             // v[c] = x[rA][c] + sB x[rB][c]
    constexpr int set = decltype(setc)::value, c = decltype(cc)::value;
    PIN(rx[set][0][c]);
    PIN(rx[set][1][c]);
#pragma unroll
    for (int e = 0; e < 4; ++e) v_[c][e] = fmaf(sB, rx[set][1][c][e], rx[set][0][c][e]);
    PIN(v_[c]);
  };
  auto xf_col = [&](auto setc, auto fc_) {            // B^T applied along the columns
    constexpr int set = decltype(setc)::value, fc = decltype(fc_)::value;
    if constexpr (fc == 0) { PIN(v_[0]); bo[set][0] = v_[0] - v_[2]; PIN(bo[set][0]); }
    if constexpr (fc == 1) { PIN(v_[2]); bo[set][1] = v_[1] + v_[2]; PIN(bo[set][1]); }
    if constexpr (fc == 2) { PIN(v_[1]); bo[set][2] = v_[2] - v_[1]; PIN(bo[set][2]); }
    if constexpr (fc == 3) { PIN(v_[3]); bo[set][3] = v_[1] - v_[3]; PIN(bo[set][3]); }
  };
  auto yf_row = [&](auto setc, auto cc) {             // r[c] = cA dy[0][c] + cB dy[1][c]
    constexpr int set = decltype(setc)::value, c = decltype(cc)::value;
    PIN(ry[set][0][c]);
    PIN(ry[set][1][c]);
#pragma unroll
    for (int e = 0; e < 4; ++e) r_[c][e] = fmaf(cB, ry[set][1][c][e], cA * ry[set][0][c][e]);
    PIN(r_[c]);
  };
  auto yf_col = [&](auto setc, auto fc_) {            // A applied along the columns: r0, r0 + r1, r0 - r1, -r1
    constexpr int set = decltype(setc)::value, fc = decltype(fc_)::value;
    if constexpr (fc == 0) { PIN(r_[0]); ao[set][0] = r_[0]; PIN(ao[set][0]); }
    if constexpr (fc == 1) { PIN(r_[1]); ao[set][1] = r_[0] + r_[1]; PIN(ao[set][1]); }
    if constexpr (fc == 2) { PIN(r_[0]); ao[set][2] = r_[0] - r_[1]; PIN(ao[set][2]); }
    if constexpr (fc == 3) { PIN(r_[1]); ao[set][3] = -r_[1]; PIN(ao[set][3]); }
  };
  auto load_all = [&](auto setc, int s) {
    addr_a(s);
    addr_b();
    addr_c(s);
    addr_d();
    load_x(setc, ic<0>{}, ic<0>{}); load_x(setc, ic<0>{}, ic<1>{}); load_x(setc, ic<0>{}, ic<2>{}); load_x(setc, ic<0>{}, ic<3>{});
    load_x(setc, ic<1>{}, ic<0>{}); load_x(setc, ic<1>{}, ic<1>{}); load_x(setc, ic<1>{}, ic<2>{}); load_x(setc, ic<1>{}, ic<3>{});
    load_y(setc, ic<0>{}, ic<0>{}); load_y(setc, ic<0>{}, ic<1>{}); load_y(setc, ic<1>{}, ic<0>{}); load_y(setc, ic<1>{}, ic<1>{});
  };
  auto transform_all = [&](auto setc) {
    xf_row(setc, ic<0>{}); xf_row(setc, ic<1>{}); xf_row(setc, ic<2>{}); xf_row(setc, ic<3>{});
    xf_col(setc, ic<0>{}); xf_col(setc, ic<1>{}); xf_col(setc, ic<2>{}); xf_col(setc, ic<3>{});
    yf_row(setc, ic<0>{}); yf_row(setc, ic<1>{});
    yf_col(setc, ic<0>{}); yf_col(setc, ic<1>{}); yf_col(setc, ic<2>{}); yf_col(setc, ic<3>{});
  };

  // ---- one k-step: 64 MFMAs on operand set CUR; in their shadow the loads of step s+2 (into raw set CUR, whose data
  //      became operand set CUR during step s-1) and the transform of raw set CUR^1 (step s+1) into operand set CUR^1 ---
  auto step = [&](auto curc, int s) {
    constexpr int cur = decltype(curc)::value, nxt = cur ^ 1;
    const ic<cur> CS{};
    const ic<nxt> NS{};
    asm volatile("s_nop 1");                          // (any accumulator copy hipcc leaves at the loop head is clear of the first MFMA)
#pragma unroll
    for (int fc = 0; fc < 4; ++fc) {
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
          const int m = fc * 16 + cb * 4 + ib;
          MFMA(acc[fc][cb][ib], ao[cur][fc][cb], bo[cur][fc][ib]);
          if (m == 0) addr_a(s + 2);
          if (m == 1) addr_b();
          if (m == 2) addr_c(s + 2);
          if (m == 3) addr_d();
          if (m == 4) load_x(CS, ic<0>{}, ic<0>{});
          if (m == 5) load_x(CS, ic<0>{}, ic<1>{});
          if (m == 6) load_x(CS, ic<0>{}, ic<2>{});
          if (m == 7) load_x(CS, ic<0>{}, ic<3>{});
          if (m == 8) load_x(CS, ic<1>{}, ic<0>{});
          if (m == 9) load_x(CS, ic<1>{}, ic<1>{});
          if (m == 10) load_x(CS, ic<1>{}, ic<2>{});
          if (m == 11) load_x(CS, ic<1>{}, ic<3>{});
          if (m == 12) load_y(CS, ic<0>{}, ic<0>{});
          if (m == 13) load_y(CS, ic<0>{}, ic<1>{});
          if (m == 14) load_y(CS, ic<1>{}, ic<0>{});
          if (m == 15) load_y(CS, ic<1>{}, ic<1>{});
          if (m == 32) xf_row(NS, ic<0>{});
          if (m == 33) xf_row(NS, ic<1>{});
          if (m == 34) xf_row(NS, ic<2>{});
          if (m == 35) xf_row(NS, ic<3>{});
          if (m == 36) xf_col(NS, ic<0>{});
          if (m == 37) xf_col(NS, ic<1>{});
          if (m == 38) xf_col(NS, ic<2>{});
          if (m == 39) xf_col(NS, ic<3>{});
          if (m == 40) yf_row(NS, ic<0>{});
          if (m == 41) yf_row(NS, ic<1>{});
          if (m == 42) yf_col(NS, ic<0>{});
          if (m == 43) yf_col(NS, ic<1>{});
          if (m == 44) yf_col(NS, ic<2>{});
          if (m == 45) yf_col(NS, ic<3>{});
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };

  load_all(ic<0>{}, s_begin);
  load_all(ic<1>{}, s_begin + 1);
  transform_all(ic<0>{});
  asm volatile("s_nop 7");                            // accumulator zeros / first operands written by the VALU: keep clear of the MFMA
  for (int s = s_begin; s < s_end; s += 2) {          // an odd count runs one dead step: its loads are all out of range, it adds zeros
    step(ic<0>{}, s);
    step(ic<1>{}, s + 1);
  }
  asm volatile("s_nop 15\n\ts_nop 15");               // last MFMA results land before the accumulators are read

  // ---- partial block: acc[fc][cb][ib][r] is dU[f = 4 fr + fc][co = co0 + 4 (4k + r) + cb][ci = ci0 + 4q + ib] ------
  float* const pb = p.part + ((size_t)split * 16 + fr * 4) * p.Cout * p.Cin + (size_t)(co0 + 16 * k) * p.Cin + ci0 + 4 * q;
#pragma unroll
  for (int fc = 0; fc < 4; ++fc)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x4 o = {acc[fc][cb][0][r], acc[fc][cb][1][r], acc[fc][cb][2][r], acc[fc][cb][3][r]};
        *reinterpret_cast<f32x4*>(pb + ((size_t)fc * p.Cout + 4 * r + cb) * p.Cin) = o;
      }
}

// Sum of the per-workgroup partial blocks, in a fixed order (deterministic).  Two levels so that the 67 MB of partials of a
// 64x64 layer are read by the whole chip: fold adds groups of FOLD consecutive partials (one thread per 16 bytes of the
// block, its FOLD loads in flight together), the final kernel adds the <= FOLD group sums and applies G^T . G:
//   dW[co][a][b][ci] (+)= sum_{f,g} G[f][a] G[g][b] U[4f+g][co][ci],   G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
constexpr int FOLD = 16;

__global__ __launch_bounds__(256) void wino_wgrad_fold(const f32x4* __restrict__ part, f32x4* __restrict__ out, int splits,
                                                        int elems4) {
  const int e = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
  if (e >= elems4) return;
  const int s0 = g * FOLD, n = splits - s0 < FOLD ? splits - s0 : FOLD;
  const f32x4* src = part + (size_t)s0 * elems4 + e;
  f32x4 v[FOLD];
#pragma unroll
  for (int i = 0; i < FOLD; ++i) v[i] = i < n ? src[(size_t)i * elems4] : f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 s = v[0];
#pragma unroll
  for (int i = 1; i < FOLD; ++i) s += v[i];
  out[(size_t)g * elems4 + e] = s;
}

// block = one co x 64 ci; thread (f = tid>>4, c4 = tid&15) sums frequency f of 4 channels over the partials, then 144
// threads (tap, c4) apply the transform
__global__ __launch_bounds__(256) void wino_wgrad_final(const float* __restrict__ part, float* __restrict__ dw, int splits,
                                                         int Cout, int Cin, int accumulate) {
  __shared__ f32x4 us[16][16];
  const int tid = threadIdx.x, c4 = tid & 15, f = tid >> 4;
  const int nci = Cin >> 6;
  const int co = blockIdx.x / nci, ci0 = (blockIdx.x - co * nci) * 64;
  const size_t fstride = (size_t)Cout * Cin;
  const float* src = part + (size_t)f * fstride + (size_t)co * Cin + ci0 + 4 * c4;
  f32x4 v[FOLD];
#pragma unroll
  for (int i = 0; i < FOLD; ++i) v[i] = i < splits ? *reinterpret_cast<const f32x4*>(src + (size_t)i * 16 * fstride) : f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 s = v[0];
#pragma unroll
  for (int i = 1; i < FOLD; ++i) s += v[i];
  us[f][c4] = s;
  __syncthreads();
  if (tid < 144) {
    const int tap = tid >> 4, a = tap / 3, b = tap - 3 * a;        // c4 = tid & 15 as above
    // column b of G per g, column a of G per f
    const float gb[4] = {b == 0 ? 1.f : 0.f, 0.5f, b == 1 ? -0.5f : 0.5f, b == 2 ? 1.f : 0.f};
    const float ga[4] = {a == 0 ? 1.f : 0.f, 0.5f, a == 1 ? -0.5f : 0.5f, a == 2 ? 1.f : 0.f};
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ff = 0; ff < 4; ++ff) {
      f32x4 h = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < 4; ++g) h += gb[g] * us[4 * ff + g][c4];
      o += ga[ff] * h;
    }
    f32x4* d = reinterpret_cast<f32x4*>(dw + ((size_t)co * 9 + tap) * Cin + ci0 + 4 * c4);
    if (accumulate) *d += o;
    else *d = o;
  }
}

struct WwPlan {
  int tilesX, tilesY, T, steps, splits, steps_per_split, nci, nco;
};

// the tile range is cut into as many pieces as fill the chip once (one 512-register workgroup per CU)
bool ww_plan(int N, int H, int W, int Cin, int Cout, WwPlan* pl) {
  if (N <= 0 || H < 3 || W < 3 || Cin <= 0 || Cout <= 0 || Cin % 64 || Cout % 64) return false;
  pl->tilesX = (W + 1) / 2;
  pl->tilesY = (H + 1) / 2;
  const long long T = (long long)N * pl->tilesY * pl->tilesX;
  if ((T + 8) * pl->tilesX >= (1ll << 32) || T + 8 >= (1ll << 30)) return false;     // umulhi division stays exact
  pl->T = (int)T;
  pl->steps = (int)((T + 3) / 4);
  pl->nci = Cin / 64;
  pl->nco = Cout / 64;
  int splits = 256 / (pl->nci * pl->nco);
  if (splits < 1) splits = 1;
  if (splits > (pl->steps + 1) / 2) splits = (pl->steps + 1) / 2;
  int sps = (pl->steps + splits - 1) / splits;
  sps += sps & 1;                                     // even: the kernel runs steps in pairs
  pl->steps_per_split = sps;
  pl->splits = (pl->steps + sps - 1) / sps;
  return true;
}

}  // namespace

extern "C" size_t bevf_wino_wgrad_workspace_floats(int N, int H, int W, int Cin, int Cout) {
  WwPlan pl;
  if (!ww_plan(N, H, W, Cin, Cout, &pl)) return 0;
  const size_t groups = pl.splits > FOLD ? (size_t)(pl.splits + FOLD - 1) / FOLD : 0;
  return ((size_t)pl.splits + groups) * 16 * Cout * Cin;
}

extern "C" int bevf_conv3x3_wgrad_wino_f32(const bevf_wgrad_desc* d, float* workspace, int accumulate, void* stream) {
  BEVF_REQUIRE(d && d->x && d->dy && d->dw && workspace, "wino wgrad: null pointer");
  BEVF_REQUIRE(d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1, "wino wgrad: 3x3 / stride 1 / pad 1 only");
  WwPlan pl;
  BEVF_REQUIRE(ww_plan(d->N, d->H, d->W, d->Cin, d->Cout, &pl),
               "wino wgrad: unsupported shape N=%d H=%d W=%d Cin=%d Cout=%d (channels in multiples of 64, H, W >= 3; "
               "bevf_wino_wgrad_workspace_floats returns 0 for these)", d->N, d->H, d->W, d->Cin, d->Cout);
  BEVF_REQUIRE(d->x_cs >= d->Cin && d->x_cs % 4 == 0 && d->dy_cs >= d->Cout && d->dy_cs % 4 == 0,
               "wino wgrad: channel strides must be multiples of 4 and cover the channels");
  BEVF_REQUIRE(bevf_aligned16(d->x) && bevf_aligned16(d->dy) && bevf_aligned16(d->dw) && bevf_aligned16(workspace),
               "wino wgrad: unaligned");
  BEVF_REQUIRE((long long)d->N * d->H * d->W * d->x_cs * 4 < (1ll << 31) && (long long)d->N * d->H * d->W * d->dy_cs * 4 < (1ll << 31),
               "wino wgrad: x / dy buffers must stay below 2 GiB (32-bit buffer offsets)");
  WwArgs a;
  a.x = d->x; a.dy = d->dy; a.part = workspace;
  a.H = d->H; a.W = d->W; a.x_cs = d->x_cs; a.dy_cs = d->dy_cs; a.Cin = d->Cin; a.Cout = d->Cout;
  a.tilesX = pl.tilesX; a.tilesY = pl.tilesY; a.T = pl.T;
  a.mX = (unsigned)((1ull << 32) / (unsigned)pl.tilesX) + 1u;
  a.mY = (unsigned)((1ull << 32) / (unsigned)pl.tilesY) + 1u;
  a.steps = pl.steps; a.steps_per_split = pl.steps_per_split; a.nci = pl.nci; a.nco = pl.nco;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(wino_wgrad_f32, dim3(pl.splits * pl.nci * pl.nco), dim3(256), 0, st, a);
  int rc = bevf_check_launch("bevf_conv3x3_wgrad_wino_f32");
  if (rc != BEVF_OK) return rc;
  const float* sums = workspace;
  int nsum = pl.splits;
  if (pl.splits > FOLD) {                              // partials [splits] -> group sums [groups] behind them in the workspace
    const int groups = (pl.splits + FOLD - 1) / FOLD, elems4 = 4 * d->Cout * d->Cin;
    float* folded = workspace + (size_t)pl.splits * 16 * d->Cout * d->Cin;
    hipLaunchKernelGGL(wino_wgrad_fold, dim3((elems4 + 255) / 256, groups), dim3(256), 0, st,
                       reinterpret_cast<const f32x4*>(workspace), reinterpret_cast<f32x4*>(folded), pl.splits, elems4);
    sums = folded;
    nsum = groups;
  }
  hipLaunchKernelGGL(wino_wgrad_final, dim3(d->Cout * pl.nci), dim3(256), 0, st, sums, d->dw, nsum, d->Cout, d->Cin, accumulate);
  return bevf_check_launch("bevf_conv3x3_wgrad_wino_f32 (reduce)");
}
