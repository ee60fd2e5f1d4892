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
// Component e of a lane's 16 bytes is channel 4q + e: MFMA block e of the operand therefore holds channels {4q + e}, a
// permutation that the store of the partial sums undoes for free.
// The fp32 MFMA shares the vector ALU (a pure-MFMA loop runs at 96 % of peak; every VALU instruction added to it costs its
// own issue time on top), so the loop is built to need almost none:
//   * pixel addresses come from a per-shape TABLE (bevf_wino_wgrad_table: per tile the 16 X and 3 x 4 dY byte offsets,
//     0x80000000 = "does not exist" for the pad ring, the odd last row / column and tiles past the end -- the buffer unit
//     answers those with 0); a lane fetches its tile's entries two steps ahead (3 x b128) and adds its channel offset: 12
//     v_add per step, no division, no compare, no select;
//   * the transforms are packed-fp32 inline asm (v_pk_fma_f32 / v_pk_add_f32 with neg modifiers): 24 per step;
//   * everything else (table / data loads, loop control) is VMEM / SALU.
// Per k-step a wave issues 64 MFMAs (2048 cycles of the matrix pipe) against 36 VALU instructions, each pinned in its
// own MFMA gap (asm volatile keeps program order; sched_barrier binds the loads).  Operands are computed one step
// ahead, pixels are loaded two steps ahead, table entries three.
// The tile range is split over 256 / (block pairs) workgroups; each writes its partial [16 f][64][64] block to a
// workspace, summed in a fixed order (deterministic, unlike the atomics of conv_wgrad.hip) by the two kernels below,
// which also apply G^T . G and write (or accumulate into) dW [Cout][3][3][Cin].
#include "conv_common.h"

#include <type_traits>

namespace {

#define MFMA(acc_, a_, b_) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc_) : "v"(a_), "v"(b_))

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// packed fp32 VALU as volatile asm: stays where it is written between the (volatile asm) MFMAs
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
  f32x2 d;
  asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
  f32x2 d;
  asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ unsigned v_add(int a, int b) {
  unsigned d;
  asm volatile("v_add_u32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2 lo2(f32x4 v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ f32x2 hi2(f32x4 v) { return __builtin_shufflevector(v, v, 2, 3); }

// table entry of one tile: 28 dwords
//   [4 r + c]      X byte offset of patch pixel (r, c), r, c = 0..3 (patch origin (2 ty - 1, 2 tx - 1))
//   [16 + 2 r + c] dY byte offset of pixel (r, c) of the tile's 2x2 block
//   [20 + ..]      the same with row 1 removed (frequency row 0 of A dY = dY row 0)
//   [24 + ..]      the same with row 0 removed (frequency row 3 = - dY row 1)
constexpr int TE = 28, TE_BYTES = TE * 4;

struct WwArgs {
  const float* x;      // NHWC, channel stride x_cs
  const float* dy;     // NHWC, channel stride dy_cs
  const int* tab;      // [4 (steps + 4)][TE]
  float* part;         // [splits][16 f][Cout][Cin]
  int Cin, Cout;
  int steps, steps_per_split;       // k-steps of 4 tiles
  int nci, nco;                     // 64-channel blocks
};

__global__ __launch_bounds__(256) void wino_wgrad_table(int* __restrict__ tab, int H, int W, int tilesX, int tilesY, int T,
                                                         int Tpad, int xs4, int ys4) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= Tpad) return;
  int e[TE];
#pragma unroll
  for (int i = 0; i < TE; ++i) e[i] = (int)kOob;
  if (t < T) {
    const int n = t / (tilesX * tilesY), rem = t - n * tilesX * tilesY, ty = rem / tilesX, tx = rem - ty * tilesX;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int y = 2 * ty - 1 + r, x = 2 * tx - 1 + c;
        if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) e[4 * r + c] = ((n * H + y) * W + x) * xs4;
      }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int y = 2 * ty + r, x = 2 * tx + c;
        if (y < H && x < W) {
          const int o = ((n * H + y) * W + x) * ys4;
          e[16 + 2 * r + c] = o;
          if (r == 0) e[20 + c] = o;
          else e[26 + c] = o;
        }
      }
  }
  i32x4* dst = reinterpret_cast<i32x4*>(tab + (size_t)t * TE);
#pragma unroll
  for (int i = 0; i < TE / 4; ++i) dst[i] = i32x4{e[4 * i], e[4 * i + 1], e[4 * i + 2], e[4 * i + 3]};
}

template <int C> using ic = std::integral_constant<int, C>;

__global__ __launch_bounds__(256, 1) void wino_wgrad_f32(const WwArgs p) {
  const int lane = threadIdx.x & 63, k = lane >> 4, q = lane & 15;
  const int fr = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int npairs = p.nci * p.nco;
  const int split = blockIdx.x / npairs, pair = blockIdx.x - split * npairs;
  const int co0 = (pair / p.nci) * 64, ci0 = (pair % p.nci) * 64;
  const int s_begin = split * p.steps_per_split;
  const int s_end = s_begin + p.steps_per_split < p.steps ? s_begin + p.steps_per_split : p.steps;
  // frequency row fr of B^T X = x[rA] + sB * x[rB];  of A dY = dy[0] + cB * dy[1] (the table removes the row a frequency
  // row does not contain: fr 0 reads dy[1] = 0, fr 3 reads dy[0] = 0)
  const int rA = fr == 0 ? 0 : (fr == 2 ? 2 : 1);
  const int rB = fr == 2 ? 1 : (fr == 3 ? 3 : 2);
  const float sBs = fr == 1 ? 1.f : -1.f, cBs = fr == 1 ? 1.f : -1.f;
  const f32x2 sB = {sBs, sBs}, cB = {cBs, cBs};
  const int selA = rA * 16, selB = rB * 16, selY = fr == 0 ? 80 : (fr == 3 ? 96 : 64);      // byte offsets inside a table entry
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + ci0), 0, (int)kOob, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy + co0), 0, (int)kOob, 0x00020000);
  const __amdgpu_buffer_rsrc_t rst =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(p.tab), 0, 4 * (p.steps + 4) * TE_BYTES, 0x00020000);
  const int q16 = q * 16, ktab = k * TE_BYTES;

  f32x4 acc[4][4][4];                                 // [fc][co block][ci block]
  f32x4 rx[2][2][4], ry[2][2][2];                     // raw pixels [set][row][col]
  i32x4 tab[2][3];                                    // table entries [set][row A, row B, dY] of the steps whose pixels are loaded next
  // MFMA operands [set]: B side bo[fc][half]; A side fc 0 = r[0], fc 1 = a1, fc 2 = a2, fc 3 = r[1] (sign folded into the final sum)
  f32x2 bo[2][4][2], rr[2][2][2], a1[2][2], a2[2][2], v_[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[a][b][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto load_tab = [&](auto setc, int s, auto wc) {    // entries of tile-step s
    constexpr int set = decltype(setc)::value, which = decltype(wc)::value;
    const int so = s * 4 * TE_BYTES + (which == 0 ? selA : which == 1 ? selB : selY);
    tab[set][which] = __builtin_bit_cast(i32x4, buf_load16(rst, ktab, so));
  };
#ifndef WGRAD_CLUSTER
#define WGRAD_CLUSTER 1
#endif
  // A lone VALU instruction between two fp32 MFMAs costs ~15 cycles, the ones right behind it ~3 (probe in conv_wino.hip / DESIGN 3.1b):
  // the twelve address adds of a step are ONE cluster (px_addr, then twelve plain loads in the following gaps), the 24 transform
  // instructions another -- 2 interruptions per 64 MFMAs instead of 36.
  int pxa[12];
  auto px_addr = [&](auto tsetc) {
    constexpr int ts = decltype(tsetc)::value;
#pragma unroll
    for (int i = 0; i < 12; ++i) pxa[i] = v_add(tab[ts][i >> 2][i & 3], q16);
  };
  auto load_px = [&](auto setc, auto tsetc, auto ic_) {   // pixel i of table set tset: 0..3 row A, 4..7 row B, 8..11 dY
    constexpr int set = decltype(setc)::value, ts = decltype(tsetc)::value, i = decltype(ic_)::value;
    if constexpr (WGRAD_CLUSTER) {
      if constexpr (i < 4) rx[set][0][i] = buf_load16(rsx, pxa[i], 0);
      else if constexpr (i < 8) rx[set][1][i - 4] = buf_load16(rsx, pxa[i], 0);
      else ry[set][(i - 8) >> 1][(i - 8) & 1] = buf_load16(rsy, pxa[i], 0);
    } else {
      if constexpr (i < 4) rx[set][0][i] = buf_load16(rsx, v_add(tab[ts][0][i], q16), 0);
      else if constexpr (i < 8) rx[set][1][i - 4] = buf_load16(rsx, v_add(tab[ts][1][i - 4], q16), 0);
      else ry[set][(i - 8) >> 1][(i - 8) & 1] = buf_load16(rsy, v_add(tab[ts][2][i - 8], q16), 0);
    }
  };
  auto xform = [&](auto setc, auto jc) {              // transform step j of raw set -> operand set, one packed instruction each
    constexpr int set = decltype(setc)::value, j = decltype(jc)::value;
    if constexpr (j < 8) {                            // rows of B^T X: v[c] = x[rA][c] + sB x[rB][c]
      constexpr int c = j >> 1, h = j & 1;
      v_[c][h] = pk_fma(sB, h ? hi2(rx[set][1][c]) : lo2(rx[set][1][c]), h ? hi2(rx[set][0][c]) : lo2(rx[set][0][c]));
    } else if constexpr (j < 16) {                    // columns: v0 - v2, v1 + v2, v2 - v1, v1 - v3
      constexpr int fc = (j - 8) >> 1, h = j & 1;
      if constexpr (fc == 0) bo[set][0][h] = pk_sub(v_[0][h], v_[2][h]);
      if constexpr (fc == 1) bo[set][1][h] = pk_add(v_[1][h], v_[2][h]);
      if constexpr (fc == 2) bo[set][2][h] = pk_sub(v_[2][h], v_[1][h]);
      if constexpr (fc == 3) bo[set][3][h] = pk_sub(v_[1][h], v_[3][h]);
    } else if constexpr (j < 20) {                    // rows of A dY: r[c] = dy[0][c] + cB dy[1][c]
      constexpr int c = (j - 16) >> 1, h = j & 1;
      rr[set][c][h] = pk_fma(cB, h ? hi2(ry[set][1][c]) : lo2(ry[set][1][c]), h ? hi2(ry[set][0][c]) : lo2(ry[set][0][c]));
    } else {                                          // columns: r0, r0 + r1, r0 - r1, (-) r1
      constexpr int h = j & 1;
      if constexpr (j < 22) a1[set][h] = pk_add(rr[set][0][h], rr[set][1][h]);
      else a2[set][h] = pk_sub(rr[set][0][h], rr[set][1][h]);
    }
  };
  auto xform_all = [&](auto setc) {
    xform(setc, ic<0>{}); xform(setc, ic<1>{}); xform(setc, ic<2>{}); xform(setc, ic<3>{}); xform(setc, ic<4>{}); xform(setc, ic<5>{});
    xform(setc, ic<6>{}); xform(setc, ic<7>{}); xform(setc, ic<8>{}); xform(setc, ic<9>{}); xform(setc, ic<10>{}); xform(setc, ic<11>{});
    xform(setc, ic<12>{}); xform(setc, ic<13>{}); xform(setc, ic<14>{}); xform(setc, ic<15>{}); xform(setc, ic<16>{}); xform(setc, ic<17>{});
    xform(setc, ic<18>{}); xform(setc, ic<19>{}); xform(setc, ic<20>{}); xform(setc, ic<21>{}); xform(setc, ic<22>{}); xform(setc, ic<23>{});
  };
  auto load_px_all = [&](auto setc, auto ts) {
    if constexpr (WGRAD_CLUSTER) px_addr(ts);
    load_px(setc, ts, ic<0>{}); load_px(setc, ts, ic<1>{}); load_px(setc, ts, ic<2>{}); load_px(setc, ts, ic<3>{});
    load_px(setc, ts, ic<4>{}); load_px(setc, ts, ic<5>{}); load_px(setc, ts, ic<6>{}); load_px(setc, ts, ic<7>{});
    load_px(setc, ts, ic<8>{}); load_px(setc, ts, ic<9>{}); load_px(setc, ts, ic<10>{}); load_px(setc, ts, ic<11>{});
  };
  auto load_tab_all = [&](auto setc, int s) { load_tab(setc, s, ic<0>{}); load_tab(setc, s, ic<1>{}); load_tab(setc, s, ic<2>{}); };

  // ---- one k-step: 64 MFMAs on operand set CUR; in their gaps the table entries of step s+3 (into table set CUR), the
  //      pixels of step s+2 (entries in table set CUR^1, fetched during step s-1) into raw set CUR, and the transform of
  //      raw set CUR^1 (step s+1, loaded during step s-1) into operand set CUR^1 ---------------------------------------
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
          const float a = fc == 0 ? rr[cur][0][cb >> 1][cb & 1] : fc == 1 ? a1[cur][cb >> 1][cb & 1]
                        : fc == 2 ? a2[cur][cb >> 1][cb & 1] : rr[cur][1][cb >> 1][cb & 1];
          MFMA(acc[fc][cb][ib], a, bo[cur][fc][ib >> 1][ib & 1]);
          if (m == 0) load_tab(CS, s + 3, ic<0>{});
          if (m == 1) load_tab(CS, s + 3, ic<1>{});
          if (m == 2) load_tab(CS, s + 3, ic<2>{});
          if (m == 3 && WGRAD_CLUSTER) px_addr(NS);
          if (m == 3) load_px(CS, NS, ic<0>{});
          if (m == 4) load_px(CS, NS, ic<1>{});
          if (m == 5) load_px(CS, NS, ic<2>{});
          if (m == 6) load_px(CS, NS, ic<3>{});
          if (m == 7) load_px(CS, NS, ic<4>{});
          if (m == 8) load_px(CS, NS, ic<5>{});
          if (m == 9) load_px(CS, NS, ic<6>{});
          if (m == 10) load_px(CS, NS, ic<7>{});
          if (m == 11) load_px(CS, NS, ic<8>{});
          if (m == 12) load_px(CS, NS, ic<9>{});
          if (m == 13) load_px(CS, NS, ic<10>{});
          if (m == 14) load_px(CS, NS, ic<11>{});
          if (WGRAD_CLUSTER) {
            if (m == 38) xform_all(NS);
          } else {
            if (m == 38) xform(NS, ic<0>{});
            if (m == 39) xform(NS, ic<1>{});
            if (m == 40) xform(NS, ic<2>{});
            if (m == 41) xform(NS, ic<3>{});
            if (m == 42) xform(NS, ic<4>{});
            if (m == 43) xform(NS, ic<5>{});
            if (m == 44) xform(NS, ic<6>{});
            if (m == 45) xform(NS, ic<7>{});
            if (m == 46) xform(NS, ic<8>{});
            if (m == 47) xform(NS, ic<9>{});
            if (m == 48) xform(NS, ic<10>{});
            if (m == 49) xform(NS, ic<11>{});
            if (m == 50) xform(NS, ic<12>{});
            if (m == 51) xform(NS, ic<13>{});
            if (m == 52) xform(NS, ic<14>{});
            if (m == 53) xform(NS, ic<15>{});
            if (m == 54) xform(NS, ic<16>{});
            if (m == 55) xform(NS, ic<17>{});
            if (m == 56) xform(NS, ic<18>{});
            if (m == 57) xform(NS, ic<19>{});
            if (m == 58) xform(NS, ic<20>{});
            if (m == 59) xform(NS, ic<21>{});
            if (m == 60) xform(NS, ic<22>{});
            if (m == 61) xform(NS, ic<23>{});
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };

  // (issue order as in the steady state -- table entries of a step before the pixels of the step before it -- so that the
  //  s_waitcnt counts hipcc derives for the loop head are the exact ones of the loop body)
  load_tab_all(ic<0>{}, s_begin);
  load_px_all(ic<0>{}, ic<0>{});
  load_tab_all(ic<0>{}, s_begin + 1);
  load_tab_all(ic<1>{}, s_begin + 2);
  load_px_all(ic<1>{}, ic<0>{});
  xform_all(ic<0>{});
  asm volatile("s_nop 7");                            // accumulator zeros / first operands written by the VALU: keep clear of the MFMA
  for (int s = s_begin; s < s_end; s += 2) {          // steps run in pairs (steps_per_split is even; past the last tile the table
    step(ic<0>{}, s);                                 // holds only out-of-range offsets: a dead step adds zeros)
    step(ic<1>{}, s + 1);
    // the last MFMAs land before anything behind the loop reads an accumulator (hipcc cannot see the inline-asm MFMAs' latency
    // and may place accumulator copies directly behind the loop: the wait has to sit inside it)
    if (s + 2 >= s_end) asm volatile("s_nop 15\n\ts_nop 15");
  }

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
    // (frequency column 3 of A dY A^T is accumulated with the opposite sign: the main kernel feeds +dY where A has -1)
    const float gb[4] = {b == 0 ? 1.f : 0.f, 0.5f, b == 1 ? -0.5f : 0.5f, b == 2 ? -1.f : 0.f};
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
  if ((T + 32) * TE_BYTES >= (1ll << 31)) return false;                               // the table is read with 32-bit offsets
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

extern "C" size_t bevf_wino_wgrad_table_bytes(int N, int H, int W) {
  WwPlan pl;
  if (!ww_plan(N, H, W, 64, 64, &pl)) return 0;
  return (size_t)4 * (pl.steps + 4) * TE_BYTES;
}

extern "C" int bevf_wino_wgrad_table(int32_t* tab, int N, int H, int W, int x_cs, int dy_cs, void* stream) {
  WwPlan pl;
  BEVF_REQUIRE(tab && ww_plan(N, H, W, 64, 64, &pl), "wino wgrad table: bad shape N=%d H=%d W=%d", N, H, W);
  BEVF_REQUIRE(x_cs > 0 && dy_cs > 0 && x_cs % 4 == 0 && dy_cs % 4 == 0, "wino wgrad table: channel strides must be multiples of 4");
  BEVF_REQUIRE((long long)N * H * W * x_cs * 4 < (1ll << 31) && (long long)N * H * W * dy_cs * 4 < (1ll << 31),
               "wino wgrad table: x / dy buffers must stay below 2 GiB (32-bit buffer offsets)");
  BEVF_REQUIRE(bevf_aligned16(tab), "wino wgrad table: unaligned");
  const int Tpad = 4 * (pl.steps + 4);
  hipLaunchKernelGGL(wino_wgrad_table, dim3((Tpad + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), tab, H, W,
                     pl.tilesX, pl.tilesY, pl.T, Tpad, x_cs * 4, dy_cs * 4);
  return bevf_check_launch("bevf_wino_wgrad_table");
}

extern "C" int bevf_conv3x3_wgrad_wino_f32(const bevf_wgrad_desc* d, float* workspace, int accumulate, void* stream) {
  BEVF_REQUIRE(d && d->x && d->dy && d->dw && d->pixtab && workspace, "wino wgrad: null pointer (pixtab = bevf_wino_wgrad_table)");
  BEVF_REQUIRE(d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1, "wino wgrad: 3x3 / stride 1 / pad 1 only");
  WwPlan pl;
  BEVF_REQUIRE(ww_plan(d->N, d->H, d->W, d->Cin, d->Cout, &pl),
               "wino wgrad: unsupported shape N=%d H=%d W=%d Cin=%d Cout=%d (channels in multiples of 64, H, W >= 3; "
               "bevf_wino_wgrad_workspace_floats returns 0 for these)", d->N, d->H, d->W, d->Cin, d->Cout);
  BEVF_REQUIRE(d->x_cs >= d->Cin && d->x_cs % 4 == 0 && d->dy_cs >= d->Cout && d->dy_cs % 4 == 0,
               "wino wgrad: channel strides must be multiples of 4 and cover the channels");
  BEVF_REQUIRE(bevf_aligned16(d->x) && bevf_aligned16(d->dy) && bevf_aligned16(d->dw) && bevf_aligned16(workspace) && bevf_aligned16(d->pixtab),
               "wino wgrad: unaligned");
  BEVF_REQUIRE((long long)d->N * d->H * d->W * d->x_cs * 4 < (1ll << 31) && (long long)d->N * d->H * d->W * d->dy_cs * 4 < (1ll << 31),
               "wino wgrad: x / dy buffers must stay below 2 GiB (32-bit buffer offsets)");
  WwArgs a;
  a.x = d->x; a.dy = d->dy; a.tab = d->pixtab; a.part = workspace;
  a.Cin = d->Cin; a.Cout = d->Cout;
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
