// bf16 direct convolution for the 3x3 / stride 1 / pad 1 layers on v_mfma_f32_16x16x32_bf16 (fp32 accumulate), gfx950.
//
// Why its own kernel (round 3): the implicit-GEMM template (conv_igemm.hip) was designed for the fp32 MFMA and is
// LDS-port-bound at the bf16 MFMA's 16x higher rate -- every K step re-stages a 128-byte A row per output pixel through
// registers and ds_write, and for a 3x3 layer the same input pixel is staged nine times (once per filter tap).  Here the
// reuse is explicit:
//   * a workgroup owns a 16x16 block of output pixels x CT output channels; per 32-channel chunk the 18x18-pixel input
//     patch is staged ONCE (LDS-DMA, `buffer_load ... lds`, zero fill for padding / image borders by out-of-range
//     offsets) and all nine taps read their fragments from it at shifted addresses: 7x less global->LDS traffic and no
//     ds_write at all;
//   * the filters are packed on the host side of the C-ABI (bevf_conv3x3_pack_bf16) into the exact MFMA fragment order,
//     one CT x 32 slice per (chunk, tap) step, so staging them is a straight LDS-DMA copy into a small ring and the
//     fragment reads are lane-linear (bank-conflict-free);
//   * the patch image is [pixel][4 x 16 B] with the 16-byte piece index XOR-swizzled by ((pixel >> 2) & 1) << 1: the
//     ds_read_b128 of a 16-pixel row segment (lane = pixel, lane>>4 = k group) then touches every bank exactly once in
//     each of the instruction's four lane groups, for every tap shift (MI355X_MICROARCH.md "LDS");
//   * C^T orientation: the MFMA's A operand is the filter fragment, B the pixels, so a lane ends up with 4 consecutive
//     output channels of one pixel -> 8-byte bf16x4 stores / residual loads instead of 2-byte ones;
//   * 64 or 128 accumulator registers per wave and 36-72 KB of LDS: two to four workgroups per CU, so one workgroup's
//     prologue / epilogue / barrier waits overlap the others' MFMAs.
// Every wait is written by hand (`s_waitcnt vmcnt(N)` + `s_barrier` in one asm): through __syncthreads() hipcc drains
// the LDS-DMA queue (vmcnt(0)) at every barrier.
#include "conv_common.h"

#include <type_traits>

namespace {

constexpr bool C3_AUTO_WIDE64 = true;                      // `tile = 0` with Cin = 64 takes the one-image kernel (tile = 5 forces it)
constexpr bool C3_AUTO_PERSIST = false;                    // `tile = 0` takes the persistent kernel (tile = 4 forces it)
constexpr int C3_AUTO_SHORTK = 1;                         // what `tile = 0` means for 64-channel tiles with Cin < 128 (1 | 2 | 3)

struct C3Args {
  const void* x;        // bf16 NHWC, channel stride x_cs
  const void* wp;       // packed filters: [ct][chunk][tap][CT/16][64 lanes][8 bf16]
  const float* scale;   // [Cout] or null
  const float* shift;
  const void* res;      // bf16 NHWC residual or null
  void* y;              // bf16 NHWC
  int N, H, W, Cin, x_cs, Cout, y_cs, res_cs, relu;
  int TBY, TBX, nct;
  unsigned wbytes;      // size of the packed filter image
  unsigned long long* stamps;   // diagnostic builds of a run (bevf_debug_conv3x3_stamps): 4 x s_memtime per workgroup, else null
};

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int C3_PW = 18;                                 // patch width in pixels (block width 16 + halo)
constexpr int C3_PITCH = 20;                              // patch row pitch in the LDS image, pixels (see the read addresses in the kernel)

// CT = output channels per workgroup (64: waves 4 x 1, 128: waves 2 x 2; a wave always owns 64 channels).
// PB = patch buffers: 2 = the next 32-channel chunk is prefetched under the current one (two workgroups per CU);
//      1 = one buffer, FOUR workgroups per CU at BH = 16: occupancy instead of prefetch.
// BH = block height in pixel rows: 16 (16 x 16 = 256 pixels) or, for 64-channel tiles, 32 (512 pixels: a wave owns 8 rows instead of 4;
//      twice the MFMA work per tile, per barrier and per filter byte against the same fixed latencies -- for the short-K layers).
template <int CT, int PB, int BH> struct C3Geo {
  static constexpr int WN = CT / 64, WM = 4 / WN, MT = BH / WM, NT = 4;
  static constexpr int PH = BH + 2;                       // patch height
  static constexpr int PPW = (PH * C3_PITCH * 4 + 255) / 256;   // LDS-DMA pieces (64 slots of 16 B) per wave and patch chunk
  static constexpr int PATCH_BYTES = 4 * PPW * 1024;
  static constexpr int WSTEP = CT * 64;                   // bytes of filters per (chunk, tap) step
#ifndef C3_RB64
#define C3_RB64 8
#endif
#ifndef C3_RB128
#define C3_RB128 4
#endif
  // ring slots; a step's filters are requested RB-1 steps ahead.  The depth is not for the filters (L2 hits): loads retire in order, so
  // a patch piece (HBM, 4-8k cycles under load) must land within RB-1 steps of its issue or the filter wait behind it stalls
  static constexpr int RB = (CT == 64 && PB == 2) ? C3_RB64 : (CT == 128 ? C3_RB128 : 3);
  static constexpr int LDS_BYTES = PB * PATCH_BYTES + RB * WSTEP;
  static constexpr int WG_PER_CU = (PB == 1 && BH == 16) ? 4 : 2;
  static constexpr bool RES_EARLY = CT == 64 && PB == 2 && BH == 16;  // residual requested in the prologue (32 registers; see the kernel)
};

template <int CT, int PB, int BH>
__global__ __launch_bounds__(256, (C3Geo<CT, PB, BH>::WG_PER_CU)) void conv3x3_bf16(const C3Args p) {
  using Geo = C3Geo<CT, PB, BH>;
  constexpr int C3_PATCH_BYTES = Geo::PATCH_BYTES;
  constexpr int WN = Geo::WN, MT = Geo::MT, NT = Geo::NT, WSTEP = Geo::WSTEP, RB = Geo::RB, D = RB - 1;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const patch = lds;                                // [PB][C3_PATCH_BYTES]
  char* const ring = lds + PB * C3_PATCH_BYTES;           // [RB][WSTEP]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int NCH = p.Cin >> 5, S = 9 * NCH;
  // Every wave requests a quarter of both the filters and the patch.  (Tried: waves 0, 1 requesting only filters and waves 2, 3 only the
  // patch, so that a filter wait never stands behind an HBM-latency patch piece in the in-order vmcnt queue -- 3-10 % SLOWER on every layer:
  // issuing an LDS-DMA costs its wave 60-180 cycles, and two waves carrying all of one kind become the step's critical path.)
  constexpr int NSH = 4;                                              // waves sharing the DMA duty
  const int role = wave;
  constexpr int PPW = Geo::PPW, FPW = (CT / 16) / NSH;                // patch pieces per wave and chunk, filter pieces per wave and step

  // ---- tile: ct-major numbering, so the workgroups running together share one filter slab in L2; XCD-contiguous ----
  const int nsp = p.N * p.TBY * p.TBX;
  const int sid = xcd_remap(blockIdx.x, gridDim.x);
  const int ct = sid / nsp;
  int sp = sid - ct * nsp;
  const int bx = sp % p.TBX;
  sp /= p.TBX;
  const int by = sp % p.TBY, n = sp / p.TBY;

  // ---- patch staging: instruction (2 j + role), j < 12, fills 64 consecutive 16-byte slots; slot i = image pixel i >> 2 (rows of
  //      20: 18 + 2 unused), piece (i & 3) ^ swz(pixel): the swizzle sits on the SOURCE address, the LDS image stays lane-linear.
  //      The 12 source offsets are recomputed per chunk (patch waves only) instead of living in 12 registers across the loop -------
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)kOob, 0x00020000);
  auto patch_dma = [&](unsigned soff, int buf, int j0, int j1, bool live) {   // this wave's pieces j0 .. j1-1 of a chunk
    int l = lane;
    asm volatile("" : "+v"(l));                                      // (opaque: keeps hipcc from hoisting the offsets out of the chunk loop)
    const int iy0 = BH * by - 1, ix0 = 16 * bx - 1;
#pragma unroll
    for (int j = j0; j < j1; ++j) {
      const int i = (NSH * j + role) * 64 + l;
      const int pix = i >> 2, kg = (i & 3) ^ (((pix >> 2) & 1) << 1);
      const int py = (pix * 3277) >> 16, px = pix - py * C3_PITCH;         // pix / 20 for pix < 16384 / 4
      const int iy = iy0 + py, ix = ix0 + px;
      const bool ok = live && px < C3_PW && py < Geo::PH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const unsigned voff = ok ? (unsigned)((((n * p.H + iy) * p.W + ix) * p.x_cs + kg * 8) * 2) : kOob;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsx, (__attribute__((address_space(3))) void*)(patch + buf * C3_PATCH_BYTES + (NSH * j + role) * 1024), 16, voff, soff, 0, 0);
    }
  };

  // ---- filter ring: the image of step s is WSTEP contiguous bytes at (ct * S + s) * WSTEP; filter wave f copies pieces f, f+2, ..;
  //      steps past the end read past num_records and arrive as zeros in a slot nobody reads any more ------------------
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wp), 0, (int)p.wbytes, 0x00020000);
  const unsigned w_lane = (unsigned)(lane * 16);
  const unsigned w_tile = (unsigned)(ct * S) * (unsigned)WSTEP;
  auto ring_dma = [&](int step, int slot) {
#pragma unroll
    for (int i = 0; i < FPW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsw, (__attribute__((address_space(3))) void*)(ring + slot * WSTEP + (role + NSH * i) * 1024), 16, w_lane,
          w_tile + (unsigned)step * (unsigned)WSTEP + (unsigned)((role + NSH * i) * 1024), 0, 0);
  };

  // ---- fragment read addresses (bytes).  Image pixel (py, px) sits at slot (py * 20 + px) * 4 + (kg ^ swz), swz = 2 * (((py * 20 + px) >> 2) & 1)
  //      = 2 * ((py & 1) ^ ((px >> 2) & 1)) because a row is 5 quads: the address of (row, tap column kw) is a per-lane term that depends on
  //      kw and the row's PARITY only (6 registers) plus row * 1280 as an instruction immediate (the wave's first row is even) ------------
  const int col = lane & 15, kgl = lane >> 4;
  int xa[3][2];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int px = kw + col;
      xa[kw][par] = wm * MT * (C3_PITCH * 64) + px * 64 + ((kgl ^ ((par ^ ((px >> 2) & 1)) << 1)) << 4);
    }
  const int wa = PB * C3_PATCH_BYTES + wn * 4096 + lane * 16;          // + slot * WSTEP + nt * 1024

  // ---- output side (round 3, after layer1's in-kernel stamps and the wide64 experiment): whole 128-byte lines.  In the MFMA layout a
  //      lane holds 4 channels of a pixel, so residual loads / output stores were 16 x 32-byte segments per instruction; that access
  //      shape, not a latency, was what bound the short-K layers (layer1 483 -> 394 us with nothing else changed).  The epilogue
  //      therefore transposes the accumulators through LDS in passes of 8 pixel rows (128 pixels x CT channels fp32, pixel pitch
  //      CT*4 + 16 bytes: conflict-free 16-byte writes) and a lane then owns 8 consecutive channels of a pixel: item i of a pass =
  //      pixel (i * 256 + tid) / (CT / 8) of the pass, channel group (i * 256 + tid) % (CT / 8); residual and output move as 16-byte
  //      pieces, 8 (or 16) lanes per pixel line.  The 64-channel / two-buffer variant still requests its residual in the prologue. ------
  typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)kOob, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, (int)kOob, 0x00020000);
  constexpr int RP = 8, NPASS = BH / RP, CG = CT / 8, IPP = RP * 16 * CG / 256;     // rows per pass, passes, channel groups, items per thread and pass
  constexpr int TPITCH = CT * 4 + 16;
  static_assert(RP * 16 * TPITCH <= Geo::LDS_BYTES, "transpose tile does not fit the kernel's LDS");
  auto item_of = [&](int pass, int i, int& oy, int& ox, int& cb) {
    const int e = i * 256 + tid, pl = e / CG;                        // pixel of the pass (row-major, 16 columns)
    cb = ct * CT + 8 * (e - pl * CG);
    oy = BH * by + pass * RP + (pl >> 4);
    ox = 16 * bx + (pl & 15);
  };
  auto res_load = [&](int pass, int i) {
    int oy, ox, cb;
    item_of(pass, i, oy, ox, cb);
    const unsigned ro = (oy < p.H && ox < p.W) ? ((unsigned)((n * p.H + oy) * p.W + ox) * (unsigned)p.res_cs + (unsigned)cb) * 2u : kOob;
    return __builtin_amdgcn_raw_buffer_load_b128(rsr, ro, 0, 0);
  };
  u32x4v rv[Geo::RES_EARLY ? NPASS * IPP : 1];

  int mt_live = p.H - (BH * by + wm * MT);                            // pixel rows of this wave inside the image (wave-uniform)
  mt_live = mt_live < 0 ? 0 : (mt_live > MT ? MT : mt_live);
  f32x4 acc[NT][MT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // (diagnostic only: p.stamps is null in every product launch; the stamps go to a buffer nothing else reads)
  auto stamp = [&](int i) {
    if (p.stamps && tid == 0) p.stamps[(size_t)blockIdx.x * 4 + i] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);
  // ---- prologue: filters of steps 0 .. D-1, patch chunk 0 (and the residual: one HBM round trip covers both) ----------------
#pragma unroll
  for (int s = 0; s < D; ++s) ring_dma(s, s);
  patch_dma(0, 0, 0, PPW, true);
  if constexpr (Geo::RES_EARLY) {
    if (p.res) {
#pragma unroll
      for (int k = 0; k < NPASS * IPP; ++k) rv[k] = res_load(k / IPP, k % IPP);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  stamp(1);

  // One step = one filter tap of one 32-channel chunk: MT x NT MFMAs per wave.  Issue order inside a step: the filter pieces of step
  // s + D, then (two-buffer variant, taps 0..5) one sixth of the NEXT chunk's patch; loads retire in order, so the wait at the step's
  // end -- vmcnt(pieces issued after the filters of step s + 1) -- names exactly those filters and, after tap 8, the whole next patch
  // chunk; everything younger stays in flight across the barrier.
  int slot = 0;                                                      // ring slot of the current step
  auto step = [&](auto tc, auto livec, const int s, const int pbuf, const unsigned psoff, const bool pnext) {
    constexpr int t = decltype(tc)::value, kh = t / 3, kw = t % 3;
    {
      int ns = slot + D;
      ns = ns >= RB ? ns - RB : ns;
      ring_dma(s + D, ns);
    }
    static_assert(PB == 1 || PPW <= 9, "one patch piece per tap");
    if constexpr (PB == 2 && t < PPW) patch_dma(psoff, pbuf ^ 1, t, t + 1, pnext);   // one piece per step, taps 0 .. PPW-1 (zeros past the last chunk)
    constexpr int LIVE = decltype(livec)::value;                    // pixel rows of this wave that take part (the rest lie below the image)
    if constexpr (LIVE > 0) {
      bf16x8 wf[NT], xf[LIVE];
      const char* wb = lds + wa + slot * WSTEP;
      const char* pb = patch + pbuf * C3_PATCH_BYTES;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) wf[nt] = *reinterpret_cast<const bf16x8*>(wb + nt * 1024);
#pragma unroll
      for (int mt = 0; mt < LIVE; ++mt)
        xf[mt] = *reinterpret_cast<const bf16x8*>(pb + xa[kw][(mt + kh) & 1] + (mt + kh) * (C3_PITCH * 64));
#pragma unroll
      for (int mt = 0; mt < LIVE; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
    }
    slot = slot + 1 == RB ? 0 : slot + 1;
    // pieces this wave has issued AFTER the filters of step s + 1 (which left first thing in step s + 1 - D): FPW per later step, one
    // patch piece in each of the D steps whose tap is < 6; after tap 8 that also covers the whole next patch chunk
    constexpr int cnt = [] {
      int c = (D - 1) * FPW;
      if (PB == 2) {
        for (int u = 0; u < D; ++u) c += ((t - u + 9) % 9) < PPW ? 1 : 0;
        // tap 8 must ALSO leave the next chunk's whole patch landed: nothing may stay in flight but what was issued after its last
        // piece (tap PPW-1), i.e. the filters of taps PPW .. 8.  (With a ring deeper than that the first count alone let patch pieces
        // fly across the chunk boundary: wrong pixels at full size only, where the DMA queue is long -- caught by the batch-invariance
        // test of config 5, not by the small-shape tests.)
        if (t == 8 && c > (9 - PPW) * FPW) c = (9 - PPW) * FPW;
      }
      return c;
    }();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(cnt) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>; using T2 = std::integral_constant<int, 2>;
  using T3 = std::integral_constant<int, 3>; using T4 = std::integral_constant<int, 4>; using T5 = std::integral_constant<int, 5>;
  using T6 = std::integral_constant<int, 6>; using T7 = std::integral_constant<int, 7>; using T8 = std::integral_constant<int, 8>;
  auto kloop = [&](auto lv) {
    for (int c = 0; c < NCH; ++c) {
      const int pbuf = PB == 2 ? (c & 1) : 0, s0 = 9 * c;
      if (PB == 1 && c > 0) {                                        // every wave has left the previous chunk (barrier): refill in place
        patch_dma((unsigned)(c * 64), 0, 0, PPW, true);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      }
      const unsigned psoff = (unsigned)((c + 1) * 64);               // next chunk: + 32 channels
      const bool pnext = c + 1 < NCH;
      step(T0{}, lv, s0 + 0, pbuf, psoff, pnext); step(T1{}, lv, s0 + 1, pbuf, psoff, pnext); step(T2{}, lv, s0 + 2, pbuf, psoff, pnext);
      step(T3{}, lv, s0 + 3, pbuf, psoff, pnext); step(T4{}, lv, s0 + 4, pbuf, psoff, pnext); step(T5{}, lv, s0 + 5, pbuf, psoff, pnext);
      step(T6{}, lv, s0 + 6, pbuf, psoff, pnext); step(T7{}, lv, s0 + 7, pbuf, psoff, pnext); step(T8{}, lv, s0 + 8, pbuf, psoff, pnext);
    }
  };
  // Bottom-edge blocks: a wave's pixel rows below the image are skipped in quarters of its MT rows (wave-uniform choice of a loop
  // specialised at compile time; per-row branches inside one loop cost hipcc 245 spilled registers).  With two to four workgroups per
  // CU the matrix-pipe and LDS cycles those rows would have burnt go to the co-resident workgroups (H = 57: 7 of 64 rows, H = 113:
  // 15 of 128).  Skipped rows keep their zero accumulators and are never stored.
  constexpr int QM = MT / 4;
  switch ((mt_live + QM - 1) / QM) {
    case 0: kloop(std::integral_constant<int, 0>{}); break;
    case 1: kloop(std::integral_constant<int, QM>{}); break;
    case 2: kloop(std::integral_constant<int, 2 * QM>{}); break;
    case 3: kloop(std::integral_constant<int, 3 * QM>{}); break;
    default: kloop(std::integral_constant<int, MT>{}); break;
  }

  stamp(2);
  // ---- epilogue: transpose through LDS, folded BN, residual, ReLU, 16-byte bf16x8 stores ----------------------------------------------
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");     // the look-ahead's last (zero-fill) DMAs still target LDS: drain before reuse
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    u32x4v rq[IPP];
    if constexpr (!Geo::RES_EARLY) {
      if (p.res) {
#pragma unroll
        for (int i = 0; i < IPP; ++i) rq[i] = res_load(pass, i);    // in flight under the transpose below
      }
    }
    if (pass) __syncthreads();                                       // the previous pass's reads are done
    if ((wm * MT) / RP == pass || (MT < RP && (wm * MT) / RP == pass)) {   // this wave's rows lie in the pass (wave-uniform)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          *reinterpret_cast<f32x4*>(lds + (((wm * MT + mt) - pass * RP) * 16 + col) * TPITCH + (wn * 64 + nt * 16 + 4 * kgl) * 4) = acc[nt][mt];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < IPP; ++i) {
      int oy, ox, cb;
      item_of(pass, i, oy, ox, cb);
      const int e = i * 256 + tid, pl = e / CG;
      const char* tp = lds + pl * TPITCH + (e - pl * CG) * 32;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(tp), a1 = *reinterpret_cast<const f32x4*>(tp + 16);
      f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, sh0 = {0.f, 0.f, 0.f, 0.f}, sh1 = sh0;
      if (p.scale) { sc0 = *reinterpret_cast<const f32x4*>(p.scale + cb); sc1 = *reinterpret_cast<const f32x4*>(p.scale + cb + 4); }
      if (p.shift) { sh0 = *reinterpret_cast<const f32x4*>(p.shift + cb); sh1 = *reinterpret_cast<const f32x4*>(p.shift + cb + 4); }
      float o[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { o[j] = fmaf(a0[j], sc0[j], sh0[j]); o[4 + j] = fmaf(a1[j], sc1[j], sh1[j]); }
      if (p.res) {
        const bf16x8 r8 = __builtin_bit_cast(bf16x8, Geo::RES_EARLY ? rv[Geo::RES_EARLY ? pass * IPP + i : 0] : rq[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += (float)r8[j];
      }
      if (p.relu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fmaxf(o[j], 0.f);
      }
      bf16x8 ob;
#pragma unroll
      for (int j = 0; j < 8; ++j) ob[j] = (__bf16)o[j];
      const unsigned yo = (oy < p.H && ox < p.W) ? ((unsigned)((n * p.H + oy) * p.W + ox) * (unsigned)p.y_cs + (unsigned)cb) * 2u : kOob;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, ob), rsy, yo, 0, 0);
    }
  }
  if (p.stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (diagnostic: the stores have been acknowledged)
    stamp(3);
  }
}

// ---- Cin = 64 (ResNet layer1): both 32-channel halves of the patch in ONE image, fetched as whole 128-byte lines ---------------------
// Hypothesis behind it: layer1 is bound by a throughput between L2 and the CU (3.1c); its patch DMA asks for 64 of every pixel's 128 bytes
// per chunk, i.e. 16 half-used cache lines per instruction, and the other halves a microsecond later.  Here a pixel's 128 bytes sit
// together in LDS ([pixel][2 halves][4 x 16 B], 46 KB, one buffer), every DMA piece is 8 pixels x 128 contiguous bytes, the whole patch is
// requested in the prologue and the K loop issues filter DMA only.  Price: 128-byte pixel pitch -> the x-fragment reads are 2-way bank
// conflicts (16 lanes over 8 distinct 16-byte slots of each 128-byte half-window).  Same arithmetic in the same order as the other variants.
template <int CT>
__global__ __launch_bounds__(256, 2) void conv3x3_bf16_wide64(const C3Args p) {
  static_assert(CT == 64, "64 output channels per workgroup");
  constexpr int MT = 4, NT = 4, WSTEP = CT * 64, RB = 8, D = RB - 1, NSH = 4, FPW = 1;
  constexpr int PH = 18, PIXB = 128, ROWB = C3_PITCH * PIXB;        // 2560 bytes per patch row
  constexpr int PPW = (PH * ROWB + 4095) / 4096;                    // DMA pieces per wave: 12
  constexpr int PATCH_BYTES = 4 * PPW * 1024;                       // 49152
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const patch = lds;
  char* const ring = lds + PATCH_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave, role = wave;
  const int S = 18;
  const int nsp = p.N * p.TBY * p.TBX;
  const int sid = xcd_remap(blockIdx.x, gridDim.x);
  const int ct = sid / nsp;
  int sp = sid - ct * nsp;
  const int bx = sp % p.TBX;
  sp /= p.TBX;
  const int by = sp % p.TBY, n = sp / p.TBY;

  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)kOob, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wp), 0, (int)p.wbytes, 0x00020000);
  const unsigned w_lane = (unsigned)(lane * 16);
  const unsigned w_tile = (unsigned)(ct * S) * (unsigned)WSTEP;
  auto ring_dma = [&](int step, int slot) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(ring + slot * WSTEP + role * 1024), 16, w_lane,
                                             w_tile + (unsigned)step * (unsigned)WSTEP + (unsigned)(role * 1024), 0, 0);
  };
  // patch: slot i = pixel i >> 3, half (i >> 2) & 1, piece (i & 3) ^ swz(pixel); swz as in the narrow image (2 * ((pixel >> 2) & 1))
#pragma unroll
  for (int s = 0; s < D; ++s) ring_dma(s, s);
  {
    const int iy0 = 16 * by - 1, ix0 = 16 * bx - 1;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int i = (NSH * j + role) * 64 + lane;
      const int pix = i >> 3, half = (i >> 2) & 1, kg = (i & 3) ^ (((pix >> 2) & 1) << 1);
      const int py = (pix * 3277) >> 16, px = pix - py * C3_PITCH;
      const int iy = iy0 + py, ix = ix0 + px;
      const bool ok = px < C3_PW && py < PH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const unsigned voff = ok ? (unsigned)((((n * p.H + iy) * p.W + ix) * p.x_cs + half * 32 + kg * 8) * 2) : kOob;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (__attribute__((address_space(3))) void*)(patch + (NSH * j + role) * 1024), 16, voff, 0, 0, 0);
    }
  }
  const int col = lane & 15, kgl = lane >> 4;
  int xa[3][2];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int px = kw + col;
      xa[kw][par] = wm * MT * ROWB + px * PIXB + ((kgl ^ ((par ^ ((px >> 2) & 1)) << 1)) << 4);
    }
  const int wa = PATCH_BYTES + lane * 16;

  // Output side: whole 128-byte pixel lines.  The accumulators (a lane: 4 channels of one pixel per MFMA tile) are transposed through LDS
  // after the K loop so that a lane then owns 8 consecutive channels (16 bytes of bf16) of a pixel: a wave instruction reads / writes
  // 8 pixels x 128 contiguous bytes of the residual / the output instead of 16 x 32-byte segments.  Item i of lane l: pixel
  // 32 i + (tid >> 3) of the 16 x 16 block (row = pixel >> 4, column = pixel & 15), channels 8 (tid & 7) ..
  typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)kOob, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, (int)kOob, 0x00020000);
  const int g8 = tid & 7, pq = tid >> 3;                              // channel group, pixel within a run of 32
  const int cbase = ct * CT + 8 * g8;
  auto item_pixel = [&](int i, int& oy, int& ox) {
    const int pb = 32 * i + pq;
    oy = 16 * by + (pb >> 4);
    ox = 16 * bx + (pb & 15);
  };
  u32x4v rv[8];
  if (p.res) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int oy, ox;
      item_pixel(i, oy, ox);
      const unsigned ro = (oy < p.H && ox < p.W) ? ((unsigned)((n * p.H + oy) * p.W + ox) * (unsigned)p.res_cs + (unsigned)cbase) * 2u : kOob;
      rv[i] = __builtin_amdgcn_raw_buffer_load_b128(rsr, ro, 0, 0);
    }
  }
  f32x4 acc[NT][MT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

  int slot = 0;
  auto step = [&](auto tc, const int s, const int half) {
    constexpr int t = decltype(tc)::value, kh = t / 3, kw = t % 3;
    {
      int ns = slot + D;
      ns = ns >= RB ? ns - RB : ns;
      ring_dma(s + D, ns);
    }
    bf16x8 wf[NT], xf[MT];
    const char* wb = lds + wa + slot * WSTEP;
    const char* pb = patch + half * 64;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wf[nt] = *reinterpret_cast<const bf16x8*>(wb + nt * 1024);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xf[mt] = *reinterpret_cast<const bf16x8*>(pb + xa[kw][(mt + kh) & 1] + (mt + kh) * ROWB);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
    slot = slot + 1 == RB ? 0 : slot + 1;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * FPW) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>; using T2 = std::integral_constant<int, 2>;
  using T3 = std::integral_constant<int, 3>; using T4 = std::integral_constant<int, 4>; using T5 = std::integral_constant<int, 5>;
  using T6 = std::integral_constant<int, 6>; using T7 = std::integral_constant<int, 7>; using T8 = std::integral_constant<int, 8>;
  for (int c = 0; c < 2; ++c) {
    const int s0 = 9 * c;
    step(T0{}, s0 + 0, c); step(T1{}, s0 + 1, c); step(T2{}, s0 + 2, c); step(T3{}, s0 + 3, c); step(T4{}, s0 + 4, c);
    step(T5{}, s0 + 5, c); step(T6{}, s0 + 6, c); step(T7{}, s0 + 7, c); step(T8{}, s0 + 8, c);
  }
  // ---- epilogue: accumulators -> LDS [pixel][64 channels fp32], pixel pitch 272 bytes (16 lanes x 16-byte writes then spread over all
  //      banks); then per lane 8 consecutive channels of a pixel: folded BN, residual, ReLU, one 16-byte bf16x8 store ------------------
  constexpr int TP = 272;
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");     // the look-ahead's last (zero-fill) filter DMAs still target the ring: drain
  {
    const int col = lane & 15, kgl = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        *reinterpret_cast<f32x4*>(lds + ((wm * MT + mt) * 16 + col) * TP + (nt * 16 + 4 * kgl) * 4) = acc[nt][mt];
  }
  __syncthreads();
  f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, sh0 = {0.f, 0.f, 0.f, 0.f}, sh1 = sh0;
  if (p.scale) { sc0 = *reinterpret_cast<const f32x4*>(p.scale + cbase); sc1 = *reinterpret_cast<const f32x4*>(p.scale + cbase + 4); }
  if (p.shift) { sh0 = *reinterpret_cast<const f32x4*>(p.shift + cbase); sh1 = *reinterpret_cast<const f32x4*>(p.shift + cbase + 4); }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int oy, ox;
    item_pixel(i, oy, ox);
    const char* tp = lds + (32 * i + pq) * TP + g8 * 32;
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(tp), a1 = *reinterpret_cast<const f32x4*>(tp + 16);
    float o[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = fmaf(a0[j], sc0[j], sh0[j]); o[4 + j] = fmaf(a1[j], sc1[j], sh1[j]); }
    if (p.res) {
      const bf16x8 r8 = __builtin_bit_cast(bf16x8, rv[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += (float)r8[j];
    }
    if (p.relu) {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = fmaxf(o[j], 0.f);
    }
    bf16x8 ob;
#pragma unroll
    for (int j = 0; j < 8; ++j) ob[j] = (__bf16)o[j];
    const unsigned yo = (oy < p.H && ox < p.W) ? ((unsigned)((n * p.H + oy) * p.W + ox) * (unsigned)p.y_cs + (unsigned)cbase) * 2u : kOob;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, ob), rsy, yo, 0, 0);
  }
}

// ---- persistent form (two patch buffers, 16-row blocks): a workgroup walks a list of tiles and the DMA schedule simply runs on ---------
// In-kernel stamps on layer1 (64 -> 64, K = 576): a tile has 4.6k cycles of MFMA work per wave and lives 29k -- 8.3k waiting for its first
// patch and residual (HBM), 15k in the K loop, 6k in the epilogue until its stores are acknowledged; occupancy (2-4 workgroups per CU) cannot
// cover that, and a bigger tile changes nothing.  Here the schedule of conv3x3_bf16 is not cut at the tile boundary: in a tile's LAST chunk
// the "next chunk" patch pieces are chunk 0 of the NEXT tile, the filter look-ahead runs on into the next tile's first steps, the residual is
// requested at the start of the last chunk (64-channel tiles) and the epilogue's stores drain under the next tile's first steps.  The waits
// stay exact counts: loads, DMAs and stores retire in order, so a wait may leave in flight whatever was issued after the filters it names --
// the residual loads in the last chunk's first D steps, the previous tile's stores in a tile's first D-1 steps.
// Each XCD owns a contiguous eighth of the tile list and its workgroups sweep it side by side (neighbouring tiles share halo rows in L2).
template <int CT>
__global__ __launch_bounds__(256, 2) void conv3x3_bf16_persist(const C3Args p, const int ntiles) {
  using Geo = C3Geo<CT, 2, 16>;
  constexpr int C3_PATCH_BYTES = Geo::PATCH_BYTES;
  constexpr int WN = Geo::WN, MT = Geo::MT, NT = Geo::NT, WSTEP = Geo::WSTEP, RB = Geo::RB, D = RB - 1;
  constexpr int NSH = 4, PPW = Geo::PPW, FPW = (CT / 16) / NSH;
  constexpr bool RES_LATE = CT == 64;                              // residual requested at the start of the last chunk (MT*NT loads per wave)
  constexpr int NRES = MT * NT, NST = MT * NT;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const patch = lds;
  char* const ring = lds + 2 * C3_PATCH_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN, role = wave;
  const int NCH = p.Cin >> 5, S = 9 * NCH;
  const int nsp = p.N * p.TBY * p.TBX;

  // ---- this workgroup's tiles: XCD x = blockIdx & 7 owns tiles [start, start + len); its gx workgroups take start + li, start + li + gx, ..
  const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
  const int gx = ((int)gridDim.x + 7 - xcd) >> 3;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int t_end = xcd * q8 + (xcd < r8 ? xcd : r8) + q8 + (xcd < r8 ? 1 : 0);
  int tile = xcd * q8 + (xcd < r8 ? xcd : r8) + li;
  if (tile >= t_end) return;                                        // (workgroup-uniform; before any barrier)
  struct Tile { int ct, n, by, bx; };
  auto decode = [&](int t) {
    Tile r;
    r.ct = t / nsp;
    int sp = t - r.ct * nsp;
    r.bx = sp % p.TBX;
    sp /= p.TBX;
    r.by = sp % p.TBY;
    r.n = sp / p.TBY;
    return r;
  };

  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)kOob, 0x00020000);
  auto patch_dma = [&](const Tile& T, unsigned soff, int buf, int j0, int j1, bool live) {
    int l = lane;
    asm volatile("" : "+v"(l));
    const int iy0 = 16 * T.by - 1, ix0 = 16 * T.bx - 1;
#pragma unroll
    for (int j = j0; j < j1; ++j) {
      const int i = (NSH * j + role) * 64 + l;
      const int pix = i >> 2, kg = (i & 3) ^ (((pix >> 2) & 1) << 1);
      const int py = (pix * 3277) >> 16, px = pix - py * C3_PITCH;
      const int iy = iy0 + py, ix = ix0 + px;
      const bool ok = live && px < C3_PW && py < Geo::PH && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const unsigned voff = ok ? (unsigned)((((T.n * p.H + iy) * p.W + ix) * p.x_cs + kg * 8) * 2) : kOob;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsx, (__attribute__((address_space(3))) void*)(patch + buf * C3_PATCH_BYTES + (NSH * j + role) * 1024), 16, voff, soff, 0, 0);
    }
  };
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wp), 0, (int)p.wbytes, 0x00020000);
  const unsigned w_lane = (unsigned)(lane * 16);
  auto ring_dma = [&](unsigned wofs, int slot) {                    // wofs: byte offset of the step's filter image (past the end: zeros)
#pragma unroll
    for (int i = 0; i < FPW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsw, (__attribute__((address_space(3))) void*)(ring + slot * WSTEP + (role + NSH * i) * 1024), 16, w_lane,
          wofs + (unsigned)((role + NSH * i) * 1024), 0, 0);
  };

  const int col = lane & 15, kgl = lane >> 4;
  int xa[3][2];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int px = kw + col;
      xa[kw][par] = wm * MT * (C3_PITCH * 64) + px * 64 + ((kgl ^ ((par ^ ((px >> 2) & 1)) << 1)) << 4);
    }
  const int wa = 2 * C3_PATCH_BYTES + wn * 4096 + lane * 16;

  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)kOob, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, (int)kOob, 0x00020000);

  // ---- prologue of the FIRST tile only: filters of its steps 0 .. D-1, patch chunk 0 ------------------------------------------------
  Tile cur = decode(tile);
  int nxt_id = tile + gx;
  Tile nxt = decode(nxt_id < t_end ? nxt_id : tile);
  bool has_next = nxt_id < t_end;
  const unsigned wslab = (unsigned)S * (unsigned)WSTEP;             // bytes of one channel tile's filters
  unsigned wofs = (unsigned)cur.ct * wslab;                         // look-ahead cursor: where the filters of step (current + D) are
  int ahead_left = S;                                               // steps left in the cursor's tile
#pragma unroll
  for (int s = 0; s < D; ++s) {
    ring_dma(wofs, s);
    wofs += WSTEP;
  }
  ahead_left -= D;
  patch_dma(cur, 0, 0, 0, PPW, true);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

  int slot = 0;
  int pbuf = 0;                                                     // patch buffer of the current chunk (runs on across tiles)
  f32x4 acc[NT][MT];
  u32x2 rv[RES_LATE ? MT : 1][NT];
  // per-tile output addressing
  int co0, ox, oy0;
  unsigned pixel0;
  auto out_ok = [&](int mt) { return ox < p.W && oy0 + mt < p.H; };
  auto load_res = [&](int mt, u32x2 (&dst)[NT]) {
    const unsigned ro = out_ok(mt) ? ((pixel0 + (unsigned)(mt * p.W)) * (unsigned)p.res_cs + (unsigned)co0) * 2u : kOob;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) dst[nt] = __builtin_amdgcn_raw_buffer_load_b64(rsr, ro, (unsigned)(nt * 32), 0);
  };

  // one step; AFTER = first chunk of a tile that follows an epilogue (its NST stores are still draining), LASTC = the tile's last chunk
  auto step = [&](auto tc, auto livec, auto afterc, auto lastc, const Tile& ptile, const unsigned psoff, const bool plive) {
    constexpr int t = decltype(tc)::value, kh = t / 3, kw = t % 3;
    constexpr bool AFTER = decltype(afterc)::value, LASTC = decltype(lastc)::value;
    {
      int ns = slot + D;
      ns = ns >= RB ? ns - RB : ns;
      ring_dma(wofs, ns);
      wofs += WSTEP;
      if (--ahead_left == 0) {                                      // the cursor enters the next tile (or runs off the list: zeros)
        wofs = has_next ? (unsigned)nxt.ct * wslab : p.wbytes;
        ahead_left = has_next ? S : (1 << 30);
      }
    }
    if constexpr (t < PPW) patch_dma(ptile, psoff, pbuf ^ 1, t, t + 1, plive);
    if constexpr (RES_LATE && LASTC && t == 0) {
      if (p.res) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) load_res(mt, rv[mt]);
      } else {                                                      // keep the queue's shape (the counts below are compile-time)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) rv[mt][nt] = __builtin_amdgcn_raw_buffer_load_b64(rsr, kOob, 0, 0);
      }
    }
    constexpr int LIVE = decltype(livec)::value;
    if constexpr (LIVE > 0) {
      bf16x8 wf[NT], xf[LIVE];
      const char* wb = lds + wa + slot * WSTEP;
      const char* pb = patch + pbuf * C3_PATCH_BYTES;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) wf[nt] = *reinterpret_cast<const bf16x8*>(wb + nt * 1024);
#pragma unroll
      for (int mt = 0; mt < LIVE; ++mt)
        xf[mt] = *reinterpret_cast<const bf16x8*>(pb + xa[kw][(mt + kh) & 1] + (mt + kh) * (C3_PITCH * 64));
#pragma unroll
      for (int mt = 0; mt < LIVE; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
    }
    slot = slot + 1 == RB ? 0 : slot + 1;
    constexpr int cnt = [] {
      int c = (D - 1) * FPW;
      for (int u = 0; u < D; ++u) c += ((t - u + 9) % 9) < PPW ? 1 : 0;
      if (t == 8 && c > (9 - PPW) * FPW) c = (9 - PPW) * FPW;      // the next chunk's (or tile's) patch has landed
      // in flight by right: what was issued AFTER the filters of step s + 1 (which left first thing in step t + 1 - D of this chunk)
      if (RES_LATE && LASTC && t <= D - 1 && t != 8) c += NRES;     // the residual loads of step 0
      if (AFTER && t <= D - 2 && t != 8) c += NST;                  // the previous tile's stores
      return c > 63 ? 63 : c;
    }();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(cnt) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>; using T2 = std::integral_constant<int, 2>;
  using T3 = std::integral_constant<int, 3>; using T4 = std::integral_constant<int, 4>; using T5 = std::integral_constant<int, 5>;
  using T6 = std::integral_constant<int, 6>; using T7 = std::integral_constant<int, 7>; using T8 = std::integral_constant<int, 8>;
  auto chunk = [&](auto lv, auto afterc, auto lastc, const Tile& ptile, unsigned psoff, bool plive) {
    step(T0{}, lv, afterc, lastc, ptile, psoff, plive); step(T1{}, lv, afterc, lastc, ptile, psoff, plive);
    step(T2{}, lv, afterc, lastc, ptile, psoff, plive); step(T3{}, lv, afterc, lastc, ptile, psoff, plive);
    step(T4{}, lv, afterc, lastc, ptile, psoff, plive); step(T5{}, lv, afterc, lastc, ptile, psoff, plive);
    step(T6{}, lv, afterc, lastc, ptile, psoff, plive); step(T7{}, lv, afterc, lastc, ptile, psoff, plive);
    step(T8{}, lv, afterc, lastc, ptile, psoff, plive);
    pbuf ^= 1;
  };
  using TT = std::true_type;
  using FF = std::false_type;
  bool first_tile = true;
  for (;;) {
    co0 = cur.ct * CT + wn * 64 + 4 * kgl;
    ox = 16 * cur.bx + col;
    oy0 = 16 * cur.by + wm * MT;
    pixel0 = (unsigned)((cur.n * p.H + oy0) * p.W + ox);
    int mt_live = p.H - oy0;
    mt_live = mt_live < 0 ? 0 : (mt_live > MT ? MT : mt_live);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto kloop = [&](auto lv) {
      for (int c = 0; c < NCH; ++c) {
        const bool lastc = c + 1 == NCH, after = c == 0 && !first_tile;
        // the patch that streams in under this chunk: the next chunk of this tile, or chunk 0 of the next tile
        const Tile& ptile = lastc ? nxt : cur;
        const unsigned psoff = lastc ? 0u : (unsigned)((c + 1) * 64);
        const bool plive = lastc ? has_next : true;
        if (lastc) { if (after) chunk(lv, TT{}, TT{}, ptile, psoff, plive); else chunk(lv, FF{}, TT{}, ptile, psoff, plive); }
        else       { if (after) chunk(lv, TT{}, FF{}, ptile, psoff, plive); else chunk(lv, FF{}, FF{}, ptile, psoff, plive); }
      }
    };
    kloop(std::integral_constant<int, MT>{});                      // (no dead-row specialisation here: every extra copy of the loop cost registers)
    (void)mt_live;

    // ---- epilogue of this tile (its stores drain under the next tile's first steps) ----------------------------------------------
    f32x4 sc[NT], sh[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      sc[nt] = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + co0 + nt * 16) : f32x4{1.f, 1.f, 1.f, 1.f};
      sh[nt] = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + co0 + nt * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    constexpr int RD = RES_LATE ? 1 : 4;
    u32x2 rq[RD][NT];
    if constexpr (!RES_LATE) {
      if (p.res) {
#pragma unroll
        for (int mt = 0; mt < RD; ++mt) load_res(mt, rq[mt]);
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const unsigned yo = out_ok(mt) ? ((pixel0 + (unsigned)(mt * p.W)) * (unsigned)p.y_cs + (unsigned)co0) * 2u : kOob;
      u32x2 rr[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) rr[nt] = RES_LATE ? rv[RES_LATE ? mt : 0][nt] : rq[mt % RD][nt];
      if constexpr (!RES_LATE) {
        if (p.res && mt + RD < MT) load_res(mt + RD, rq[mt % RD]);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = fmaf(acc[nt][mt][j], sc[nt][j], sh[nt][j]);
        if (p.res) {
          const bf16x4 r4 = __builtin_bit_cast(bf16x4, rr[nt]);
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] += (float)r4[j];
        }
        if (p.relu) {
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.f);
        }
        bf16x4 ob;
#pragma unroll
        for (int j = 0; j < 4; ++j) ob[j] = (__bf16)o[j];
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ob), rsy, yo, (unsigned)(nt * 32), 0);
      }
    }
    if (!has_next) break;
    first_tile = false;
    tile = nxt_id;
    cur = nxt;
    nxt_id = tile + gx;
    has_next = nxt_id < t_end;
    nxt = decode(has_next ? nxt_id : tile);
  }
}

// OHWI bf16 filters [Cout][3][3][Cin] -> [ct][chunk][tap][CT/16][lane][8]: element j of lane (m = lane & 15, kg = lane >> 4)
// of fragment nt is w[ct*CT + nt*16 + m][tap][chunk*32 + kg*8 + j]
__global__ __launch_bounds__(256) void conv3x3_pack(const unsigned short* __restrict__ w, unsigned short* __restrict__ out, int Cout,
                                                    int Cin, int CT, long long total) {
  const long long idx = blockIdx.x * 256ll + threadIdx.x;            // one 16-byte fragment piece each
  if (idx >= total) return;
  const int NCH = Cin >> 5, nfr = CT / 16;
  long long r = idx;
  const int lane = (int)(r & 63); r >>= 6;
  const int nt = (int)(r % nfr); r /= nfr;
  const int tap = (int)(r % 9); r /= 9;
  const int c = (int)(r % NCH);
  const int ct = (int)(r / NCH);
  const int co = ct * CT + nt * 16 + (lane & 15), ci = c * 32 + (lane >> 4) * 8;
  typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
  u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
  if (co < Cout) v = *reinterpret_cast<const u16x8*>(w + ((size_t)co * 9 + tap) * Cin + ci);
  *reinterpret_cast<u16x8*>(out + idx * 8) = v;
}

}  // namespace

// Output-channel tile the kernel will use for a layer (the packed filter image depends on it).
extern "C" int bevf_conv3x3_bf16_ct(int Cout) { return Cout % 128 == 0 ? 128 : 64; }

extern "C" size_t bevf_conv3x3_pack_elems(int Cout, int Cin) {
  const int CT = bevf_conv3x3_bf16_ct(Cout);
  return (size_t)((Cout + CT - 1) / CT) * CT * 9 * (size_t)Cin;
}

extern "C" int bevf_conv3x3_pack_bf16(const void* w_ohwi, void* packed, int Cout, int Cin, void* stream) {
  BEVF_REQUIRE(w_ohwi && packed, "conv3x3_pack: null pointer");
  BEVF_REQUIRE(Cout > 0 && Cin > 0 && Cin % 32 == 0, "conv3x3_pack: Cin=%d must be a positive multiple of 32", Cin);
  BEVF_REQUIRE(bevf_aligned16(w_ohwi) && bevf_aligned16(packed), "conv3x3_pack: pointers must be 16-byte aligned");
  const int CT = bevf_conv3x3_bf16_ct(Cout);
  const long long total = (long long)bevf_conv3x3_pack_elems(Cout, Cin) / 8;
  hipLaunchKernelGGL(conv3x3_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned short*>(w_ohwi), static_cast<unsigned short*>(packed), Cout, Cin, CT, total);
  return bevf_check_launch("bevf_conv3x3_pack_bf16");
}

static unsigned long long* g_c3_stamps = nullptr;
// Diagnostic (tools/conv3x3_bench.py stamps): buf = device buffer of 4 x 8 bytes per workgroup of the NEXT launches, or null to stop
extern "C" int bevf_debug_conv3x3_stamps(void* buf) {
  g_c3_stamps = static_cast<unsigned long long*>(buf);
  return BEVF_OK;
}

extern "C" int bevf_conv3x3_bf16(const bevf_conv_desc* d, void* stream) {
  BEVF_REQUIRE(d && d->x && d->w && d->y, "conv3x3_bf16: null x / w / y");
  BEVF_REQUIRE(d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1, "conv3x3_bf16: 3x3, stride 1, pad 1 only (got %dx%d s%d p%d)",
               d->KH, d->KW, d->stride, d->pad);
  BEVF_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cout > 0 && d->Ho == d->H && d->Wo == d->W, "conv3x3_bf16: bad shape");
  BEVF_REQUIRE(d->Cin > 0 && d->Cin % 32 == 0, "conv3x3_bf16: Cin=%d must be a positive multiple of 32", d->Cin);
  BEVF_REQUIRE(d->Cout % 64 == 0, "conv3x3_bf16: Cout=%d must be a multiple of 64", d->Cout);
  BEVF_REQUIRE(d->x_cs >= d->Cin && d->x_cs % 8 == 0, "conv3x3_bf16: x_cs=%d must be >= Cin and a multiple of 8", d->x_cs);
  BEVF_REQUIRE(d->y_cs >= d->Cout && d->y_cs % 4 == 0, "conv3x3_bf16: y_cs=%d must be >= Cout and a multiple of 4", d->y_cs);
  BEVF_REQUIRE(!d->res || (d->res_cs >= d->Cout && d->res_cs % 4 == 0), "conv3x3_bf16: res_cs must be >= Cout and a multiple of 4");
  BEVF_REQUIRE(!d->colmax && !d->stats && !d->bnb_x, "conv3x3_bf16: no column max / BatchNorm epilogues (inference kernel)");
  BEVF_REQUIRE(bevf_aligned16(d->x) && bevf_aligned16(d->w) && (reinterpret_cast<uintptr_t>(d->y) & 7u) == 0 &&
                   (!d->res || (reinterpret_cast<uintptr_t>(d->res) & 7u) == 0) &&
                   (!d->scale || bevf_aligned16(d->scale)) && (!d->shift || bevf_aligned16(d->shift)),
               "conv3x3_bf16: x / w / scale / shift must be 16-byte aligned, y / res 8-byte aligned");
  BEVF_REQUIRE((long long)d->N * d->H * d->W * d->x_cs * 2 < (1ll << 31) && (long long)d->N * d->H * d->W * d->y_cs * 2 < (1ll << 31) &&
                   (!d->res || (long long)d->N * d->H * d->W * d->res_cs * 2 < (1ll << 31)),
               "conv3x3_bf16: activations must stay below 2 GiB (32-bit buffer offsets)");
  const size_t wbytes = bevf_conv3x3_pack_elems(d->Cout, d->Cin) * 2;
  BEVF_REQUIRE(wbytes < (1ull << 31), "conv3x3_bf16: packed filters must stay below 2 GiB");
  C3Args a;
  a.x = d->x; a.wp = d->w; a.scale = d->scale; a.shift = d->shift; a.res = d->res; a.y = d->y;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.x_cs = d->x_cs; a.Cout = d->Cout; a.y_cs = d->y_cs; a.res_cs = d->res_cs;
  a.relu = d->relu;
  a.TBY = (d->H + 15) / 16; a.TBX = (d->W + 15) / 16;
  a.wbytes = (unsigned)wbytes;
  a.stamps = g_c3_stamps;
  const int CT = bevf_conv3x3_bf16_ct(d->Cout);
  a.nct = d->Cout / CT;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // tile (64-channel tiles only): 0 = auto, 1 = two patch buffers (2 workgroups per CU), 2 = one (4 per CU), 3 = one buffer and 32-row
  // blocks (2 per CU).  Auto: tools/conv3x3_bench.py
  constexpr int lds128 = C3Geo<128, 2, 16>::LDS_BYTES, lds64 = C3Geo<64, 2, 16>::LDS_BYTES, lds64s = C3Geo<64, 1, 16>::LDS_BYTES,
                lds64t = C3Geo<64, 1, 32>::LDS_BYTES;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_bf16<64, 2, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, lds64);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_bf16<128, 2, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, lds128);
    attr_done = true;
  }
  const int variant = CT == 128 ? 1 : ((d->tile && d->tile < 4) ? d->tile : (d->Cin >= 128 ? 2 : C3_AUTO_SHORTK));
  if (variant == 3) { a.TBY = (d->H + 31) / 32; }
  const long long ntiles = (long long)d->N * a.TBY * a.TBX * a.nct;
  BEVF_REQUIRE(ntiles < (1ll << 31), "conv3x3_bf16: too many tiles");
  const dim3 grid((unsigned)ntiles), block(256);
  if (CT == 64 && d->Cin == 64 && (d->tile == 5 || (d->tile == 0 && C3_AUTO_WIDE64))) {      // both channel halves in one patch image
    static bool wattr = false;
    constexpr int ldsw = 4 * 12 * 1024 + 8 * 4096;
    if (!wattr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_bf16_wide64<64>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsw);
      wattr = true;
    }
    a.TBY = (d->H + 15) / 16;
    const long long nt16 = (long long)d->N * a.TBY * a.TBX * a.nct;
    hipLaunchKernelGGL((conv3x3_bf16_wide64<64>), dim3((unsigned)nt16), block, ldsw, st, a);
    return bevf_check_launch("bevf_conv3x3_bf16");
  }
  if (CT == 64 && (d->tile == 4 || (d->tile == 0 && C3_AUTO_PERSIST && d->Cin < 128))) {                // persistent form (two patch buffers, 16-row blocks)
    static bool pattr = false;
    if (!pattr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_bf16_persist<64>), hipFuncAttributeMaxDynamicSharedMemorySize, lds64);
      pattr = true;
    }
    a.TBY = (d->H + 15) / 16;
    const long long nt16 = (long long)d->N * a.TBY * a.TBX * a.nct;
    const dim3 pgrid((unsigned)(nt16 < 512 ? nt16 : 512));          // 2 workgroups per CU
    hipLaunchKernelGGL((conv3x3_bf16_persist<64>), pgrid, block, lds64, st, a, (int)nt16);
    return bevf_check_launch("bevf_conv3x3_bf16");
  }
  if (CT == 128) hipLaunchKernelGGL((conv3x3_bf16<128, 2, 16>), grid, block, lds128, st, a);
  else if (variant == 3) hipLaunchKernelGGL((conv3x3_bf16<64, 1, 32>), grid, block, lds64t, st, a);
  else if (variant == 2) hipLaunchKernelGGL((conv3x3_bf16<64, 1, 16>), grid, block, lds64s, st, a);
  else hipLaunchKernelGGL((conv3x3_bf16<64, 2, 16>), grid, block, lds64, st, a);
  return bevf_check_launch("bevf_conv3x3_bf16");
}
