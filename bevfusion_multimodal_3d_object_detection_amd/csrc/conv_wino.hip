// fp32 Winograd F(2x2,3x3) convolution (3x3, stride 1, pad 1) on v_mfma_f32_16x16x4_f32, gfx950 -- fully fused:
// input transform in registers from a raw LDS patch, the 16 "frequency" GEMMs on the matrix pipe, output transform
// in registers, the same fused epilogue as conv_igemm (folded BN scale/shift, residual, ReLU, channel-strided store).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A      per 2x2 output tile, summed over input channels
//
// 16 multiplies per 4 outputs instead of 36: 2.25x fewer MFMA FLOPs than the implicit GEMM, and the fp32 MFMA pipe is
// what bounds these layers.  The products are fp32 FMAs with fp32 accumulation like the direct kernel;
// what changes is the summation structure, so results agree with it to a few 1e-7 relative, not bit for bit.
//
// Workgroup = 256 threads = 4 waves = 8x8 Winograd tiles (16x16 output pixels) x 64 output channels.  Wave w owns
// tile rows 2w, 2w+1 (16 tiles) x 64 channels x all 16 frequencies: 64 accumulator tiles of 16x16 = all 256 AGPRs
// (one wave per SIMD).  K loop over groups of 8 input channels = 2 MFMA k-steps of 128 MFMAs:
//   * A side: the 18x18-pixel input patch sits in LDS as raw NHWC rows (32-channel chunks, double-buffered, staged
//     global -> registers -> LDS with out-of-range pixels answered by the buffer unit with zeros); lane (t = l&15,
//     kq = l>>4) reads its tile's 4x4 patch for channels 2kq, 2kq+1 (16 ds_read_b64) and transforms it (64 VALU)
//     into the A fragments of all 16 frequencies, one group ahead of their use;
//   * B side: the transformed filters are stored in HBM in the exact LDS image order per (64-channel slab, group),
//     so staging is a straight 32 KB LDS-DMA copy (global_load_lds), double-buffered; fragments are ds_read_b128,
//     bank-conflict-free.
// With one wave per SIMD nothing else fills an issue gap, so every non-MFMA instruction is pinned BETWEEN two MFMAs
// (one small piece per gap, sched_barrier after each) and issues in the shadow of the MFMA executing.  The MFMAs are
// inline asm with "+a" accumulators: with the builtin, hipcc's allocator moved accumulators between AGPRs and VGPRs
// inside the loop (44-296 v_accvgpr moves per group); pinned, the loop has none and no spills.
#include "conv_common.h"

#include <type_traits>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// a - b on two floats at once (same rounding as two v_sub_f32)
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

#define MFMA(acc_, a_, b_) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc_) : "v"(a_), "v"(b_))
#define MFMA0(acc_, a_, b_) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(acc_) : "v"(a_), "v"(b_))

struct WinoArgs {
  const float* x;      // NHWC, channel stride x_cs
  const float* u;      // [ct][G][16 f][2 nb pair][4 kq][16 n][2 nb][2 cin]: bevf_wino_filter_transform_f32
  const float* scale;  // [Cout] or null
  const float* shift;
  const float* res;    // NHWC residual or null
  float* y;
  float* stats;        // STATS: [nsp * 4][Cout][2] partial sums of (y - pivot), (y - pivot)^2;  BNB: of dy, dy * xhat
  const float* pivot;  // [Cout] or null
  // BNB (training backward): this conv's output is the gradient dY of a train-mode BatchNorm(+ReLU) layer; the epilogue applies
  // that layer's ReLU mask and leaves its backward sums (csrc/norm_train.hip: bn_bwd_partials) -- the layer's own raw
  // input bnb_x [pixels][Cout], its output bnb_y (mask source when the layer had a residual; else the mask is recomputed
  // from bnb_x with bn_apply's fma), and its per-channel mean / invstd / gamma / beta
  const float *bnb_x, *bnb_y, *bnb_mean, *bnb_invstd, *bnb_gamma, *bnb_beta;
  int N, H, W, Cin, x_cs, Cout, y_cs, res_cs;
  int TBY, TBX, nct;   // tile blocks per image (rows, cols), 64-channel slabs
  // Stacked rows (HS > 0): the N images are treated as ONE map of N * HS rows -- image n at rows n*HS .. n*HS+H-1, the HS - H rows
  // between two images dead (they read as zeros: they ARE the padding row both neighbours need) -- cut into SB block rows that
  // ignore image boundaries.  A 57-row map then wastes 1 row in 58 instead of 7 in 64.  HS is even (a tile's 2x2 alignment inside
  // its image, and with it every rounding, is what it is in the per-image tiling: the two are bit-identical) and > block height
  // (a patch spans at most two images).  HS = 0: block rows per image, SB = N * TBY.
  int HS, SB;
  unsigned div_mul;    // ceil(2^32 / HS) (stacked) or ceil(2^32 / TBY) (0 for TBY = 1): row -> image by one s_mul_hi (the host checks the range)
  unsigned long long* stamps;   // DIAG instantiations only (bevf_debug_wino_stamps): 5 x s_memtime per workgroup
};

constexpr int PITCH = 36;                                      // floats per patch pixel in LDS (32 + 4 pad); 18x18 or 34x10 pixels
constexpr int PDMA = 12;                                       // LDS-DMA instructions per wave and chunk (4 waves x 12 x 64 slots)
constexpr int PATCH_FLOATS = 4 * PDMA * 64 * 4;                // 12288 floats = 48 KiB: 9 16-byte slots per pixel (8 data + 1 pad), 2916 / 3060 slots rounded up to 48 x 64
constexpr int BG_FLOATS = 16 * 4 * 16 * 8;                     // 8192 floats = 32 KB per channel group
constexpr int LDS_BYTES = (2 * PATCH_FLOATS + 2 * BG_FLOATS) * 4;

// GEO = block geometry: 0 = 8x8 tiles (16x16 pixels; wave w: tile rows 2w, 2w+1), 1 = 16x4 tiles (32 rows x 8 columns; wave w:
// tile rows 4w .. 4w+3) -- the host takes whichever covers the map with fewer blocks (57x100: 28 -> 26, 113x200: 104 -> 100).
template <bool RES, bool RELU, bool STATS = false, int BNB = 0, int GEO = 0, bool DIAG = false>
__global__ __launch_bounds__(256, 1) void wino_f32(const WinoArgs p) {
  auto stamp = [&](int i) {                                         // diagnostic launches only; the buffer is read by nothing else
    if constexpr (DIAG) {
      if (threadIdx.x == 0)
        p.stamps[(size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 5 + i] = __builtin_amdgcn_s_memtime();
    }
  };
  stamp(0);
  constexpr int BHP = GEO ? 32 : 16, BWP = GEO ? 8 : 16;          // block height / width in pixels
  constexpr int PWP = BWP + 2, PIXG = (BHP + 2) * PWP;            // patch width, patch pixels (324 or 340)
  constexpr int PSLOTSG = PIXG * 9;
  static_assert(PSLOTSG <= 4 * PDMA * 64, "patch does not fit the DMA image");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const patch = lds;                                     // [2][patch pixels][PITCH] (padded to PATCH_FLOATS)
  float* const bbuf = lds + 2 * PATCH_FLOATS;                   // [2][BG_FLOATS]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = p.Cin >> 3;                                     // channel groups of 8
  const int NCH = p.Cin >> 5;                                   // patch chunks of 32 channels
  // grid = (TBX, SB, nct): dispatch order x, y, z = spatial blocks of one 64-channel slab first, so concurrent workgroups share one slab
  // of transformed filters (L2-resident); no index arithmetic to decode a tile
  // ---- patch staging by LDS-DMA (buffer_load ... lds: an out-of-range lane writes zeros -- the pad ring, the pad slot of
  //      every pixel, the slots past the patch): instruction (4 j + wave), j < 12, fills 64 consecutive 16-byte slots;
  //      slot s = pixel s / 9, piece s % 9 (8 = pad) --------------------------------------------------------------------
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)kOob, 0x00020000);
  // n / rb: image and in-image row of the block's first pixel row; hw: rows after which a row index wraps into the next image
  // The 12 slot offsets of a wave are what stands between the workgroup's start and its first HBM request, beside fp32 MFMAs that
  // share the vector ALU with them: 24-bit multiplies only (full rate; every factor is < 2^24: a pixel index because the tensor stays
  // below 2 GiB with >= 32 channels of 4 bytes), no division (3D grid, one s_mul_hi for the image of a row).
  auto make_pv = [&](unsigned (&pv)[PDMA], int& n, int& rb, int& hw, int& bx, int& ct) {
    bx = blockIdx.x;
    const int sp = blockIdx.y;
    ct = blockIdx.z;
    int n0, r0;                                                     // image / in-image row of the patch's first row (block row - 1)
    if (p.HS) {
      const int s0 = BHP * sp - 1 + p.HS, q0 = (int)__umulhi((unsigned)s0, p.div_mul);   // s0 / HS (+ HS: the division never sees -1)
      n0 = q0 - 1;
      r0 = s0 - q0 * p.HS;
      hw = p.HS;
      const bool top = r0 + 1 == p.HS;                              // the patch's first row is the last (dead) row of image n0
      n = top ? q0 : n0;
      rb = top ? 0 : r0 + 1;
    } else {
      n0 = p.div_mul ? (int)__umulhi((unsigned)sp, p.div_mul) : sp;   // sp / TBY (TBY = 1: the multiplier 2^32 is passed as 0)
      rb = BHP * (sp - n0 * p.TBY);
      r0 = rb - 1;
      hw = 0x7fffffff;
      n = n0;
    }
    const int ix0 = BWP * bx - 1;
    const unsigned nh0 = (unsigned)(n0 * p.H), xcs4 = (unsigned)(p.x_cs * 4);
#pragma unroll
    for (int j = 0; j < PDMA; ++j) {
      const unsigned sl = (unsigned)((4 * j + wave) * 64 + lane);
      const unsigned pix = __umul24(sl, 7282u) >> 16, piece = sl - 9u * pix;                    // sl / 9 for sl < 3072
      const unsigned py = __umul24(pix, GEO ? 6554u : 3641u) >> 16, px = pix - __umul24(py, (unsigned)PWP);   // pix / 10 (or / 18) for pix < 1024
      int iy = r0 + (int)py;
      const int ix = ix0 + (int)px;
      const bool wr = iy >= hw;
      iy -= wr ? hw : 0;
      const bool ok = sl < (unsigned)PSLOTSG && piece < 8u && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W &&
                      (unsigned)(n0 + (wr ? 1 : 0)) < (unsigned)p.N;
      const unsigned row = nh0 + (wr ? (unsigned)p.H : 0u) + (unsigned)iy;                      // image row index over the batch
      pv[j] = ok ? __umul24(__umul24(row, (unsigned)p.W) + (unsigned)ix, xcs4) + 16u * piece : kOob;
    }
  };
  auto patch_dma_from = [&](__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, int buf, int j) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(patch + buf * PATCH_FLOATS + (4 * j + wave) * 256),
                                             16, voff, soff, 0, 0);
  };
  auto patch_dma = [&](unsigned voff, unsigned soff, int buf, int j) { patch_dma_from(rsx, voff, soff, buf, j); };
  // RES: in a tile's LAST chunk the "next chunk" slots have nothing to fetch; two of them (group 0's first two) then touch one 16-byte
  // piece of every 128-byte line of this wave's residual pixels -- the data lands in the idle patch buffer and is never read, but the
  // lines are in L2 when the epilogue asks for them ~3 groups later (the epilogue waited 4.7k cycles for HBM here: tools/wino_stamps.py)
  const __amdgpu_buffer_rsrc_t rsr_pf = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(RES ? p.res : p.x), 0, (int)kOob, 0x00020000);
  bool pf_res = false;                                              // group 0's part of the patch comes through rsr_pf

  // ---- B staging by LDS-DMA: group image = 32 pieces of 1 KiB; wave w issues pieces 8w .. 8w+7 ----------------------
  // (buffer form: the per-lane part of the address is one constant VGPR, the image / piece offset a scalar -- the global
  //  form cost a 64-bit vector add per piece, and the fp32 MFMA shares the vector ALU)
  const __amdgpu_buffer_rsrc_t rsu = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, (int)kOob, 0x00020000);
  const unsigned u_lane = (unsigned)((wave * 8 * 256 + lane * 4) * 4);
  auto dma_piece = [&](unsigned img, int buf, int i) {              // img: byte offset of the group image in p.u
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsu, (__attribute__((address_space(3))) void*)(bbuf + buf * BG_FLOATS + wave * 8 * 256 + i * 256),
                                             16, u_lane, img + i * 1024, 0, 0);
  };

  // ---- compute roles --------------------------------------------------------------------------------------------
  const int t = lane & 15, kq = lane >> 4;
  const int ty = GEO ? 4 * wave + (t >> 2) : 2 * wave + (t >> 3), tx = GEO ? (t & 3) : (t & 7);
  const int a_lane = ((2 * ty) * PWP + 2 * tx) * PITCH + 2 * kq;    // floats: patch pixel (2ty, 2tx), channels 2kq..
  const int b_lane = kq * 64 + t * 4;                              // floats within a [4 kq][16 n][2 nb][2 cin] block: lane * 16 bytes, linear

  f32x4 acc[16][4];
#ifndef WINO_ONE_CLUSTER
#define WINO_ONE_CLUSTER 1
#endif
  // A fragments: two sets (WINO_ONE_CLUSTER: group g multiplies from set g & 1 while the WHOLE input transform of group g + 1 is written
  // into the other set in ONE MFMA gap -- a lone VALU instruction between fp32 MFMAs costs ~15 cycles, the ones right behind it ~3
  // (probe), so 32 packed adds cost ~120 cycles as one cluster and ~250 as the twelve they used to be)
  f32x2 vv[WINO_ONE_CLUSTER ? 2 : 1][16], dn[16];
  f32x2 (&v)[16] = vv[0];
  auto load_patch = [&](int buf, int gl, int q) {                  // row q of the 4x4 patch -> dn[4q..4q+3]
    const float* pa = patch + buf * PATCH_FLOATS + a_lane + gl * 8;
#pragma unroll
    for (int c = 0; c < 4; ++c) dn[4 * q + c] = *reinterpret_cast<const f32x2*>(pa + (q * PWP + c) * PITCH);
  };
  auto rows_col = [&](int c) {                                      // B^T d, column c (in place)
    const f32x2 t0 = dn[0 + c] - dn[8 + c], t1 = dn[4 + c] + dn[8 + c], t2 = dn[8 + c] - dn[4 + c], t3 = dn[4 + c] - dn[12 + c];
    dn[0 + c] = t0; dn[4 + c] = t1; dn[8 + c] = t2; dn[12 + c] = t3;
  };
  auto cols_row = [&](int r) {                                      // (B^T d) B, row r -> v[4r..4r+3]
    v[4 * r + 0] = dn[4 * r + 0] - dn[4 * r + 2];
    v[4 * r + 1] = dn[4 * r + 1] + dn[4 * r + 2];
    v[4 * r + 2] = dn[4 * r + 2] - dn[4 * r + 1];
    v[4 * r + 3] = dn[4 * r + 1] - dn[4 * r + 3];
  };

  // ---- prologue: first tile's patch chunk 0, B group 0, A fragments of group 0 -----------------------------------------
  unsigned pv[PDMA], pvl[PDMA];
  int n, rb, hw, bx, ct;
  {                                                                 // filter DMA first: it flies while the slot offsets are computed
    const int ct0 = blockIdx.z;
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_piece((unsigned)(ct0 * G) * (BG_FLOATS * 4), 0, i);
  }
  make_pv(pv, n, rb, hw, bx, ct);
#pragma unroll
  for (int j = 0; j < PDMA; ++j) patch_dma(pv[j], 0, 0, j);
  // Staging schedule (filter image of group g+1 and a third of a later patch chunk per group, everything by LDS-DMA, no
  // register staging and no ds_write in the loop): chunk c+1's patch is fetched in three parts, in the last group of
  // chunk c-1 and groups 0, 1 of chunk c, always AFTER the group's filter pieces.  Loads retire in order, so the
  // hand-written wait at a group's end, vmcnt(4), names exactly the filter image and leaves the newest patch part in
  // flight: every patch load has at least a whole group (1.7 us) to land.  (A __syncthreads() waits vmcnt(0), and hipcc
  // adds vmcnt(0) before any ds_write that follows an LDS-DMA: in-kernel ablation put 13 % of the kernel on those waits.)
#pragma unroll
  for (int j = 0; j < 4; ++j) patch_dma(NCH > 1 ? pv[j] : kOob, 128, 1, j);
  asm volatile("s_waitcnt vmcnt(4)");
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) load_patch(0, 0, q);
#pragma unroll
  for (int c = 0; c < 4; ++c) rows_col(c);
#pragma unroll
  for (int r = 0; r < 4; ++r) cols_row(r);
  int pbuf = 0;                                                     // patch buffer holding the current chunk
  stamp(1);

  // one channel group (8 channels = 2 MFMA k-steps); gl = position in the 32-channel patch chunk (static)
  //   pvl / psoff: where the NEXT chunk's patch comes from (this tile's next chunk, or the next tile's chunk 0)
  //   bimg: byte offset of the NEXT group's filter image in p.u (bnext false: nothing follows)
  auto group = [&](auto glc, auto firstc, const unsigned psoff, const unsigned bimg, const bool bnext) {
    constexpr int gl = decltype(glc)::value;
    constexpr bool kFirst = decltype(firstc)::value;      // a tile's first group: its k-step-0 MFMAs start the accumulators (C = 0)
    const int abuf = gl == 3 ? pbuf ^ 1 : pbuf;                     // A fragments of the next group: next chunk after gl 3
    constexpr int agl = (gl + 1) & 3;
    const float* pb = bbuf + (gl & 1) * BG_FLOATS + b_lane;
    const float* pa = patch + abuf * PATCH_FLOATS + a_lane + agl * 8;
    f32x4 b0[2], b1[2];
    b0[0] = *reinterpret_cast<const f32x4*>(pb);
    b0[1] = *reinterpret_cast<const f32x4*>(pb + 256);
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      f32x4 (&bc)[2] = (f & 1) ? b1 : b0;
      f32x4 (&bn)[2] = (f & 1) ? b0 : b1;
      const f32x2 a = vv[WINO_ONE_CLUSTER ? (gl & 1) : 0][f];
      // 8 MFMAs; after each one a small piece of the other work, pinned in place, so that every non-MFMA instruction
      // issues in the shadow of an executing MFMA (one wave per SIMD: nothing else would fill the gap)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int j = k >> 2, nb = k & 3;
        if (kFirst && j == 0) MFMA0(acc[f][nb], a[j], bc[nb >> 1][(nb & 1) * 2 + j]);
        else MFMA(acc[f][nb], a[j], bc[nb >> 1][(nb & 1) * 2 + j]);
        if (k == 0 && f < 15) bn[0] = *reinterpret_cast<const f32x4*>(pb + ((f + 1) * 2) * 256);
        if (k == 1 && f < 15) bn[1] = *reinterpret_cast<const f32x4*>(pb + ((f + 1) * 2 + 1) * 256);
        if (f < 4) {                                                // patch row f of the next group: 4 x b64
          if (k == 2) { dn[4 * f + 0] = *reinterpret_cast<const f32x2*>(pa + (f * PWP + 0) * PITCH);
                        dn[4 * f + 1] = *reinterpret_cast<const f32x2*>(pa + (f * PWP + 1) * PITCH); }
          if (k == 3) { dn[4 * f + 2] = *reinterpret_cast<const f32x2*>(pa + (f * PWP + 2) * PITCH);
                        dn[4 * f + 3] = *reinterpret_cast<const f32x2*>(pa + (f * PWP + 3) * PITCH); }
        } else if (WINO_ONE_CLUSTER) {
          if (f == 5 && k == 2) {                                   // patch rows landed (read at f < 4): B^T d B, all of it
            f32x2 (&vn)[16] = vv[WINO_ONE_CLUSTER ? ((gl + 1) & 1) : 0];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const f32x2 t0 = dn[0 + c] - dn[8 + c], t1 = dn[4 + c] + dn[8 + c];
              const f32x2 t2 = dn[8 + c] - dn[4 + c], t3 = dn[4 + c] - dn[12 + c];
              dn[0 + c] = t0; dn[4 + c] = t1; dn[8 + c] = t2; dn[12 + c] = t3;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              vn[4 * r + 0] = dn[4 * r + 0] - dn[4 * r + 2]; vn[4 * r + 1] = dn[4 * r + 1] + dn[4 * r + 2];
              vn[4 * r + 2] = dn[4 * r + 2] - dn[4 * r + 1]; vn[4 * r + 3] = dn[4 * r + 1] - dn[4 * r + 3];
            }
          }
        } else if (f < 8) {                                         // B^T d, column c = f - 4: four f32x2 ops
          const int c = f - 4;
          if (k == 2) { const f32x2 t0 = dn[0 + c] - dn[8 + c], t1 = dn[4 + c] + dn[8 + c];
                        const f32x2 t2 = dn[8 + c] - dn[4 + c], t3 = dn[4 + c] - dn[12 + c];
                        dn[0 + c] = t0; dn[4 + c] = t1; dn[8 + c] = t2; dn[12 + c] = t3; }
        } else if (f & 1) {                                         // (B^T d) B, row r: v[4r..4r+3] are dead by now
          const int r = (f - 9) >> 1;
          if (k == 2) { v[4 * r + 0] = dn[4 * r + 0] - dn[4 * r + 2]; v[4 * r + 1] = dn[4 * r + 1] + dn[4 * r + 2]; }
          if (k == 3) { v[4 * r + 2] = dn[4 * r + 2] - dn[4 * r + 1]; v[4 * r + 3] = dn[4 * r + 1] - dn[4 * r + 3]; }
        }
        // staging: the next group's filter image first, then this group's part of a later patch chunk (see above)
        if (f == 0 && bnext) dma_piece(bimg, (gl + 1) & 1, k);
        if (f == 1 && k >= 4 && gl != 2) {
          constexpr int part = gl == 3 ? 0 : gl + 1;
          if (RES && gl == 0 && pf_res) patch_dma_from(rsr_pf, pvl[part * 4 + (k - 4)], psoff, pbuf ^ 1, part * 4 + (k - 4));
          else patch_dma(pvl[part * 4 + (k - 4)], psoff, gl == 3 ? pbuf : pbuf ^ 1, part * 4 + (k - 4));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // group end: the filter image (and, after group 2, the whole next patch chunk) has landed; the patch part issued in
    // this group stays in flight.  Wait and barrier are ONE asm with a memory clobber (no LDS access moves across it):
    // through __syncthreads() or a workgroup fence hipcc waits vmcnt(0) here, the DMA being an LDS write to it.
    // (a chunk's last group also lets its MFMAs land: they are inline asm, unseen by hipcc's hazard pass, and hipcc puts accumulator
    //  copies / the epilogue's v_accvgpr_read right behind the loop)
    if constexpr (gl == 3) asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(gl == 2 ? 0 : 4) : "memory");
  };
  using G0 = std::integral_constant<int, 0>;
  using G1 = std::integral_constant<int, 1>;
  using G2 = std::integral_constant<int, 2>;
  using G3 = std::integral_constant<int, 3>;

  {                                                                 // one tile per workgroup
    using T = std::true_type;
    using F = std::false_type;
#pragma unroll
    for (int j = 0; j < PDMA; ++j) pvl[j] = pv[j];
    for (int chunk = 0; chunk < NCH; ++chunk) {                     // 32 channels
      const bool lastc = chunk + 1 == NCH;
      const unsigned ug = (unsigned)(ct * G + chunk * 4) * (BG_FLOATS * 4), gb = BG_FLOATS * 4;
      const unsigned soff = (unsigned)((chunk + 1) * 128);
      if (lastc) {                                                  // no next chunk: out-of-range VGPR offsets, the DMA writes zeros
#pragma unroll                                                      // into the idle buffer (the SGPR offset is not range-checked)
        for (int j = 4; j < PDMA; ++j) pvl[j] = kOob;
        if constexpr (RES) {                                        // ... except the two residual-prefetch slots (see rsr_pf)
          pf_res = true;
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const int line = s2 * 64 + lane, pw = line >> 1;        // this wave's 64 pixels x two 128-byte halves of their 64 channels
            const int prow = GEO ? (pw >> 3) : (pw >> 4), pcol = GEO ? (pw & 7) : (pw & 15);
            int ry = rb + (GEO ? 8 : 4) * wave + prow;
            const bool wr = ry >= hw;
            ry -= wr ? hw : 0;
            const int rn = n + (wr ? 1 : 0), rx = BWP * bx + pcol;
            const bool ok = rn < p.N && ry < p.H && rx < p.W && ct * 64 + (line & 1) * 32 < p.Cout;
            // (the group adds soff to every address of this part: taken off here; an offset below soff wraps out of range and that
            //  one line is simply not prefetched)
            pvl[4 + s2] = ok ? (unsigned)((((rn * p.H + ry) * p.W + rx) * p.res_cs + ct * 64 + (line & 1) * 32) * 4) - soff : kOob;
          }
        }
      }
      if (chunk + 2 >= NCH) {                                       // no chunk after next: the same for the part group 3 fetches
#pragma unroll
        for (int j = 0; j < 4; ++j) pvl[j] = kOob;
      }
      if (chunk == 0) group(G0{}, T{}, soff, ug + 1 * gb, true); else group(G0{}, F{}, soff, ug + 1 * gb, true);
      group(G1{}, F{}, soff, ug + 2 * gb, true);
      group(G2{}, F{}, soff, ug + 3 * gb, true);
      group(G3{}, F{}, soff + 128, ug + 4 * gb, !lastc);
      pbuf ^= 1;
    }

    asm volatile("s_waitcnt vmcnt(0)");                             // (the last, all-out-of-range patch part: LDS is reused below)
    stamp(2);
    if constexpr (STATS || BNB != 0) __syncthreads();              // ... by EVERY wave's zero-writing DMAs before any wave's partial sums land there
    // ---- epilogue: output transform per (tile, channel), scale/shift (+res) (+relu), store ----------------------------
    // acc[f][nb][r]: tile 4 kq + r of this wave = (tile row kq>>1, tile column 4 (kq&1) + r), channel ct*64 + nb*16 + (lane&15)
    {
      // (GEO 1: tile 4 kq + r = tile row kq of the wave's four, tile column r)
      // this lane's two pixel rows (oy even, HS even: both in the same image); a row may be dead (below the map, or the row between
      // two stacked images): its base offset is out of range and every access through it is dropped by the buffer hardware
      int oy = rb + (GEO ? 8 * wave + 2 * kq : 4 * wave + 2 * (kq >> 1));
      const int ox = GEO ? 8 * bx : 16 * bx + 8 * (kq & 1);
      const bool wr = oy >= hw;
      oy -= wr ? hw : 0;
      const int nl = n + (wr ? 1 : 0);
      const bool rok[2] = {nl < p.N && oy < p.H, nl < p.N && oy + 1 < p.H};
      const bool interior = BWP * bx + BWP <= p.W && ct * 64 + 64 <= p.Cout;
      const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)kOob, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res), 0, (int)kOob, 0x00020000);
      const unsigned y_lane = (unsigned)((((nl * p.H + oy) * p.W + ox) * p.y_cs + ct * 64 + t) * 4);
      const unsigned r_lane = (unsigned)((((nl * p.H + oy) * p.W + ox) * p.res_cs + ct * 64 + t) * 4);
      const unsigned yb[2] = {rok[0] ? y_lane : kOob, rok[1] ? y_lane + (unsigned)(p.W * p.y_cs * 4) : kOob};
      const unsigned rbs[2] = {rok[0] ? r_lane : kOob, rok[1] ? r_lane + (unsigned)(p.W * p.res_cs * 4) : kOob};
      // interior tile blocks (all columns and all 64 channels exist; rows go by the two lane bases): no per-element predicate at
      // all; the scalar part of every address is an SGPR offset.  Edge blocks: out-of-range elements get an out-of-range VGPR offset.
      auto emit = [&](auto interiorc) {
        constexpr bool kInt = decltype(interiorc)::value;
        float sc[4], sh[4];
        bool cok[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const int co = ct * 64 + nb * 16 + t;
          cok[nb] = kInt || co < p.Cout;
          sc[nb] = (p.scale && cok[nb]) ? p.scale[co] : 1.f;
          sh[nb] = (p.shift && cok[nb]) ? p.shift[co] : 0.f;
        }
        auto colok = [&](int nb, int q) {                            // q = r*4 + i*2 + j: pixel row i, column 2r + j of the lane's 2 x 8
          return kInt || (cok[nb] && ox + 2 * (q >> 2) + (q & 1) < p.W);
        };
        auto live = [&](int nb, int q) { return rok[(q >> 1) & 1] && colok(nb, q); };
        auto vo = [&](int nb, int q, const unsigned (&base)[2]) { return colok(nb, q) ? base[(q >> 1) & 1] : kOob; };
        auto soff_of = [&](int nb, int q, int cs) {                  // scalar byte offset of element q of channel block nb (within its row)
          return (unsigned)(((2 * (q >> 2) + (q & 1)) * cs + nb * 16) * 4);
        };
        float rv[2][16];                                             // residual: two channel blocks in flight
        float lx[BNB ? 2 : 1][16], ly[BNB == 2 ? 2 : 1][16];         // BNB: the BatchNorm layer's raw input / its output
        const __amdgpu_buffer_rsrc_t rbx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bnb_x), 0, (int)kOob, 0x00020000);
        const __amdgpu_buffer_rsrc_t rby = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bnb_y), 0, (int)kOob, 0x00020000);
        auto fetch = [&](int nb, int set) {                          // BNB: one channel block's operands, a block ahead of their use
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const unsigned vy = vo(nb, q, yb);                       // bnb_x / bnb_y are laid out like y (y_cs == Cout)
            lx[set][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbx, vy, soff_of(nb, q, p.y_cs), 0));
            if constexpr (BNB == 2)
              ly[set][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rby, vy, soff_of(nb, q, p.y_cs), 0));
            if constexpr (RES)
              rv[set][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                               rsr, vo(nb, q, rbs), soff_of(nb, q, p.res_cs), 0));
          }
        };
        auto res_fetch = [&](int nb, int set) {
#pragma unroll
          for (int q = 0; q < 16; ++q)
            rv[set][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsr, vo(nb, q, rbs), soff_of(nb, q, p.res_cs), 0));
        };
        if constexpr (BNB != 0) {
          fetch(0, 0);
        } else if constexpr (RES) {                                  // the residual of channel blocks 0, 1 now, of 2, 3 as their sets
          res_fetch(0, 0);                                           // free up (L2 hits: the last chunk prefetched the lines); 32
          res_fetch(1, 1);                                           // registers instead of 64
        }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (BNB != 0) {
            if (nb < 3) fetch(nb + 1, (nb + 1) & 1);
          }
          const int set = nb & 1;
          float st1 = 0.f, st2 = 0.f;                                // STATS / BNB: this lane's sums for channel nb*16 + t
          const int cch = ct * 64 + nb * 16 + t;
          const float pvt = (STATS && p.pivot && cok[nb]) ? p.pivot[cch] : 0.f;
          float mu = 0.f, is = 1.f, fa = 1.f, fb = 0.f;
          if constexpr (BNB != 0) {
            if (cok[nb]) {
              mu = p.bnb_mean[cch];
              is = p.bnb_invstd[cch];
              fa = (p.bnb_gamma ? p.bnb_gamma[cch] : 1.f) * is;
              fb = (p.bnb_beta ? p.bnb_beta[cch] : 0.f) - mu * fa;
            }
          }
          // A^T M over the frequency rows, all four tiles r at once -- as two-float halves with the subtractions as explicit packed
          // instructions: left to hipcc every vector subtraction here became scalar v_sub_f32 (192 per epilogue beside fp32 MFMAs' shared pipe)
          f32x2 sr[2][4][2];
#pragma unroll
          for (int nu = 0; nu < 4; ++nu)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              const f32x2 m0 = {acc[0 + nu][nb][2 * hh], acc[0 + nu][nb][2 * hh + 1]}, m1 = {acc[4 + nu][nb][2 * hh], acc[4 + nu][nb][2 * hh + 1]};
              const f32x2 m2 = {acc[8 + nu][nb][2 * hh], acc[8 + nu][nb][2 * hh + 1]}, m3 = {acc[12 + nu][nb][2 * hh], acc[12 + nu][nb][2 * hh + 1]};
              sr[0][nu][hh] = m0 + m1 + m2;
              sr[1][nu][hh] = pk_sub(pk_sub(m1, m2), m3);
            }
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            f32x2 yh[2][2];                                          // [j][half]
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              yh[0][hh] = sr[i][0][hh] + sr[i][1][hh] + sr[i][2][hh];
              yh[1][hh] = pk_sub(pk_sub(sr[i][1][hh], sr[i][2][hh]), sr[i][3][hh]);
            }
            const f32x2 sc2 = {sc[nb], sc[nb]}, sh2 = {sh[nb], sh[nb]};
            f32x2 of[2][2];                                          // folded BatchNorm on pairs (v_pk_fma_f32)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int hh = 0; hh < 2; ++hh) of[j][hh] = __builtin_elementwise_fma(yh[j][hh], sc2, sh2);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int q = r * 4 + i * 2 + j;
                float o = of[j][r >> 1][r & 1];
                if constexpr (RES) o += rv[set][q];
                if constexpr (RELU) o = fmaxf(o, 0.f);
                if constexpr (BNB != 0) {                             // the consumer's ReLU mask, then its backward sums
                  const float tv = BNB == 2 ? ly[set][q] : fmaf(lx[set][q], fa, fb);
                  o = tv > 0.f ? o : 0.f;
                  if (live(nb, q)) {
                    st1 += o;
                    st2 = fmaf(o, (lx[set][q] - mu) * is, st2);
                  }
                }
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rsy, vo(nb, q, yb), soff_of(nb, q, p.y_cs), 0);
                if constexpr (STATS) {
                  const float dv = live(nb, q) ? o - pvt : 0.f;
                  st1 += dv;
                  st2 = fmaf(dv, dv, st2);
                }
              }
          }
          if constexpr (RES && BNB == 0) {
            if (nb < 2) res_fetch(nb + 2, set);
          }
          if constexpr (STATS || BNB != 0) {                        // the four lane groups kq hold the same channel: fixed xor tree
            st1 += __shfl_xor(st1, 16); st2 += __shfl_xor(st2, 16);
            st1 += __shfl_xor(st1, 32); st2 += __shfl_xor(st2, 32);
            if (kq == 0) {                                          // per-wave sums -> LDS (the patch buffers are dead by now)
              lds[(wave * 64 + nb * 16 + t) * 2] = st1;
              lds[(wave * 64 + nb * 16 + t) * 2 + 1] = st2;
            }
          }
        }
      };
      if (interior) emit(std::true_type{}); else emit(std::false_type{});
      if constexpr (DIAG) {
        stamp(3);                                                   // everything issued ...
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(4);                                                   // ... and every store acknowledged
      }
      if constexpr (STATS || BNB != 0) {                            // one partial row per tile block: the four waves' sums in fixed order
        __syncthreads();
        if (tid < 64 && ct * 64 + tid < p.Cout) {
          float a1 = 0.f, a2 = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) { a1 += lds[(w * 64 + tid) * 2]; a2 += lds[(w * 64 + tid) * 2 + 1]; }
          float* dst = p.stats + ((size_t)(blockIdx.y * p.TBX + blockIdx.x) * p.Cout + ct * 64 + tid) * 2;
          dst[0] = a1;
          dst[1] = a2;
        }
      }
    }
  }
}


// OHWI filters [Cout][3][3][Cin] -> U image [ct][G][16 f][2 np][4 kq][16 n][2 nbl][2 cin]; U = G g G^T in double
__global__ __launch_bounds__(256) void wino_filter_transform(const float* __restrict__ w, float* __restrict__ u, int Cout,
                                                             int Cin, int nct) {
  const long long idx = blockIdx.x * 256ll + threadIdx.x;
  const long long total = (long long)nct * 64 * Cin;
  if (idx >= total) return;
  const int ci = (int)(idx % Cin), co = (int)(idx / Cin);
  double g[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) g[i][j] = co < Cout ? (double)w[(((size_t)co * 3 + i) * 3 + j) * Cin + ci] : 0.0;
  const double Gm[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
  double tmp[4][3], U[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) tmp[i][j] = Gm[i][0] * g[0][j] + Gm[i][1] * g[1][j] + Gm[i][2] * g[2][j];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) U[i][j] = tmp[i][0] * Gm[j][0] + tmp[i][1] * Gm[j][1] + tmp[i][2] * Gm[j][2];
  const int G = Cin >> 3, ct = co >> 6, nb = (co & 63) >> 4, nn = co & 15, g8 = ci >> 3, q = ci & 7;
#pragma unroll
  for (int f = 0; f < 16; ++f)
    u[((((size_t)ct * G + g8) * 16 + f) * 2 + (nb >> 1)) * 256 + ((q >> 1) * 16 + nn) * 4 + (nb & 1) * 2 + (q & 1)] =
        (float)U[f >> 2][f & 3];
}

}  // namespace

static unsigned long long* g_wino_stamps = nullptr;
// Diagnostic (tools/wino_stamps.py): buf = device buffer of 5 x 8 bytes per workgroup of the NEXT plain launches (relu, optional residual),
// or null to stop
extern "C" int bevf_debug_wino_stamps(void* buf) {
  g_wino_stamps = static_cast<unsigned long long*>(buf);
  return BEVF_OK;
}

extern "C" int bevf_wino_stat_rows(int N, int H, int W) { return N * ((H + 15) / 16) * ((W + 15) / 16); }

extern "C" size_t bevf_wino_filter_floats(int Cout, int Cin) {
  return (size_t)((Cout + 63) / 64) * 64 * 16 * (size_t)Cin;
}

extern "C" int bevf_wino_filter_transform_f32(const float* w_ohwi, float* u, int Cout, int Cin, void* stream) {
  BEVF_REQUIRE(w_ohwi && u, "wino_filter_transform: null pointer");
  BEVF_REQUIRE(Cout > 0 && Cin > 0 && Cin % 32 == 0, "wino_filter_transform: Cin=%d must be a positive multiple of 32", Cin);
  const int nct = (Cout + 63) / 64;
  const long long total = (long long)nct * 64 * Cin;
  hipLaunchKernelGGL(wino_filter_transform, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     w_ohwi, u, Cout, Cin, nct);
  return bevf_check_launch("bevf_wino_filter_transform_f32");
}

extern "C" int bevf_conv3x3_wino_f32(const bevf_conv_desc* d, void* stream) {
  BEVF_REQUIRE(d && d->x && d->w && d->y, "conv_wino: null x / w / y");
  BEVF_REQUIRE(d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1, "conv_wino: 3x3, stride 1, pad 1 only (got %dx%d s%d p%d)",
               d->KH, d->KW, d->stride, d->pad);
  BEVF_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cout > 0 && d->Ho == d->H && d->Wo == d->W, "conv_wino: bad shape");
  BEVF_REQUIRE(d->Cin > 0 && d->Cin % 32 == 0, "conv_wino: Cin=%d must be a positive multiple of 32", d->Cin);
  BEVF_REQUIRE(d->x_cs >= d->Cin && d->x_cs % 4 == 0 && d->y_cs >= d->Cout, "conv_wino: channel strides");
  BEVF_REQUIRE(!d->res || d->res_cs >= d->Cout, "conv_wino: res_cs < Cout");
  BEVF_REQUIRE(!d->colmax, "conv_wino: the fused column max belongs to the 1x1 layers (bevf_conv2d_nhwc_f32)");
  BEVF_REQUIRE(bevf_aligned16(d->x) && bevf_aligned16(d->w), "conv_wino: x / w must be 16-byte aligned");
  BEVF_REQUIRE((long long)d->N * d->H * d->W * d->x_cs * 4 < (1ll << 31) && (long long)d->N * d->H * d->W * d->y_cs * 4 < (1ll << 31) &&
                   (!d->res || (long long)d->N * d->H * d->W * d->res_cs * 4 < (1ll << 31)),
               "conv_wino: activations must stay below 2 GiB (32-bit buffer offsets)");
  BEVF_REQUIRE((long long)((d->Cout + 63) / 64) * 64 * 16 * d->Cin * 4 < (1ll << 31), "conv_wino: transformed filters must stay below 2 GiB");
  WinoArgs a;
  a.x = d->x; a.u = d->w; a.scale = d->scale; a.shift = d->shift; a.res = d->res; a.y = d->y;
  a.stats = d->stats; a.pivot = d->stats_pivot;
  a.bnb_x = d->bnb_x; a.bnb_y = d->bnb_y; a.bnb_mean = d->bnb_mean; a.bnb_invstd = d->bnb_invstd;
  a.bnb_gamma = d->bnb_gamma; a.bnb_beta = d->bnb_beta;
  if (d->bnb_x) {
    BEVF_REQUIRE(d->stats && d->bnb_mean && d->bnb_invstd && !d->relu && !d->stats_pivot && d->y_cs == d->Cout,
                 "conv_wino: the BatchNorm-backward epilogue needs stats (partials out), mean, invstd, relu = 0 and y_cs == Cout");
  } else {
    BEVF_REQUIRE(!d->stats || (!d->res && !d->relu), "conv_wino: stats need relu = 0 and no residual (they describe the raw conv output)");
  }
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.x_cs = d->x_cs; a.Cout = d->Cout; a.y_cs = d->y_cs; a.res_cs = d->res_cs;
  a.TBY = (d->H + 15) / 16; a.TBX = (d->W + 15) / 16; a.nct = (d->Cout + 63) / 64;
  a.HS = 0; a.SB = d->N * a.TBY; a.stamps = nullptr;
  // Block geometry (16x16 or 32x8 pixels) and row stacking (WinoArgs::HS): whichever covers the batch with the fewest blocks; plain
  // epilogues only (the statistics rows are per 16x16 block of one image).  tile: 0 = auto, 1 / 2 = 16x16 / 32x8 per image,
  // 3 / 4 = the same two over stacked rows (tests: all four are bit-identical)
  bool geo1 = false;
  if (!d->stats && !d->bnb_x && d->tile != 1) {
    const int hs = d->H + 1 + ((d->H + 1) & 1);                     // even, >= H + 1
    long long best = (long long)a.SB * a.TBX;
    for (int cand = 1; cand < 4; ++cand) {                          // bit 0: 32x8 blocks, bit 1: stacked rows
      const int bh = (cand & 1) ? 32 : 16, bw = (cand & 1) ? 8 : 16, tbx = (d->W + bw - 1) / bw;
      if ((cand & 2) && (hs <= bh || (long long)d->N * hs >= (1ll << 30))) continue;
      const long long sb = (cand & 2) ? ((long long)d->N * hs - (hs - d->H) + bh - 1) / bh : (long long)d->N * ((d->H + bh - 1) / bh);
      const bool forced = d->tile == cand + 1;
      if (d->tile != 0 && !forced) continue;
      if (forced || sb * tbx < best) {
        best = sb * tbx;
        geo1 = cand & 1;
        a.TBY = (d->H + bh - 1) / bh; a.TBX = tbx; a.SB = (int)sb; a.HS = (cand & 2) ? hs : 0;
      }
    }
  }
  const long long ntiles = (long long)a.SB * a.TBX * a.nct;
  BEVF_REQUIRE(ntiles < (1ll << 31) && a.SB < 65536 && a.nct < 65536, "conv_wino: too many tiles");
  {                                                                 // row -> image by multiply-high: exact while (rows + divisor) * divisor < 2^32
    const unsigned long long dv = a.HS ? (unsigned long long)a.HS : (unsigned long long)a.TBY;
    const unsigned long long top = a.HS ? (unsigned long long)a.SB * (geo1 ? 32 : 16) + dv : (unsigned long long)a.SB;
    BEVF_REQUIRE((top + dv) * dv < (1ull << 32), "conv_wino: map too tall for the row arithmetic");
    a.div_mul = (unsigned)(((1ull << 32) + dv - 1) / dv);
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_f32<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_f32<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_f32<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_f32<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_f32<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_done = true;
  }
  const dim3 grid((unsigned)a.TBX, (unsigned)a.SB, (unsigned)a.nct), block(256);
  if (d->bnb_x) {
    static bool bnb_attr = false;
    if (!bnb_attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_f32<false, false, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_f32<false, false, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_f32<true, false, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_f32<true, false, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      bnb_attr = true;
    }
    if (d->res) {
      if (d->bnb_y) hipLaunchKernelGGL((wino_f32<true, false, false, 2>), grid, block, LDS_BYTES, st, a);
      else hipLaunchKernelGGL((wino_f32<true, false, false, 1>), grid, block, LDS_BYTES, st, a);
    } else {
      if (d->bnb_y) hipLaunchKernelGGL((wino_f32<false, false, false, 2>), grid, block, LDS_BYTES, st, a);
      else hipLaunchKernelGGL((wino_f32<false, false, false, 1>), grid, block, LDS_BYTES, st, a);
    }
    return bevf_check_launch("bevf_conv3x3_wino_f32");
  }
  if (d->stats) {
    hipLaunchKernelGGL((wino_f32<false, false, true>), grid, block, LDS_BYTES, st, a);
    return bevf_check_launch("bevf_conv3x3_wino_f32");
  }
  if (g_wino_stamps && d->relu) {                                  // diagnostic launch: same kernel, five time stamps per workgroup
    a.stamps = g_wino_stamps;
    auto go = [&](auto kern) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      hipLaunchKernelGGL(kern, grid, block, LDS_BYTES, st, a);
    };
    if (d->res) { if (geo1) go(&wino_f32<true, true, false, 0, 1, true>); else go(&wino_f32<true, true, false, 0, 0, true>); }
    else { if (geo1) go(&wino_f32<false, true, false, 0, 1, true>); else go(&wino_f32<false, true, false, 0, 0, true>); }
    return bevf_check_launch("bevf_conv3x3_wino_f32");
  }
  if (geo1) {
    static bool attr1 = false;
    if (!attr1) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_f32<false, false, false, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_f32<false, true, false, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_f32<true, false, false, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_f32<true, true, false, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      attr1 = true;
    }
    if (d->res) {
      if (d->relu) hipLaunchKernelGGL((wino_f32<true, true, false, 0, 1>), grid, block, LDS_BYTES, st, a);
      else hipLaunchKernelGGL((wino_f32<true, false, false, 0, 1>), grid, block, LDS_BYTES, st, a);
    } else {
      if (d->relu) hipLaunchKernelGGL((wino_f32<false, true, false, 0, 1>), grid, block, LDS_BYTES, st, a);
      else hipLaunchKernelGGL((wino_f32<false, false, false, 0, 1>), grid, block, LDS_BYTES, st, a);
    }
    return bevf_check_launch("bevf_conv3x3_wino_f32");
  }
  if (d->res) {
    if (d->relu) hipLaunchKernelGGL((wino_f32<true, true>), grid, block, LDS_BYTES, st, a);
    else hipLaunchKernelGGL((wino_f32<true, false>), grid, block, LDS_BYTES, st, a);
  } else {
    if (d->relu) hipLaunchKernelGGL((wino_f32<false, true>), grid, block, LDS_BYTES, st, a);
    else hipLaunchKernelGGL((wino_f32<false, false>), grid, block, LDS_BYTES, st, a);
  }
  return bevf_check_launch("bevf_conv3x3_wino_f32");
}
