// ResNet stem: 7x7 stride-2 pad-3 convolution of a planar 3-channel image, fused BN + ReLU,
// on v_mfma_f32_32x32x2_f32.  ref src/encoders.py:154-156.
//
// One workgroup = TH = 4 output rows x 128 pixels x all 64 channels.  The 3 x (2*TH+5) input
// rows it needs are staged once in LDS straight from the NCHW image (16-B loads along W,
// zero-filled outside the image); the whole filter bank sits beside it as [k][channel] (packed
// once by the host) so the B-operand reads are conflict-free, and is amortised over the rows.
// The A operand is read element-wise out of the patch:
//     A[pixel i of row ro][k=(c,kh,kw)] = patch[c][2*ro + kh][2*i + kw + 1]
// (patch column 0 is image column 2*ow0 - 4, which keeps the 16-B loads aligned).
//
// v_mfma_f32_32x32x2_f32 consumes two k per step (lane half h takes one each) and runs at the
// fp32 VALU rate, so the k loop must not spend VALU on addresses.  The 147 taps are therefore
// paired so that the h=1 tap is the h=0 tap shifted by a constant: 63 pairs one patch row apart
// (kh, kh+1), 9 pairs one column apart in the kh=6 row, one pair a channel apart, and the last
// tap alone against a zero filter row.  Each step's addresses are then "per-group base VGPR
// (holding the h shift) + compile-time immediate", fully unrolled: 3 ds_read_b32 + 2 MFMA per step.
#include "common.h"

#include <stdlib.h>

#include <type_traits>

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int TP = 128;              // output pixels per workgroup row (along W)
constexpr int TH = 4;                // output rows per workgroup
constexpr int PW = 2 * TP + 8;       // patch row pitch in floats (cols 1..261 used)
constexpr int PR = 2 * TH + 5;       // patch rows per input channel
constexpr int KPAD = 148, NSTEP = 74;

struct StemStep { int a_imm, b_imm, grp; };
__host__ __device__ constexpr StemStep stem_step(int p) {
  if (p < 63) {                                     // (c,kh,kw) | (c,kh+1,kw), kh even
    const int c = p / 21, r = p % 21, kh = 2 * (r / 7), kw = r % 7;
    return {((c * PR + kh) * PW + kw + 1) * 4, (c * 49 + kh * 7 + kw) * 256, 0};
  }
  if (p < 72) {                                     // (c,6,kw) | (c,6,kw+1), kw even
    const int q = p - 63, c = q / 3, kw = 2 * (q % 3);
    return {((c * PR + 6) * PW + kw + 1) * 4, (c * 49 + 42 + kw) * 256, 1};
  }
  if (p == 72) return {((0 * PR + 6) * PW + 6 + 1) * 4, 48 * 256, 2};          // (0,6,6) | (1,6,6)
  return {((2 * PR + 6) * PW + 6 + 1) * 4, 146 * 256, 3};                      // (2,6,6) | zero row 147
}

// Stage the 3 x PR input rows a tile of TH output rows x TP pixels needs into LDS, zero-filled outside the image.
// patch row (c, pr) <-> image row ih = 2*oh0 - 3 + pr of plane c; patch col <-> iw = 2*ow0 - 4 + col
__device__ __forceinline__ void stage_patch(const float* __restrict__ x, float* patch, int n, int H, int W, int oh0, int ow0,
                                            int vec_ok, int tid, int lane, int wave) {
  const float* img = x + (size_t)n * 3 * H * W;
  const int iw0 = 2 * ow0 - 4;
  if (vec_ok) {                          // W % 4 == 0 and 16-B aligned rows: whole float4s are in or out
    // one patch row per wave and pass: lane -> float4 column (66 per row: lanes 0,1 take a second one)
    const int iwa = iw0 + 4 * lane, iwb = iw0 + 4 * (lane + 64);
    const unsigned va = (unsigned)iwa < (unsigned)W ? (unsigned)(iwa * 4) : 0x80000000u;
    const unsigned vb = (lane < 2 && (unsigned)iwb < (unsigned)W) ? (unsigned)(iwb * 4) : 0x80000000u;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    // ALL of the wave's rows are requested before the first one is written (the row-by-row loop paid one memory latency per row);
    // a row outside the image is a zero-length buffer: its loads return zeros, no branch
    constexpr int NR = (3 * PR + 3) / 4;
    u32x4 v0[NR], v1[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int r = wv + 4 * k;
      const int c = r / PR, pr = r - c * PR;
      const int ih = 2 * oh0 - 3 + pr;
      const bool ok = r < 3 * PR && (unsigned)ih < (unsigned)H;      // wave-uniform
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(img) + (ok ? ((size_t)c * H + ih) * W : 0), 0, ok ? W * 4 : 0, 0x00020000);
      v0[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, va, 0, 0);
      v1[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, vb, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int r = wv + 4 * k;
      if (r < 3 * PR) {
        char* const dst = reinterpret_cast<char*>(patch) + r * (PW * 4) + lane * 16;
        *reinterpret_cast<u32x4*>(dst) = v0[k];
        if (lane < 2) *reinterpret_cast<u32x4*>(dst + 1024) = v1[k];
      }
    }
  } else {
#pragma unroll 8
    for (int i = tid; i < 3 * PR * PW; i += 256) {
      const int r = i / PW, col = i - r * PW;
      const int c = r / PR, pr = r - c * PR;
      const int ih = 2 * oh0 - 3 + pr, iw = iw0 + col;
      const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
      patch[i] = ok ? img[((size_t)c * H + ih) * W + iw] : 0.f;
    }
  }
}

template <typename TO>
__global__ __launch_bounds__(256) void stem_conv7x7(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ scale,
                                                     const float* __restrict__ shift, TO* __restrict__ y,
                                                         int H, int W, int Ho, int Wo, int tilesW, int tilesH,
                                                         int vec_ok, int relu) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wl = smem;                      // [KPAD][64]
  float* patch = smem + KPAD * 64;       // [3][PR][PW]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tw = blockIdx.x % tilesW;
  const int th = (blockIdx.x / tilesW) % tilesH;
  const int n = blockIdx.x / (tilesW * tilesH);
  const int ow0 = tw * TP, oh0 = th * TH;

  // Staging is written for a minimal VALU count (every vector-ALU instruction here queues behind the other
  // workgroup's 64-cycle MFMAs): per-lane offsets are computed once, rows advance on the scalar unit.
  // filter bank, pre-packed by the host as [k = c*49+kh*7+kw][co] with a zero row k = 147
  {
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, KPAD * 64 * 4, 0x00020000);
    const unsigned voff = (unsigned)tid * 16u;
#pragma unroll
    for (int it = 0; it < (KPAD * 16 + 255) / 256; ++it) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, voff, it * 4096, 0);
      if (it * 256 + 255 < KPAD * 16 || tid < KPAD * 16 - it * 256)
        *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(wl) + tid * 16 + it * 4096) = v;
    }
  }
  stage_patch(x, patch, n, H, W, oh0, ow0, vec_ok, tid, lane, wave);
  __syncthreads();

  const int h = lane >> 5, l31 = lane & 31;
  const int pix = wave * 32 + l31;             // pixel within the tile row
  const int wv32 = __builtin_amdgcn_readfirstlane(wave) * 32;
  const unsigned y_lane = (unsigned)((4 * h * 64 + l31) * (int)sizeof(TO)), y_lane2 = y_lane + 4096u;
  const float sc0 = scale[l31], sh0 = shift[l31], sc1 = scale[l31 + 32], sh1 = shift[l31 + 32];
  const char* const bb = reinterpret_cast<const char*>(wl) + l31 * 4;
  const char* const pb[3] = {bb + h * 7 * 256, bb + h * 256, bb + h * 49 * 256};

  for (int ro = 0; ro < TH; ++ro) {
    const int oh = oh0 + ro;
    if (oh >= Ho) break;
    const char* const ab = reinterpret_cast<const char*>(patch) + (2 * ro * PW + 2 * pix) * 4;
    const char* const pa[4] = {ab + h * PW * 4, ab + h * 4, ab + h * PR * PW * 4, ab};
    f32x16 acc0, acc1;                               // started by the first MFMA (C = 0)
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // software pipeline over chunks of 4 steps: the 12 operand reads of chunk c+1 are issued before
    // the 8 MFMAs of chunk c (two register sets, statically indexed), so LDS latency hides under MFMA
    constexpr int CH = 4, NCH = (NSTEP + CH - 1) / CH;
    float av[2][CH], b0v[2][CH], b1v[2][CH];
    auto load_chunk = [&](int cidx, int set) {
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int p = cidx * CH + u;
        if (p < NSTEP) {
          const StemStep st = stem_step(p);
          const char* const bp = pb[st.grp == 3 ? 1 : st.grp];
          av[set][u] = *reinterpret_cast<const float*>(pa[st.grp] + st.a_imm);
          b0v[set][u] = *reinterpret_cast<const float*>(bp + st.b_imm);
          b1v[set][u] = *reinterpret_cast<const float*>(bp + st.b_imm + 128);
        }
      }
    };
    load_chunk(0, 0);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (c + 1 < NCH) load_chunk(c + 1, (c + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        if (c * CH + u < NSTEP) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c & 1][u], b0v[c & 1][u], c == 0 && u == 0 ? zero : acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c & 1][u], b1v[c & 1][u], c == 0 && u == 0 ? zero : acc1, 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // epilogue: column j = lane&31 -> channel, row i -> pixel wave*32 + i.  Buffer stores: the lane offset is fixed,
    // the pixel offsets are instruction immediates, and pixels past the end of the row fall outside the descriptor.
    TO* const ywave = y + ((size_t)(n * Ho + oh) * Wo + ow0 + wv32) * 64;
    const int left = Wo - ow0 - wv32;                 // pixels of this wave's 32 that exist
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        ywave, 0, left > 0 ? (left > 32 ? 32 : left) * 64 * (int)sizeof(TO) : 0, 0x00020000);
    // (two lane bases so that every pixel offset fits the 12-bit instruction immediate, which -- unlike the scalar
    // offset -- takes part in the descriptor's range check)
    constexpr int ES = (int)sizeof(TO);
    auto put = [&](auto relu_c) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ioff = ((r & 3) + 8 * (r >> 2)) * 64 * ES;
        const unsigned base = ioff < 4096 - 32 * ES ? y_lane : y_lane2;
        const int imm = ioff < 4096 - 32 * ES ? ioff : ioff - 4096;
        float v0 = fmaf(acc0[r], sc0, sh0), v1 = fmaf(acc1[r], sc1, sh1);
        if constexpr (decltype(relu_c)::value) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
        if constexpr (sizeof(TO) == 4) {
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), ry, base + imm, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), ry, base + imm + 32 * ES, 0, 0);
        } else {
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (__bf16)v0), ry, base + imm, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (__bf16)v1), ry, base + imm + 32 * ES, 0, 0);
        }
      }
    };
    if (relu) put(std::true_type{}); else put(std::false_type{});
  }
}

// ---- stem + 3x3/s2 max-pool in one kernel (inference, fp32): the 64-channel stem map never reaches HBM -------------------
// ref src/encoders.py:154-157 (conv1, bn1, relu, maxpool).  The stem output is the largest activation of the whole path
// (4.4 GB at 48 images of 900x1600); written and read back it costs the separate max-pool 1 ms of pure HBM streaming.
// Here a workgroup streams DOWN the image: it owns a strip of stem columns and a segment of pooled rows, computes its
// stem rows four at a time exactly like stem_conv7x7 (same patch staging, same MFMA schedule, same fma / max, so the
// result is bit-identical to stem -> max-pool), and pools in registers:
//   * columns: wave w computes stem pixels [30w, 30w+32) of the strip -- two pixels shared with its neighbour -- so the
//     three columns of every window it owns (centres at its local even pixels 2..30) are its own: no exchange between
//     waves.  In the 32x32 accumulator layout a lane holds pixels 8g+4h+{0..3}; the window centres are its registers
//     4g and 4g+2, the only value living in the other lane half (the left neighbour of 4g) comes with four shuffles;
//   * rows: the horizontal maxima of the last two stem rows stay in registers; every odd stem row 2p+1 completes pooled
//     row p.  Rows / columns outside the stem map count as 0 (= -inf after a ReLU).
// A strip is 120 new stem columns (122 computed: +1.7 %), pooled 60; segments of pooled rows are sized on the host so
// that the grid fills whole rounds of 512 resident workgroups.
constexpr int WPX = 30, TPS = 4 * WPX;

__global__ __launch_bounds__(256) void stem_pool7x7(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     float* __restrict__ y, int H, int W, int Ho, int Wo, int Hp, int Wp,
                                                     int tilesW, int nseg, int rows_per_seg, int vec_ok) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wl = smem;                      // [KPAD][64]
  float* patch = smem + KPAD * 64;       // [3][PR][PW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tw = blockIdx.x % tilesW;
  const int seg = (blockIdx.x / tilesW) % nseg;
  const int n = blockIdx.x / (tilesW * nseg);
  const int ow0 = tw * TPS - 2;                                    // even: the patch's 16-byte column loads stay aligned
  const int p0 = seg * rows_per_seg, p1 = p0 + rows_per_seg < Hp ? p0 + rows_per_seg : Hp;
  const int r_first = 2 * p0 - 1, r_last = 2 * (p1 - 1) + 1;       // stem rows this segment needs
  {
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, KPAD * 64 * 4, 0x00020000);
    const unsigned voff = (unsigned)tid * 16u;
#pragma unroll
    for (int it = 0; it < (KPAD * 16 + 255) / 256; ++it) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, voff, it * 4096, 0);
      if (it * 256 + 255 < KPAD * 16 || tid < KPAD * 16 - it * 256)
        *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(wl) + tid * 16 + it * 4096) = v;
    }
  }
  const int h = lane >> 5, l31 = lane & 31;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int pix = wv * WPX + l31;                                  // stem pixel of the strip this lane feeds to the MFMA
  const float sc0 = scale[l31], sh0 = shift[l31], sc1 = scale[l31 + 32], sh1 = shift[l31 + 32];
  const char* const bb = reinterpret_cast<const char*>(wl) + l31 * 4;
  const char* const pb[3] = {bb + h * 7 * 256, bb + h * 256, bb + h * 49 * 256};
  const int col0 = ow0 + wv * WPX + 4 * h;                         // stem column of register 0 of this lane
  const bool edge_cols = ow0 < 0 || ow0 + TPS + 2 > Wo;            // workgroup-uniform
  // pooled column of centre register rc = 2c: (col0 + (rc&3) + 8*(rc>>2)) / 2
  float prev1[16], prev2[16];                                      // horizontal maxima of the last two stem rows: [acc][centre]
#pragma unroll
  for (int i = 0; i < 16; ++i) { prev1[i] = 0.f; prev2[i] = 0.f; }

  for (int oh0 = r_first; oh0 <= r_last; oh0 += TH) {
    __syncthreads();                                               // the previous chunk's patch is fully consumed
    stage_patch(x, patch, n, H, W, oh0, ow0, vec_ok, tid, lane, wave);
    __syncthreads();
    for (int ro = 0; ro < TH; ++ro) {
      const int oh = oh0 + ro;
      if (oh > r_last) break;
      const char* const ab = reinterpret_cast<const char*>(patch) + (2 * ro * PW + 2 * pix) * 4;
      const char* const pa[4] = {ab + h * PW * 4, ab + h * 4, ab + h * PR * PW * 4, ab};
      f32x16 acc0, acc1;
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      constexpr int CH = 4, NCH = (NSTEP + CH - 1) / CH;
      float av[2][CH], b0v[2][CH], b1v[2][CH];
      auto load_chunk = [&](int cidx, int set) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int p = cidx * CH + u;
          if (p < NSTEP) {
            const StemStep st = stem_step(p);
            const char* const bp = pb[st.grp == 3 ? 1 : st.grp];
            av[set][u] = *reinterpret_cast<const float*>(pa[st.grp] + st.a_imm);
            b0v[set][u] = *reinterpret_cast<const float*>(bp + st.b_imm);
            b1v[set][u] = *reinterpret_cast<const float*>(bp + st.b_imm + 128);
          }
        }
      };
      load_chunk(0, 0);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH) load_chunk(c + 1, (c + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          if (c * CH + u < NSTEP) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c & 1][u], b0v[c & 1][u], c == 0 && u == 0 ? zero : acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c & 1][u], b1v[c & 1][u], c == 0 && u == 0 ? zero : acc1, 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // BatchNorm + ReLU exactly as stem_conv7x7 stores them; stem rows / columns that do not exist count as 0
      const bool row_ok = (unsigned)oh < (unsigned)Ho;             // uniform
      float v0[16], v1[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = row_ok && (!edge_cols || (unsigned)(col0 + (r & 3) + 8 * (r >> 2)) < (unsigned)Wo);
        v0[r] = ok ? fmaxf(fmaf(acc0[r], sc0, sh0), 0.f) : 0.f;
        v1[r] = ok ? fmaxf(fmaf(acc1[r], sc1, sh1), 0.f) : 0.f;
      }
      // horizontal 3-max at the centres rc = 0, 2, .., 14 (pixel 8g + 4h + {0, 2}); the left neighbour of rc = 4g lives in
      // the other lane half: register 4g+3 of half 0 for h = 1, register 4g-1 of half 1 for h = 0 (g = 0, h = 0: no centre)
      float hm[16];
      float x0[4], x1[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        x0[g] = __shfl_xor(v0[4 * g + 3], 32);
        x1[g] = __shfl_xor(v1[4 * g + 3], 32);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float l0 = h ? x0[g] : (g > 0 ? x0[g - 1] : 0.f), l1 = h ? x1[g] : (g > 0 ? x1[g - 1] : 0.f);
        hm[2 * g] = fmaxf(fmaxf(l0, v0[4 * g]), v0[4 * g + 1]);
        hm[2 * g + 1] = fmaxf(fmaxf(v0[4 * g + 1], v0[4 * g + 2]), v0[4 * g + 3]);
        hm[8 + 2 * g] = fmaxf(fmaxf(l1, v1[4 * g]), v1[4 * g + 1]);
        hm[8 + 2 * g + 1] = fmaxf(fmaxf(v1[4 * g + 1], v1[4 * g + 2]), v1[4 * g + 3]);
      }
      if (oh & 1) {                                                // stem row 2p+1 completes pooled row p (uniform branch)
        const int pr = (oh - 1) >> 1;
        if (pr >= p0 && pr < p1) {
          float* const yrow = y + ((size_t)(n * Hp + pr) * Wp) * 64 + l31;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const int rc = 2 * c;
            const int col = col0 + (rc & 3) + 8 * (rc >> 2);       // stem column of the window centre (even)
            const bool own = (h != 0 || c != 0) && col >= 0 && (col >> 1) < Wp;   // (h = 0, rc = 0 is the neighbour's window)
            if (own) {
              float* const dst = yrow + (size_t)(col >> 1) * 64;
              dst[0] = fmaxf(fmaxf(prev2[c], prev1[c]), hm[c]);
              dst[32] = fmaxf(fmaxf(prev2[8 + c], prev1[8 + c]), hm[8 + c]);
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) { prev2[i] = prev1[i]; prev1[i] = hm[i]; }
    }
  }
}

// ---- weight gradient of the stem: dW[co][k] = sum over pixels dY[pixel][co] * patch(pixel, tap k) --------------------
// Same tile (TH rows x TP pixels of one image, patch in LDS) as the forward.  Per output row the dY row tile
// [128 pixels][64 channels] is staged next to the patch; MFMA tiles C[i = channel][j = tap] (2 x 5 tiles of 32x32, 160
// accumulator registers per wave), the reduction dimension is the pixel: A = dY (lane = channel), B = the patch value
// under the lane's tap (per-lane tap base + immediate pixel offset).  Each wave owns 32 of the 128 pixels; a
// workgroup walks over many tiles (persistent) and adds its partial dW with float atomics once at the end, so the
// image and dY are read once and no im2col matrix is ever written.  dw: [64][160] (k >= 147 unused), zero-filled.
__global__ __launch_bounds__(256) void stem_wgrad(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw,
                                                   int N, int H, int W, int Ho, int Wo, int tilesW, int tilesH, int vec_ok) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* patch = smem;                   // [3][PR][PW]
  float* dyt = smem + 3 * PR * PW;       // [TP][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  // lane's tap of tile t: k = 32 t + l31 (clamped: the pad taps repeat tap 146, their columns are never read back)
  unsigned tapoff[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    int k = 32 * t + l31;
    k = k < 147 ? k : 146;
    const int c = k / 49, r = k - c * 49, kh = r / 7, kw = r - kh * 7;
    tapoff[t] = (unsigned)((((c * PR + kh) * PW + kw + 1) + 2 * (wv * 32 + h)) * 4);
  }
  const unsigned aoff = (unsigned)(((wv * 32 + h) * 64 + l31) * 4);
  f32x16 acc[2][5];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

  const int ntiles = N * tilesH * tilesW;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tw = tile % tilesW, th = (tile / tilesW) % tilesH, n = tile / (tilesW * tilesH);
    const int ow0 = tw * TP, oh0 = th * TH;
    __syncthreads();                                  // previous tile fully consumed
    stage_patch(x, patch, n, H, W, oh0, ow0, vec_ok, tid, lane, wave);
    for (int ro = 0; ro < TH; ++ro) {
      const int oh = oh0 + ro;
      if (oh >= Ho) break;
      if (ro) __syncthreads();                        // previous dY row consumed
      {                                               // dY row tile: 128 pixels x 64 channels, zero past the row end
        const int left = Wo - ow0;
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(dy) + ((size_t)(n * Ho + oh) * Wo + ow0) * 64, 0, (left > TP ? TP : left) * 256, 0x00020000);
#pragma unroll
        for (int it = 0; it < TP * 64 * 4 / (256 * 16); ++it) {
          // (whole offset in the vector operand: the scalar offset is not range-checked, and pixels past the row end must read 0)
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rd, (unsigned)(tid * 16 + it * 4096), 0, 0);
          *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(dyt) + tid * 16 + it * 4096) = v;
        }
      }
      __syncthreads();
      const char* const pa = reinterpret_cast<const char*>(dyt) + aoff;
      const char* const pb = reinterpret_cast<const char*>(patch) + 2 * ro * PW * 4;
#pragma unroll 4
      for (int st = 0; st < 16; ++st) {               // this wave's 32 pixels, two per MFMA
        const float a0 = *reinterpret_cast<const float*>(pa + st * 512);
        const float a1 = *reinterpret_cast<const float*>(pa + st * 512 + 128);
        float b[5];
#pragma unroll
        for (int t = 0; t < 5; ++t) b[t] = *reinterpret_cast<const float*>(pb + tapoff[t] + st * 16);
#pragma unroll
        for (int t = 0; t < 5; ++t) {
          acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[t], acc[0][t], 0, 0, 0);
          acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[t], acc[1][t], 0, 0, 0);
        }
      }
    }
  }
  // C[i][j]: i = channel within tile (rows from r and h), j = tap = 32 t + l31: 32 lanes write 128 contiguous bytes
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h, k = 32 * t + l31;
        if (k < 147) atomicAdd(&dw[co * 160 + k], acc[i][t][r]);
      }
}

constexpr size_t kStemWgradLds = (size_t)(3 * PR * PW + TP * 64) * sizeof(float);

constexpr size_t kStemLds = (size_t)(KPAD * 64 + 3 * PR * PW) * sizeof(float);

// 3x3 stride-2 pad-1 max-pool on NHWC, one 16-byte channel vector per thread.  ref src/encoders.py:157.
// Neighbouring output rows share an input row; the workgroups are renumbered so that each XCD (own L2) owns a
// contiguous band of output rows and fetches the shared rows from HBM once.
template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_nhwc(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W,
                                                          int C, int Ho, int Wo) {
  constexpr int V = vec16<T>::N;
  const int cv = C / V;
  const long long total = (long long)N * Ho * Wo * cv;
  const long long i = (long long)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % cv);
  long long pix = i / cv;
  const int ow = (int)(pix % Wo);
  pix /= Wo;
  const int oh = (int)(pix % Ho), n = (int)(pix / Ho);
  float m[V], v[V];
#pragma unroll
  for (int j = 0; j < V; ++j) m[j] = -INFINITY;
#pragma unroll
  for (int dh = 0; dh < 3; ++dh) {
    const int ih = 2 * oh - 1 + dh;
    if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
    for (int dw = 0; dw < 3; ++dw) {
      const int iw = 2 * ow - 1 + dw;
      if ((unsigned)iw >= (unsigned)W) continue;
      load16(x + ((size_t)(n * H + ih) * W + iw) * C + c * V, v);
#pragma unroll
      for (int j = 0; j < V; ++j) m[j] = fmaxf(m[j], v[j]);
    }
  }
  store16(y + (size_t)i * V, m);
}

}  // namespace

template <typename TO>
static int stem_entry(const float* x, const float* w, const float* scale, const float* shift, void* y, int N, int H,
                      int W, int relu, void* stream) {
  BEVF_REQUIRE(x && w && scale && shift && y, "stem: null pointer");
  BEVF_REQUIRE(bevf_aligned16(w), "stem: packed filter bank must be 16-byte aligned");
  BEVF_REQUIRE(N > 0 && H >= 1 && W >= 1, "stem: empty shape");
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  const int tilesW = (Wo + TP - 1) / TP, tilesH = (Ho + TH - 1) / TH;
  const long long grid = (long long)N * tilesH * tilesW;
  BEVF_REQUIRE(grid < (1ll << 31), "stem: grid too large");
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_conv7x7<TO>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStemLds);
    attr_done = true;
  }
  hipLaunchKernelGGL(stem_conv7x7<TO>, dim3((unsigned)grid), dim3(256), kStemLds, static_cast<hipStream_t>(stream), x,
                     w, scale, shift, static_cast<TO*>(y), H, W, Ho, Wo, tilesW, tilesH,
                     (W % 4 == 0 && bevf_aligned16(x)) ? 1 : 0, relu);
  return bevf_check_launch("bevf_stem_conv7x7");
}
extern "C" int bevf_stem_conv7x7_f32(const float* x, const float* w, const float* scale, const float* shift,
                                     float* y, int N, int H, int W, int relu, void* stream) {
  return stem_entry<float>(x, w, scale, shift, y, N, H, W, relu, stream);
}
// same computation (fp32 MFMA on the fp32 image), bf16 NHWC output for the bf16 path
extern "C" int bevf_stem_conv7x7_bf16out(const float* x, const float* w, const float* scale, const float* shift,
                                         void* y, int N, int H, int W, int relu, void* stream) {
  return stem_entry<__bf16>(x, w, scale, shift, y, N, H, W, relu, stream);
}

// stem + max-pool fused (fp32 in, fp32 pooled NHWC out [N][Hp][Wp][64]); bit-identical to bevf_stem_conv7x7_f32 (relu) followed
// by bevf_maxpool3x3s2_nhwc_f32
extern "C" int bevf_stem_pool_f32(const float* x, const float* w, const float* scale, const float* shift, float* y, int N,
                                  int H, int W, void* stream) {
  BEVF_REQUIRE(x && w && scale && shift && y, "stem_pool: null pointer");
  BEVF_REQUIRE(bevf_aligned16(w), "stem_pool: packed filter bank must be 16-byte aligned");
  BEVF_REQUIRE(N > 0 && H >= 1 && W >= 1, "stem_pool: empty shape");
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  const int Hp = (Ho + 2 - 3) / 2 + 1, Wp = (Wo + 2 - 3) / 2 + 1;
  const int tilesW = (Wp + TPS / 2 - 1) / (TPS / 2);
  // segments of pooled rows: fewest (rounds of 512 resident workgroups) x (chunks of 4 stem rows per workgroup)
  int best_seg = 1;
  long long best_cost = -1;
  for (int ns = 1; ns <= 32 && ns <= Hp; ++ns) {
    const int rps = (Hp + ns - 1) / ns, real = (Hp + rps - 1) / rps;
    const long long wgs = (long long)N * tilesW * real;
    const long long cost = ((wgs + 511) / 512) * ((2 * rps + 1 + TH - 1) / TH);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_seg = real; }
  }
  const int rps = (Hp + best_seg - 1) / best_seg, nseg = (Hp + rps - 1) / rps;
  const long long grid = (long long)N * tilesW * nseg;
  BEVF_REQUIRE(grid < (1ll << 31), "stem_pool: grid too large");
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_pool7x7), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStemLds);
    attr_done = true;
  }
  hipLaunchKernelGGL(stem_pool7x7, dim3((unsigned)grid), dim3(256), kStemLds, static_cast<hipStream_t>(stream), x, w, scale, shift, y,
                     H, W, Ho, Wo, Hp, Wp, tilesW, nseg, rps, (W % 4 == 0 && bevf_aligned16(x)) ? 1 : 0);
  return bevf_check_launch("bevf_stem_pool_f32");
}

template <typename T>
static int maxpool_entry(const void* x, void* y, int N, int H, int W, int C, void* stream) {
  constexpr int V = vec16<T>::N;
  BEVF_REQUIRE(x && y, "maxpool: null pointer");
  BEVF_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % V == 0, "maxpool: C=%d must be a positive multiple of %d", C, V);
  BEVF_REQUIRE(bevf_aligned16(x) && bevf_aligned16(y), "maxpool: unaligned");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long long total = (long long)N * Ho * Wo * (C / V);
  BEVF_REQUIRE((total + 255) / 256 < (1ll << 31), "maxpool: grid too large");
  const unsigned grid = (unsigned)((total + 255) / 256);
  hipLaunchKernelGGL(maxpool3x3s2_nhwc<T>, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const T*>(x), static_cast<T*>(y), N, H, W, C, Ho, Wo);
  return bevf_check_launch("bevf_maxpool3x3s2_nhwc");
}
extern "C" int bevf_maxpool3x3s2_nhwc_f32(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  return maxpool_entry<float>(x, y, N, H, W, C, stream);
}
extern "C" int bevf_maxpool3x3s2_nhwc_bf16(const void* x, void* y, int N, int H, int W, int C, void* stream) {
  return maxpool_entry<__bf16>(x, y, N, H, W, C, stream);
}

// dW [64][160] += stem weight gradient (dw zero-filled by the caller; k = c*49+kh*7+kw, columns >= 147 stay zero)
extern "C" int bevf_stem_wgrad_f32(const float* x, const float* dy, float* dw, int N, int H, int W, void* stream) {
  BEVF_REQUIRE(x && dy && dw && N > 0 && H >= 1 && W >= 1, "stem_wgrad: bad arguments");
  BEVF_REQUIRE(bevf_aligned16(dy), "stem_wgrad: dy must be 16-byte aligned");
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  const int tilesW = (Wo + TP - 1) / TP, tilesH = (Ho + TH - 1) / TH;
  const long long tiles = (long long)N * tilesH * tilesW;
  BEVF_REQUIRE(tiles < (1ll << 31), "stem_wgrad: too many tiles");
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_wgrad), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)kStemWgradLds);
    attr_done = true;
  }
  const unsigned grid = (unsigned)(tiles < 512 ? tiles : 512);          // 2 workgroups per CU, each over many tiles
  hipLaunchKernelGGL(stem_wgrad, dim3(grid), dim3(256), kStemWgradLds, static_cast<hipStream_t>(stream), x, dy, dw, N, H, W, Ho,
                     Wo, tilesW, tilesH, (W % 4 == 0 && bevf_aligned16(x)) ? 1 : 0);
  return bevf_check_launch("bevf_stem_wgrad_f32");
}

// ---- bf16 stem (bf16-storage models): 7x7 stride-2 conv on v_mfma_f32_32x32x16_bf16 -----------------------------------
// The bf16 MFMA wants 8 consecutive k per lane, which the strided patch cannot supply directly.  Per half output row
// (64 pixels) the workgroup therefore expands the bf16 patch into an explicit column tile in LDS,
//     col[pixel][k],  k = (c*7 + kh)*8 + kw   (kw = 7 and k >= 168 are zero columns; 176 = 11 MFMA k-groups),
// with four aligned dword reads + three v_alignbit per (pixel, c, kh) item and one 16-byte write, then runs 11 MFMAs
// per wave on aligned 16-byte fragments.  The expansion is vector-ALU / LDS work that the fp32 stem could not afford
// (there every VALU instruction queues behind a 64-cycle MFMA); the bf16 MFMA is 16x faster per flop, so even with the
// expansion the stem drops from ~24 % to a few % of the bf16 step.
namespace {

constexpr int KP = 176;                  // padded K (bf16 elements)
constexpr int CP = 368;                  // row pitch in bytes of the column / weight tiles (176*2 + 16: conflict-free b128)
constexpr int HP = 64;                   // pixels per expansion (half a tile row)

typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4s __attribute__((ext_vector_type(4)));

template <typename TO>
__global__ __launch_bounds__(256) void stem_conv7x7_bf16mma(const float* __restrict__ x, const __bf16* __restrict__ w,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             TO* __restrict__ y, int H, int W, int Ho, int Wo, int tilesW,
                                                             int tilesH, int relu) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // The filter bank only passes through LDS (each wave keeps its fragments in registers), so it shares the column
  // tile's storage: 44 KB per workgroup, 3 workgroups per CU.
  char* const col = reinterpret_cast<char*>(smem);                 // [HP][CP]   bf16 column tile
  char* const wt = col;                                            // [64][CP]   bf16 filter, k contiguous per channel (transient)
  char* const patch = col + HP * CP;                               // [3][PR][PW] bf16
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tw = blockIdx.x % tilesW;
  const int th = (blockIdx.x / tilesW) % tilesH;
  const int n = blockIdx.x / (tilesW * tilesH);
  const int ow0 = tw * TP, oh0 = th * TH;

  // filter bank [64][176] bf16 -> LDS rows of pitch CP
  for (int i = tid; i < 64 * (KP / 8); i += 256) {
    const int ch = i / (KP / 8), q = i - ch * (KP / 8);
    *reinterpret_cast<u32x4*>(wt + ch * CP + q * 16) = *reinterpret_cast<const u32x4*>(w + (size_t)ch * KP + q * 8);
  }
  // patch: fp32 image rows -> bf16 (patch row (c, pr) <-> image row 2*oh0 - 3 + pr, patch col <-> iw = 2*ow0 - 4 + col);
  // one patch row per wave and pass, 16-byte loads that the descriptor zero-fills outside the image row
  {
    const float* img = x + (size_t)n * 3 * H * W;
    const int iw0 = 2 * ow0 - 4;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const bool vec_ok = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15u) == 0);
    for (int r = wv; r < 3 * PR; r += 4) {
      const int c = r / PR, pr = r - c * PR;
      const int ih = 2 * oh0 - 3 + pr;
      const bool row_ok = (unsigned)ih < (unsigned)H;                      // wave-uniform
      const float* row = img + ((size_t)c * H + (row_ok ? ih : 0)) * W;
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row), 0, row_ok ? W * 4 : 0, 0x00020000);
      for (int c4 = lane; c4 < PW / 4; c4 += 64) {
        const int iw = iw0 + 4 * c4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (vec_ok) {
          v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, iw >= 0 ? (unsigned)(iw * 4) : 0x80000000u, 0, 0));
        } else if (row_ok) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if ((unsigned)(iw + j) < (unsigned)W) v[j] = row[iw + j];
        }
        bf16x4s b;
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = (__bf16)v[j];
        *reinterpret_cast<bf16x4s*>(patch + ((size_t)r * PW + 4 * c4) * 2) = b;
      }
    }
  }
  __syncthreads();

  const int h = lane >> 5, l31 = lane & 31;
  const int mi = wave >> 1, ni = wave & 1;                         // wave tile: 32 pixels x 32 channels of the half row
  const float sc = scale[ni * 32 + l31], sh = shift[ni * 32 + l31];
  const char* const a_rd = col + (mi * 32 + l31) * CP + h * 16;
  const char* const b_rd = wt + (ni * 32 + l31) * CP + h * 16;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  constexpr int NG = KP / 16;
  bf16x8s bfrag[NG];                                               // the wave's filter fragments: read once per tile
#pragma unroll
  for (int g = 0; g < NG; ++g) bfrag[g] = *reinterpret_cast<const bf16x8s*>(b_rd + g * 32);
  __syncthreads();                                                 // filter bank consumed: its storage becomes the column tile
  if (tid < HP) *reinterpret_cast<u32x4*>(col + tid * CP + 168 * 2) = u32x4{0u, 0u, 0u, 0u};   // zero columns 168..175, once
  // expansion roles: item i = tid + 256*j <-> (pixel p = i & 63, patch row r = i >> 6); p and the r-step are fixed per thread
  const int xp = tid & (HP - 1), xr0 = tid >> 6;                   // r = xr0 + 4*j, j < 6 (r < 21)

  for (int ro = 0; ro < TH; ++ro) {
    const int oh = oh0 + ro;
    if (oh >= Ho) break;
    for (int hf = 0; hf < TP / HP; ++hf) {
      if (ow0 + hf * HP >= Wo) break;
      // ---- expand: all window reads first, then the shifts and the 16-byte writes -----------------------------------
      {
        unsigned d[6][4];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int r = xr0 + 4 * j;
          if (r < 21) {
            const int c = r / 7, kh = r - c * 7;
            const unsigned* src = reinterpret_cast<const unsigned*>(patch + ((size_t)(c * PR + 2 * ro + kh) * PW + 2 * (hf * HP + xp)) * 2);
#pragma unroll
            for (int q = 0; q < 4; ++q) d[j][q] = src[q];
          }
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int r = xr0 + 4 * j;
          if (r < 21) {
            u32x4 o;
            o[0] = __builtin_amdgcn_alignbit(d[j][1], d[j][0], 16);
            o[1] = __builtin_amdgcn_alignbit(d[j][2], d[j][1], 16);
            o[2] = __builtin_amdgcn_alignbit(d[j][3], d[j][2], 16);
            o[3] = d[j][3] >> 16;                                  // kw = 6, then the zero column kw = 7
            *reinterpret_cast<u32x4*>(col + xp * CP + r * 16) = o;
          }
        }
      }
      __syncthreads();
      // ---- 11 k-groups of 16: fragments first, then the MFMAs back to back --------------------------------------------
      bf16x8s afrag[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) afrag[g] = *reinterpret_cast<const bf16x8s*>(a_rd + g * 32);
      f32x16 acc;
#pragma unroll
      for (int g = 0; g < NG; ++g) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[g], bfrag[g], g == 0 ? zero : acc, 0, 0, 0);
      // ---- epilogue: rows i = pixel, column j = channel ------------------------------------------------------
      const int px0 = ow0 + hf * HP + mi * 32;
      const int left = Wo - px0;
      TO* const ybase = y + ((size_t)(n * Ho + oh) * Wo + px0) * 64 + ni * 32;
      const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
          ybase, 0, left > 0 ? ((left > 32 ? 32 : left) * 64 - ni * 32) * (int)sizeof(TO) : 0, 0x00020000);
      const unsigned lane_off = (unsigned)((4 * h * 64 + l31) * (int)sizeof(TO));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ioff = ((r & 3) + 8 * (r >> 2)) * 64 * (int)sizeof(TO);
        float v = fmaf(acc[r], sc, sh);
        if (relu) v = fmaxf(v, 0.f);
        if constexpr (sizeof(TO) == 4)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, lane_off + ioff, 0, 0);
        else
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (__bf16)v), ry, lane_off + ioff, 0, 0);
      }
      __syncthreads();                                             // column tile free for the next expansion
    }
  }
}

constexpr size_t kStemBf16Lds = (size_t)HP * CP + (size_t)3 * PR * PW * 2;

// ---- bf16 stem + 3x3/s2 max-pool in one kernel (bf16-storage inference): the streaming structure of stem_pool7x7 on the
// bf16 MFMA path above.  A workgroup owns a strip of 120 new stem columns and a segment of pooled rows; per stem row it runs
// two half rows of 62 pixels (the column tile still has 64 rows; wave (mi, ni) takes pixels 30 mi .. 30 mi + 31 of the
// half and channels 32 ni ..), so the four 32-pixel wave tiles of a row start 30 pixels apart and every pooling window
// lies inside one wave's tile; vertical state (the horizontal maxima of the last two stem rows) stays in registers.
// max commutes with the (monotonic) bf16 rounding and a missing neighbour counts as 0 after the ReLU, so the result is
// bit-identical to bevf_stem_conv7x7_bf16mma followed by bevf_maxpool3x3s2_nhwc_bf16 -- without the 2.2 GB bf16 stem map.
__global__ __launch_bounds__(256) void stem_pool7x7_bf16mma(const float* __restrict__ x, const __bf16* __restrict__ w,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             __bf16* __restrict__ y, int H, int W, int Ho, int Wo, int Hp, int Wp,
                                                             int tilesW, int nseg, int rows_per_seg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* const col = reinterpret_cast<char*>(smem);                 // [HP][CP]   bf16 column tile
  char* const wt = col;                                            // [64][CP]   bf16 filter (transient)
  char* const patch = col + HP * CP;                               // [3][PR][PW] bf16
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tw = blockIdx.x % tilesW;
  const int seg = (blockIdx.x / tilesW) % nseg;
  const int n = blockIdx.x / (tilesW * nseg);
  const int ow0 = tw * TPS - 2;
  const int p0 = seg * rows_per_seg, p1 = p0 + rows_per_seg < Hp ? p0 + rows_per_seg : Hp;
  const int r_first = 2 * p0 - 1, r_last = 2 * (p1 - 1) + 1;       // stem rows this segment needs
  for (int i = tid; i < 64 * (KP / 8); i += 256) {
    const int ch = i / (KP / 8), q = i - ch * (KP / 8);
    *reinterpret_cast<u32x4*>(wt + ch * CP + q * 16) = *reinterpret_cast<const u32x4*>(w + (size_t)ch * KP + q * 8);
  }
  __syncthreads();
  const int h = lane >> 5, l31 = lane & 31;
  const int mi = wave >> 1, ni = wave & 1;
  const float sc = scale[ni * 32 + l31], sh = shift[ni * 32 + l31];
  const char* const a_rd = col + (mi * WPX + l31) * CP + h * 16;
  const char* const b_rd = wt + (ni * 32 + l31) * CP + h * 16;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  constexpr int NG = KP / 16;
  bf16x8s bfrag[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) bfrag[g] = *reinterpret_cast<const bf16x8s*>(b_rd + g * 32);
  __syncthreads();                                                 // filter bank consumed: its storage becomes the column tile
  if (tid < HP) *reinterpret_cast<u32x4*>(col + tid * CP + 168 * 2) = u32x4{0u, 0u, 0u, 0u};
  const int xp = tid & (HP - 1), xr0 = tid >> 6;
  const bool vec_ok = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15u) == 0);
  float prev1[2][8], prev2[2][8];                                  // [half][centre]: horizontal maxima of the last two stem rows
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 8; ++i) { prev1[a][i] = 0.f; prev2[a][i] = 0.f; }

  for (int oh0 = r_first; oh0 <= r_last; oh0 += TH) {
    __syncthreads();                                               // the previous chunk's patch is fully consumed
    {                                                              // patch rows 2*oh0 - 3 .., columns 2*ow0 - 4 .. as bf16
      const float* img = x + (size_t)n * 3 * H * W;
      const int iw0 = 2 * ow0 - 4;
      const int wv = __builtin_amdgcn_readfirstlane(wave);
      for (int r = wv; r < 3 * PR; r += 4) {
        const int c = r / PR, pr = r - c * PR;
        const int ih = 2 * oh0 - 3 + pr;
        const bool row_ok = (unsigned)ih < (unsigned)H;
        const float* row = img + ((size_t)c * H + (row_ok ? ih : 0)) * W;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row), 0, row_ok ? W * 4 : 0, 0x00020000);
        for (int c4 = lane; c4 < PW / 4; c4 += 64) {
          const int iw = iw0 + 4 * c4;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (vec_ok) {
            v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, iw >= 0 ? (unsigned)(iw * 4) : 0x80000000u, 0, 0));
          } else if (row_ok) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if ((unsigned)(iw + j) < (unsigned)W) v[j] = row[iw + j];
          }
          bf16x4s b;
#pragma unroll
          for (int j = 0; j < 4; ++j) b[j] = (__bf16)v[j];
          *reinterpret_cast<bf16x4s*>(patch + ((size_t)r * PW + 4 * c4) * 2) = b;
        }
      }
    }
    __syncthreads();
    for (int ro = 0; ro < TH; ++ro) {
      const int oh = oh0 + ro;
      if (oh > r_last) break;
      const bool row_ok = (unsigned)oh < (unsigned)Ho;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        if (ow0 + hf * 2 * WPX >= Wo) break;                       // (uniform) nothing of this half exists
        {
          unsigned d[6][4];
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            const int r = xr0 + 4 * j;
            if (r < 21) {
              const int c = r / 7, kh = r - c * 7;
              const unsigned* src = reinterpret_cast<const unsigned*>(patch + ((size_t)(c * PR + 2 * ro + kh) * PW + 2 * (hf * 2 * WPX + xp)) * 2);
#pragma unroll
              for (int q = 0; q < 4; ++q) d[j][q] = src[q];
            }
          }
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            const int r = xr0 + 4 * j;
            if (r < 21) {
              u32x4 o;
              o[0] = __builtin_amdgcn_alignbit(d[j][1], d[j][0], 16);
              o[1] = __builtin_amdgcn_alignbit(d[j][2], d[j][1], 16);
              o[2] = __builtin_amdgcn_alignbit(d[j][3], d[j][2], 16);
              o[3] = d[j][3] >> 16;
              *reinterpret_cast<u32x4*>(col + xp * CP + r * 16) = o;
            }
          }
        }
        __syncthreads();
        bf16x8s afrag[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) afrag[g] = *reinterpret_cast<const bf16x8s*>(a_rd + g * 32);
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < NG; ++g) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[g], bfrag[g], g == 0 ? zero : acc, 0, 0, 0);
        // BatchNorm + ReLU as the unfused kernel stores them (before its bf16 rounding); missing rows / columns count as 0
        const int col0 = ow0 + hf * 2 * WPX + mi * WPX + 4 * h;    // stem column of register 0 of this lane
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool ok = row_ok && (unsigned)(col0 + (r & 3) + 8 * (r >> 2)) < (unsigned)Wo;
          v[r] = ok ? fmaxf(fmaf(acc[r], sc, sh), 0.f) : 0.f;
        }
        float hm[8], xl[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) xl[g] = __shfl_xor(v[4 * g + 3], 32);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float l0 = h ? xl[g] : (g > 0 ? xl[g - 1] : 0.f);
          hm[2 * g] = fmaxf(fmaxf(l0, v[4 * g]), v[4 * g + 1]);
          hm[2 * g + 1] = fmaxf(fmaxf(v[4 * g + 1], v[4 * g + 2]), v[4 * g + 3]);
        }
        if (oh & 1) {                                              // stem row 2p+1 completes pooled row p (uniform branch)
          const int pr = (oh - 1) >> 1;
          if (pr >= p0 && pr < p1) {
            __bf16* const yrow = y + ((size_t)(n * Hp + pr) * Wp) * 64 + ni * 32 + l31;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              const int rc = 2 * c;
              const int cc = col0 + (rc & 3) + 8 * (rc >> 2);      // stem column of the window centre (even)
              const bool own = (h != 0 || c != 0) && cc >= 0 && (cc >> 1) < Wp;
              if (own) yrow[(size_t)(cc >> 1) * 64] = (__bf16)fmaxf(fmaxf(prev2[hf][c], prev1[hf][c]), hm[c]);
            }
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { prev2[hf][i] = prev1[hf][i]; prev1[hf][i] = hm[i]; }
        __syncthreads();                                           // column tile free for the next expansion
      }
    }
  }
}

// ---- round 3: the same fused bf16 stem + max-pool WITHOUT the column-tile expansion ------------------------------------------
// stem_pool7x7_bf16mma spends its time expanding: per half row every thread makes ~5 x (4 ds_read_b32 + 3 v_alignbit + ds_write_b128)
// and two barriers frame 11 MFMAs per wave -- PMC: matrix pipe 13 % busy, 55 % of the wave cycles parked, the LDS array saturated by
// three co-resident workgroups.  Here the A fragment of a pixel is read STRAIGHT from the staged patch: the 8 k of a group
// (c, kh) are the 8 consecutive input pixels 2q+1 .. 2q+8 of one patch row (q = strip pixel; the 8th meets the filter's zero
// column), an odd element offset.  The patch is therefore kept as TWO bf16 copies shifted by 1 and by 3 elements: pixel q reads copy
// q & 1 at byte 8 (q >> 1) -- two aligned ds_read_b64 per fragment, address = lane base + immediate, no VALU, no barrier.  With the
// copies 128 B (mod 256) apart the 32 lanes of a ds_read_b64 group cover all 64 banks once.  Staging makes both copies from two
// aligned 16-byte global loads per lane (columns 4m .. 4m+7: the second is the neighbour's first, an L1 hit), three v_cvt_pk and
// two ds_write_b64.  Barriers: two per chunk of four stem rows (was 18).  Same fragments, same MFMA order, same epilogue ->
// bit-identical to stem_pool7x7_bf16mma for finite images (the kw = 7 product is pixel x 0 instead of 0 x 0).
constexpr int RPB = PW * 2;                                        // bytes per patch row of one copy (264 bf16)
constexpr int COPYB = ((3 * PR * RPB + 255) / 256) * 256 + 128;    // copy stride: == 128 (mod 256)
// The pooled row leaves through a per-wave LDS transpose: a lane holds ONE channel of 8 pooled pixels (2-byte stores, 8 per lane and
// half row: in-kernel ablation put 40 % of the kernel on them); written as bf16 to [16 pixels][32 channels] and read back as 16-byte
// pieces, a wave stores its 16 x 64 bytes with ONE b128 store per lane.  Wave-local: no barrier.
constexpr int TPITCH = 80;                                         // bytes per pixel of a transpose tile (64 + 16: the two lane halves hit different banks)
constexpr int TTILE = 16 * TPITCH;                                 // one (wave, half row) tile
constexpr size_t kStemBf16V2Lds = (size_t)2 * COPYB + (size_t)4 * 2 * TTILE;
constexpr int QPR = PW / 4;                                        // 16-byte quads per patch row (66)
constexpr int NITEM = 3 * PR * QPR, NIT = (NITEM + 255) / 256;     // staging items (row, quad) per chunk: 2574 = 10 x 256 + 14
__host__ __device__ constexpr int stem_row_of(int r) { return r >= 21 ? (2 * PR + 6) : (r / 7) * PR + (r % 7); }   // k-group -> patch row (group 21: zero filter)

__global__ __launch_bounds__(256) void stem_pool7x7_bf16v2(const float* __restrict__ x, const __bf16* __restrict__ w,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            __bf16* __restrict__ y, int H, int W, int Ho, int Wo, int Hp, int Wp,
                                                            int tilesW, int nseg, int rows_per_seg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* const lds = reinterpret_cast<char*>(smem);                 // [2 copies][3 * PR rows][RPB]; the filter bank passes through first
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tw = blockIdx.x % tilesW;
  const int seg = (blockIdx.x / tilesW) % nseg;
  const int n = blockIdx.x / (tilesW * nseg);
  const int ow0 = tw * TPS - 2;
  const int p0 = seg * rows_per_seg, p1 = p0 + rows_per_seg < Hp ? p0 + rows_per_seg : Hp;
  const int r_first = 2 * p0 - 1, r_last = 2 * (p1 - 1) + 1;       // stem rows this segment needs
  for (int i = tid; i < 64 * (KP / 8); i += 256) {
    const int ch = i / (KP / 8), q = i - ch * (KP / 8);
    *reinterpret_cast<u32x4*>(lds + ch * CP + q * 16) = *reinterpret_cast<const u32x4*>(w + (size_t)ch * KP + q * 8);
  }
  __syncthreads();
  const int h = lane >> 5, l31 = lane & 31;
  const int mi = wave >> 1, ni = wave & 1;
  const float sc = scale[ni * 32 + l31], sh = shift[ni * 32 + l31];
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  constexpr int NG = KP / 16;
  bf16x8s bfrag[NG];
  {
    const char* const b_rd = lds + (ni * 32 + l31) * CP + h * 16;
#pragma unroll
    for (int g = 0; g < NG; ++g) bfrag[g] = *reinterpret_cast<const bf16x8s*>(b_rd + g * 32);
  }
  // A-fragment addresses: k-group 2g + h of pixel q -> copy (q & 1), patch row stem_row_of(2g + h) + 2 ro, byte 8 (q >> 1).
  // Row of the odd group = row of the even one + 1, except 6|7 (next channel: + PR - 6), 20|21 (the zero group: same row, any finite data)
  // (the reads are inline asm: left to hipcc, pairs of them -- the two halves of a fragment, or the same half of two rows -- are fused into
  //  ds_read2_b64, which runs at half rate and is banked mod 32: PMC showed 41 % conflict cycles.  Plain ds_read_b64 are conflict-free.)
  unsigned abase[2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    const int q = hf * 2 * WPX + mi * WPX + l31;
    abase[hf] = (unsigned)(uintptr_t)lds + (unsigned)((q & 1) * COPYB + 8 * (q >> 1));
  }
  const int hrow1 = h * RPB, hrow7 = h * (PR - 6) * RPB;
  const bool all_cols = ow0 >= 0 && ow0 + 3 * WPX + 32 <= Wo;      // every stem column the four wave tiles touch exists
  const bool vec_ok = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15u) == 0) && (long long)3 * H * W * 4 < (1ll << 31);
  const __amdgpu_buffer_rsrc_t rimg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (size_t)n * 3 * H * W), 0, 3 * H * W * 4, 0x00020000);
  float prev1[2][8], prev2[2][8];                                  // [half][centre]: horizontal maxima of the last two stem rows
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 8; ++i) { prev1[a][i] = 0.f; prev2[a][i] = 0.f; }

  for (int oh0 = r_first; oh0 <= r_last; oh0 += TH) {
    __syncthreads();                                               // the previous chunk's patch (or the filter bank) is consumed
    const float* img = x + (size_t)n * 3 * H * W;
    const int iw0 = 2 * ow0 - 4;
    if (vec_ok) {
      // patch rows 2*oh0 - 3 .., columns 2*ow0 - 4 .. -> the two shifted bf16 copies.  ALL loads of the chunk first (a thread owns up to
      // NIT (row, quad) items; 2 x 16 bytes each: columns 4 c4 .. + 7, the second load is the neighbour's first -- a cache hit), ONE
      // wait, then the conversions: the per-row loop this replaces paid a memory latency per row (in-kernel ablation: 53 % of the kernel)
      f32x4 sv[NIT], su[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int i = it * 256 + tid;
        const int r = (i * 993) >> 16, c4 = i - r * QPR;              // i / 66 for i < 2816
        const int c = (r * 5042) >> 16, pr = r - c * PR;              // r / 13 for r < 42
        const int ih = 2 * oh0 - 3 + pr, iw = iw0 + 4 * c4;
        // (one unsigned compare per bound, folded into the offset at once: no lane masks kept across the loads)
        unsigned ro = (unsigned)(((c * H + ih) * W + iw) * 4);
        ro = ((it < NIT - 1 || i < NITEM) && (unsigned)ih < (unsigned)H) ? ro : 0x80000000u;
        sv[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rimg, (unsigned)iw < (unsigned)W ? ro : 0x80000000u, 0, 0));
        su[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rimg, (unsigned)(iw + 4) < (unsigned)W ? ro + 16u : 0x80000000u, 0, 0));
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int i = it * 256 + tid;
        const int r = (i * 993) >> 16, c4 = i - r * QPR;
        const f32x4 v = sv[it], u = su[it];
        // copy 0 holds patch column e + 1 at element e, copy 1 column e + 3: elements 4 c4 .. 4 c4 + 3 of both
        bf16x4s a0, a1;
        a0[0] = (__bf16)v[1]; a0[1] = (__bf16)v[2]; a0[2] = (__bf16)v[3]; a0[3] = (__bf16)u[0];
        a1[0] = a0[2]; a1[1] = a0[3]; a1[2] = (__bf16)u[1]; a1[3] = (__bf16)u[2];
        if (it < NIT - 1 || i < NITEM) {
          *reinterpret_cast<bf16x4s*>(lds + (size_t)r * RPB + 8 * c4) = a0;
          *reinterpret_cast<bf16x4s*>(lds + COPYB + (size_t)r * RPB + 8 * c4) = a1;
        }
      }
    } else {                                                       // unaligned image / odd width: the scalar path, row by row
      const int wv = __builtin_amdgcn_readfirstlane(wave);
      for (int r = wv; r < 3 * PR; r += 4) {
        const int c = r / PR, pr = r - c * PR;
        const int ih = 2 * oh0 - 3 + pr;
        const bool row_ok = (unsigned)ih < (unsigned)H;
        const float* row = img + ((size_t)c * H + (row_ok ? ih : 0)) * W;
        for (int c4 = lane; c4 < QPR; c4 += 64) {
          const int iw = iw0 + 4 * c4;
          f32x4 v = {0.f, 0.f, 0.f, 0.f}, u = {0.f, 0.f, 0.f, 0.f};    // patch columns 4 c4 .. + 3 and the next four
          if (row_ok) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if ((unsigned)(iw + j) < (unsigned)W) v[j] = row[iw + j];
              if ((unsigned)(iw + 4 + j) < (unsigned)W) u[j] = row[iw + 4 + j];
            }
          }
          bf16x4s a0, a1;
          a0[0] = (__bf16)v[1]; a0[1] = (__bf16)v[2]; a0[2] = (__bf16)v[3]; a0[3] = (__bf16)u[0];
          a1[0] = a0[2]; a1[1] = a0[3]; a1[2] = (__bf16)u[1]; a1[3] = (__bf16)u[2];
          *reinterpret_cast<bf16x4s*>(lds + (size_t)r * RPB + 8 * c4) = a0;
          *reinterpret_cast<bf16x4s*>(lds + COPYB + (size_t)r * RPB + 8 * c4) = a1;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int ro = 0; ro < TH; ++ro) {
      const int oh = oh0 + ro;
      if (oh > r_last) break;
      const bool row_ok = (unsigned)oh < (unsigned)Ho;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        if (ow0 + hf * 2 * WPX >= Wo) break;                       // (uniform) nothing of this half exists
        typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
        bf16x8s afrag[NG];
        {
          const unsigned a0 = abase[hf], a1 = a0 + (unsigned)hrow1, a7 = a0 + (unsigned)hrow7;
          bf16x4v lo[NG], hi[NG];
#define STEM_RD(g, base)                                                                                                            \
          asm volatile("ds_read_b64 %0, %2 offset:%3\n\tds_read_b64 %1, %2 offset:%4"                                            \
                       : "=&v"(lo[g]), "=&v"(hi[g])                                                                                  \
                       : "v"(base), "n"(stem_row_of(2 * (g)) * RPB + ro * 2 * RPB), "n"(stem_row_of(2 * (g)) * RPB + ro * 2 * RPB + 8) \
                       : "memory")
          STEM_RD(0, a1); STEM_RD(1, a1); STEM_RD(2, a1); STEM_RD(3, a7); STEM_RD(4, a1); STEM_RD(5, a1);
          STEM_RD(6, a1); STEM_RD(7, a1); STEM_RD(8, a1); STEM_RD(9, a1); STEM_RD(10, a0);
#undef STEM_RD
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int g = 0; g < NG; ++g) afrag[g] = __builtin_shufflevector(lo[g], hi[g], 0, 1, 2, 3, 4, 5, 6, 7);
        }
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < NG; ++g) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[g], bfrag[g], g == 0 ? zero : acc, 0, 0, 0);
        // BatchNorm + ReLU as the unfused kernel stores them (before its bf16 rounding); missing rows / columns count as 0
        const int col0 = ow0 + hf * 2 * WPX + mi * WPX + 4 * h;    // stem column of register 0 of this lane
        float v[16];
        if (row_ok && all_cols) {                                  // (uniform) interior strip: no per-element predicate
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = fmaxf(fmaf(acc[r], sc, sh), 0.f);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool ok = row_ok && (unsigned)(col0 + (r & 3) + 8 * (r >> 2)) < (unsigned)Wo;
            v[r] = ok ? fmaxf(fmaf(acc[r], sc, sh), 0.f) : 0.f;
          }
        }
        float hm[8], xl[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) xl[g] = __shfl_xor(v[4 * g + 3], 32);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float l0 = h ? xl[g] : (g > 0 ? xl[g - 1] : 0.f);
          hm[2 * g] = fmaxf(fmaxf(l0, v[4 * g]), v[4 * g + 1]);
          hm[2 * g + 1] = fmaxf(fmaxf(v[4 * g + 1], v[4 * g + 2]), v[4 * g + 3]);
        }
        if (oh & 1) {                                              // stem row 2p+1 completes pooled row p (uniform branch)
          const int pr = (oh - 1) >> 1;
          if (pr >= p0 && pr < p1) {
            // this wave's 16 pooled pixels x 32 channels through its transpose tile (see TPITCH): lane (l31, h) holds channel l31 of
            // pixels 4 (c >> 1) + (c & 1) + 2 h; lane (pixel = lane >> 2, piece = lane & 3) then stores 8 channels of one pixel
            char* const tt = lds + 2 * COPYB + (wave * 2 + hf) * TTILE;
#pragma unroll
            for (int c = 0; c < 8; ++c)
              *reinterpret_cast<__bf16*>(tt + (4 * (c >> 1) + (c & 1) + 2 * h) * TPITCH + 2 * l31) =
                  (__bf16)fmaxf(fmaxf(prev2[hf][c], prev1[hf][c]), hm[c]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int px = lane >> 2, piece = lane & 3;
            const u32x4 q = *reinterpret_cast<const u32x4*>(tt + px * TPITCH + piece * 16);
            const int cc = ow0 + hf * 2 * WPX + mi * WPX + 2 * px;      // stem column of the window centre (even); pixel 0 is the left neighbour's
            if (px != 0 && cc >= 0 && (cc >> 1) < Wp)
              *reinterpret_cast<u32x4*>(y + ((size_t)(n * Hp + pr) * Wp + (cc >> 1)) * 64 + ni * 32 + piece * 8) = q;
            asm volatile("" ::: "memory");
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { prev2[hf][i] = prev1[hf][i]; prev1[hf][i] = hm[i]; }
      }
    }
  }
}

__global__ __launch_bounds__(256) void stem_pack_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 64 * KP) return;
  const int ch = i / KP, k = i - ch * KP, r = k >> 3, kw = k & 7;
  float v = 0.f;
  if (r < 21 && kw < 7) v = w[(ch * 21 + r) * 7 + kw];
  out[i] = (__bf16)v;
}

}  // namespace

// Packs the fp32 OIHW stem filter (64,3,7,7) into the bf16 [64][176] bank of the bf16 stem: k = (c*7+kh)*8 + kw.
extern "C" int bevf_stem_pack_bf16(const float* w_oihw, void* packed, void* stream) {
  BEVF_REQUIRE(w_oihw && packed, "stem_pack_bf16: null pointer");
  hipLaunchKernelGGL(stem_pack_bf16_kernel, dim3((64 * KP + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), w_oihw,
                     static_cast<__bf16*>(packed));
  return bevf_check_launch("bevf_stem_pack_bf16");
}

extern "C" int bevf_stem_conv7x7_bf16mma(const float* x, const void* w_packed, const float* scale, const float* shift, void* y,
                                         int N, int H, int W, int relu, void* stream) {
  BEVF_REQUIRE(x && w_packed && scale && shift && y, "stem bf16: null pointer");
  BEVF_REQUIRE(bevf_aligned16(w_packed) && bevf_aligned16(y), "stem bf16: unaligned");
  BEVF_REQUIRE(N > 0 && H >= 1 && W >= 1, "stem bf16: empty shape");
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  const int tilesW = (Wo + TP - 1) / TP, tilesH = (Ho + TH - 1) / TH;
  const long long grid = (long long)N * tilesH * tilesW;
  BEVF_REQUIRE(grid < (1ll << 31), "stem bf16: grid too large");
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_conv7x7_bf16mma<__bf16>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStemBf16Lds);
    attr_done = true;
  }
  hipLaunchKernelGGL(stem_conv7x7_bf16mma<__bf16>, dim3((unsigned)grid), dim3(256), kStemBf16Lds,
                     static_cast<hipStream_t>(stream), x, static_cast<const __bf16*>(w_packed), scale, shift,
                     static_cast<__bf16*>(y), H, W, Ho, Wo, tilesW, tilesH, relu);
  return bevf_check_launch("bevf_stem_conv7x7_bf16mma");
}

// bf16 stem + max-pool fused (fp32 image in, bf16 pooled NHWC out [N][Hp][Wp][64]); bit-identical to
// bevf_stem_conv7x7_bf16mma (relu) followed by bevf_maxpool3x3s2_nhwc_bf16
extern "C" int bevf_stem_pool_bf16mma(const float* x, const void* w_packed, const float* scale, const float* shift, void* y, int N,
                                      int H, int W, void* stream) {
  BEVF_REQUIRE(x && w_packed && scale && shift && y, "stem_pool bf16: null pointer");
  BEVF_REQUIRE(bevf_aligned16(w_packed), "stem_pool bf16: packed filter bank must be 16-byte aligned");
  BEVF_REQUIRE(N > 0 && H >= 1 && W >= 1, "stem_pool bf16: empty shape");
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  const int Hp = (Ho + 2 - 3) / 2 + 1, Wp = (Wo + 2 - 3) / 2 + 1;
  const int tilesW = (Wp + TPS / 2 - 1) / (TPS / 2);
  // segments of pooled rows: fewest (rounds of 768 resident workgroups: 44 KB of LDS each) x (chunks of 4 stem rows per workgroup)
  int best_seg = 1;
  long long best_cost = -1;
  for (int ns = 1; ns <= 32 && ns <= Hp; ++ns) {
    const int rps = (Hp + ns - 1) / ns, real = (Hp + rps - 1) / rps;
    const long long wgs = (long long)N * tilesW * real;
    const long long cost = ((wgs + 767) / 768) * ((2 * rps + 1 + TH - 1) / TH);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_seg = real; }
  }
  const int rps = (Hp + best_seg - 1) / best_seg, nseg = (Hp + rps - 1) / rps;
  const long long grid = (long long)N * tilesW * nseg;
  BEVF_REQUIRE(grid < (1ll << 31), "stem_pool bf16: grid too large");
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_pool7x7_bf16mma), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)kStemBf16Lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_pool7x7_bf16v2), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)kStemBf16V2Lds);
    attr_done = true;
  }
  // BEVF_STEM_BF16_EXPAND=1: the round-2 kernel (column tile expanded in LDS), kept for A/B; default: fragments straight from the patch
  static const bool expand = getenv("BEVF_STEM_BF16_EXPAND") != nullptr;
  if (expand)
    hipLaunchKernelGGL(stem_pool7x7_bf16mma, dim3((unsigned)grid), dim3(256), kStemBf16Lds, static_cast<hipStream_t>(stream), x,
                       static_cast<const __bf16*>(w_packed), scale, shift, static_cast<__bf16*>(y), H, W, Ho, Wo, Hp, Wp, tilesW, nseg, rps);
  else
    hipLaunchKernelGGL(stem_pool7x7_bf16v2, dim3((unsigned)grid), dim3(256), kStemBf16V2Lds, static_cast<hipStream_t>(stream), x,
                       static_cast<const __bf16*>(w_packed), scale, shift, static_cast<__bf16*>(y), H, W, Ho, Wo, Hp, Wp, tilesW, nseg, rps);
  return bevf_check_launch("bevf_stem_pool_bf16mma");
}
