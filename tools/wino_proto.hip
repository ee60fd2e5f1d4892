// Prototype + microbenchmark: fp32 Winograd F(2x2,3x3) convolution on v_mfma_f32_16x16x4_f32 (gfx950), fully fused
// (input transform in registers from a raw LDS patch, 16 "frequency" GEMMs on the matrix pipe, output transform in
// registers) -- the measurement VERDICT r1 item 7 asks for.  Stand-alone: builds with hipcc, runs on the GPU box.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wino_proto.hip -o gpurun_out/wino_proto && gpurun_out/wino_proto
//
// Workgroup = 256 threads = 4 waves = 8x8 Winograd tiles (16x16 output pixels) x 64 output channels.  Wave w owns
// tile rows 2w, 2w+1 (16 tiles) x 64 channels x all 16 frequencies: 64 accumulator tiles of 16x16 (256 AGPRs).
// K loop over groups of 8 input channels = 2 MFMA k-steps: lane (t = l&15, kq = l>>4) reads its tile's 4x4 patch for
// channels 2kq, 2kq+1 (16 ds_read_b64), transforms it (64 VALU) into the A fragments of all 16 frequencies; B fragments
// (pre-transformed filters, laid out in HBM in the exact LDS image order) are staged per group by LDS-DMA.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

#ifndef ASM_MFMA
#define ASM_MFMA 1
#endif
#ifndef VAR
#define VAR 0
#endif
#if ASM_MFMA
#define MFMA(acc_, a_, b_) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc_) : "v"(a_), "v"(b_))
#else
#define MFMA(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_16x16x4f32(a_, b_, acc_, 0, 0, 0)
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOob = 0x80000000u;

struct WinoArgs {
  const float* x;      // NHWC, channel stride x_cs
  const float* u;      // [ct][G][16 f][4 nb][16 n][8 cin]
  const float* scale;  // [Cout] or null
  const float* shift;
  const float* res;    // NHWC residual or null
  float* y;
  int N, H, W, Cin, x_cs, Cout, y_cs, res_cs, relu;
  int TBY, TBX, nct;   // tile blocks per image (rows, cols), cout tiles
};

constexpr int PIX = 18 * 18, PITCH = 36;                       // patch pixels, floats per pixel in LDS (32 + 4 pad)
constexpr int PATCH_FLOATS = PIX * PITCH;                      // 11664 floats = 46656 B
constexpr int BG_FLOATS = 16 * 4 * 16 * 8;                     // 8192 floats = 32 KB per channel group
constexpr int LDS_BYTES = (2 * PATCH_FLOATS + 2 * BG_FLOATS) * 4;

__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
  return __builtin_bit_cast(f32x4, v);
}

template <bool RES, bool RELU>
__global__ __launch_bounds__(256, 1) void wino_f32(const WinoArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const patch = lds;                                     // [2][PIX][PITCH]
  float* const bbuf = lds + 2 * PATCH_FLOATS;                   // [2][BG_FLOATS]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = p.Cin >> 3;                                     // channel groups of 8
  const int NCH = p.Cin >> 5;                                   // patch chunks of 32 channels
  const int nsp = p.N * p.TBY * p.TBX, ntiles = nsp * p.nct;    // tile index = ct * nsp + spatial: concurrent workgroups share
                                                                // one 64-channel slab of transformed filters (L2-resident)
  // ---- patch staging role: slot s = tid + 256 i -> pixel s>>3, 16-byte piece s&7 ---------------------------------
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)kOob, 0x00020000);
  const int sl4 = 4 * (tid & 7);
  auto make_pv = [&](int tile, unsigned (&pv)[11], int& n, int& by, int& bx, int& ct) {
    ct = tile / nsp;
    int sp = tile - ct * nsp;
    bx = sp % p.TBX;
    sp /= p.TBX;
    by = sp % p.TBY;
    n = sp / p.TBY;
    const int iy0 = 16 * by - 1, ix0 = 16 * bx - 1;
    const bool live = tile < ntiles;
#pragma unroll
    for (int i = 0; i < 11; ++i) {
      const int pix = (tid >> 3) + 32 * i, py = (pix * 3641) >> 16, px = pix - py * 18;      // pix / 18 for pix < 1024
      const int iy = iy0 + py, ix = ix0 + px;
      const bool ok = live && pix < PIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      pv[i] = ok ? (unsigned)((((n * p.H + iy) * p.W + ix) * p.x_cs + sl4) * 4) : kOob;
    }
  };
  const int pw_off = (tid >> 3) * PITCH + sl4;                  // floats; + i * 32 * PITCH
  const bool last_ok = tid < (PIX * 8 - 2560);                  // slot i = 10 exists for 32 threads only

  // ---- B staging by LDS-DMA: group image = 32 pieces of 1 KiB; wave w issues pieces 8w .. 8w+7 ----------------------
  const float* const ub = p.u + (size_t)wave * 8 * 256 + lane * 4;
  auto dma_piece = [&](const float* src, int buf, int i) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 256),
                                     (__attribute__((address_space(3))) void*)(bbuf + buf * BG_FLOATS + wave * 8 * 256 + i * 256), 16, 0, 0);
  };

  // ---- compute roles --------------------------------------------------------------------------------------------
  const int t = lane & 15, kq = lane >> 4;
  const int ty = 2 * wave + (t >> 3), tx = t & 7;
  const int a_lane = ((2 * ty) * 18 + 2 * tx) * PITCH + 2 * kq;    // floats: patch pixel (2ty, 2tx), channels 2kq..
  const int b_lane = ((kq >> 1) * 16 + t) * 8 + (kq & 1) * 4;      // floats within a [2 kh][16 n][2][2 nb][2 cin] block

  f32x4 acc[16][4];
  f32x2 v[16], dn[16];
  auto load_patch = [&](int buf, int gl, int q) {                  // row q of the 4x4 patch -> dn[4q..4q+3]
    const float* pa = patch + buf * PATCH_FLOATS + a_lane + gl * 8;
#pragma unroll
    for (int c = 0; c < 4; ++c) dn[4 * q + c] = *reinterpret_cast<const f32x2*>(pa + (q * 18 + c) * PITCH);
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
  int tile = blockIdx.x;
  unsigned pv[11], pvl[11];
  int n, by, bx, ct, nn, nby, nbx, nct_;
  make_pv(tile, pv, n, by, bx, ct);
#pragma unroll
  for (int i = 0; i < 8; ++i) dma_piece(ub + (size_t)ct * G * BG_FLOATS, 0, i);
  {
    f32x4 r[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) r[i] = buf_load16(rsx, pv[i], 0);
#pragma unroll
    for (int i = 0; i < 10; ++i) *reinterpret_cast<f32x4*>(patch + pw_off + i * 32 * PITCH) = r[i];
    if (last_ok) *reinterpret_cast<f32x4*>(patch + pw_off + 10 * 32 * PITCH) = r[10];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) load_patch(0, 0, q);
#pragma unroll
  for (int c = 0; c < 4; ++c) rows_col(c);
#pragma unroll
  for (int r = 0; r < 4; ++r) cols_row(r);
  int pbuf = 0;                                                     // patch buffer holding the current chunk

  // one channel group (8 channels = 2 MFMA k-steps); gl = position in the 32-channel patch chunk (static)
  //   pvl / psoff: where the NEXT chunk's patch comes from (this tile's next chunk, or the next tile's chunk 0)
  //   bsrc: the NEXT group's filter image (null: nothing follows)
  auto group = [&](auto glc, const unsigned psoff, const float* bsrc) {
    constexpr int gl = decltype(glc)::value;
    const int abuf = gl == 3 ? pbuf ^ 1 : pbuf;                     // A fragments of the next group: next chunk after gl 3
    constexpr int agl = (gl + 1) & 3;
    f32x4 rr[4];
    const float* pb = bbuf + (gl & 1) * BG_FLOATS + b_lane;
    const float* pa = patch + abuf * PATCH_FLOATS + a_lane + agl * 8;
    f32x4 b0[2], b1[2];
    b0[0] = *reinterpret_cast<const f32x4*>(pb);
    b0[1] = *reinterpret_cast<const f32x4*>(pb + 256);
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      f32x4 (&bc)[2] = (f & 1) ? b1 : b0;
      f32x4 (&bn)[2] = (f & 1) ? b0 : b1;
      const f32x2 a = v[f];
      // 8 MFMAs; after each one a small piece of the other work, pinned in place, so that every non-MFMA instruction
      // issues in the shadow of an executing MFMA (one wave per SIMD: nothing else would fill the gap)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int j = k >> 2, nb = k & 3;
        MFMA(acc[f][nb], a[j], bc[nb >> 1][(nb & 1) * 2 + j]);
        if (k == 0 && f < 15) bn[0] = *reinterpret_cast<const f32x4*>(pb + ((f + 1) * 2) * 256);
        if (k == 1 && f < 15) bn[1] = *reinterpret_cast<const f32x4*>(pb + ((f + 1) * 2 + 1) * 256);
        if (f < 4) {                                                // patch row f of the next group: 4 x b64
          if (k == 2) { dn[4 * f + 0] = *reinterpret_cast<const f32x2*>(pa + (f * 18 + 0) * PITCH);
                        dn[4 * f + 1] = *reinterpret_cast<const f32x2*>(pa + (f * 18 + 1) * PITCH); }
          if (k == 3) { dn[4 * f + 2] = *reinterpret_cast<const f32x2*>(pa + (f * 18 + 2) * PITCH);
                        dn[4 * f + 3] = *reinterpret_cast<const f32x2*>(pa + (f * 18 + 3) * PITCH); }
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
        // staging: patch loads early, filter DMA in the first half of the group, LDS writes late
        if (k == 5 && gl < 3 && (f & 1) == 0 && f < 8 && gl * 4 + (f >> 1) < 11) rr[f >> 1] = buf_load16(rsx, pvl[gl * 4 + (f >> 1)], psoff);
        if (k == 6 && f < 8 && bsrc) dma_piece(bsrc, (gl + 1) & 1, f);
        if (k == 5 && gl < 3 && (f & 1) == 0 && f >= 8) {
          const int i = (f - 8) >> 1;
          float* pd = patch + (pbuf ^ 1) * PATCH_FLOATS + pw_off + (gl * 4 + i) * 32 * PITCH;
          if (gl * 4 + i < 10) *reinterpret_cast<f32x4*>(pd) = rr[i];
          else if (gl * 4 + i == 10) { if (last_ok) *reinterpret_cast<f32x4*>(pd) = rr[i]; }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  };
  using G0 = std::integral_constant<int, 0>;
  using G1 = std::integral_constant<int, 1>;
  using G2 = std::integral_constant<int, 2>;
  using G3 = std::integral_constant<int, 3>;

  {                                                                 // one tile per workgroup
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) acc[f][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 11; ++i) pvl[i] = pv[i];
    for (int chunk = 0; chunk < NCH; ++chunk) {                     // 32 channels
      const bool lastc = chunk + 1 == NCH;
      const float* ug = ub + ((size_t)ct * G + chunk * 4) * BG_FLOATS;
      const unsigned soff = lastc ? kOob : (unsigned)((chunk + 1) * 128);
      group(G0{}, soff, ug + 1 * BG_FLOATS);
      group(G1{}, soff, ug + 2 * BG_FLOATS);
      group(G2{}, soff, ug + 3 * BG_FLOATS);
      group(G3{}, soff, lastc ? nullptr : ug + 4 * BG_FLOATS);
      pbuf ^= 1;
    }

    // ---- epilogue: output transform per (tile, channel), scale/shift (+res) (+relu), store ----------------------------
    // acc[f][nb][r]: tile 4 kq + r of this wave = (tile row kq>>1, tile column 4 (kq&1) + r), channel ct*64 + nb*16 + (lane&15)
    {
      const int oy = 16 * by + 4 * wave + 2 * (kq >> 1), ox = 16 * bx + 8 * (kq & 1);
      const bool interior = 16 * by + 16 <= p.H && 16 * bx + 16 <= p.W && ct * 64 + 64 <= p.Cout;
      const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)kOob, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res), 0, (int)kOob, 0x00020000);
      const unsigned y_lane = (unsigned)((((n * p.H + oy) * p.W + ox) * p.y_cs + ct * 64 + t) * 4);
      const unsigned r_lane = (unsigned)((((n * p.H + oy) * p.W + ox) * p.res_cs + ct * 64 + t) * 4);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        __builtin_amdgcn_sched_barrier(0);
        const int co = ct * 64 + nb * 16 + t;
        const bool cok = co < p.Cout;
        const float sc = (p.scale && cok) ? p.scale[co] : 1.f;
        const float sh = (p.shift && cok) ? p.shift[co] : 0.f;
        unsigned voy[16];
        float rv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {                             // q = r*4 + i*2 + j
          const int r = q >> 2, i = (q >> 1) & 1, j = q & 1;
          const bool ok = interior || (cok && oy + i < p.H && ox + 2 * r + j < p.W);
          voy[q] = ok ? y_lane : kOob;
          if constexpr (RES)
            rv[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        rsr, ok ? r_lane : kOob, (unsigned)(((i * p.W + 2 * r + j) * p.res_cs + nb * 16) * 4), 0));
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s[2][4];
#pragma unroll
          for (int nu = 0; nu < 4; ++nu) {
            const float m0 = acc[0 + nu][nb][r], m1 = acc[4 + nu][nb][r], m2 = acc[8 + nu][nb][r], m3 = acc[12 + nu][nb][r];
            s[0][nu] = m0 + m1 + m2;
            s[1][nu] = m1 - m2 - m3;
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const float yv[2] = {s[i][0] + s[i][1] + s[i][2], s[i][1] - s[i][2] - s[i][3]};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int q = r * 4 + i * 2 + j;
              float o = fmaf(yv[j], sc, sh);
              if constexpr (RES) o += rv[q];
              if constexpr (RELU) o = fmaxf(o, 0.f);
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rsy, voy[q],
                                                    (unsigned)(((i * p.W + 2 * r + j) * p.y_cs + nb * 16) * 4), 0);
            }
          }
        }
      }
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------
static uint64_t rng_state = 0x1234567ull;
static float frand() {
  rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
  return (float)((rng_state >> 33) & 0xFFFFFF) / (float)0x1000000 * 2.f - 1.f;
}

// filters OIHW [Cout][Cin][3][3] -> U image [ct][G][16][4][16][8]
static std::vector<float> transform_filters(const std::vector<float>& w, int Cout, int Cin) {
  const int nct = (Cout + 63) / 64, G = Cin / 8;
  std::vector<float> u((size_t)nct * G * 8192, 0.f);
  const double Gm[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
  for (int co = 0; co < Cout; ++co)
    for (int ci = 0; ci < Cin; ++ci) {
      const float* g = &w[((size_t)co * Cin + ci) * 9];
      double tmp[4][3], U[4][4];
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 3; ++j) tmp[i][j] = Gm[i][0] * g[0 * 3 + j] + Gm[i][1] * g[1 * 3 + j] + Gm[i][2] * g[2 * 3 + j];
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) U[i][j] = tmp[i][0] * Gm[j][0] + tmp[i][1] * Gm[j][1] + tmp[i][2] * Gm[j][2];
      const int ct = co / 64, nb = (co % 64) / 16, nn = co % 16, g8 = ci / 8, q = ci % 8;
      for (int f = 0; f < 16; ++f)
        u[((((size_t)ct * G + g8) * 16 + f) * 2 + (nb >> 1)) * 256 + ((q >> 2) * 16 + nn) * 8 + ((q >> 1) & 1) * 4 + (nb & 1) * 2 + (q & 1)] = (float)U[f >> 2][f & 3];
    }
  return u;
}

static void cpu_conv(const std::vector<float>& x, const std::vector<float>& w, std::vector<double>& y, int N, int H, int W, int Cin, int Cout) {
  y.assign((size_t)N * H * W * Cout, 0.0);
  for (int n = 0; n < N; ++n)
    for (int oy = 0; oy < H; ++oy)
      for (int ox = 0; ox < W; ++ox)
        for (int co = 0; co < Cout; ++co) {
          double s = 0;
          for (int kh = 0; kh < 3; ++kh)
            for (int kw = 0; kw < 3; ++kw) {
              const int iy = oy + kh - 1, ix = ox + kw - 1;
              if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
              const float* xp = &x[(((size_t)n * H + iy) * W + ix) * Cin];
              const float* wp = &w[(size_t)co * Cin * 9 + kh * 3 + kw];
              for (int ci = 0; ci < Cin; ++ci) s += (double)xp[ci] * wp[(size_t)ci * 9];
            }
          y[(((size_t)n * H + oy) * W + ox) * Cout + co] = s;
        }
}

static double run(int N, int H, int W, int Cin, int Cout, bool check, int iters) {
  std::vector<float> x((size_t)N * H * W * Cin), w((size_t)Cout * Cin * 9), scale(Cout), shift(Cout);
  for (auto& v : x) v = fmaxf(frand() * 3.f, 0.f);              // post-ReLU-like activations
  const float ws = sqrtf(2.f / (9.f * Cin));
  for (auto& v : w) v = frand() * ws * 1.7f;
  for (int i = 0; i < Cout; ++i) { scale[i] = 0.8f + 0.4f * fabsf(frand()); shift[i] = 0.1f * frand(); }
  std::vector<float> u = transform_filters(w, Cout, Cin);
  float *dx, *du, *dy, *dsc, *dsh;
  CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&du, u.size() * 4)); CK(hipMalloc(&dy, (size_t)N * H * W * Cout * 4));
  CK(hipMalloc(&dsc, Cout * 4)); CK(hipMalloc(&dsh, Cout * 4));
  CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(du, u.data(), u.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsc, scale.data(), Cout * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsh, shift.data(), Cout * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dy, 0xFF, (size_t)N * H * W * Cout * 4));
  WinoArgs a;
  a.x = dx; a.u = du; a.scale = dsc; a.shift = dsh; a.res = nullptr; a.y = dy;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.x_cs = Cin; a.Cout = Cout; a.y_cs = Cout; a.res_cs = 0; a.relu = check ? 0 : 1;
  a.TBY = (H + 15) / 16; a.TBX = (W + 15) / 16; a.nct = (Cout + 63) / 64;
  const int ntiles = N * a.TBY * a.TBX * a.nct;
  const int grid = ntiles;
  auto kern = check ? wino_f32<false, false> : wino_f32<false, true>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS_BYTES, 0, a);
  CK(hipDeviceSynchronize());
  double worst = 0;
  if (check) {
    std::vector<float> y((size_t)N * H * W * Cout);
    CK(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
    std::vector<double> ref;
    cpu_conv(x, w, ref, N, H, W, Cin, Cout);
    double mx = 0;
    for (size_t i = 0; i < ref.size(); ++i) {
      const double r = ref[i] * scale[i % Cout] + shift[i % Cout];
      mx = fmax(mx, fabs(r));
      worst = fmax(worst, fabs((double)y[i] - r));
    }
    printf("check N=%d %dx%d Cin=%d Cout=%d: max|err| %.3e  max|ref| %.3e  rel %.3e\n", N, H, W, Cin, Cout, worst, mx, worst / mx);
  }
  if (iters > 0) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS_BYTES, 0, a);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS_BYTES, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    const double flops = 2.0 * N * H * W * Cout * 9.0 * Cin;
    printf("bench N=%d %dx%d Cin=%d Cout=%d: %.3f ms  %.1f TF direct-equivalent (%d tiles)\n", N, H, W, Cin, Cout, ms,
           flops / ms * 1e-9, ntiles);
    worst = ms;
  }
  CK(hipFree(dx)); CK(hipFree(du)); CK(hipFree(dy)); CK(hipFree(dsc)); CK(hipFree(dsh));
  return worst;
}

int main(int argc, char** argv) {
  printf("VAR=%d\n", VAR);
  run(2, 13, 21, 64, 64, true, 0);
  run(1, 33, 18, 96, 80, true, 0);
  if (argc > 1 && !strcmp(argv[1], "check")) return 0;
  run(48, 225, 400, 64, 64, false, 10);      // ResNet layer1 at B=8
  run(48, 113, 200, 128, 128, false, 10);    // layer2
  run(48, 57, 100, 256, 256, false, 10);     // layer3
  run(8, 128, 128, 512, 512, false, 10);     // bev_fusion conv1 (M=2)
  run(8, 128, 128, 512, 256, false, 10);     // bev_fusion conv2
  run(8, 128, 128, 256, 320, false, 10);     // fused head conv
  return 0;
}
