// PointNet front: conv1 -> conv2 -> conv3 (K -> 64 -> 128 -> 256, each + folded BatchNorm + ReLU; ref src/encoders.py:289-291)
// as ONE kernel whose 64- and 128-wide activations never leave the register file.
//
// A wave owns 32 points and computes the TRANSPOSED problem  H_out^T [channels x 32 points] = W [c_out x c_in] . H_in^T  with
// v_mfma_f32_32x32x2_f32: A = a 32-row block of W, B = the activations (k = input channel, column = point).  The 32x32
// accumulator of a lane holds, for ITS point (column lane % 32), the channels  32*mb + 8*j + 4*(lane / 32) + r  in register
// 4*j + r -- and the next layer's B operand at k-step t wants "the lane half's channel of step t" for that same point.  The
// order in which a GEMM walks its reduction index is free as long as A and B agree, so k-step t = 4*j + r of input block kb
// is DEFINED as channel 32*kb + 8*j + 4*(lane / 32) + r: register t of the previous layer's accumulator (after scale / shift /
// ReLU, in place) IS the B operand of step t.  No LDS, no shuffle, no barrier between the layers.  The A side pays for it on
// the host, once per weight version: bevf_pointnet_front_pack_f32 stores W as [mb][kb][j][lane][r] fragments, so a lane's A
// operands for four consecutive steps are one 16-byte load and a wave instruction reads 1 KB contiguous (L2-resident: 32 +
// 128 KB for the two layers, shared by every wave of the chip).
// Layer 1 (K <= 8 inputs) is 4..8 FMAs per output on the vector ALU, straight into that register layout; its 64 x K filter and
// the three layers' scale / shift vectors sit in LDS (broadcast reads).
#include "common.h"

namespace {

constexpr int C1 = 64, C2 = 128, C3 = 256;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 frag_load(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
}

__device__ __forceinline__ float act_relu(float v) { return !(v > 0.f) ? 0.f : v; }   // the unfused kernels' expression

struct FrontArgs {
  const float* x;          // [M][K]
  const float* w1;         // [64][K]
  const float* s1; const float* b1;
  const float* w2f;        // fragments of [128][64]
  const float* s2; const float* b2;
  const float* w3f;        // fragments of [256][128]
  const float* s3; const float* b3;
  float* y;                // [M][256]
  int M, K, ntiles;
};

// KP: the layer-1 reduction length the code is unrolled for (4, 5, or 8 with zero-padded filter columns): same FMA order as
// pointwise_smallk, so layer 1 is bit-identical to the separate kernel.  Two workgroups per CU (<= 256 registers a lane).
template <int KP>
__global__ __launch_bounds__(256, 2) void pointnet_front(FrontArgs a) {
  __shared__ float lw1[C1 * KP];
  __shared__ float ls1[C1], lb1[C1], ls2[C2], lb2[C2], ls3[C3], lb3[C3];
  for (int i = threadIdx.x; i < C1 * KP; i += 256) lw1[i] = (i % KP) < a.K ? a.w1[(i / KP) * a.K + i % KP] : 0.f;
  for (int i = threadIdx.x; i < C1; i += 256) { ls1[i] = a.s1[i]; lb1[i] = a.b1[i]; }
  for (int i = threadIdx.x; i < C2; i += 256) { ls2[i] = a.s2[i]; lb2[i] = a.b2[i]; }
  for (int i = threadIdx.x; i < C3; i += 256) { ls3[i] = a.s3[i]; lb3[i] = a.b3[i]; }
  __syncthreads();                                          // the only barrier; every wave reaches it before its tile loop
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, h = lane >> 5;
  // a lane's channels are 4*h + (compile-time constant): one LDS base per table, everything else an immediate offset
  const float* lw1h = lw1 + 4 * h * KP;
  const float *ls1h = ls1 + 4 * h, *lb1h = lb1 + 4 * h, *ls2h = ls2 + 4 * h, *lb2h = lb2 + 4 * h, *ls3h = ls3 + 4 * h, *lb3h = lb3 + 4 * h;
  // fragment loads are buffer loads: descriptor + byte offset of the group in scalar registers (constants, or advanced on the
  // scalar unit with the row-block loop), the lane's 16-byte slot as the ONE vector offset shared by all 64 loads of a tile
  // (plain pointers made hipcc keep a 64-bit vector address per load, which spilled to scratch)
  const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w2f), 0, C2 * C1 * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t r3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w3f), 0, C3 * C2 * 4, 0x00020000);
  const unsigned ulane = (unsigned)lane * 16u;

#define BEVF_LOAD_GROUP(W, PTR)                                     \
    _Pragma("unroll") for (int q = 0; q < 8; ++q) W[q] = frag_load(r2, ulane, (unsigned)(((PTR) + q) * 1024)); \
    __builtin_amdgcn_sched_barrier(0);
  // a lane's point of the wave's first tile; the next tile's point is requested as soon as layer 1 has consumed this one
  float xv[KP];
  {
    const int m0 = (blockIdx.x * 4 + wave) * 32 + col;
    const float* xp = a.x + (size_t)(m0 < a.M ? m0 : a.M - 1) * a.K;
#pragma unroll
    for (int k = 0; k < KP; ++k) xv[k] = xp[(KP != 8 || k < a.K) ? k : 0];       // KP = 8: columns >= K meet zero filter entries
  }
  for (int tile = blockIdx.x * 4 + wave; tile < a.ntiles; tile += gridDim.x * 4) {
    const int m = tile * 32 + col;
    const bool ok = m < a.M;
    f32x4 wa[8], wb[8];
    BEVF_LOAD_GROUP(wa, 0)                               // layer-2 fragments are [mb][kb][j]: 8 consecutive per row block
    BEVF_LOAD_GROUP(wb, 8)

    // ---- layer 1 on the vector ALU, into the accumulator layout ------------------------------------------------------
    float h1[2][16];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int c = 32 * kb + 8 * (t >> 2) + (t & 3);          // + 4*h in the bases
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < KP; ++k) acc = fmaf(xv[k], lw1h[c * KP + k], acc);
        h1[kb][t] = act_relu(fmaf(acc, ls1h[c], lb1h[c]));
      }

    {
      const int mn = (tile + (int)gridDim.x * 4) * 32 + col;      // next tile's point (clamped: a finished wave re-reads the last row)
      const float* xp = a.x + (size_t)(mn < a.M ? mn : a.M - 1) * a.K;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < KP; ++k) xv[k] = xp[(KP != 8 || k < a.K) ? k : 0];       // KP = 8: columns >= K meet zero filter entries
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- layers 2 and 3 ---------------------------------------------------------------------------------------------
    // One GROUP = 32 MFMAs fed by 8 fragment loads (16 B a lane each).  The fragments of group g+1 are requested before the
    // MFMAs of group g are issued (two register sets, wa / wb); sched_barriers keep hipcc from hoisting every load of the
    // unrolled code to the top (it did: 426 registers, one wave per SIMD).  Layer 2: a group is one 32-row block of the 128
    // outputs (2 input blocks x 4 j).  Layer 3: a group is one input block kb for TWO row blocks (independent accumulators).
    f32x16 h2[4];
#define BEVF_MFMA_GROUP_L2(W, MB)                                                                          \
    {                                                                                                      \
      f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      \
      _Pragma("unroll") for (int q = 0; q < 8; ++q)                                                        \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                      \
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W[q][r], h1[q >> 2][4 * (q & 3) + r], acc, 0, 0, 0);  \
      _Pragma("unroll") for (int t = 0; t < 16; ++t) {                                                     \
        const int c = 32 * (MB) + 8 * (t >> 2) + (t & 3);                                                  \
        acc[t] = act_relu(fmaf(acc[t], ls2h[c], lb2h[c]));                                                 \
      }                                                                                                    \
      h2[MB] = acc;                                                                                        \
    }
    BEVF_MFMA_GROUP_L2(wa, 0)
    __builtin_amdgcn_sched_barrier(0);
    BEVF_LOAD_GROUP(wa, 16)
    BEVF_MFMA_GROUP_L2(wb, 1)
    __builtin_amdgcn_sched_barrier(0);
    BEVF_LOAD_GROUP(wb, 24)
    BEVF_MFMA_GROUP_L2(wa, 2)
    __builtin_amdgcn_sched_barrier(0);
    // Layer 3 walks 16 STEPS per pair of row blocks (step s: input block s / 4, j = s % 4; two fragments, 8 MFMAs on two
    // independent accumulators) behind a RING of 8 fragment pairs: the pair of step s+7 is requested before the MFMAs of step s
    // are issued.  Stores share `vmcnt` with the loads on gfx950, so a wait for fragments requested AFTER a row block's stores
    // also waits for their write acknowledgements: with the ring that first happens 7 steps (56 MFMAs) after the stores, not
    // one group (32) as with two alternating register sets -- same 64 registers.
    f32x4 ring[8][2];
#define BEVF_LOAD_STEP(SLOT, MP, S)                                                                            \
    ring[SLOT][0] = frag_load(r3, ulane, (unsigned)(((2 * (MP)) * 16 + (S)) * 1024));                         \
    ring[SLOT][1] = frag_load(r3, ulane, (unsigned)(((2 * (MP) + 1) * 16 + (S)) * 1024));                     \
    __builtin_amdgcn_sched_barrier(0);
    BEVF_LOAD_STEP(0, 0, 0)                                // wa's registers are free from here on
    BEVF_LOAD_STEP(1, 0, 1)
    BEVF_LOAD_STEP(2, 0, 2)
    BEVF_LOAD_STEP(3, 0, 3)
    BEVF_MFMA_GROUP_L2(wb, 3)
    __builtin_amdgcn_sched_barrier(0);
    BEVF_LOAD_STEP(4, 0, 4)
    BEVF_LOAD_STEP(5, 0, 5)
    BEVF_LOAD_STEP(6, 0, 6)

    float* yp = a.y + (size_t)m * C3 + 4 * h;
#define BEVF_MFMA_STEP(S)                                                                                      \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                            \
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[(S) & 7][0][r], h2[(S) >> 2][4 * ((S) & 3) + r], acc0, 0, 0, 0); \
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[(S) & 7][1][r], h2[(S) >> 2][4 * ((S) & 3) + r], acc1, 0, 0, 0); \
    }                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int mp = 0; mp < 4; ++mp) {
      f32x16 acc0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      f32x16 acc1 = acc0;
      const int mpn = mp < 3 ? mp + 1 : 3;               // the last round re-reads its own first steps (in bounds, unused)
#pragma unroll
      for (int st = 0; st < 16; ++st) {
        if (st < 9) {
          BEVF_LOAD_STEP((st + 7) & 7, mp, st + 7)
        } else {
          BEVF_LOAD_STEP((st + 7) & 7, mpn, st - 9)
        }
        BEVF_MFMA_STEP(st)
      }
      if (ok) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int mb = 2 * mp + half;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int c = 32 * mb + 8 * j;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = act_relu(fmaf(half ? acc1[4 * j + r] : acc0[4 * j + r], ls3h[c + r], lb3h[c + r]));
            *reinterpret_cast<f32x4*>(yp + 32 * mb + 8 * j) = v;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef BEVF_LOAD_STEP
#undef BEVF_MFMA_STEP
#undef BEVF_MFMA_GROUP_L2
#undef BEVF_LOAD_GROUP
  }
}

// W [Cout][Cin] row-major -> fragments [Cout/32][Cin/32][4][64 lanes][4]:  element (mb, kb, j, lane, r) = W[32 mb + lane % 32][32 kb + 8 j + 4 (lane / 32) + r]
__global__ void pack_fragments(const float* __restrict__ w, float* __restrict__ wf, int Cout, int Cin) {
  const int total = Cout * Cin;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int r = i & 3, lane = (i >> 2) & 63, j = (i >> 8) & 3;
    const int blk = i >> 10, kb = blk % (Cin / 32), mb = blk / (Cin / 32);
    wf[i] = w[(size_t)(32 * mb + (lane & 31)) * Cin + 32 * kb + 8 * j + 4 * (lane >> 5) + r];
  }
}

}  // namespace

extern "C" int bevf_pointnet_front_pack_f32(const float* w, float* wf, int Cout, int Cin, void* stream) {
  BEVF_REQUIRE(w && wf, "pointnet_front_pack: null pointer");
  BEVF_REQUIRE(Cout > 0 && Cin > 0 && Cout % 32 == 0 && Cin % 32 == 0, "pointnet_front_pack: Cout and Cin must be multiples of 32 (Cout=%d Cin=%d)",
               Cout, Cin);
  const int total = Cout * Cin;
  hipLaunchKernelGGL(pack_fragments, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), w, wf, Cout, Cin);
  return bevf_check_launch("bevf_pointnet_front_pack_f32");
}

extern "C" int bevf_pointnet_front_f32(const float* x, int M, int K, const float* w1, const float* s1, const float* b1,
                                       const float* w2f, const float* s2, const float* b2, const float* w3f, const float* s3,
                                       const float* b3, float* y, void* stream) {
  BEVF_REQUIRE(x && w1 && s1 && b1 && w2f && s2 && b2 && w3f && s3 && b3 && y, "pointnet_front: null pointer");
  BEVF_REQUIRE(M > 0 && K > 0 && K <= 8, "pointnet_front: need M > 0 and 0 < K <= 8 (M=%d K=%d)", M, K);
  BEVF_REQUIRE(bevf_aligned16(w2f) && bevf_aligned16(w3f) && bevf_aligned16(y), "pointnet_front: fragments and output must be 16-byte aligned");
  FrontArgs a{x, w1, s1, b1, w2f, s2, b2, w3f, s3, b3, y, M, K, (M + 31) / 32};
  const int blocks = (a.ntiles + 3) / 4;
  const dim3 grid(blocks < 512 ? blocks : 512);          // two resident workgroups per CU, each wave strides over the tiles
  if (K == 4)
    hipLaunchKernelGGL(pointnet_front<4>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), a);
  else if (K == 5)
    hipLaunchKernelGGL(pointnet_front<5>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), a);
  else
    hipLaunchKernelGGL(pointnet_front<8>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return bevf_check_launch("bevf_pointnet_front_f32");
}
