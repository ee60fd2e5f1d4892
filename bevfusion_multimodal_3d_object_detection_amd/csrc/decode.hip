// CenterNet post-processing on device: 3x3 keep mask, per-class top-K, top-K of the C*K pool,
// gather + box assembly.  ref src/centernet_target.py:326-452, src/fusion_detection.py:695-820.
//
// Every candidate gets a 64-bit composite key (order-preserving float bits << 32 | ~position):
// larger key = larger score, ties -> lower flattened position (a stable descending order; the
// reference's torch.topk leaves tie order unspecified).  Top-K = 4-pass byte-wise radix select
// on the score bits + an index-ordered pick among the scores equal to the K-th + a bitonic sort
// of the K survivors in LDS.
#include "common.h"

namespace {

__device__ __forceinline__ uint32_t ord_bits(float v) {
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord_float(uint32_t o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

// descending bitonic sort of n (power of two) 64-bit keys in LDS by the whole workgroup
__device__ void bitonic_desc(unsigned long long* s, int n) {
  for (int k = 2; k <= n; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = s[i], b = s[ixj];
          const bool desc = (i & k) == 0;
          if (desc ? (a < b) : (a > b)) { s[i] = b; s[ixj] = a; }
        }
      }
      __syncthreads();
    }
  }
}

struct DecArgs {
  const float* heat; const float* offset; const float* size; const float* rot; const float* vel;
  float* boxes; float* scores; long long* labels; float* velocities; int* count;
  uint32_t* keys;                 // [B*C][H*W] masked, order-preserving score bits
  unsigned long long* cls_top;    // [B*C][K]   composite keys of the per-class winners
  long long* pool_ind;
  int B, C, H, W, K, Kp, poolp, true_labels, raw_scores;
  float thresh, voxel, x_min, y_min;
};

// stage A: one workgroup per (frame, class), 1024 threads (round 3: 256 before -- a map of 128^2 .. 256^2 scores is walked five times by ONE
// workgroup and only B*C of them exist, so the walks are latency-bound and the wider workgroup is what parallelism there is)
constexpr int NTA = 1024;
// keep mask (ref _nms) -> order-preserving score bits, over the whole chip: a value survives iff it equals the 3x3 max around it, else 0
__global__ __launch_bounds__(256) void decode_keys(const DecArgs a) {
  const int n = a.H * a.W;
  const long long total = (long long)a.B * a.C * n;
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long bc = e / n;
    const int i = (int)(e - bc * n);
    const float* hp = a.heat + (size_t)bc * n;
    const int y = i / a.W, x = i - y * a.W;
    const float v = hp[i];
    float m = v;
    if (!a.raw_scores) for (int dy = -1; dy <= 1; ++dy) {
      const int yy = y + dy;
      if ((unsigned)yy >= (unsigned)a.H) continue;
      for (int dx = -1; dx <= 1; ++dx) {
        const int xx = x + dx;
        if ((unsigned)xx < (unsigned)a.W) m = fmaxf(m, hp[yy * a.W + xx]);
      }
    }
    a.keys[e] = ord_bits(m == v ? v : 0.f);
  }
}
__global__ __launch_bounds__(NTA) void decode_class_topk(const DecArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* sortbuf = reinterpret_cast<unsigned long long*>(smem);       // [Kp]
  int* hist = reinterpret_cast<int*>(smem + (size_t)a.Kp * 8);                     // [256]
  int* sh = hist + 256;                                                            // scratch [8 + NTA + 16]
  const int bc = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = a.H * a.W;
  const uint32_t* keys = a.keys + (size_t)bc * n;

  const int K = a.K < n ? a.K : n;
  // radix select: T = score bits of the K-th largest, need = how many == T still to take
  uint32_t prefix = 0, mask = 0;
  int need = K;
  for (int pass = 3; pass >= 0; --pass) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const int shift = pass * 8;
    for (int i = tid; i < n; i += NTA) {
      const uint32_t k = keys[i];
      if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255], 1);
    }
    __syncthreads();
    if (tid == 0) {
      int cum = 0, d = 255;
      for (; d > 0; --d) {
        if (cum + hist[d] >= need) break;
        cum += hist[d];
      }
      sh[0] = d;
      sh[1] = need - cum;
    }
    __syncthreads();
    prefix |= (uint32_t)sh[0] << shift;
    mask |= 0xFFu << shift;
    need = sh[1];
    __syncthreads();
  }
  const uint32_t T = prefix;
  const int n_gt = K - need;

  // collect: all > T (any order), then the first `need` (by index) of those == T
  for (int i = tid; i < a.Kp; i += NTA) sortbuf[i] = 0ull;
  if (tid == 0) sh[0] = 0;
  __syncthreads();
  const int chunk = (n + NTA - 1) / NTA, lo = tid * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
  int eq = 0;
  for (int i = lo; i < hi; ++i) {
    const uint32_t k = keys[i];
    if (k > T) sortbuf[atomicAdd(&sh[0], 1)] = ((unsigned long long)k << 32) | (uint32_t)(~(uint32_t)i);
    else if (k == T) ++eq;
  }
  // exclusive scan of the per-thread counts in thread (= index) order: wave scan by shuffles, the 16 wave totals by every thread
  int incl = eq;
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  if (lane == 63) sh[8 + wave] = incl;
  __syncthreads();
  int rank = incl - eq;
  for (int w = 0; w < wave; ++w) rank += sh[8 + w];
  for (int i = lo; i < hi && rank < need; ++i) {
    if (keys[i] == T) {
      sortbuf[n_gt + rank] = ((unsigned long long)T << 32) | (uint32_t)(~(uint32_t)i);
      ++rank;
    }
  }
  __syncthreads();
  bitonic_desc(sortbuf, a.Kp);
  for (int i = tid; i < a.K; i += NTA) a.cls_top[(size_t)bc * a.K + i] = i < K ? sortbuf[i] : 0ull;
}

// stage B: one workgroup per frame: top-K of the C*K pool, gather, boxes
__global__ __launch_bounds__(256) void decode_frame(const DecArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* pool = reinterpret_cast<unsigned long long*>(smem);          // [poolp]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int CK = a.C * a.K, n = a.H * a.W;
  const unsigned long long* src = a.cls_top + (size_t)b * CK;
  for (int j = tid; j < a.poolp; j += 256)
    pool[j] = j < CK ? ((src[j] & 0xFFFFFFFF00000000ull) | (uint32_t)(~(uint32_t)j)) : 0ull;
  __syncthreads();
  bitonic_desc(pool, a.poolp);
  __shared__ int cnt;
  if (tid == 0) cnt = 0;
  __syncthreads();
  for (int k = tid; k < a.K; k += 256) {
    const unsigned long long e = pool[k];
    const int j = (int)(~(uint32_t)e);                    // position in the (C,K) pool
    const bool valid = k < CK && j >= 0 && j < CK;
    float score = 0.f, box[7] = {0, 0, 0, 0, 0, 0, 0}, v2[2] = {0, 0};
    long long label = 0;
    if (valid) {
      score = ord_float((uint32_t)(e >> 32));
      const int c = j / a.K;
      int idx = (int)(~(uint32_t)src[j]);
      if (idx < 0 || idx >= n) idx = 0;
      const int y = idx / a.W, x = idx - y * a.W;
      const size_t p = (size_t)y * a.W + x;
      const float* off = a.offset + (size_t)b * 2 * n;
      const float* sz = a.size + (size_t)b * 3 * n;
      const float* rt = a.rot + (size_t)b * 2 * n;
      const float* vl = a.vel + (size_t)b * 2 * n;
      box[0] = ((float)x + off[p]) * a.voxel + a.x_min;
      box[1] = ((float)y + off[n + p]) * a.voxel + a.y_min;
      box[2] = 0.f - 1.0f;
      box[3] = sz[p]; box[4] = sz[n + p]; box[5] = sz[2 * (size_t)n + p];
      box[6] = atan2f(rt[p], rt[n + p]);
      v2[0] = vl[p]; v2[1] = vl[n + p];
      // the reference derives the class from an index that is already < H*W, so it is always 0
      label = a.true_labels ? c : 0;
      if (score > a.thresh) atomicAdd(&cnt, 1);
    }
    const size_t o = (size_t)b * a.K + k;
    a.scores[o] = score;
    a.labels[o] = label;
    if (a.pool_ind) a.pool_ind[o] = valid ? j : 0;
    for (int q = 0; q < 7; ++q) a.boxes[o * 7 + q] = box[q];
    a.velocities[o * 2] = v2[0];
    a.velocities[o * 2 + 1] = v2[1];
  }
  __syncthreads();
  if (tid == 0) a.count[b] = cnt;
}

static int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

}  // namespace

extern "C" size_t bevf_centernet_decode_work_bytes(int B, int C, int H, int W, int K) {
  return (size_t)B * C * H * W * sizeof(uint32_t) + (size_t)B * C * K * sizeof(unsigned long long) + 64;
}

extern "C" int bevf_centernet_decode_f32(const bevf_decode_desc* d, void* stream) {
  BEVF_REQUIRE(d && d->heat && d->offset && d->size && d->rot && d->vel, "decode: null prediction pointer");
  BEVF_REQUIRE(d->boxes && d->scores && d->labels && d->velocities && d->count && d->work, "decode: null output/work pointer");
  BEVF_REQUIRE(d->B > 0 && d->C > 0 && d->H > 0 && d->W > 0 && d->K > 0, "decode: empty shape");
  BEVF_REQUIRE((long long)d->H * d->W < (1ll << 30), "decode: map too large");
  BEVF_REQUIRE(d->K <= d->H * d->W, "decode: K=%d exceeds H*W=%d (torch.topk would raise)", d->K, d->H * d->W);
  DecArgs a;
  a.heat = d->heat; a.offset = d->offset; a.size = d->size; a.rot = d->rot; a.vel = d->vel;
  a.boxes = d->boxes; a.scores = d->scores; a.labels = (long long*)d->labels; a.velocities = d->velocities;
  a.count = d->count;
  a.B = d->B; a.C = d->C; a.H = d->H; a.W = d->W; a.K = d->K; a.true_labels = d->true_labels; a.raw_scores = d->raw_scores;
  a.pool_ind = (long long*)d->pool_ind;
  a.thresh = d->thresh; a.voxel = d->voxel; a.x_min = d->x_min; a.y_min = d->y_min;
  a.Kp = next_pow2(d->K);
  a.poolp = next_pow2(d->C * d->K);
  BEVF_REQUIRE(a.Kp <= 4096 && a.poolp <= 8192, "decode: K=%d / C*K=%d too large for the LDS sort", d->K, d->C * d->K);
  uintptr_t w = (reinterpret_cast<uintptr_t>(d->work) + 7) & ~uintptr_t(7);
  a.cls_top = reinterpret_cast<unsigned long long*>(w);
  a.keys = reinterpret_cast<uint32_t*>(w + (size_t)d->B * d->C * d->K * sizeof(unsigned long long));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t ldsA = (size_t)a.Kp * 8 + (256 + 8 + 16) * sizeof(int);
  {
    const long long total = (long long)d->B * d->C * d->H * d->W, g = (total + 255) / 256;
    hipLaunchKernelGGL(decode_keys, dim3((unsigned)(g < 8192 ? g : 8192)), dim3(256), 0, st, a);
  }
  hipLaunchKernelGGL(decode_class_topk, dim3(d->B * d->C), dim3(NTA), ldsA, st, a);
  hipLaunchKernelGGL(decode_frame, dim3(d->B), dim3(256), (size_t)a.poolp * 8, st, a);
  return bevf_check_launch("bevf_centernet_decode_f32");
}
