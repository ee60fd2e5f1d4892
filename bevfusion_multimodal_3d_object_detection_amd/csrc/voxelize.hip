// Hard voxelisation of LiDAR / radar sweeps: points -> (voxel_features, voxel_coords,
// num_points) in the layout VFELayer / VoxelNetLiDAREncoder take (ref src/encoders.py:313-321, :385-387).
// The reference has NO voxel assignment anywhere, so the semantics are the standard
// deterministic "hard voxelisation" and parity is against the oracle's sequential restatement:
//   * cell = floor((p - range_min) / voxel_size) per axis in fp32; points outside the grid are dropped;
//   * voxels are numbered in order of their FIRST point; at most max_voxels are kept;
//   * a voxel keeps its first max_points points in input order, zero padded.
// Pipeline (all frames of the batch at once):
//   keys  (frame, cell, point index) packed in 64 bits                          -- one thread per point
//   sort  stable LSD radix sort on the cell bits, 8 bits a pass, MANY workgroups per frame (round 3; rounds 1-2 walked a frame with one
//         workgroup = 8 of 256 CUs at B = 8): per pass  vox_hist  (digit histogram of every 1024-key tile)  ->  vox_offsets  (per frame:
//         where each tile's keys of each digit go: digits below + same digit in earlier tiles)  ->  vox_scatter  (a key's rank among
//         its wave's keys of the same digit from 8 ballots + a popcount, the tile's waves / rounds chained through LDS counters) --
//         groups cells, keeps point order
//   heads a sorted position is a segment head when its (frame,cell) differs from its predecessor; the head
//         marks its first point in a per-point flag array                         -- one thread per position
//   scan  per-frame exclusive scan of the flags in POINT order = voxel id in first-appearance order: vox_flagcount (heads per
//         1024-point tile) -> vox_vid (a tile adds up the counts of the tiles before it, then scans its own flags)
//   fill  one wave per voxel copies its first max_points points with 16-byte-free, coalesced row copies
#include "common.h"

namespace {

struct VoxArgs {
  const float* pts;      // [B][N][C]
  int B, N, C;
  float x0, y0, z0, vx, vy, vz;
  int gx, gy, gz;
  int max_points, max_voxels;
};

constexpr unsigned long long kInvalidCell = 0xFFFFFFFFull;

__global__ __launch_bounds__(256) void vox_keys(const VoxArgs a, unsigned long long* __restrict__ keys) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  const long long total = (long long)a.B * a.N;
  if (i >= total) return;
  const int b = (int)(i / a.N), n = (int)(i - (long long)b * a.N);
  const float* p = a.pts + (size_t)i * a.C;
  // fp32, operation for operation what the oracle does: floor((x - x0) / vx)
  const float fx = floorf(__fdiv_rn(__fsub_rn(p[0], a.x0), a.vx));
  const float fy = floorf(__fdiv_rn(__fsub_rn(p[1], a.y0), a.vy));
  const float fz = floorf(__fdiv_rn(__fsub_rn(p[2], a.z0), a.vz));
  unsigned long long cell = kInvalidCell;
  if (fx >= 0.f && fx < (float)a.gx && fy >= 0.f && fy < (float)a.gy && fz >= 0.f && fz < (float)a.gz)
    cell = ((unsigned long long)(int)fz * a.gy + (int)fy) * a.gx + (int)fx;
  // frame in the top bits so one sort serves the whole batch; invalid points sort to the end of their frame
  keys[i] = ((unsigned long long)b << 52) | (cell << 20) | (unsigned long long)n;
}

// ---- stable LSD radix sort of every frame's keys by cell (bits 20..51), 8 bits per pass --------------------------------------
// The point index in the low 20 bits is ascending in the input and a stable sort keeps it so.  A tile = 1024 consecutive keys
// of one frame, one 256-thread workgroup (4 waves x 4 rounds of 64 keys, in order).
constexpr int VT = 1024;                                         // keys per tile

// hist[frame][tile][digit] = keys of the tile whose digit q is `digit`
__global__ __launch_bounds__(256) void vox_hist(const unsigned long long* __restrict__ keys, int* __restrict__ hist, int N, int T,
                                                  int q) {
  __shared__ int h[256];
  const int tid = threadIdx.x, tile = blockIdx.x % T, b = blockIdx.x / T;
  h[tid] = 0;
  __syncthreads();
  const unsigned long long* src = keys + (size_t)b * N;
#pragma unroll
  for (int r = 0; r < VT / 256; ++r) {
    const int i = tile * VT + r * 256 + tid;
    if (i < N) atomicAdd(&h[((unsigned)(src[i] >> 20) >> (8 * q)) & 255u], 1);
  }
  __syncthreads();
  hist[((size_t)b * T + tile) * 256 + tid] = h[tid];
}

// per frame (one workgroup, thread = digit): hist -> the position of each tile's first key of each digit
__global__ __launch_bounds__(256) void vox_offsets(int* __restrict__ hist, int T) {
  __shared__ int wtot[4];
  const int d = threadIdx.x, lane = d & 63, wave = d >> 6;
  int* h = hist + (size_t)blockIdx.x * T * 256;
  int total = 0;
  for (int t = 0; t < T; ++t) total += h[t * 256 + d];            // coalesced across the digits
  int incl = total;                                               // exclusive scan over the 256 digit totals
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    const int up = __shfl_up(incl, s);
    if (lane >= s) incl += up;
  }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  int run = incl - total;
  for (int w = 0; w < wave; ++w) run += wtot[w];
  for (int t = 0; t < T; ++t) {                                   // same digit, earlier tiles
    const int c = h[t * 256 + d];
    h[t * 256 + d] = run;
    run += c;
  }
}

// every key to  offs[tile][digit] + (keys of that digit earlier in the tile)
__global__ __launch_bounds__(256) void vox_scatter(const unsigned long long* __restrict__ src_all, unsigned long long* __restrict__ dst_all,
                                                     const int* __restrict__ offs, int N, int T, int q) {
  __shared__ int base[256];
  __shared__ int wcnt[4][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tile = blockIdx.x % T, b = blockIdx.x / T;
  const unsigned long long* src = src_all + (size_t)b * N;
  unsigned long long* dst = dst_all + (size_t)b * N;
  base[tid] = offs[((size_t)b * T + tile) * 256 + tid];
#pragma unroll 1
  for (int r = 0; r < VT / 256; ++r) {
#pragma unroll
    for (int w = 0; w < 4; ++w) wcnt[w][tid] = 0;
    __syncthreads();
    const int i = tile * VT + r * 256 + tid;
    const bool valid = i < N;
    const unsigned long long key = valid ? src[i] : 0ull;
    const unsigned d = ((unsigned)(key >> 20) >> (8 * q)) & 255u;
    unsigned long long peers = __ballot(valid);                   // narrowed down to the lanes holding the same digit
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const bool set = (d >> bit) & 1u;
      const unsigned long long bb = __ballot(set);
      peers &= set ? bb : ~bb;
    }
    const int rank = __popcll(peers & ((1ull << lane) - 1ull));
    if (valid && rank == 0) wcnt[wave][d] = __popcll(peers);
    __syncthreads();
    {                                                             // thread = digit: this round's waves in order, then advance the base
      int run = base[tid];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int c = wcnt[w][tid];
        wcnt[w][tid] = run;
        run += c;
      }
      base[tid] = run;
    }
    __syncthreads();
    if (valid) dst[wcnt[wave][d] + rank] = key;
    __syncthreads();
  }
}

// head_flag[b][n] = sorted position + 1 of the segment whose first point is n (0 elsewhere)
__global__ __launch_bounds__(256) void vox_heads(const unsigned long long* __restrict__ keys, int* __restrict__ head_flag,
                                                  long long total, int N) {
  const long long j = blockIdx.x * 256ll + threadIdx.x;
  if (j >= total) return;
  const unsigned long long k = keys[j];
  const unsigned long long cell = (k >> 20) & 0xFFFFFFFFull;
  if (cell == kInvalidCell) return;
  const bool head = j == 0 || (keys[j - 1] >> 20) != (k >> 20);
  if (head) {
    const int b = (int)(k >> 52), n = (int)(k & 0xFFFFF);
    head_flag[(size_t)b * N + n] = (int)(j - (long long)b * N) + 1;       // position inside the frame's slice
  }
}

// tcount[frame][tile] = segment heads among the tile's 1024 points
__global__ __launch_bounds__(256) void vox_flagcount(const int* __restrict__ head_flag, int* __restrict__ tcount, int N, int T) {
  __shared__ int wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tile = blockIdx.x % T, b = blockIdx.x / T;
  int c = 0;
#pragma unroll
  for (int r = 0; r < VT / 256; ++r) {
    const int n = tile * VT + r * 256 + tid;
    c += (n < N && head_flag[(size_t)b * N + n] != 0) ? 1 : 0;
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) c += __shfl_xor(c, s);
  if (lane == 0) wsum[wave] = c;
  __syncthreads();
  if (tid == 0) tcount[(size_t)b * T + tile] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// voxel id of every point = heads before it in POINT order: the tiles before this one (summed here, at most 1024 counts) + an
// in-tile scan in point order; the frame's last tile also leaves num_voxels
__global__ __launch_bounds__(256) void vox_vid(const int* __restrict__ head_flag, const int* __restrict__ tcount, int* __restrict__ vid,
                                                 int* __restrict__ num_voxels, int N, int T, int max_voxels) {
  __shared__ int wsum[4];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tile = blockIdx.x % T, b = blockIdx.x / T;
  int before = 0;
  for (int t = tid; t < tile; t += 256) before += tcount[(size_t)b * T + t];
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) before += __shfl_xor(before, s);
  if (lane == 0) wsum[wave] = before;
  __syncthreads();
  if (tid == 0) carry = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
#pragma unroll 1
  for (int r = 0; r < VT / 256; ++r) {
    const int n = tile * VT + r * 256 + tid;
    const int f = (n < N && head_flag[(size_t)b * N + n] != 0) ? 1 : 0;
    int incl = f;                                                 // inclusive scan inside the wave
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      const int t = __shfl_up(incl, s);
      if (lane >= s) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const int excl = carry + woff + incl - f;
    if (n < N) vid[(size_t)b * N + n] = f ? excl : -1;
    __syncthreads();
    if (tid == 255) carry = excl + f;
    __syncthreads();
  }
  if (tile == T - 1 && tid == 0) num_voxels[b] = carry < max_voxels ? carry : max_voxels;
}

// one wave per segment head: copy the first max_points points of the voxel, write coords / counts
__global__ __launch_bounds__(256) void vox_fill(const VoxArgs a, const unsigned long long* __restrict__ keys,
                                                 const int* __restrict__ head_flag, const int* __restrict__ vid,
                                                 float* __restrict__ feats, long long* __restrict__ coords,
                                                 int* __restrict__ npts) {
  const long long wave_global = (blockIdx.x * 256ll + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const long long total = (long long)a.B * a.N;
  if (wave_global >= total) return;
  const int b = (int)(wave_global / a.N);
  const int hf = head_flag[wave_global];
  if (hf == 0) return;                                        // wave-uniform: the wave handles point n as a voxel head
  const int v = vid[wave_global];
  if (v < 0 || v >= a.max_voxels) return;
  const long long j0 = (long long)b * a.N + (hf - 1);
  const unsigned long long bc = keys[j0] >> 20;
  // segment length, capped at max_points: lanes probe consecutive sorted positions
  int cnt = 0;
  for (int base = 0; base < a.max_points; base += 64) {
    const long long j = j0 + base + lane;
    const bool in = base + lane < a.max_points && j < (long long)(b + 1) * a.N && (keys[j] >> 20) == bc;
    const unsigned long long m = __ballot(in);
    cnt += __popcll(m);
    if (m != ~0ull) break;
  }
  const unsigned long long cell = bc & 0xFFFFFFFFull;
  if (lane == 0) {
    npts[(size_t)b * a.max_voxels + v] = cnt;
    const int cx = (int)(cell % a.gx), cy = (int)((cell / a.gx) % a.gy), cz = (int)(cell / ((unsigned long long)a.gx * a.gy));
    long long* c = coords + ((size_t)b * a.max_voxels + v) * 3;
    c[0] = cz; c[1] = cy; c[2] = cx;                          // (D,H,W) index order of ref src/encoders.py:399-410
  }
  float* dst = feats + ((size_t)b * a.max_voxels + v) * a.max_points * a.C;
  for (int e = lane; e < cnt * a.C; e += 64) {
    const int s = e / a.C, c = e - s * a.C;
    const int pn = (int)(keys[j0 + s] & 0xFFFFF);
    dst[s * a.C + c] = a.pts[((size_t)b * a.N + pn) * a.C + c];
  }
}

// ---- dense scatter of per-voxel features into the (D,H,W) grid: ref src/encoders.py:399-410 ---------------------------------
// `feature_grid[b, :, c0, c1, c2] = features.T`: rows are written in order, so when several rows name one cell the LAST
// one wins.  Pass 1 records the highest row index per cell (integer atomicMax: order-free), pass 2 writes each cell
// from its owner (or zeros) -- coalesced along the cells for every channel.
__global__ __launch_bounds__(256) void scatter_owner(const long long* __restrict__ coords, const int* __restrict__ num_voxels,
                                                      int* __restrict__ owner, int B, int Nv, int D, int H, int W) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= (long long)B * Nv) return;
  const int b = (int)(i / Nv), v = (int)(i - (long long)b * Nv);
  if (num_voxels && v >= num_voxels[b]) return;
  const long long* c = coords + i * 3;
  const long long z = c[0], y = c[1], x = c[2];
  if (z < 0 || z >= D || y < 0 || y >= H || x < 0 || x >= W) return;
  atomicMax(&owner[(size_t)b * D * H * W + ((size_t)z * H + y) * W + x], v);
}

__global__ __launch_bounds__(256) void scatter_write(const float* __restrict__ feats, const int* __restrict__ owner,
                                                      float* __restrict__ out, int B, int Nv, int C, long long cells) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= (long long)B * cells) return;
  const int b = (int)(i / cells);
  const long long cell = i - (long long)b * cells;
  const int v = owner[i];
  const float* f = v >= 0 ? feats + ((size_t)b * Nv + v) * C : nullptr;
  float* o = out + (size_t)b * C * cells + cell;
  for (int c = 0; c < C; ++c) o[(size_t)c * cells] = f ? f[c] : 0.f;
}

}  // namespace

extern "C" size_t bevf_voxelize_work_bytes(int B, int N) {
  const long long n = (long long)B * N, T = (N + VT - 1) / VT;
  return (size_t)n * 16 + (size_t)n * 8 + (size_t)B * T * 257 * 4 + 256;   // keys (two buffers), head_flag + vid, tile histograms + counts
}

extern "C" int bevf_voxelize_f32(const bevf_voxelize_desc* d, void* stream) {
  BEVF_REQUIRE(d && d->points && d->voxel_features && d->voxel_coords && d->num_points && d->num_voxels && d->work,
               "voxelize: null pointer");
  BEVF_REQUIRE(d->B > 0 && d->B < 4096 && d->N > 0 && d->N < (1 << 20) && d->C >= 3, "voxelize: need B < 4096, N < 2^20, C >= 3");
  BEVF_REQUIRE(d->max_points > 0 && d->max_voxels > 0, "voxelize: max_points / max_voxels must be positive");
  VoxArgs a;
  a.pts = d->points; a.B = d->B; a.N = d->N; a.C = d->C;
  a.x0 = d->pc_range[0]; a.y0 = d->pc_range[1]; a.z0 = d->pc_range[2];
  a.vx = d->voxel_size[0]; a.vy = d->voxel_size[1]; a.vz = d->voxel_size[2];
  BEVF_REQUIRE(a.vx > 0 && a.vy > 0 && a.vz > 0, "voxelize: voxel sizes must be positive");
  a.gx = (int)lroundf((d->pc_range[3] - d->pc_range[0]) / a.vx);
  a.gy = (int)lroundf((d->pc_range[4] - d->pc_range[1]) / a.vy);
  a.gz = (int)lroundf((d->pc_range[5] - d->pc_range[2]) / a.vz);
  BEVF_REQUIRE(a.gx > 0 && a.gy > 0 && a.gz > 0 && (long long)a.gx * a.gy * a.gz < 0xFFFFFFFFll, "voxelize: bad grid");
  a.max_points = d->max_points; a.max_voxels = d->max_voxels;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long n = (long long)d->B * d->N;
  char* w = static_cast<char*>(d->work);
  w = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(w) + 255) & ~uintptr_t(255));
  unsigned long long* keys_in = reinterpret_cast<unsigned long long*>(w);
  unsigned long long* keys = keys_in + n;
  int* head_flag = reinterpret_cast<int*>(keys + n);
  int* vid = head_flag + n;
  const unsigned grid = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(vox_keys, dim3(grid), dim3(256), 0, st, a, keys_in);
  // frame (12 bits) | cell (32 bits) | point (20 bits).  Frames are independent slices; each is sorted on as many 8-bit digits as
  // the grid's cell count needs (the invalid cell, all ones, then still sorts behind every valid cell)
  const unsigned long long ncells = (unsigned long long)a.gx * a.gy * a.gz;
  int passes = 1;
  while (passes < 4 && (1ull << (8 * passes)) <= ncells) ++passes;
  const int T = (d->N + VT - 1) / VT;
  int* hist = vid + n;
  int* tcount = hist + (size_t)d->B * T * 256;
  const dim3 tiles((unsigned)(d->B * T));
  unsigned long long *src = keys_in, *dst = keys;
  for (int q = 0; q < passes; ++q) {
    hipLaunchKernelGGL(vox_hist, tiles, dim3(256), 0, st, src, hist, d->N, T, q);
    hipLaunchKernelGGL(vox_offsets, dim3(d->B), dim3(256), 0, st, hist, T);
    hipLaunchKernelGGL(vox_scatter, tiles, dim3(256), 0, st, src, dst, hist, d->N, T, q);
    unsigned long long* t = src; src = dst; dst = t;
  }
  keys = src;                                                    // the buffer the last pass wrote
  if (hipMemsetAsync(head_flag, 0, (size_t)n * sizeof(int), st) != hipSuccess) {
    bevf_set_error("voxelize: memset failed");
    return BEVF_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(vox_heads, dim3(grid), dim3(256), 0, st, keys, head_flag, n, d->N);
  hipLaunchKernelGGL(vox_flagcount, tiles, dim3(256), 0, st, head_flag, tcount, d->N, T);
  hipLaunchKernelGGL(vox_vid, tiles, dim3(256), 0, st, head_flag, tcount, vid, d->num_voxels, d->N, T, d->max_voxels);
  hipLaunchKernelGGL(vox_fill, dim3((unsigned)((n * 64 + 255) / 256)), dim3(256), 0, st, a, keys, head_flag, vid,
                     d->voxel_features, (long long*)d->voxel_coords, d->num_points);
  return bevf_check_launch("bevf_voxelize_f32");
}

extern "C" int bevf_scatter_voxels_f32(const float* features, const int64_t* coords, const int32_t* num_voxels, int32_t* owner,
                                       float* out, int B, int Nv, int C, int D, int H, int W, void* stream) {
  BEVF_REQUIRE(features && coords && owner && out, "scatter_voxels: null pointer");
  BEVF_REQUIRE(B > 0 && Nv > 0 && C > 0 && D > 0 && H > 0 && W > 0, "scatter_voxels: empty shape");
  const long long cells = (long long)D * H * W;
  BEVF_REQUIRE((long long)B * cells < (1ll << 31) && (long long)B * Nv < (1ll << 31), "scatter_voxels: grid too large");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(owner, 0xFF, (size_t)B * cells * sizeof(int), st) != hipSuccess) {       // -1: no voxel names the cell
    bevf_set_error("scatter_voxels: memset failed");
    return BEVF_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(scatter_owner, dim3((unsigned)(((long long)B * Nv + 255) / 256)), dim3(256), 0, st,
                     reinterpret_cast<const long long*>(coords), num_voxels, owner, B, Nv, D, H, W);
  hipLaunchKernelGGL(scatter_write, dim3((unsigned)((B * cells + 255) / 256)), dim3(256), 0, st, features, owner, out, B, Nv, C,
                     cells);
  return bevf_check_launch("bevf_scatter_voxels_f32");
}
