// CenterNet training targets and losses on device.
//   targets: ref src/centernet_target.py:118-324   losses: ref src/centernet_target.py:476-622
//   keep mask (_nms): ref src/centernet_target.py:416-421
//
// Bit-exactness plan for the integer pins (ind, mask, reg_mask) and the radius: the reference runs this
// arithmetic on numpy float32 scalars (boxes are float32; python-float constants are converted to float32
// by numpy>=2 promotion), so every step is a correctly rounded fp32 op in the reference's own order.
// The __f*_rn intrinsics below keep that order and forbid FMA contraction.  The gaussian itself is
// float64 in the reference (np.ogrid / np.exp) and is rounded to fp32 by the max-merge.
#include "common.h"

namespace {

__device__ __forceinline__ float mulf(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float addf(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float subf(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float divf(float a, float b) { return __fdiv_rn(a, b); }

// ref :128-150 with height = box_l, width = box_w (ref :272), operation for operation
__device__ float gaussian_radius_f32(float height, float width, float c_1m, float c_1p, float c_m1, float c_neg2mo,
                                     float c_4a3) {
  const float hw = addf(height, width);
  const float b1 = hw;
  const float c1 = divf(mulf(mulf(width, height), c_1m), c_1p);           // w*h*(1-mo)/(1+mo)
  const float sq1 = __fsqrt_rn(subf(mulf(b1, b1), mulf(4.f, c1)));       // 4*a1 = 4
  const float r1 = divf(addf(b1, sq1), 2.f);
  const float b2 = mulf(2.f, hw);
  const float c2 = mulf(mulf(c_1m, width), height);                       // (1-mo)*w*h
  const float sq2 = __fsqrt_rn(subf(mulf(b2, b2), mulf(16.f, c2)));       // 4*a2 = 16
  const float r2 = divf(addf(b2, sq2), 2.f);
  const float b3 = mulf(c_neg2mo, hw);                                    // (-2*mo)*(h+w)
  const float c3 = mulf(mulf(c_m1, width), height);                       // (mo-1)*w*h
  const float sq3 = __fsqrt_rn(subf(mulf(b3, b3), mulf(c_4a3, c3)));      // (4*a3)*c3, a3 = 4*mo
  const float r3 = divf(addf(b3, sq3), 2.f);
  return fminf(r1, fminf(r2, r3));
}

struct TgtArgs {
  const float* boxes;   // [B][nmax][9]
  const int* labels;    // [B][nmax]
  const int* has_vel;   // [B]
  float* heatmap; float* offset; float* size; float* rot; float* vel;
  unsigned char* mask; long long* ind; unsigned char* reg_mask;
  float* t_offset; float* t_size; float* t_rot; float* t_vel;
  int* owner;           // [B][H*W] scratch, zeroed by the caller
  int B, nmax, H, W, C, max_objects, min_radius;
  float x_min, y_min, vx, vy;                      // float32(python floats)
  float c_1m, c_1p, c_m1, c_neg2mo, c_4a3;          // float32((1-mo)), float32(1+mo), float32(mo-1), float32(-2*mo), float32(4*4*mo)
};

// one workgroup per frame
__global__ __launch_bounds__(256) void centernet_targets(const TgtArgs a) {
  extern __shared__ int sm[];          // per object: cx, cy, radius (or -1), cls
  int* s_cx = sm; int* s_cy = sm + a.nmax; int* s_r = sm + 2 * a.nmax; int* s_cls = sm + 3 * a.nmax;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int HW = a.H * a.W;
  const float* bx = a.boxes + (size_t)b * a.nmax * 9;
  const float Wf = (float)a.W, Hf = (float)a.H;

  for (int k = tid; k < a.nmax; k += 256) {
    int r = -1, cx = 0, cy = 0;
    const int cls = a.labels[b * a.nmax + k];
    if (cls >= 0 && cls < a.C) {
      const float* p = bx + k * 9;
      const float px = divf(subf(p[0], a.x_min), a.vx), py = divf(subf(p[1], a.y_min), a.vy);
      if (!(px < 0.f || px >= Wf || py < 0.f || py >= Hf)) {
        cx = (int)px; cy = (int)py;
        if (cx >= 0 && cx < a.W && cy >= 0 && cy < a.H) {
          const float box_w = divf(p[3], a.vx), box_l = divf(p[4], a.vy);
          const float rad = gaussian_radius_f32(box_l, box_w, a.c_1m, a.c_1p, a.c_m1, a.c_neg2mo, a.c_4a3);
          r = (int)rad;
          if (r < a.min_radius) r = a.min_radius;
          // per-object records (ref :285-309)
          const size_t o = (size_t)b * a.max_objects + k;
          a.ind[o] = (long long)cy * a.W + cx;
          a.mask[o] = 1; a.reg_mask[o] = 1;
          a.t_offset[o * 2] = px - (float)cx; a.t_offset[o * 2 + 1] = py - (float)cy;
          a.t_size[o * 3] = p[3]; a.t_size[o * 3 + 1] = p[4]; a.t_size[o * 3 + 2] = p[5];
          a.t_rot[o * 2] = sinf(p[6]); a.t_rot[o * 2 + 1] = cosf(p[6]);
          if (a.has_vel[b]) { a.t_vel[o * 2] = p[7]; a.t_vel[o * 2 + 1] = p[8]; }
          atomicMax(&a.owner[(size_t)b * HW + cy * a.W + cx], k + 1);     // dense maps: the last object wins
        }
      }
    }
    s_cx[k] = cx; s_cy[k] = cy; s_r[k] = r; s_cls[k] = cls;
  }
  __syncthreads();

  // dense regression maps (ref :292-309): written by the highest-index object of each cell
  for (int k = tid; k < a.nmax; k += 256) {
    if (s_r[k] < 0) continue;
    const int cell = s_cy[k] * a.W + s_cx[k];
    if (a.owner[(size_t)b * HW + cell] != k + 1) continue;
    const float* p = bx + k * 9;
    const float px = divf(subf(p[0], a.x_min), a.vx), py = divf(subf(p[1], a.y_min), a.vy);
    float* o2 = a.offset + (size_t)b * 2 * HW; float* s3 = a.size + (size_t)b * 3 * HW;
    float* r2 = a.rot + (size_t)b * 2 * HW;
    o2[cell] = px - (float)s_cx[k]; o2[HW + cell] = py - (float)s_cy[k];
    s3[cell] = p[3]; s3[HW + cell] = p[4]; s3[2 * HW + cell] = p[5];
    r2[cell] = sinf(p[6]); r2[HW + cell] = cosf(p[6]);
  }
  // velocity map: the reference writes it only for boxes that carry vx,vy, so an earlier 9-wide object is not
  // overwritten by a later 7-wide one; inside one frame all boxes have the same width, so owner == last is right
  if (a.has_vel[b]) {
    for (int k = tid; k < a.nmax; k += 256) {
      if (s_r[k] < 0) continue;
      const int cell = s_cy[k] * a.W + s_cx[k];
      if (a.owner[(size_t)b * HW + cell] != k + 1) continue;
      const float* p = bx + k * 9;
      float* v2 = a.vel + (size_t)b * 2 * HW;
      v2[cell] = p[7]; v2[HW + cell] = p[8];
    }
  }

  // heatmap: element-wise max of clipped gaussians (ref :118-125, :152-168); float64 like numpy, then fp32
  for (int k = 0; k < a.nmax; ++k) {
    const int r = s_r[k];
    if (r < 0) continue;
    const int cx = s_cx[k], cy = s_cy[k], d = 2 * r + 1;
    const double sigma = (double)d / 6.0;
    const double den = (2.0 * sigma) * sigma;
    unsigned* plane = reinterpret_cast<unsigned*>(a.heatmap + ((size_t)b * a.C + s_cls[k]) * HW);
    for (int i = tid; i < d * d; i += 256) {
      const int dy = i / d - r, dx = i - (i / d) * d - r;
      const int x = cx + dx, y = cy + dy;
      if ((unsigned)x >= (unsigned)a.W || (unsigned)y >= (unsigned)a.H) continue;
      double g = exp(-((double)(dx * dx) + (double)(dy * dy)) / den);
      if (g < 2.220446049250313e-16) g = 0.0;            // h[h < eps * h.max()] = 0, h.max() == 1 at the centre
      atomicMax(&plane[y * a.W + x], __float_as_uint((float)g));
    }
  }
}

// ---- keep mask: out = (max3x3(heat) == heat) ? heat : 0 --------------------------------------------------
__global__ __launch_bounds__(256) void nms_keep(const float* __restrict__ heat, float* __restrict__ out, int H, int W,
                                                 long long total) {
  const int HW = H * W;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long plane = i / HW;
    const int p = (int)(i - plane * HW), y = p / W, x = p - y * W;
    const float* hp = heat + plane * HW;
    const float v = hp[p];
    float m = v;
    for (int dy = -1; dy <= 1; ++dy) {
      const int yy = y + dy;
      if ((unsigned)yy >= (unsigned)H) continue;
      for (int dx = -1; dx <= 1; ++dx) {
        const int xx = x + dx;
        if ((unsigned)xx < (unsigned)W) m = fmaxf(m, hp[yy * W + xx]);
      }
    }
    out[i] = (m == v) ? v : v * 0.f;
  }
}

// ---- losses ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

constexpr int kLossGrid = 512;

// focal partial sums: part[g] = {pos_loss, neg_loss, num_pos}   (ref :563-575, sigmoid applied AGAIN as the ref does)
__global__ __launch_bounds__(256) void focal_partials(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                       float* __restrict__ part, long long n) {
  __shared__ float red[4];
  float pos = 0.f, neg = 0.f, cnt = 0.f;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float p = 1.f / (1.f + expf(-pred[i]));
    p = fminf(fmaxf(p, 1e-4f), 1.f - 1e-4f);
    const float t = tgt[i];
    if (t == 1.f) {
      const float q = 1.f - p;
      pos += logf(p) * (q * q);
      cnt += 1.f;
    } else if (t < 1.f) {
      const float q = 1.f - t, q2 = q * q;
      neg += logf(1.f - p) * (p * p) * (q2 * q2);
    }
  }
  const float a = block_sum(pos, red), b = block_sum(neg, red), c = block_sum(cnt, red);
  if (threadIdx.x == 0) { part[blockIdx.x * 3] = a; part[blockIdx.x * 3 + 1] = b; part[blockIdx.x * 3 + 2] = c; }
}

struct LossArgs {
  const float* part; int nparts;
  const float* pred[4];      // offset,size,rot,vel (B,C,H,W)
  const float* tgt[4];       // target_* (B,K,C)
  int c[4];
  const long long* ind; const unsigned char* reg_mask;
  int B, K, HW;
  float w[5];
  float* out;                // total, heatmap, offset, size, rot, vel
};

__global__ __launch_bounds__(256) void loss_final(const LossArgs a) {
  __shared__ float red[4];
  __shared__ float res[5];
  // focal: fixed-order sum of the partials in double
  if (threadIdx.x == 0) {
    double pos = 0, neg = 0, cnt = 0;
    for (int g = 0; g < a.nparts; ++g) { pos += a.part[g * 3]; neg += a.part[g * 3 + 1]; cnt += a.part[g * 3 + 2]; }
    res[0] = (float)(cnt == 0 ? -neg : -(pos + neg) / cnt);             // ref :577-580
  }
  float msum = 0.f;
  for (int i = threadIdx.x; i < a.B * a.K; i += 256) msum += a.reg_mask[i] ? 1.f : 0.f;
  msum = block_sum(msum, red);
  for (int q = 0; q < 4; ++q) {                                         // ref :584-622
    const int C = a.c[q];
    float s = 0.f;
    for (int i = threadIdx.x; i < a.B * a.K * C; i += 256) {
      const int ch = i % C, bk = i / C, b = bk / a.K;
      if (!a.reg_mask[bk]) continue;
      const long long cell = a.ind[bk];
      const float pv = a.pred[q][((size_t)b * C + ch) * a.HW + cell];
      s += fabsf(pv - a.tgt[q][i]);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) res[1 + q] = s / (msum * (float)C + 1e-4f);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float total = 0.f;
    for (int q = 0; q < 5; ++q) { total += a.w[q] * res[q]; a.out[1 + q] = res[q]; }
    a.out[0] = total;
  }
}

}  // namespace

extern "C" int bevf_centernet_targets_f32(const bevf_targets_desc* d, void* stream) {
  BEVF_REQUIRE(d && d->boxes && d->labels && d->has_vel && d->owner_scratch, "targets: null input pointer");
  BEVF_REQUIRE(d->heatmap && d->offset && d->size && d->rot && d->vel && d->mask && d->ind && d->reg_mask &&
                   d->target_offset && d->target_size && d->target_rot && d->target_vel, "targets: null output pointer");
  BEVF_REQUIRE(d->B > 0 && d->nmax > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->max_objects >= d->nmax,
               "targets: bad shape (nmax=%d max_objects=%d)", d->nmax, d->max_objects);
  BEVF_REQUIRE((size_t)d->nmax * 4 * sizeof(int) <= 64 * 1024, "targets: more than 4096 objects per frame");
  TgtArgs a;
  a.boxes = d->boxes; a.labels = d->labels; a.has_vel = d->has_vel;
  a.heatmap = d->heatmap; a.offset = d->offset; a.size = d->size; a.rot = d->rot; a.vel = d->vel;
  a.mask = d->mask; a.ind = (long long*)d->ind; a.reg_mask = d->reg_mask;
  a.t_offset = d->target_offset; a.t_size = d->target_size; a.t_rot = d->target_rot; a.t_vel = d->target_vel;
  a.owner = d->owner_scratch;
  a.B = d->B; a.nmax = d->nmax; a.H = d->H; a.W = d->W; a.C = d->C; a.max_objects = d->max_objects;
  a.min_radius = d->min_radius;
  // python-float constants exactly as the reference forms them, then rounded to float32 once (numpy>=2 promotion)
  const double vx = ((double)d->pc_range[3] - (double)d->pc_range[0]) / d->W;
  const double vy = ((double)d->pc_range[4] - (double)d->pc_range[1]) / d->H;
  const double mo = d->gaussian_overlap;
  a.x_min = (float)d->pc_range[0]; a.y_min = (float)d->pc_range[1]; a.vx = (float)vx; a.vy = (float)vy;
  a.c_1m = (float)(1 - mo); a.c_1p = (float)(1 + mo); a.c_m1 = (float)(mo - 1); a.c_neg2mo = (float)(-2 * mo);
  a.c_4a3 = (float)(4 * (4 * mo));
  hipLaunchKernelGGL(centernet_targets, dim3(d->B), dim3(256), (size_t)d->nmax * 4 * sizeof(int),
                     static_cast<hipStream_t>(stream), a);
  return bevf_check_launch("bevf_centernet_targets_f32");
}

extern "C" int bevf_nms_keep_f32(const float* heat, float* out, int planes, int H, int W, void* stream) {
  BEVF_REQUIRE(heat && out && planes > 0 && H > 0 && W > 0, "nms_keep: bad arguments");
  const long long total = (long long)planes * H * W;
  const unsigned grid = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(nms_keep, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), heat, out, H, W, total);
  return bevf_check_launch("bevf_nms_keep_f32");
}

extern "C" size_t bevf_centernet_loss_work_floats(void) { return (size_t)kLossGrid * 3; }

extern "C" int bevf_centernet_loss_f32(const bevf_loss_desc* d, void* stream) {
  BEVF_REQUIRE(d && d->pred_heatmap && d->tgt_heatmap && d->ind && d->reg_mask && d->work && d->out, "loss: null pointer");
  BEVF_REQUIRE(d->B > 0 && d->C > 0 && d->H > 0 && d->W > 0 && d->K > 0, "loss: empty shape");
  const long long n = (long long)d->B * d->C * d->H * d->W;
  const int grid = (int)((n + 255) / 256 > kLossGrid ? kLossGrid : (n + 255) / 256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(focal_partials, dim3(grid), dim3(256), 0, st, d->pred_heatmap, d->tgt_heatmap, d->work, n);
  LossArgs a;
  a.part = d->work; a.nparts = grid;
  const int cs[4] = {2, 3, 2, 2};
  for (int q = 0; q < 4; ++q) {
    BEVF_REQUIRE(d->pred_reg[q] && d->tgt_reg[q], "loss: regression branch %d missing", q);
    a.pred[q] = d->pred_reg[q]; a.tgt[q] = d->tgt_reg[q]; a.c[q] = cs[q];
  }
  a.ind = (const long long*)d->ind; a.reg_mask = d->reg_mask; a.B = d->B; a.K = d->K; a.HW = d->H * d->W;
  for (int q = 0; q < 5; ++q) a.w[q] = d->weights[q];
  a.out = d->out;
  hipLaunchKernelGGL(loss_final, dim3(1), dim3(256), 0, st, a);
  return bevf_check_launch("bevf_centernet_loss_f32");
}
