// Train-mode BatchNorm on NHWC rows ([M][C], channel stride cs): batch statistics, apply (+residual, +ReLU),
// and the backward pair (per-channel reductions, then the element-wise input gradient).
//   forward : torch.nn.BatchNorm{1,2}d in training mode, as every conv/bn pair of ref src/encoders.py and
//             src/fusion.py runs under model.train() (ref src/train_detect.py:395-434)
// Reductions are two-stage and order-fixed (deterministic): fp32 partial sums per workgroup, merged in
// double.  Variance uses sums shifted by the first row (no catastrophic cancellation when |mean| >> std).
#include "common.h"

namespace {

constexpr int kStatGrid = 1024;

// thread (cq, rl): channel quad cq = tid % C4, row lane rl = tid / C4; C4 = C/4 <= 256 and a power of two * ...
struct RowMap {
  int c4, lanes;
  __device__ RowMap(int C) : c4(C >> 2), lanes(256 / (C >> 2) > 0 ? 256 / (C >> 2) : 1) {}
};

// partial[g][c] = {sum(x-s), sum((x-s)^2)} ; shift s[c] = x[0][c]
__global__ __launch_bounds__(256) void stats_partials(const float* __restrict__ x, float* __restrict__ part, int M, int C,
                                                       int cs) {
  extern __shared__ float red[];                      // [256][8]
  const int c4 = C >> 2;
  const int lanes = c4 >= 256 ? 1 : 256 / c4;
  float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  for (int cq = threadIdx.x % (c4 < 256 ? c4 : 256); cq < c4; cq += 256) {   // C > 1024: loop over quads
    const int rl = c4 >= 256 ? 0 : threadIdx.x / c4;
    if (rl >= lanes) break;
    const f32x4 sh = *reinterpret_cast<const f32x4*>(x + cq * 4);
    float a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
    const long long step = (long long)gridDim.x * lanes;
    long long m = (long long)blockIdx.x * lanes + rl;
    for (; m + 3 * step < M; m += 4 * step) {           // four independent 16-byte loads in flight per thread
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(x + (size_t)(m + u * step) * cs + cq * 4);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[u][j] - sh[j]; a1[j] += d; a2[j] = fmaf(d, d, a2[j]); }
    }
    for (; m < M; m += step) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)m * cs + cq * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = v[j] - sh[j]; a1[j] += d; a2[j] = fmaf(d, d, a2[j]); }
    }
    if (c4 >= 256) {                                  // one row lane: write directly
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        part[((size_t)blockIdx.x * C + cq * 4 + j) * 2] = a1[j];
        part[((size_t)blockIdx.x * C + cq * 4 + j) * 2 + 1] = a2[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) { s1[j] = a1[j]; s2[j] = a2[j]; }
    }
  }
  if (c4 < 256) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[threadIdx.x * 8 + j] = s1[j]; red[threadIdx.x * 8 + 4 + j] = s2[j]; }
    __syncthreads();
    if ((int)threadIdx.x < c4) {                      // fixed-order merge over the row lanes
      float t1[4] = {0, 0, 0, 0}, t2[4] = {0, 0, 0, 0};
      for (int rl = 0; rl < lanes; ++rl)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          t1[j] += red[(rl * c4 + threadIdx.x) * 8 + j];
          t2[j] += red[(rl * c4 + threadIdx.x) * 8 + 4 + j];
        }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        part[((size_t)blockIdx.x * C + threadIdx.x * 4 + j) * 2] = t1[j];
        part[((size_t)blockIdx.x * C + threadIdx.x * 4 + j) * 2 + 1] = t2[j];
      }
    }
  }
}

// merge of the G partial rows of one channel by one 256-thread workgroup: thread l takes g = l, l+256, ... (fixed order), a
// fixed xor tree merges the 64 lane totals of each wave, and the four wave totals are added in wave order through LDS.
// Every thread of the workgroup must call it; the result is valid in thread 0.
__device__ __forceinline__ void merge_partials(const float* __restrict__ part, int C, int G, int c, double& s1, double& s2) {
  __shared__ double wsum[4][2];
  s1 = 0; s2 = 0;
  if (c < C)
    for (int g = threadIdx.x; g < G; g += 256) { s1 += part[((size_t)g * C + c) * 2]; s2 += part[((size_t)g * C + c) * 2 + 1]; }
#pragma unroll
  for (int sft = 1; sft < 64; sft <<= 1) {
    s1 += __shfl_xor(s1, sft);
    s2 += __shfl_xor(s2, sft);
  }
  if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6][0] = s1; wsum[threadIdx.x >> 6][1] = s2; }
  __syncthreads();
  s1 = ((wsum[0][0] + wsum[1][0]) + wsum[2][0]) + wsum[3][0];
  s2 = ((wsum[0][1] + wsum[1][1]) + wsum[2][1]) + wsum[3][1];
}

// mean, biased var, invstd from the partials (double, fixed order)
__global__ __launch_bounds__(256) void stats_finalize(const float* __restrict__ x, const float* __restrict__ part,
                                                       float* __restrict__ mean, float* __restrict__ var,
                                                       float* __restrict__ invstd, int M, int C, int G, float eps) {
  const int c = blockIdx.x;
  double s1, s2;
  merge_partials(part, C, G, c, s1, s2);
  if (c >= C || threadIdx.x != 0) return;
  const double sh = x[c], d = s1 / M;
  const double v = s2 / M - d * d;
  mean[c] = (float)(sh + d);
  var[c] = (float)(v > 0 ? v : 0);
  invstd[c] = (float)(1.0 / sqrt((v > 0 ? v : 0) + (double)eps));
}

// y = act((x - mean) * invstd * gamma + beta (+ res)).  Thread (cq, rl) keeps channel quad cq for its whole life, so
// the per-channel scale/shift live in registers and the row loop is pure 16-byte streaming with 4 rows in flight.
__global__ __launch_bounds__(256) void bn_apply(const float* __restrict__ x, const float* __restrict__ mean,
                                                 const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, const float* __restrict__ res,
                                                 float* __restrict__ y, long long M, int C, int cs, int relu) {
  const int c4 = C >> 2;
  const int lanes = c4 >= 256 ? 1 : 256 / c4;
  const int rl = c4 >= 256 ? 0 : threadIdx.x / c4;
  if (rl >= lanes) return;
  const long long step = (long long)gridDim.x * lanes;
  for (int cq = threadIdx.x % (c4 < 256 ? c4 : 256); cq < c4; cq += 256) {
    const int c = cq * 4;
    float a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[j] = (gamma ? gamma[c + j] : 1.f) * invstd[c + j];
      b[j] = (beta ? beta[c + j] : 0.f) - mean[c + j] * a[j];
    }
    long long m = (long long)blockIdx.x * lanes + rl;
    for (; m + 3 * step < M; m += 4 * step) {
      f32x4 v[4], r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v[u] = *reinterpret_cast<const f32x4*>(x + (size_t)(m + u * step) * cs + c);
        if (res) r[u] = *reinterpret_cast<const f32x4*>(res + (size_t)(m + u * step) * C + c);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float t = fmaf(v[u][j], a[j], b[j]);
          if (res) t += r[u][j];
          o[j] = relu ? fmaxf(t, 0.f) : t;
        }
        *reinterpret_cast<f32x4*>(y + (size_t)(m + u * step) * C + c) = o;
      }
    }
    for (; m < M; m += step) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)m * cs + c);
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float t = fmaf(v[j], a[j], b[j]);
        if (res) t += res[(size_t)m * C + c + j];
        o[j] = relu ? fmaxf(t, 0.f) : t;
      }
      *reinterpret_cast<f32x4*>(y + (size_t)m * C + c) = o;
    }
  }
}

// backward stage 1: dy <- dy * (y > 0) when relu; partial[g][c] = {sum dy, sum dy*xhat}.
// relu == 2: the forward output is not read; the mask is recomputed from the raw input with the very operations of
// bn_apply (t = fma(x, gamma*invstd, beta - mean*gamma*invstd) > 0) -- valid for layers without a residual input.
__global__ __launch_bounds__(256) void bn_bwd_partials(float* __restrict__ dy, const float* __restrict__ y,
                                                        const float* __restrict__ x, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ part,
                                                        int M, int C, int cs, int relu) {
  extern __shared__ float red[];
  const int c4 = C >> 2;
  const int lanes = c4 >= 256 ? 1 : 256 / c4;
  float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  for (int cq = threadIdx.x % (c4 < 256 ? c4 : 256); cq < c4; cq += 256) {
    const int rl = c4 >= 256 ? 0 : threadIdx.x / c4;
    if (rl >= lanes) break;
    float mu[4], is[4], fa[4], fb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mu[j] = mean ? mean[cq * 4 + j] : 0.f;
      is[j] = invstd ? invstd[cq * 4 + j] : 1.f;
      fa[j] = (gamma ? gamma[cq * 4 + j] : 1.f) * is[j];
      fb[j] = (beta ? beta[cq * 4 + j] : 0.f) - mu[j] * fa[j];
    }
    float a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
    const long long step = (long long)gridDim.x * lanes;
    long long m = (long long)blockIdx.x * lanes + rl;
    for (; m + 3 * step < M; m += 4 * step) {           // 4 rows x up to 3 streams of 16-byte loads in flight
      f32x4 g[4], yy[4], xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t r = (size_t)(m + u * step);
        g[u] = *reinterpret_cast<const f32x4*>(dy + r * C + cq * 4);
        if (relu == 1) yy[u] = *reinterpret_cast<const f32x4*>(y + r * C + cq * 4);
        if (x) xv[u] = *reinterpret_cast<const f32x4*>(x + r * cs + cq * 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (relu) {
          if (relu >= 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) yy[u][j] = fmaf(xv[u][j], fa[j], fb[j]);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) g[u][j] = yy[u][j] > 0.f ? g[u][j] : 0.f;
          if (relu != 3) *reinterpret_cast<f32x4*>(dy + (size_t)(m + u * step) * C + cq * 4) = g[u];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a1[j] += g[u][j];
          if (x) a2[j] = fmaf(g[u][j], (xv[u][j] - mu[j]) * is[j], a2[j]);
        }
      }
    }
    for (; m < M; m += step) {
      f32x4 g = *reinterpret_cast<const f32x4*>(dy + (size_t)m * C + cq * 4);
      if (relu) {
        f32x4 yy;
        if (relu == 1) {
          yy = *reinterpret_cast<const f32x4*>(y + (size_t)m * C + cq * 4);
        } else {
          const f32x4 xr = *reinterpret_cast<const f32x4*>(x + (size_t)m * cs + cq * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) yy[j] = fmaf(xr[j], fa[j], fb[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = yy[j] > 0.f ? g[j] : 0.f;
        if (relu != 3) *reinterpret_cast<f32x4*>(dy + (size_t)m * C + cq * 4) = g;
      }
      if (x) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)m * cs + cq * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { a1[j] += g[j]; a2[j] = fmaf(g[j], (xv[j] - mu[j]) * is[j], a2[j]); }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) a1[j] += g[j];
      }
    }
    if (c4 >= 256) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        part[((size_t)blockIdx.x * C + cq * 4 + j) * 2] = a1[j];
        part[((size_t)blockIdx.x * C + cq * 4 + j) * 2 + 1] = a2[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) { s1[j] = a1[j]; s2[j] = a2[j]; }
    }
  }
  if (c4 < 256) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[threadIdx.x * 8 + j] = s1[j]; red[threadIdx.x * 8 + 4 + j] = s2[j]; }
    __syncthreads();
    if ((int)threadIdx.x < c4) {
      float t1[4] = {0, 0, 0, 0}, t2[4] = {0, 0, 0, 0};
      for (int rl = 0; rl < lanes; ++rl)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          t1[j] += red[(rl * c4 + threadIdx.x) * 8 + j];
          t2[j] += red[(rl * c4 + threadIdx.x) * 8 + 4 + j];
        }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        part[((size_t)blockIdx.x * C + threadIdx.x * 4 + j) * 2] = t1[j];
        part[((size_t)blockIdx.x * C + threadIdx.x * 4 + j) * 2 + 1] = t2[j];
      }
    }
  }
}

__global__ __launch_bounds__(256) void sums_finalize(const float* __restrict__ part, float* __restrict__ s_dy,
                                                      float* __restrict__ s_dyx, int C, int G) {
  const int c = blockIdx.x;
  double s1, s2;
  merge_partials(part, C, G, c, s1, s2);
  if (c >= C || threadIdx.x != 0) return;
  s_dy[c] = (float)s1;
  if (s_dyx) s_dyx[c] = (float)s2;
}

// backward stage 2: dx = gamma*invstd * (dy - sum_dy/M - xhat * sum_dyx/M) = k1*dy + k2*x + k3 per channel
// (contraction switched off: the kernels that share this formula -- bn_bwd_apply with and without re-masking, pool_bn_bwd_apply
//  -- must agree bit for bit, which contraction decisions that depend on the surrounding code would not guarantee)
__device__ __forceinline__ float bn_dx(float gi, float g, float sd, float x, float mu, float is, float sx) {
#pragma clang fp contract(off)                         // (HIP's __fmul_rn / __fsub_rn are plain operators: they do not stop contraction)
  const float t = ((x - mu) * is) * sx;
  const float u = (g - sd) - t;
  return gi * u;
}
// (remask: dy arrives WITHOUT the ReLU mask -- stage 1 ran in mode 3 and did not write it back -- and is masked here with the same
//  recomputed fma(x, gamma*invstd, beta - mean*gamma*invstd) > 0)
__global__ __launch_bounds__(256) void bn_bwd_apply(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ s_dy,
                                                     const float* __restrict__ s_dyx, float* __restrict__ dx,
                                                     long long M, int C, int cs, const float* __restrict__ beta = nullptr,
                                                     int remask = 0, int frozen = 0) {
  const int c4 = C >> 2;
  const int lanes = c4 >= 256 ? 1 : 256 / c4;
  const int rl = c4 >= 256 ? 0 : threadIdx.x / c4;
  if (rl >= lanes) return;
  const long long step = (long long)gridDim.x * lanes;
  // frozen: the layer normalised with FIXED statistics (eval-mode BatchNorm inside a training module): mean / invstd do not depend on
  // x, so the two mean-subtraction terms vanish and dx = gamma * invstd * dy
  const float invM = frozen ? 0.f : 1.f / (float)M;
  for (int cq = threadIdx.x % (c4 < 256 ? c4 : 256); cq < c4; cq += 256) {
    const int c = cq * 4;
    float gi[4], is[4], mu[4], sd[4], sx[4], fb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      is[j] = invstd[c + j];
      mu[j] = mean[c + j];
      gi[j] = (gamma ? gamma[c + j] : 1.f) * is[j];
      fb[j] = (beta ? beta[c + j] : 0.f) - mu[j] * gi[j];
      sd[j] = s_dy[c + j] * invM;
      sx[j] = s_dyx[c + j] * invM;
    }
    long long m = (long long)blockIdx.x * lanes + rl;
    for (; m + 3 * step < M; m += 4 * step) {
      f32x4 g[4], xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        g[u] = dy ? *reinterpret_cast<const f32x4*>(dy + (size_t)(m + u * step) * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        xv[u] = *reinterpret_cast<const f32x4*>(x + (size_t)(m + u * step) * cs + c);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f32x4 o;
        if (remask) {
#pragma unroll
          for (int j = 0; j < 4; ++j) g[u][j] = fmaf(xv[u][j], gi[j], fb[j]) > 0.f ? g[u][j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = bn_dx(gi[j], g[u][j], sd[j], xv[u][j], mu[j], is[j], sx[j]);
        *reinterpret_cast<f32x4*>(dx + (size_t)(m + u * step) * cs + c) = o;
      }
    }
    for (; m < M; m += step) {
      f32x4 g = dy ? *reinterpret_cast<const f32x4*>(dy + (size_t)m * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)m * cs + c);
      f32x4 o;
      if (remask) {
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = fmaf(xv[j], gi[j], fb[j]) > 0.f ? g[j] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = bn_dx(gi[j], g[j], sd[j], xv[j], mu[j], is[j], sx[j]);
      *reinterpret_cast<f32x4*>(dx + (size_t)m * cs + c) = o;
    }
  }
}

// running statistics, as torch.nn.BatchNorm updates them in training mode (one launch instead of five tiny ones)
__global__ __launch_bounds__(256) void bn_update_running(const float* __restrict__ mean, const float* __restrict__ var,
                                                          float* __restrict__ rmean, float* __restrict__ rvar,
                                                          long long* __restrict__ nbt, int C, float keep, float mom, float mom_unbiased) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c == 0 && nbt) *nbt += 1;
  if (c >= C) return;
  rmean[c] = fmaf(mom, mean[c], rmean[c] * keep);
  rvar[c] = fmaf(mom_unbiased, var[c], rvar[c] * keep);
}

// ---- BatchNorm backward fed by a max over rows (PointNet's last layer) ---------------------------------------------------
// The gradient of y = relu(bn(x)) arriving from g[b][c] = max_n y[b][n][c] is non-zero in ONE row per (frame, channel).
// Instead of scattering it into a dense [M][C] tensor and reducing that again, the per-channel sums are gathered from
// the B*C non-zeros, the dense part of dx is written without reading any dy, and the B*C entries are added afterwards.
__global__ __launch_bounds__(256) void gmax_bn_sums(const float* __restrict__ dg, const float* __restrict__ gmax,
                                                     const int* __restrict__ idx, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                     float* __restrict__ dgm, float* __restrict__ s_dy, float* __restrict__ s_dyx,
                                                     int B, int P, int C, int cs) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float mu = mean[c], is = invstd[c];
  double a1 = 0, a2 = 0;
  for (int b = 0; b < B; ++b) {
    const float g = gmax[(size_t)b * C + c] > 0.f ? dg[(size_t)b * C + c] : 0.f;      // ReLU: relu'(0) = 0, as torch
    dgm[(size_t)b * C + c] = g;
    const float xv = x[((size_t)b * P + idx[(size_t)b * C + c]) * cs + c];
    a1 += g;
    a2 += (double)g * ((xv - mu) * is);
  }
  s_dy[c] = (float)a1;
  s_dyx[c] = (float)a2;
}

__global__ __launch_bounds__(256) void gmax_bn_scatter(const float* __restrict__ dgm, const int* __restrict__ idx,
                                                        const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                        float* __restrict__ dx, int P, int C, int cs, long long total) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= total) return;
  const long long b = i / C;
  const int c = (int)(i - b * C);
  dx[((size_t)b * P + idx[i]) * cs + c] += (gamma ? gamma[c] : 1.f) * invstd[c] * dgm[i];
}

// ---- BatchNorm(+ReLU) backward fed by a 3x3/s2 max-pool backward (the ResNet stem in training) -----------------------------
// dY of the dense stem map [N][H][W][C] is never materialised: each pass gathers it from the pooled gradient and the saved
// argmax codes (the <= 4 windows that contain a pixel, in the order of train_misc.hip: maxpool_bwd, so the values -- and with
// the loop structure of bn_bwd_partials / bn_bwd_apply the sums and dx -- are bit-identical to the unfused chain).  The ReLU
// mask is recomputed from the raw input with the forward's fma.  Traffic at 48 images of 448x800: 9.2 GB -> 4 GB.
__device__ __forceinline__ f32x4 pool_gather(const float* __restrict__ dpool, const unsigned char* __restrict__ idx, long long m,
                                             int c, int H, int W, int C, int Ho, int Wo) {
  const int iw = (int)(m % W);
  const long long t = m / W;
  const int ih = (int)(t % H), n = (int)(t / H);
  f32x4 g = {0.f, 0.f, 0.f, 0.f};
  for (int oh = ih / 2; oh <= (ih + 1) / 2; ++oh) {                            // 2*oh-1 <= ih <= 2*oh+1
    if (oh >= Ho) continue;
    const int dh = ih - (2 * oh - 1);
    for (int ow = iw / 2; ow <= (iw + 1) / 2; ++ow) {
      if (ow >= Wo) continue;
      const unsigned code = (unsigned)(dh * 3 + (iw - (2 * ow - 1)));
      const size_t o = ((size_t)(n * Ho + oh) * Wo + ow) * C + c;
      const unsigned id4 = *reinterpret_cast<const unsigned*>(idx + o);
      const f32x4 d = *reinterpret_cast<const f32x4*>(dpool + o);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (((id4 >> (8 * j)) & 0xff) == code) g[j] += d[j];
    }
  }
  return g;
}

__global__ __launch_bounds__(256) void pool_bn_bwd_partials(const float* __restrict__ dpool, const unsigned char* __restrict__ idx,
                                                             const float* __restrict__ x, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ part,
                                                             long long M, int H, int W, int C, int Ho, int Wo) {
  extern __shared__ float red[];
  const int c4 = C >> 2;                                                       // host: c4 <= 256 and 256 % c4 == 0
  const int lanes = 256 / c4, cq = threadIdx.x % c4, rl = threadIdx.x / c4;
  float mu[4], is[4], fa[4], fb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    mu[j] = mean[cq * 4 + j];
    is[j] = invstd[cq * 4 + j];
    fa[j] = (gamma ? gamma[cq * 4 + j] : 1.f) * is[j];
    fb[j] = (beta ? beta[cq * 4 + j] : 0.f) - mu[j] * fa[j];
  }
  float a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
  const long long step = (long long)gridDim.x * lanes;
  long long m = (long long)blockIdx.x * lanes + rl;
  auto one = [&](long long r, const f32x4 xv) {
    f32x4 g = pool_gather(dpool, idx, r, cq * 4, H, W, C, Ho, Wo);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      g[j] = fmaf(xv[j], fa[j], fb[j]) > 0.f ? g[j] : 0.f;
      a1[j] += g[j];
      a2[j] = fmaf(g[j], (xv[j] - mu[j]) * is[j], a2[j]);
    }
  };
  for (; m + 3 * step < M; m += 4 * step) {
    f32x4 xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xv[u] = *reinterpret_cast<const f32x4*>(x + (size_t)(m + u * step) * C + cq * 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) one(m + u * step, xv[u]);
  }
  for (; m < M; m += step) one(m, *reinterpret_cast<const f32x4*>(x + (size_t)m * C + cq * 4));
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[threadIdx.x * 8 + j] = a1[j]; red[threadIdx.x * 8 + 4 + j] = a2[j]; }
  __syncthreads();
  if ((int)threadIdx.x < c4) {
    float t1[4] = {0, 0, 0, 0}, t2[4] = {0, 0, 0, 0};
    for (int r = 0; r < lanes; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        t1[j] += red[(r * c4 + threadIdx.x) * 8 + j];
        t2[j] += red[(r * c4 + threadIdx.x) * 8 + 4 + j];
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      part[((size_t)blockIdx.x * C + threadIdx.x * 4 + j) * 2] = t1[j];
      part[((size_t)blockIdx.x * C + threadIdx.x * 4 + j) * 2 + 1] = t2[j];
    }
  }
}

__global__ __launch_bounds__(256) void pool_bn_bwd_apply(const float* __restrict__ dpool, const unsigned char* __restrict__ idx,
                                                          const float* __restrict__ x, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ s_dy,
                                                          const float* __restrict__ s_dyx, float* __restrict__ dx, long long M,
                                                          int H, int W, int C, int Ho, int Wo) {
  const int c4 = C >> 2;
  const int lanes = 256 / c4, cq = threadIdx.x % c4, rl = threadIdx.x / c4, c = cq * 4;
  const long long step = (long long)gridDim.x * lanes;
  const float invM = 1.f / (float)M;
  float gi[4], is[4], mu[4], sd[4], sx[4], fa[4], fb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    is[j] = invstd[c + j];
    mu[j] = mean[c + j];
    gi[j] = (gamma ? gamma[c + j] : 1.f) * is[j];
    fa[j] = gi[j];
    fb[j] = (beta ? beta[c + j] : 0.f) - mu[j] * fa[j];
    sd[j] = s_dy[c + j] * invM;
    sx[j] = s_dyx[c + j] * invM;
  }
  for (long long m = (long long)blockIdx.x * lanes + rl; m < M; m += step) {
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)m * C + c);
    f32x4 g = pool_gather(dpool, idx, m, c, H, W, C, Ho, Wo);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      g[j] = fmaf(xv[j], fa[j], fb[j]) > 0.f ? g[j] : 0.f;
      o[j] = bn_dx(gi[j], g[j], sd[j], xv[j], mu[j], is[j], sx[j]);
    }
    *reinterpret_cast<f32x4*>(dx + (size_t)m * C + c) = o;
  }
}

static inline unsigned row_grid(long long M, int C) {      // workgroups for the row-streaming kernels
  const int lanes = C / 4 >= 256 ? 1 : 256 / (C / 4);
  long long g = (M + (long long)lanes * 4 - 1) / ((long long)lanes * 4);
  return (unsigned)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" size_t bevf_bn_work_floats(int C) { return (size_t)kStatGrid * C * 2; }

extern "C" int bevf_bn_stats_f32(const float* x, float* work, float* mean, float* var, float* invstd, int M, int C,
                                 int cs, float eps, void* stream) {
  BEVF_REQUIRE(x && work && mean && var && invstd, "bn_stats: null pointer");
  BEVF_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && cs >= C && cs % 4 == 0, "bn_stats: bad shape (M=%d C=%d cs=%d)", M, C, cs);
  BEVF_REQUIRE(bevf_aligned16(x), "bn_stats: unaligned");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int lanes = C / 4 >= 256 ? 1 : 256 / (C / 4);
  int G = (M + lanes - 1) / lanes;
  if (G > kStatGrid) G = kStatGrid;
  hipLaunchKernelGGL(stats_partials, dim3(G), dim3(256), 256 * 8 * sizeof(float), st, x, work, M, C, cs);
  hipLaunchKernelGGL(stats_finalize, dim3(C), dim3(256), 0, st, x, work, mean, var, invstd, M, C, G, eps);
  return bevf_check_launch("bevf_bn_stats_f32");
}

// Batch statistics from partial sums a producer left behind (bevf_conv3x3_wino_f32 with `stats`): part [G][C][2] =
// {sum(x - pivot), sum((x - pivot)^2)} over disjoint row sets covering all M rows.  Same fixed-order double merge as above.
extern "C" int bevf_bn_stats_from_partials_f32(const float* part, int G, const float* pivot, float* mean, float* var,
                                               float* invstd, int M, int C, float eps, void* stream) {
  BEVF_REQUIRE(part && pivot && mean && var && invstd && G > 0 && M > 0 && C > 0, "bn_stats_from_partials: bad arguments");
  hipLaunchKernelGGL(stats_finalize, dim3(C), dim3(256), 0, static_cast<hipStream_t>(stream), pivot, part, mean, var,
                     invstd, M, C, G, eps);
  return bevf_check_launch("bevf_bn_stats_from_partials_f32");
}

extern "C" int bevf_bn_update_running_f32(const float* mean, const float* var, float* running_mean, float* running_var,
                                          int64_t* num_batches_tracked, int C, int M, float momentum, void* stream) {
  BEVF_REQUIRE(mean && var && running_mean && running_var && C > 0 && M > 0, "bn_update_running: bad arguments");
  const float unbias = M > 1 ? (float)((double)M / (double)(M - 1)) : 1.f;
  hipLaunchKernelGGL(bn_update_running, dim3((C + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), mean, var,
                     running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked), C, 1.f - momentum, momentum,
                     momentum * unbias);
  return bevf_check_launch("bevf_bn_update_running_f32");
}

extern "C" int bevf_bn_apply_f32(const float* x, const float* mean, const float* invstd, const float* gamma,
                                 const float* beta, const float* res, float* y, int M, int C, int cs, int relu,
                                 void* stream) {
  BEVF_REQUIRE(x && mean && invstd && y, "bn_apply: null pointer");
  BEVF_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && cs >= C && cs % 4 == 0, "bn_apply: bad shape");
  BEVF_REQUIRE(bevf_aligned16(x) && bevf_aligned16(y) && (!res || bevf_aligned16(res)), "bn_apply: unaligned");
  hipLaunchKernelGGL(bn_apply, dim3(row_grid(M, C)), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     mean, invstd, gamma, beta, res, y, (long long)M, C, cs, relu);
  return bevf_check_launch("bevf_bn_apply_f32");
}

extern "C" int bevf_bn_backward_f32(float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                                    const float* gamma, const float* beta, float* work, float* dgamma, float* dbeta,
                                    float* dx, int M, int C, int cs, int relu, void* stream) {
  BEVF_REQUIRE(dy && work && dbeta, "bn_backward: null pointer");
  BEVF_REQUIRE(!relu || y || (x && mean && invstd), "bn_backward: relu needs the forward output, or x/mean/invstd to recompute it");
  BEVF_REQUIRE(!dx || (x && mean && invstd && dgamma), "bn_backward: dx needs x, mean, invstd, dgamma");
  BEVF_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && cs >= C && cs % 4 == 0, "bn_backward: bad shape");
  // y == NULL: mask recomputed from x (no residual in the forward); relu == 2: additionally dy is left untouched (nobody reads the
  // masked gradient of a layer without a skip connection) and the second pass masks again: one write of dy less.
  // relu | 4: frozen statistics (mean / invstd are constants, e.g. the running buffers of an eval-mode layer): dx = gamma invstd dy
  const int frozen = (relu & 4) ? 1 : 0;
  relu &= 3;
  BEVF_REQUIRE(relu != 2 || (!y && x && mean && invstd), "bn_backward: relu = 2 recomputes the mask from x (y must be NULL)");
  const int relu_mode = relu ? (y ? 1 : (relu == 2 ? 3 : 2)) : 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int lanes = C / 4 >= 256 ? 1 : 256 / (C / 4);
  int G = (M + lanes - 1) / lanes;
  if (G > kStatGrid) G = kStatGrid;
  hipLaunchKernelGGL(bn_bwd_partials, dim3(G), dim3(256), 256 * 8 * sizeof(float), st, dy, y, x, mean, invstd, gamma, beta,
                     work, M, C, cs, relu_mode);
  hipLaunchKernelGGL(sums_finalize, dim3(C), dim3(256), 0, st, work, dbeta, dgamma, C, G);
  if (dx)
    hipLaunchKernelGGL(bn_bwd_apply, dim3(row_grid(M, C)), dim3(256), 0, st, dy, x, mean, invstd, gamma,
                       dbeta, dgamma, dx, (long long)M, C, cs, beta, relu_mode == 3 ? 1 : 0, frozen);
  return bevf_check_launch("bevf_bn_backward_f32");
}

extern "C" int bevf_bn_backward_from_partials_f32(const float* dy, const float* x, const float* mean, const float* invstd,
                                                  const float* gamma, const float* part, int G, float* dgamma, float* dbeta,
                                                  float* dx, int M, int C, int cs, void* stream) {
  BEVF_REQUIRE(dy && x && mean && invstd && part && dgamma && dbeta, "bn_backward_from_partials: null pointer");
  BEVF_REQUIRE(G > 0 && M > 0 && C > 0 && C % 4 == 0 && cs >= C && cs % 4 == 0, "bn_backward_from_partials: bad shape");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(sums_finalize, dim3(C), dim3(256), 0, st, part, dbeta, dgamma, C, G);
  if (dx)
    hipLaunchKernelGGL(bn_bwd_apply, dim3(row_grid(M, C)), dim3(256), 0, st, dy, x, mean, invstd, gamma, dbeta, dgamma, dx,
                       (long long)M, C, cs);
  return bevf_check_launch("bevf_bn_backward_from_partials_f32");
}

// BatchNorm(+ReLU) backward whose dY comes out of a 3x3/s2/p1 max-pool backward (the ResNet stem, ref src/encoders.py:154-157 in
// training): dpool [N][Ho][Wo][C] gradient of the pooled map, idx the argmax codes of bevf_maxpool3x3s2_idx_f32, x the raw conv
// output [N][H][W][C].  Bit-identical to bevf_maxpool3x3s2_bwd_f32 followed by bevf_bn_backward_f32(relu = 1, y = NULL) without the
// dense dY in HBM.
extern "C" int bevf_pool_bn_backward_f32(const float* dpool, const uint8_t* idx, const float* x, const float* mean,
                                         const float* invstd, const float* gamma, const float* beta, float* work, float* dgamma,
                                         float* dbeta, float* dx, int N, int H, int W, int C, void* stream) {
  BEVF_REQUIRE(dpool && idx && x && mean && invstd && work && dgamma && dbeta && dx, "pool_bn_backward: null pointer");
  BEVF_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0,
               "pool_bn_backward: C=%d must be a multiple of 4 with 256 %% (C/4) == 0", C);
  BEVF_REQUIRE(bevf_aligned16(dpool) && bevf_aligned16(x) && bevf_aligned16(dx), "pool_bn_backward: unaligned");
  const long long M = (long long)N * H * W;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int lanes = 256 / (C / 4);
  long long G = (M + lanes - 1) / lanes;
  if (G > kStatGrid) G = kStatGrid;
  hipLaunchKernelGGL(pool_bn_bwd_partials, dim3((unsigned)G), dim3(256), 256 * 8 * sizeof(float), st, dpool, idx, x, mean, invstd, gamma,
                     beta, work, M, H, W, C, Ho, Wo);
  hipLaunchKernelGGL(sums_finalize, dim3(C), dim3(256), 0, st, work, dbeta, dgamma, C, (int)G);
  hipLaunchKernelGGL(pool_bn_bwd_apply, dim3(row_grid(M, C)), dim3(256), 0, st, dpool, idx, x, mean, invstd, gamma, beta, dbeta, dgamma,
                     dx, M, H, W, C, Ho, Wo);
  return bevf_check_launch("bevf_pool_bn_backward_f32");
}

// The first stage of bevf_gmax_bn_backward_f32 alone: dgm [B][C] = dg masked by the ReLU, dbeta = sum dgm, dgamma = sum dgm * xhat
// (gathered from the B argmax rows per channel).  For callers that never build the dense dx (training.py: the low-rank form of
// PointNet's last layer).
extern "C" int bevf_gmax_bn_sums_f32(const float* dg, const float* gmax, const int32_t* idx, const float* x, const float* mean,
                                     const float* invstd, float* dgm, float* dgamma, float* dbeta, int B, int P, int C, int cs,
                                     void* stream) {
  BEVF_REQUIRE(dg && gmax && idx && x && mean && invstd && dgm && dgamma && dbeta, "gmax_bn_sums: null pointer");
  BEVF_REQUIRE(B > 0 && P > 0 && C > 0 && cs >= C, "gmax_bn_sums: bad shape");
  BEVF_REQUIRE((long long)B * P < (1ll << 31), "gmax_bn_sums: too many rows");
  hipLaunchKernelGGL(gmax_bn_sums, dim3((C + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), dg, gmax, idx, x, mean,
                     invstd, dgm, dbeta, dgamma, B, P, C, cs);
  return bevf_check_launch("bevf_gmax_bn_sums_f32");
}

// Backward of y = relu(batchnorm(x)) followed by a max over the P rows of each of B groups (ref src/encoders.py:296-299
// in training mode): dg [B][C] gradient of the max, gmax its forward value, idx its argmax row.  Writes dgamma, dbeta and
// the dense dx [B*P][cs]; dgm [B][C] is scratch.  Equivalent to group_max_bwd + bn_backward without the dense dy.
extern "C" int bevf_gmax_bn_backward_f32(const float* dg, const float* gmax, const int32_t* idx, const float* x,
                                         const float* mean, const float* invstd, const float* gamma, float* dgm,
                                         float* dgamma, float* dbeta, float* dx, int B, int P, int C, int cs, void* stream) {
  BEVF_REQUIRE(dg && gmax && idx && x && mean && invstd && dgm && dgamma && dbeta && dx, "gmax_bn_backward: null pointer");
  BEVF_REQUIRE(B > 0 && P > 0 && C > 0 && C % 4 == 0 && cs >= C && cs % 4 == 0, "gmax_bn_backward: bad shape");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long M = (long long)B * P;
  BEVF_REQUIRE(M < (1ll << 31), "gmax_bn_backward: too many rows");
  hipLaunchKernelGGL(gmax_bn_sums, dim3((C + 255) / 256), dim3(256), 0, st, dg, gmax, idx, x, mean, invstd, dgm, dbeta, dgamma,
                     B, P, C, cs);
  hipLaunchKernelGGL(bn_bwd_apply, dim3(row_grid(M, C)), dim3(256), 0, st, (const float*)nullptr, x, mean, invstd, gamma, dbeta,
                     dgamma, dx, M, C, cs);
  const long long total = (long long)B * C;
  hipLaunchKernelGGL(gmax_bn_scatter, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dgm, idx, gamma, invstd, dx, P, C,
                     cs, total);
  return bevf_check_launch("bevf_gmax_bn_backward_f32");
}
