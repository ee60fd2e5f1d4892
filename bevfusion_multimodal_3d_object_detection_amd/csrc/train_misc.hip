// Backward / optimiser kernels around the MFMA convolutions (training step).
// Everything here is HBM- or latency-bound; 16 B per lane where the layout allows.
#include "common.h"

namespace {

static inline unsigned ew_grid(long long items) {
  long long g = (items + 255) / 256;
  return (unsigned)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}
#define GRID_STRIDE(i, total) for (long long i = blockIdx.x * 256ll + threadIdx.x; i < (total); i += (long long)gridDim.x * 256)

// ---- element-wise helpers ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add_inplace(float* __restrict__ y, const float* __restrict__ x, long long n4) {
  GRID_STRIDE(i, n4) {
    f32x4 a = reinterpret_cast<f32x4*>(y)[i];
    const f32x4 b = reinterpret_cast<const f32x4*>(x)[i];
    a += b;
    reinterpret_cast<f32x4*>(y)[i] = a;
  }
}
__global__ __launch_bounds__(256) void relu_mask(float* __restrict__ dy, const float* __restrict__ y, long long n4) {
  GRID_STRIDE(i, n4) {
    f32x4 g = reinterpret_cast<f32x4*>(dy)[i];
    const f32x4 v = reinterpret_cast<const f32x4*>(y)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = v[j] > 0.f ? g[j] : 0.f;
    reinterpret_cast<f32x4*>(dy)[i] = g;
  }
}

// ---- max-pool 3x3 s2 p1 with argmax (first maximum in scan order, as torch) and its gather backward ---------------
__global__ __launch_bounds__(256) void maxpool_fwd_idx(const float* __restrict__ x, float* __restrict__ y,
                                                        unsigned char* __restrict__ idx, int N, int H, int W, int C, int Ho,
                                                        int Wo) {
  const int c4 = C >> 2;
  const long long total = (long long)N * Ho * Wo * c4;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % c4) * 4;
    long long pix = i / c4;
    const int ow = (int)(pix % Wo);
    pix /= Wo;
    const int oh = (int)(pix % Ho), n = (int)(pix / Ho);
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int best[4] = {0, 0, 0, 0};
    for (int dh = 0; dh < 3; ++dh) {
      const int ih = 2 * oh - 1 + dh;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int dw = 0; dw < 3; ++dw) {
        const int iw = 2 * ow - 1 + dw;
        if ((unsigned)iw >= (unsigned)W) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((size_t)(n * H + ih) * W + iw) * C + c);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (v[j] > m[j] || (v[j] != v[j])) { m[j] = v[j]; best[j] = dh * 3 + dw; }
      }
    }
    *reinterpret_cast<f32x4*>(y + (size_t)i * 4) = m;
    *reinterpret_cast<unsigned*>(idx + (size_t)i * 4) =
        (unsigned)best[0] | ((unsigned)best[1] << 8) | ((unsigned)best[2] << 16) | ((unsigned)best[3] << 24);
  }
}
// max-pool over relu(batchnorm(x)) evaluated on the fly (the ResNet stem in training): the normalised 64-channel stem map -- the
// largest activation of the step -- is never written; values and argmax codes are those of bn_apply (same fma, same max(t, 0))
// followed by maxpool_fwd_idx
__global__ __launch_bounds__(256) void bn_relu_maxpool_idx(const float* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            unsigned char* __restrict__ idx, int N, int H, int W, int C, int Ho,
                                                            int Wo) {
  const int c4 = C >> 2;
  const long long total = (long long)N * Ho * Wo * c4;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % c4) * 4;
    long long pix = i / c4;
    const int ow = (int)(pix % Wo);
    pix /= Wo;
    const int oh = (int)(pix % Ho), n = (int)(pix / Ho);
    float a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[j] = (gamma ? gamma[c + j] : 1.f) * invstd[c + j];
      b[j] = (beta ? beta[c + j] : 0.f) - mean[c + j] * a[j];
    }
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int best[4] = {0, 0, 0, 0};
    for (int dh = 0; dh < 3; ++dh) {
      const int ih = 2 * oh - 1 + dh;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int dw = 0; dw < 3; ++dw) {
        const int iw = 2 * ow - 1 + dw;
        if ((unsigned)iw >= (unsigned)W) continue;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + ((size_t)(n * H + ih) * W + iw) * C + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = fmaxf(fmaf(xv[j], a[j], b[j]), 0.f);
          if (v > m[j] || (v != v)) { m[j] = v; best[j] = dh * 3 + dw; }
        }
      }
    }
    *reinterpret_cast<f32x4*>(y + (size_t)i * 4) = m;
    *reinterpret_cast<unsigned*>(idx + (size_t)i * 4) =
        (unsigned)best[0] | ((unsigned)best[1] << 8) | ((unsigned)best[2] << 16) | ((unsigned)best[3] << 24);
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd(const float* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                    float* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo) {
  const int c4 = C >> 2;
  const long long total = (long long)N * H * W * c4;
  GRID_STRIDE(i, total) {                              // gather: every output window that contains this input pixel
    const int c = (int)(i % c4) * 4;
    long long pix = i / c4;
    const int iw = (int)(pix % W);
    pix /= W;
    const int ih = (int)(pix % H), n = (int)(pix / H);
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    for (int oh = ih / 2; oh <= (ih + 1) / 2; ++oh) {                          // 2*oh-1 <= ih <= 2*oh+1
      if (oh >= Ho) continue;
      const int dh = ih - (2 * oh - 1);
      for (int ow = iw / 2; ow <= (iw + 1) / 2; ++ow) {
        if (ow >= Wo) continue;
        const unsigned code = (unsigned)(dh * 3 + (iw - (2 * ow - 1)));
        const size_t o = ((size_t)(n * Ho + oh) * Wo + ow) * C + c;
        const unsigned id4 = *reinterpret_cast<const unsigned*>(idx + o);
        const f32x4 d = *reinterpret_cast<const f32x4*>(dy + o);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (((id4 >> (8 * j)) & 0xff) == code) g[j] += d[j];
      }
    }
    *reinterpret_cast<f32x4*>(dx + (size_t)i * 4) = g;
  }
}

// ---- bilinear backward (scatter with float atomics; dx zero-filled by the caller) -----------------------------------
__device__ __forceinline__ void lin_coord(int o, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
  float s = scale * ((float)o + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = fminf(fmaxf(s - (float)i0, 0.f), 1.f);
  l0 = 1.f - l1;
}
__global__ __launch_bounds__(256) void bilinear_bwd(const float* __restrict__ dy, float* __restrict__ dx, int Hi, int Wi,
                                                     int C, int x_cs, int Ho, int Wo, int y_cs, float sh, float sw,
                                                     long long total) {
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    long long pix = i / C;
    const int ow = (int)(pix % Wo);
    const long long t = pix / Wo;
    const int oh = (int)(t % Ho), b = (int)(t / Ho);
    int h0, h1, w0, w1;
    float lh0, lh1, lw0, lw1;
    lin_coord(oh, sh, Hi, h0, h1, lh0, lh1);
    lin_coord(ow, sw, Wi, w0, w1, lw0, lw1);
    const float g = dy[(size_t)pix * y_cs + c];
    float* base = dx + (size_t)b * Hi * Wi * x_cs + c;
    atomicAdd(base + ((size_t)h0 * Wi + w0) * x_cs, lh0 * lw0 * g);
    atomicAdd(base + ((size_t)h0 * Wi + w1) * x_cs, lh0 * lw1 * g);
    atomicAdd(base + ((size_t)h1 * Wi + w0) * x_cs, lh1 * lw0 * g);
    atomicAdd(base + ((size_t)h1 * Wi + w1) * x_cs, lh1 * lw1 * g);
  }
}

// ---- camera mean backward: dx[b][n][p][c] = dy[b][p][c] / ncam ------------------------------------------------------
__global__ __launch_bounds__(256) void cam_mean_bwd(const f32x4* __restrict__ dy, f32x4* __restrict__ dx, int ncam,
                                                     long long pc4, long long total) {
  const float div = (float)ncam;
  GRID_STRIDE(i, total) {
    const long long bn = i / pc4, r = i - bn * pc4, b = bn / ncam;
    f32x4 g = dy[b * pc4 + r];
    g.x /= div; g.y /= div; g.z /= div; g.w /= div;
    dx[i] = g;
  }
}

// ---- group max with argmax, and its scatter backward (dx zero-filled by the caller) -----------------------------------
// stage 1: one workgroup per (group, chunk of GM_CHUNK points): threads own channel quads, rows are read coalesced
constexpr int GM_CHUNK = 128;
// AFFINE: the rows are raw pre-BatchNorm values and the maximum is taken over relu(fma(x, a, b)) with a = gamma*invstd,
// b = beta - mean*a -- bn_apply's own operations, so value and argmax equal those of the materialised activation.
template <bool AFFINE>
__global__ __launch_bounds__(256) void group_max_partial(const float* __restrict__ x, float* __restrict__ pmax,
                                                          int* __restrict__ pidx, int P, int C, int nchunks,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta) {
  const int g = blockIdx.x / nchunks, ch = blockIdx.x - g * nchunks;
  const int p0 = ch * GM_CHUNK, p1 = (p0 + GM_CHUNK < P) ? p0 + GM_CHUNK : P;
  const int c4 = C >> 2;
  for (int cq = threadIdx.x; cq < c4; cq += 256) {
    float fa[4] = {1.f, 1.f, 1.f, 1.f}, fb[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (AFFINE) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        fa[j] = (gamma ? gamma[cq * 4 + j] : 1.f) * invstd[cq * 4 + j];
        fb[j] = (beta ? beta[cq * 4 + j] : 0.f) - mean[cq * 4 + j] * fa[j];
      }
    }
    auto act = [&](f32x4 v) {
      if constexpr (AFFINE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaf(v[j], fa[j], fb[j]), 0.f);
      }
      return v;
    };
    const float* src = x + ((size_t)g * P + p0) * C + cq * 4;
    f32x4 m = act(*reinterpret_cast<const f32x4*>(src));
    int bi[4] = {p0, p0, p0, p0};
    for (int p = p0 + 1; p < p1; ++p) {
      const f32x4 v = act(*reinterpret_cast<const f32x4*>(src + (size_t)(p - p0) * C));
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (v[j] > m[j]) { m[j] = v[j]; bi[j] = p; }
    }
    const size_t o = ((size_t)g * nchunks + ch) * C + cq * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) { pmax[o + j] = m[j]; pidx[o + j] = bi[j]; }
  }
}
// stage 2: first maximum over the chunks (chunk order = point order, strict > keeps the earliest)
__global__ __launch_bounds__(256) void group_max_final(const float* __restrict__ pmax, const int* __restrict__ pidx,
                                                        float* __restrict__ y, int* __restrict__ idx, int C, int nchunks,
                                                        long long total) {
  GRID_STRIDE(i, total) {
    const long long g = i / C;
    const int c = (int)(i - g * C);
    const size_t base = (size_t)g * nchunks * C + c;
    float m = pmax[base];
    int best = pidx[base];
    for (int k = 1; k < nchunks; ++k) {
      const float v = pmax[base + (size_t)k * C];
      if (v > m) { m = v; best = pidx[base + (size_t)k * C]; }
    }
    y[i] = m;
    idx[i] = best;
  }
}
__global__ __launch_bounds__(256) void group_max_bwd(const float* __restrict__ dy, const int* __restrict__ idx,
                                                      float* __restrict__ dx, int P, int C, long long total) {
  GRID_STRIDE(i, total) {
    const long long g = i / C;
    const int c = (int)(i - g * C);
    dx[((size_t)g * P + idx[i]) * C + c] = dy[i];
  }
}

// ---- the sparse rows of the low-rank group-max backward (training._backward_from_groupmax_lowrank): S [G][C] holds one entry per (group,
// channel), sitting at row g*P + idx[g][c] of the [G*P][K] activation.  Both kernels add in a FIXED order (no atomics): run-to-run bit-identical.
// sparse_rows_wgrad: out[c][k] = sum_g S[g][c] * A[g*P + idx[g][c]][k]                       (groups in ascending order)
__global__ __launch_bounds__(256) void sparse_rows_wgrad(const float* __restrict__ S, const int* __restrict__ idx,
                                                          const float* __restrict__ A, float* __restrict__ out, int G, int P, int C,
                                                          int K) {
  const int c = blockIdx.x;
  for (int k = threadIdx.x; k < K; k += 256) {
    float acc = 0.f;
    for (int g = 0; g < G; ++g) acc = fmaf(S[(size_t)g * C + c], A[((size_t)g * P + idx[(size_t)g * C + c]) * K + k], acc);
    out[(size_t)c * K + k] = acc;
  }
}
// sparse_rows_scatter: dA[g*P + idx[g][c]][k] += sum over the channels c' of group g that share that row (ascending c') of S[g][c'] * W[c'][k].
// One workgroup per (g, c): the FIRST channel of a row owns it and walks the later ones; the others leave.  Rows of different owners differ.
__global__ __launch_bounds__(256) void sparse_rows_scatter(const float* __restrict__ S, const int* __restrict__ idx,
                                                            const float* __restrict__ Wt, float* __restrict__ dA, int P, int C, int K) {
  extern __shared__ int srow[];                                    // idx[g][0..C)
  const int g = blockIdx.y, c = blockIdx.x;
  for (int i = threadIdx.x; i < C; i += 256) srow[i] = idx[(size_t)g * C + i];
  __syncthreads();
  const int row = srow[c];
  for (int j = 0; j < c; ++j)
    if (srow[j] == row) return;                                    // (uniform: every thread reads the same LDS words)
  float* const dst = dA + ((size_t)g * P + row) * K;
  for (int k = threadIdx.x; k < K; k += 256) {
    float acc = 0.f;
    for (int j = c; j < C; ++j)
      if (srow[j] == row) acc = fmaf(S[(size_t)g * C + j], Wt[(size_t)j * K + k], acc);
    dst[k] += acc;
  }
}

// ---- zero stuffing for the data gradient of strided convs: out[n][s*oh][s*ow][c] = dy[n][oh][ow][c], 0 elsewhere -----
__global__ __launch_bounds__(256) void zero_stuff(const float* __restrict__ dy, float* __restrict__ out, int N, int Ho,
                                                   int Wo, int C, int H, int W, int s) {
  const int c4 = C >> 2;
  const long long total = (long long)N * H * W * c4;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % c4) * 4;
    long long pix = i / c4;
    const int iw = (int)(pix % W);
    pix /= W;
    const int ih = (int)(pix % H), n = (int)(pix / H);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ih % s == 0 && iw % s == 0 && ih / s < Ho && iw / s < Wo)
      v = *reinterpret_cast<const f32x4*>(dy + ((size_t)(n * Ho + ih / s) * Wo + iw / s) * C + c);
    *reinterpret_cast<f32x4*>(out + (size_t)i * 4) = v;
  }
}

// ---- data gradient of a stride-2 conv, assembled from its four parity classes ------------------------------------------
// dx[n][ih][iw][:] = cls[(ih&1)*2 + (iw&1)][n][ih>>1][iw>>1][:]  (class q stored with spatial size hq x wq; null = zeros)
struct Interleave4 {
  const float* cls[4];
  int hq[4], wq[4];
};
__global__ __launch_bounds__(256) void interleave2x2(const Interleave4 a, float* __restrict__ dx, int N, int H, int W, int C) {
  const int c4 = C >> 2;
  const long long total = (long long)N * H * W * c4;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % c4) * 4;
    long long pix = i / c4;
    const int iw = (int)(pix % W);
    pix /= W;
    const int ih = (int)(pix % H), n = (int)(pix / H);
    const int q = (ih & 1) * 2 + (iw & 1);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (a.cls[q]) v = *reinterpret_cast<const f32x4*>(a.cls[q] + ((size_t)(n * a.hq[q] + (ih >> 1)) * a.wq[q] + (iw >> 1)) * C + c);
    *reinterpret_cast<f32x4*>(dx + (size_t)i * 4) = v;
  }
}

// ---- dense layer backward ------------------------------------------------------------------------------------------
// dx[b][k] = sum_o dy[b][perm(o)] * W[o][k]: workgroup = chunk of outputs, partial sums -> part[g][b][k]
// Streams W once with 16-byte loads: thread (kq, osub) owns k = 4kq..4kq+3 and every `lanes`-th row of the workgroup's
// chunk, 4 rows in flight; each (workgroup, osub) pair writes its own partial slot, merged in fixed order afterwards.
__global__ __launch_bounds__(256) void linear_bwd_dx_partials(const float* __restrict__ dy, const float* __restrict__ w,
                                                               float* __restrict__ part, int B, int K, int O, int chunk,
                                                               int perm_inner, int perm_outer) {
  extern __shared__ float sdy[];                          // [chunk][B] gradients of this workgroup's outputs
  const int o0 = blockIdx.x * chunk, o1 = (o0 + chunk < O) ? o0 + chunk : O;
  for (int i = threadIdx.x; i < (o1 - o0) * B; i += 256) {
    const int o = o0 + i / B, b = i - (i / B) * B;
    const int oo = perm_inner > 0 ? (o % perm_inner) * perm_outer + o / perm_inner : o;
    sdy[i] = dy[(size_t)b * O + oo];
  }
  __syncthreads();
  const int k4 = K >> 2, per = k4 < 256 ? k4 : 256, lanes = 256 / per;
  const int osub = threadIdx.x / per;
  if (osub >= lanes) return;
  for (int kq = threadIdx.x % per; kq < k4; kq += per) {
    for (int b0 = 0; b0 < B; b0 += 8) {
      const int nb = B - b0 < 8 ? B - b0 : 8;
      f32x4 acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      int o = o0 + osub;
      for (; o + 3 * lanes < o1; o += 4 * lanes) {
        f32x4 wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) wv[u] = *reinterpret_cast<const f32x4*>(w + (size_t)(o + u * lanes) * K + kq * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* g = sdy + (o + u * lanes - o0) * B + b0;
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (j < nb) acc[j] += g[j] * wv[u];
        }
      }
      for (; o < o1; o += lanes) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (size_t)o * K + kq * 4);
        const float* g = sdy + (o - o0) * B + b0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j < nb) acc[j] += g[j] * wv;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < nb) *reinterpret_cast<f32x4*>(part + (((size_t)blockIdx.x * lanes + osub) * B + b0 + j) * K + kq * 4) = acc[j];
    }
  }
}
// out[slice][i] = sum of part[g][i] over the slice's range of g (fixed order, double); slices = gridDim.y
__global__ __launch_bounds__(256) void linear_bwd_dx_final(const float* __restrict__ part, float* __restrict__ out, int BK,
                                                            int G) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= BK) return;
  const int per = (G + gridDim.y - 1) / gridDim.y, g0 = blockIdx.y * per, g1 = g0 + per < G ? g0 + per : G;
  double s = 0;
  for (int g = g0; g < g1; ++g) s += part[(size_t)g * BK + i];
  out[(size_t)blockIdx.y * BK + i] = (float)s;
}
// dW[o][k] = sum_b dy[b][perm(o)] * x[b][k];  db[o] = sum_b dy[b][perm(o)]
__global__ __launch_bounds__(256) void linear_bwd_dw(const float* __restrict__ dy, const float* __restrict__ x,
                                                      float* __restrict__ dw, float* __restrict__ db, int B, int K, int O,
                                                      int perm_inner, int perm_outer) {
  const int k4 = K >> 2;
  const long long total = (long long)O * k4;
  GRID_STRIDE(i, total) {
    const int o = (int)(i / k4), kq = (int)(i - (long long)o * k4);
    const int oo = perm_inner > 0 ? (o % perm_inner) * perm_outer + o / perm_inner : o;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    for (int b = 0; b < B; ++b) {
      const float g = dy[(size_t)b * O + oo];
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)b * K + kq * 4);
      acc += g * xv;
      bsum += g;
    }
    *reinterpret_cast<f32x4*>(dw + (size_t)o * K + kq * 4) = acc;
    if (kq == 0 && db) db[o] = bsum;
  }
}

// ---- CenterNet head tail backward: dpre = dout * (sigmoid' for the heatmap), dhid, dW, db -----------------------------
struct HeadBwdArgs {
  const float* hid; const float* w; const float* out0;   // out0 = post-sigmoid heatmap
  const float* dout[5];
  float* dhid; float* dw; float* db;                     // dw/db zero-filled by the caller (atomics)
  int B, P, hc, c[5], n_sigmoid;
};
// Tiled form (round 2): per tile of 256 pixels the masked gradients g [256][ctot] and, head by head, the hidden activations
// [256][hc] sit in LDS; dhid is computed by thread = pixel, dW / db by thread = (output, hidden channel) summing over the tile's
// pixels in order -- no cross-lane reduction at all (the first version spent 6 shuffles + an LDS atomic per (output, channel) and
// wave: 0.51 ms per step for a 20 000-pixel map).  Workgroup sums go to dw / db with one atomic per element.
__global__ __launch_bounds__(256) void head_tail_bwd_tiled(const HeadBwdArgs a) {
  extern __shared__ float sm[];
  const int ctot = a.c[0] + a.c[1] + a.c[2] + a.c[3] + a.c[4];
  const int gp = ctot | 1;                               // odd pitch: thread = pixel reads its row without bank conflicts
  float* wl = sm;                                        // [ctot][hc]
  float* accw = wl + ctot * a.hc;                        // [ctot][hc]
  float* accb = accw + ctot * a.hc;                      // [ctot]
  float* gt = accb + ctot;                               // [256][gp]
  float* ht = gt + 256 * gp;                             // [256][hc]
  const int tid = threadIdx.x;
  for (int i = tid; i < ctot * a.hc; i += 256) { wl[i] = a.w[i]; accw[i] = 0.f; }
  for (int i = tid; i < ctot; i += 256) accb[i] = 0.f;
  const long long total = (long long)a.B * a.P;
  for (long long pix0 = blockIdx.x * 256ll; pix0 < total; pix0 += (long long)gridDim.x * 256) {
    __syncthreads();                                     // previous tile fully consumed (first pass: wl / acc initialised)
    const long long pix = pix0 + tid;
    const bool ok = pix < total;
    const int b = ok ? (int)(pix / a.P) : 0, p = ok ? (int)(pix - (long long)b * a.P) : 0;
    int oc0 = 0;
    for (int k = 0; k < 5; ++k) {
      for (int c = 0; c < a.c[k]; ++c) {
        float gv = ok ? a.dout[k][((size_t)b * a.c[k] + c) * a.P + p] : 0.f;
        if (oc0 + c < a.n_sigmoid && ok) { const float s = a.out0[((size_t)b * a.c[0] + c) * a.P + p]; gv *= s * (1.f - s); }
        gt[tid * gp + oc0 + c] = gv;
      }
      oc0 += a.c[k];
    }
    const int npx = total - pix0 < 256 ? (int)(total - pix0) : 256;
    oc0 = 0;
    for (int k = 0; k < 5; ++k) {
      __syncthreads();                                   // gt complete / previous head's ht consumed
      for (int i = tid; i < npx * a.hc; i += 256) {      // hidden activations of head k, coalesced
        const int pp = i / a.hc, j = i - pp * a.hc;
        ht[i] = a.hid[(size_t)(pix0 + pp) * 5 * a.hc + k * a.hc + j];
      }
      if (ok) {                                          // dhid of my pixel
        for (int j = 0; j < a.hc; ++j) {
          float dh = 0.f;
          for (int c = 0; c < a.c[k]; ++c) dh = fmaf(gt[tid * gp + oc0 + c], wl[(oc0 + c) * a.hc + j], dh);
          a.dhid[(size_t)pix * 5 * a.hc + k * a.hc + j] = dh;
        }
      }
      __syncthreads();
      for (int q = tid; q < a.c[k] * a.hc; q += 256) {   // dW: fixed (output, channel) -> thread mapping, sequential over pixels
        const int c = q / a.hc, j = q - c * a.hc;
        float sacc = 0.f;
        for (int pp = 0; pp < npx; ++pp) sacc = fmaf(gt[pp * gp + oc0 + c], ht[pp * a.hc + j], sacc);
        accw[(oc0 + c) * a.hc + j] += sacc;
      }
      if (tid < a.c[k]) {
        float sacc = 0.f;
        for (int pp = 0; pp < npx; ++pp) sacc += gt[pp * gp + oc0 + tid];
        accb[oc0 + tid] += sacc;
      }
      oc0 += a.c[k];
    }
  }
  __syncthreads();
  for (int i = tid; i < ctot * a.hc; i += 256) atomicAdd(&a.dw[i], accw[i]);
  for (int i = tid; i < ctot; i += 256) atomicAdd(&a.db[i], accb[i]);
}

__global__ __launch_bounds__(256) void head_tail_bwd(const HeadBwdArgs a) {
  extern __shared__ float sm[];                          // weights [ctot][hc], then per-wave dW/db accumulators
  const int ctot = a.c[0] + a.c[1] + a.c[2] + a.c[3] + a.c[4];
  float* wl = sm;
  float* accw = sm + ctot * a.hc;                        // [ctot][hc] block-level accumulation
  float* accb = accw + ctot * a.hc;                      // [ctot]
  for (int i = threadIdx.x; i < ctot * a.hc; i += 256) { wl[i] = a.w[i]; accw[i] = 0.f; }
  for (int i = threadIdx.x; i < ctot; i += 256) accb[i] = 0.f;
  __syncthreads();
  const long long total = (long long)a.B * a.P;
  for (long long pix0 = blockIdx.x * 256ll; pix0 < total; pix0 += (long long)gridDim.x * 256) {
    const long long pix = pix0 + threadIdx.x;
    const bool ok = pix < total;
    const int b = ok ? (int)(pix / a.P) : 0, p = ok ? (int)(pix - (long long)b * a.P) : 0;
    int oc0 = 0;
    for (int k = 0; k < 5; ++k) {
      float g[16];                                        // c[k] <= 16 outputs per branch (statically indexed)
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        float gv = 0.f;
        if (c < a.c[k]) {
          gv = ok ? a.dout[k][((size_t)b * a.c[k] + c) * a.P + p] : 0.f;
          if (oc0 + c < a.n_sigmoid && ok) { const float s = a.out0[((size_t)b * a.c[0] + c) * a.P + p]; gv *= s * (1.f - s); }
          float gs = gv;                                  // db: wave reduce, one LDS atomic per wave
#pragma unroll
          for (int s = 32; s >= 1; s >>= 1) gs += __shfl_xor(gs, s);
          if ((threadIdx.x & 63) == 0) atomicAdd(&accb[oc0 + c], gs);
        }
        g[c] = gv;
      }
      for (int j = 0; j < a.hc; ++j) {
        const float hv = ok ? a.hid[(size_t)pix * 5 * a.hc + k * a.hc + j] : 0.f;
        float dh = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          if (c < a.c[k]) {
            dh = fmaf(g[c], wl[(oc0 + c) * a.hc + j], dh);
            float t = g[c] * hv;
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) t += __shfl_xor(t, s);
            if ((threadIdx.x & 63) == 0) atomicAdd(&accw[(oc0 + c) * a.hc + j], t);
          }
        }
        if (ok) a.dhid[(size_t)pix * 5 * a.hc + k * a.hc + j] = dh;
      }
      oc0 += a.c[k];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ctot * a.hc; i += 256) atomicAdd(&a.dw[i], accw[i]);
  for (int i = threadIdx.x; i < ctot; i += 256) atomicAdd(&a.db[i], accb[i]);
}

// ---- loss backward (d total_loss / d predictions), ref src/centernet_target.py:544-622 -----------------------------------
// focal: p = clamp(sigmoid(x), 1e-4, 1-1e-4); loss = -(sum pos + sum neg)/num_pos (or -sum neg when num_pos == 0)
__global__ __launch_bounds__(256) void focal_bwd(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                  const float* __restrict__ num_pos, float* __restrict__ dpred,
                                                  float wscale, long long n) {
  const float np = num_pos[0];
  const float k = -wscale / (np > 0.f ? np : 1.f);
  GRID_STRIDE(i, n) {
    const float s = 1.f / (1.f + expf(-pred[i]));
    const bool clamped = s < 1e-4f || s > 1.f - 1e-4f;
    const float p = fminf(fmaxf(s, 1e-4f), 1.f - 1e-4f);
    const float t = tgt[i];
    float dLdp = 0.f;
    if (t == 1.f) {
      const float q = 1.f - p;
      dLdp = q * q / p - 2.f * q * logf(p);                       // d/dp [log p (1-p)^2]
    } else if (t < 1.f) {
      const float q = 1.f - t, q2 = q * q;
      dLdp = (2.f * p * logf(1.f - p) - p * p / (1.f - p)) * (q2 * q2);   // d/dp [log(1-p) p^2] (1-t)^4
    }
    dpred[i] = clamped ? 0.f : k * dLdp * s * (1.f - s);
  }
}
// gather-L1: d/dpred[b][c][ind] += w * sign(pred - tgt) * mask / (C*sum(mask) + 1e-4)
__global__ __launch_bounds__(256) void reg_l1_bwd(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                   const long long* __restrict__ ind, const unsigned char* __restrict__ mask,
                                                   const float* __restrict__ msum, float* __restrict__ dpred, int B, int K,
                                                   int C, int HW, float w) {
  const float denom = msum[0] * (float)C + 1e-4f;
  const int total = B * K * C;
  GRID_STRIDE(i, total) {
    const int ch = (int)(i % C), bk = (int)(i / C), b = bk / K;
    if (!mask[bk]) continue;
    const size_t o = ((size_t)b * C + ch) * HW + ind[bk];
    const float d = pred[o] - tgt[i];
    const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    atomicAdd(&dpred[o], w * sg / denom);
  }
}
__global__ __launch_bounds__(256) void mask_sum(const unsigned char* __restrict__ mask, const float* __restrict__ tgt_heat,
                                                 float* __restrict__ out, int BK, long long nheat) {
  __shared__ float red[8];
  float m = 0.f, np = 0.f;
  for (int i = threadIdx.x; i < BK; i += 256) m += mask[i] ? 1.f : 0.f;
  for (long long i = threadIdx.x; i < nheat; i += 256) np += tgt_heat[i] == 1.f ? 1.f : 0.f;
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) { m += __shfl_xor(m, s); np += __shfl_xor(np, s); }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = m; red[4 + (threadIdx.x >> 6)] = np; }
  __syncthreads();
  if (threadIdx.x == 0) { out[0] = red[0] + red[1] + red[2] + red[3]; out[1] = red[4] + red[5] + red[6] + red[7]; }
}

// ---- stem im2col (for the weight gradient of the 7x7 stem through the generic wgrad GEMM) ------------------------------
// col[m][k], k = c*49 + kh*7 + kw for k < 147, zero for 147 <= k < 160
__global__ __launch_bounds__(256) void stem_im2col(const float* __restrict__ x, float* __restrict__ col, int N, int H, int W,
                                                    int Ho, int Wo) {
  const long long total = (long long)N * Ho * Wo * 160;
  GRID_STRIDE(i, total) {
    const int k = (int)(i % 160);
    long long m = i / 160;
    const int ow = (int)(m % Wo);
    m /= Wo;
    const int oh = (int)(m % Ho), n = (int)(m / Ho);
    float v = 0.f;
    if (k < 147) {
      const int c = k / 49, r = k - c * 49, kh = r / 7, kw = r - kh * 7;
      const int ih = 2 * oh - 3 + kh, iw = 2 * ow - 3 + kw;
      if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) v = x[((size_t)(n * 3 + c) * H + ih) * W + iw];
    }
    col[i] = v;
  }
}

// ---- PointNet conv1 (K <= 16) weight gradient: dw[n][k] = sum_m dy[m][n] x[m][k] (block partials + atomics) -------------
__global__ __launch_bounds__(256) void smallk_wgrad(const float* __restrict__ dy, const float* __restrict__ x,
                                                     float* __restrict__ dw, int M, int K, int Cout) {
  extern __shared__ float acc[];                         // [Cout][K]
  for (int i = threadIdx.x; i < Cout * K; i += 256) acc[i] = 0.f;
  __syncthreads();
  const int n = threadIdx.x % Cout, lanes = 256 / Cout, rl = threadIdx.x / Cout;
  if (rl < lanes) {
    float a[16];
    for (int k = 0; k < 16; ++k) a[k] = 0.f;
    for (long long m = (long long)blockIdx.x * lanes + rl; m < M; m += (long long)gridDim.x * lanes) {
      const float g = dy[(size_t)m * Cout + n];
      for (int k = 0; k < K; ++k) a[k] = fmaf(g, x[(size_t)m * K + k], a[k]);
    }
    for (int k = 0; k < K; ++k) atomicAdd(&acc[n * K + k], a[k]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Cout * K; i += 256) atomicAdd(&dw[i], acc[i]);
}

// ---- optimiser: sum of squares (for clip_grad_norm_), scale, AdamW -------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_partials(const float* __restrict__ g, double* __restrict__ part, long long n) {
  __shared__ double red[4];
  double s = 0;
  GRID_STRIDE(i, n) { const double v = g[i]; s += v * v; }
#pragma unroll
  for (int sh = 32; sh >= 1; sh >>= 1) s += __shfl_xor(s, sh);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// total_norm = sqrt(sum of partials); clip_coef = min(1, max_norm / (total_norm + 1e-6))  (torch.nn.utils.clip_grad_norm_)
__global__ void norm_finalize(const double* __restrict__ part, int G, float max_norm, float* __restrict__ out) {
  double s = 0;
  for (int g = 0; g < G; ++g) s += part[g];
  const float norm = (float)sqrt(s);
  out[0] = norm;
  const float coef = max_norm / (norm + 1e-6f);
  out[1] = coef < 1.f ? coef : 1.f;
}
// torch.optim.AdamW (amsgrad off, maximize off): p *= 1 - lr*wd; m,v update; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adamw_step(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, const float* __restrict__ clip, long long n,
                                                   float lr, float b1, float b2, float eps, float wd, float bc1,
                                                   float bc2_sqrt) {
  const float coef = clip ? clip[1] : 1.f;
  GRID_STRIDE(i, n) {
    const float gi = g[i] * coef;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);             // lerp_, as torch's single-tensor path
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

}  // namespace

#define ST static_cast<hipStream_t>(stream)

extern "C" int bevf_add_inplace_f32(float* y, const float* x, size_t n, void* stream) {
  BEVF_REQUIRE(y && x && n % 4 == 0 && bevf_aligned16(y) && bevf_aligned16(x), "add: needs aligned buffers, n %% 4 == 0");
  if (n) hipLaunchKernelGGL(add_inplace, dim3(ew_grid(n / 4)), dim3(256), 0, ST, y, x, (long long)(n / 4));
  return bevf_check_launch("bevf_add_inplace_f32");
}
extern "C" int bevf_relu_mask_f32(float* dy, const float* y, size_t n, void* stream) {
  BEVF_REQUIRE(dy && y && n % 4 == 0 && bevf_aligned16(dy) && bevf_aligned16(y), "relu_mask: needs aligned buffers, n %% 4 == 0");
  if (n) hipLaunchKernelGGL(relu_mask, dim3(ew_grid(n / 4)), dim3(256), 0, ST, dy, y, (long long)(n / 4));
  return bevf_check_launch("bevf_relu_mask_f32");
}
extern "C" int bevf_maxpool3x3s2_idx_f32(const float* x, float* y, uint8_t* idx, int N, int H, int W, int C, void* stream) {
  BEVF_REQUIRE(x && y && idx && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool_idx: bad arguments (C %% 4)");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(maxpool_fwd_idx, dim3(ew_grid((long long)N * Ho * Wo * (C / 4))), dim3(256), 0, ST, x, y, idx, N, H, W, C, Ho, Wo);
  return bevf_check_launch("bevf_maxpool3x3s2_idx_f32");
}
extern "C" int bevf_bn_relu_maxpool3x3s2_idx_f32(const float* x, const float* mean, const float* invstd, const float* gamma,
                                                 const float* beta, float* y, uint8_t* idx, int N, int H, int W, int C,
                                                 void* stream) {
  BEVF_REQUIRE(x && mean && invstd && y && idx && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0,
               "bn_relu_maxpool_idx: bad arguments (C %% 4)");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(bn_relu_maxpool_idx, dim3(ew_grid((long long)N * Ho * Wo * (C / 4))), dim3(256), 0, ST, x, mean, invstd, gamma, beta,
                     y, idx, N, H, W, C, Ho, Wo);
  return bevf_check_launch("bevf_bn_relu_maxpool3x3s2_idx_f32");
}
extern "C" int bevf_maxpool3x3s2_bwd_f32(const float* dy, const uint8_t* idx, float* dx, int N, int H, int W, int C,
                                         void* stream) {
  BEVF_REQUIRE(dy && idx && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool_bwd: bad arguments (C %% 4)");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(maxpool_bwd, dim3(ew_grid((long long)N * H * W * (C / 4))), dim3(256), 0, ST, dy, idx, dx, N, H, W, C, Ho, Wo);
  return bevf_check_launch("bevf_maxpool3x3s2_bwd_f32");
}
extern "C" int bevf_bilinear_bwd_nhwc_f32(const float* dy, float* dx, int B, int Hi, int Wi, int C, int x_cs, int Ho,
                                          int Wo, int y_cs, void* stream) {
  BEVF_REQUIRE(dy && dx && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && x_cs >= C && y_cs >= C, "bilinear_bwd: bad arguments");
  const long long total = (long long)B * Ho * Wo * C;
  hipLaunchKernelGGL(bilinear_bwd, dim3(ew_grid(total)), dim3(256), 0, ST, dy, dx, Hi, Wi, C, x_cs, Ho, Wo, y_cs,
                     (float)Hi / (float)Ho, (float)Wi / (float)Wo, total);
  return bevf_check_launch("bevf_bilinear_bwd_nhwc_f32");
}
extern "C" int bevf_cam_mean_bwd_f32(const float* dy, float* dx, int B, int ncam, int P, int C, void* stream) {
  BEVF_REQUIRE(dy && dx && B > 0 && ncam > 0 && P > 0 && C > 0 && C % 4 == 0, "cam_mean_bwd: bad arguments");
  const long long pc4 = (long long)P * C / 4, total = pc4 * B * ncam;
  hipLaunchKernelGGL(cam_mean_bwd, dim3(ew_grid(total)), dim3(256), 0, ST, reinterpret_cast<const f32x4*>(dy),
                     reinterpret_cast<f32x4*>(dx), ncam, pc4, total);
  return bevf_check_launch("bevf_cam_mean_bwd_f32");
}
extern "C" size_t bevf_group_max_idx_work_bytes(int G, int P, int C) {
  return (size_t)G * ((P + GM_CHUNK - 1) / GM_CHUNK) * C * 8;
}
extern "C" int bevf_group_max_idx_f32(const float* x, float* y, int32_t* idx, void* work, int G, int P, int C,
                                      void* stream) {
  BEVF_REQUIRE(x && y && idx && work && G > 0 && P > 0 && C > 0 && C % 4 == 0, "group_max_idx: bad arguments");
  BEVF_REQUIRE(bevf_aligned16(x), "group_max_idx: unaligned");
  const int nchunks = (P + GM_CHUNK - 1) / GM_CHUNK;
  float* pmax = static_cast<float*>(work);
  int* pidx = reinterpret_cast<int*>(pmax + (size_t)G * nchunks * C);
  hipLaunchKernelGGL(group_max_partial<false>, dim3(G * nchunks), dim3(256), 0, ST, x, pmax, pidx, P, C, nchunks,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr);
  const long long total = (long long)G * C;
  hipLaunchKernelGGL(group_max_final, dim3(ew_grid(total)), dim3(256), 0, ST, pmax, pidx, y, idx, C, nchunks, total);
  return bevf_check_launch("bevf_group_max_idx_f32");
}
// max and argmax over the P rows of each group of relu(batchnorm(x)) computed on the fly from the raw rows (training
// forward of PointNet's last layer: the activation is neither written nor re-read)
extern "C" int bevf_bn_relu_group_max_idx_f32(const float* x, const float* mean, const float* invstd, const float* gamma,
                                              const float* beta, float* y, int32_t* idx, void* work, int G, int P, int C,
                                              void* stream) {
  BEVF_REQUIRE(x && mean && invstd && y && idx && work && G > 0 && P > 0 && C > 0 && C % 4 == 0, "bn_relu_group_max_idx: bad arguments");
  BEVF_REQUIRE(bevf_aligned16(x), "bn_relu_group_max_idx: unaligned");
  const int nchunks = (P + GM_CHUNK - 1) / GM_CHUNK;
  float* pmax = static_cast<float*>(work);
  int* pidx = reinterpret_cast<int*>(pmax + (size_t)G * nchunks * C);
  hipLaunchKernelGGL(group_max_partial<true>, dim3(G * nchunks), dim3(256), 0, ST, x, pmax, pidx, P, C, nchunks, mean, invstd,
                     gamma, beta);
  const long long total = (long long)G * C;
  hipLaunchKernelGGL(group_max_final, dim3(ew_grid(total)), dim3(256), 0, ST, pmax, pidx, y, idx, C, nchunks, total);
  return bevf_check_launch("bevf_bn_relu_group_max_idx_f32");
}
extern "C" int bevf_group_max_bwd_f32(const float* dy, const int32_t* idx, float* dx, int G, int P, int C, void* stream) {
  BEVF_REQUIRE(dy && idx && dx && G > 0 && P > 0 && C > 0, "group_max_bwd: bad arguments");
  const long long total = (long long)G * C;
  hipLaunchKernelGGL(group_max_bwd, dim3(ew_grid(total)), dim3(256), 0, ST, dy, idx, dx, P, C, total);
  return bevf_check_launch("bevf_group_max_bwd_f32");
}
extern "C" int bevf_sparse_rows_wgrad_f32(const float* S, const int32_t* idx, const float* A, float* out, int G, int P, int C, int K,
                                          void* stream) {
  BEVF_REQUIRE(S && idx && A && out && G > 0 && P > 0 && C > 0 && K > 0, "sparse_rows_wgrad: bad arguments");
  hipLaunchKernelGGL(sparse_rows_wgrad, dim3((unsigned)C), dim3(256), 0, ST, S, idx, A, out, G, P, C, K);
  return bevf_check_launch("bevf_sparse_rows_wgrad_f32");
}
extern "C" int bevf_sparse_rows_scatter_add_f32(const float* S, const int32_t* idx, const float* W, float* dA, int G, int P, int C, int K,
                                                void* stream) {
  BEVF_REQUIRE(S && idx && W && dA && G > 0 && G < 65536 && P > 0 && C > 0 && C <= 16384 && K > 0, "sparse_rows_scatter_add: bad arguments");
  hipLaunchKernelGGL(sparse_rows_scatter, dim3((unsigned)C, (unsigned)G), dim3(256), (size_t)C * sizeof(int), ST, S, idx, W, dA, P, C, K);
  return bevf_check_launch("bevf_sparse_rows_scatter_add_f32");
}
extern "C" int bevf_zero_stuff_nhwc_f32(const float* dy, float* out, int N, int Ho, int Wo, int C, int H, int W, int s,
                                        void* stream) {
  BEVF_REQUIRE(dy && out && N > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 4 == 0 && H > 0 && W > 0 && s > 0, "zero_stuff: bad arguments");
  hipLaunchKernelGGL(zero_stuff, dim3(ew_grid((long long)N * H * W * (C / 4))), dim3(256), 0, ST, dy, out, N, Ho, Wo, C, H, W, s);
  return bevf_check_launch("bevf_zero_stuff_nhwc_f32");
}
extern "C" int bevf_interleave2x2_nhwc_f32(const float* const* cls4, const int32_t* hq4, const int32_t* wq4, float* dx, int N,
                                           int H, int W, int C, void* stream) {
  BEVF_REQUIRE(cls4 && hq4 && wq4 && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "interleave2x2: bad arguments");
  Interleave4 a;
  for (int q = 0; q < 4; ++q) {
    a.cls[q] = cls4[q]; a.hq[q] = hq4[q]; a.wq[q] = wq4[q];
    const int need_h = (H - (q >> 1) + 1) / 2, need_w = (W - (q & 1) + 1) / 2;
    BEVF_REQUIRE(!cls4[q] || (hq4[q] >= need_h && wq4[q] >= need_w), "interleave2x2: class %d is %dx%d, needs %dx%d", q, hq4[q],
                 wq4[q], need_h, need_w);
  }
  hipLaunchKernelGGL(interleave2x2, dim3(ew_grid((long long)N * H * W * (C / 4))), dim3(256), 0, ST, a, dx, N, H, W, C);
  return bevf_check_launch("bevf_interleave2x2_nhwc_f32");
}
constexpr int kLinSlices = 32;
static inline void linear_bwd_geometry(int K, int O, int* G, int* chunk, int* lanes) {
  const int k4 = K >> 2, per = k4 < 256 ? k4 : 256;
  *lanes = 256 / per;
  int c = (O + 1023) / 1024;                               // ~1000 workgroups stream W, at least 8 rows each
  if (c < 8) c = 8;
  if (c > 128) c = 128;
  *chunk = c;
  *G = (O + c - 1) / c;
}
extern "C" size_t bevf_linear_bwd_work_floats(int B, int K, int O) {
  int G, chunk, lanes;
  linear_bwd_geometry(K, O, &G, &chunk, &lanes);
  return (size_t)G * lanes * B * K + (size_t)kLinSlices * B * K;
}
extern "C" int bevf_linear_bwd_f32(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db,
                                   float* work, int B, int K, int O, int perm_inner, int perm_outer, void* stream) {
  BEVF_REQUIRE(dy && x && w && dw && work && B > 0 && B <= 64 && K > 0 && K % 4 == 0 && O > 0, "linear_bwd: bad arguments (B <= 64)");
  int G, chunk, lanes;
  linear_bwd_geometry(K, O, &G, &chunk, &lanes);
  if (dx) {
    float* part2 = work + (size_t)G * lanes * B * K;
    const unsigned gx = (unsigned)((B * K + 255) / 256);
    hipLaunchKernelGGL(linear_bwd_dx_partials, dim3(G), dim3(256), (size_t)chunk * B * sizeof(float), ST, dy, w, work, B, K, O, chunk, perm_inner, perm_outer);
    hipLaunchKernelGGL(linear_bwd_dx_final, dim3(gx, kLinSlices), dim3(256), 0, ST, work, part2, B * K, G * lanes);
    hipLaunchKernelGGL(linear_bwd_dx_final, dim3(gx, 1), dim3(256), 0, ST, part2, dx, B * K, kLinSlices);
  }
  hipLaunchKernelGGL(linear_bwd_dw, dim3(ew_grid((long long)O * (K / 4))), dim3(256), 0, ST, dy, x, dw, db, B, K, O, perm_inner, perm_outer);
  return bevf_check_launch("bevf_linear_bwd_f32");
}
extern "C" int bevf_head_tail_bwd_f32(const bevf_head_bwd_desc* d, void* stream) {
  BEVF_REQUIRE(d && d->hid && d->w && d->out0 && d->dhid && d->dw && d->db, "head_tail_bwd: null pointer");
  BEVF_REQUIRE(d->B > 0 && d->P > 0 && d->hc > 0, "head_tail_bwd: empty shape");
  for (int k = 0; k < 5; ++k) BEVF_REQUIRE(d->c[k] >= 0 && d->c[k] <= 16, "head_tail_bwd: at most 16 outputs per branch");
  HeadBwdArgs a;
  a.hid = d->hid; a.w = d->w; a.out0 = d->out0; a.dhid = d->dhid; a.dw = d->dw; a.db = d->db;
  a.B = d->B; a.P = d->P; a.hc = d->hc; a.n_sigmoid = d->n_sigmoid;
  int ctot = 0;
  for (int k = 0; k < 5; ++k) {
    BEVF_REQUIRE(d->dout[k], "head_tail_bwd: missing gradient of branch %d", k);
    a.dout[k] = d->dout[k]; a.c[k] = d->c[k]; ctot += d->c[k];
  }
  const size_t lds = (size_t)(2 * ctot * d->hc + ctot) * sizeof(float);
  BEVF_REQUIRE(lds <= 64 * 1024, "head_tail_bwd: LDS");
  const long long total = (long long)d->B * d->P;
  unsigned grid = (unsigned)((total + 255) / 256 > 512 ? 512 : (total + 255) / 256);
  const size_t lds_tiled = lds + (size_t)(256 * (ctot | 1) + 256 * d->hc) * sizeof(float);
  if (lds_tiled <= 160 * 1024) {                         // tile of 256 pixels fits: the shuffle-free kernel
    static bool attr_done = false;
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&head_tail_bwd_tiled), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_done = true;
    }
    hipLaunchKernelGGL(head_tail_bwd_tiled, dim3(grid), dim3(256), lds_tiled, ST, a);
    return bevf_check_launch("bevf_head_tail_bwd_f32");
  }
  hipLaunchKernelGGL(head_tail_bwd, dim3(grid), dim3(256), lds, ST, a);
  return bevf_check_launch("bevf_head_tail_bwd_f32");
}
extern "C" int bevf_centernet_loss_bwd_f32(const bevf_loss_desc* d, float* const dpred[5], float* scratch2, void* stream) {
  BEVF_REQUIRE(d && d->pred_heatmap && d->tgt_heatmap && d->ind && d->reg_mask && dpred && scratch2, "loss_bwd: null pointer");
  const long long n = (long long)d->B * d->C * d->H * d->W;
  hipLaunchKernelGGL(mask_sum, dim3(1), dim3(256), 0, ST, d->reg_mask, d->tgt_heatmap, scratch2, d->B * d->K, n);
  hipLaunchKernelGGL(focal_bwd, dim3(ew_grid(n)), dim3(256), 0, ST, d->pred_heatmap, d->tgt_heatmap, scratch2 + 1, dpred[0],
                     d->weights[0], n);
  const int cs[4] = {2, 3, 2, 2};
  for (int q = 0; q < 4; ++q) {
    BEVF_REQUIRE(dpred[1 + q] && d->pred_reg[q] && d->tgt_reg[q], "loss_bwd: regression branch %d missing", q);
    hipLaunchKernelGGL(reg_l1_bwd, dim3(ew_grid((long long)d->B * d->K * cs[q])), dim3(256), 0, ST, d->pred_reg[q],
                       d->tgt_reg[q], (const long long*)d->ind, d->reg_mask, scratch2, dpred[1 + q], d->B, d->K, cs[q],
                       d->H * d->W, d->weights[1 + q]);
  }
  return bevf_check_launch("bevf_centernet_loss_bwd_f32");
}
extern "C" int bevf_stem_im2col_f32(const float* x, float* col, int N, int H, int W, void* stream) {
  BEVF_REQUIRE(x && col && N > 0 && H > 0 && W > 0, "stem_im2col: bad arguments");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(stem_im2col, dim3(ew_grid((long long)N * Ho * Wo * 160)), dim3(256), 0, ST, x, col, N, H, W, Ho, Wo);
  return bevf_check_launch("bevf_stem_im2col_f32");
}
extern "C" int bevf_smallk_wgrad_f32(const float* dy, const float* x, float* dw, int M, int K, int Cout, void* stream) {
  BEVF_REQUIRE(dy && x && dw && M > 0 && K > 0 && K <= 16 && Cout > 0 && Cout <= 256 && 256 % Cout == 0, "smallk_wgrad: K <= 16, Cout | 256");
  const int lanes = 256 / Cout;
  int G = (M + lanes - 1) / lanes;
  if (G > 512) G = 512;
  hipLaunchKernelGGL(smallk_wgrad, dim3(G), dim3(256), (size_t)Cout * K * sizeof(float), ST, dy, x, dw, M, K, Cout);
  return bevf_check_launch("bevf_smallk_wgrad_f32");
}
extern "C" int bevf_grad_norm_f32(const float* g, size_t n, double* work512, float max_norm, float* out2, void* stream) {
  BEVF_REQUIRE(g && work512 && out2 && n > 0, "grad_norm: bad arguments");
  const unsigned G = ew_grid((long long)n) > 512 ? 512 : ew_grid((long long)n);
  hipLaunchKernelGGL(sumsq_partials, dim3(G), dim3(256), 0, ST, g, work512, (long long)n);
  hipLaunchKernelGGL(norm_finalize, dim3(1), dim3(1), 0, ST, work512, (int)G, max_norm, out2);
  return bevf_check_launch("bevf_grad_norm_f32");
}
extern "C" int bevf_adamw_step_f32(float* p, const float* g, float* m, float* v, const float* clip2, size_t n, float lr,
                                   float beta1, float beta2, float eps, float weight_decay, int step, void* stream) {
  BEVF_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adamw: bad arguments");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);   // python floats in torch
  hipLaunchKernelGGL(adamw_step, dim3(ew_grid((long long)n)), dim3(256), 0, ST, p, g, m, v, clip2, (long long)n, lr, beta1,
                     beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2));
  return bevf_check_launch("bevf_adamw_step_f32");
}
