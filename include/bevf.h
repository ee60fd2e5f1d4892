/*
 * bevf.h -- C-ABI of libbevf_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * BEV-fusion detector hot path of meg89/bevfusion_multimodal_3d_object_detection.
 *
 * The reference has no FFI of its own: its boundary is the Python module API
 * of src/encoders.py, src/fusion.py, src/fusion_detection.py and src/centernet_target.py.  Each
 * entry point below names the reference lines whose arithmetic it replaces; the Python host
 * in bevfusion_multimodal_3d_object_detection_amd/ binds them with ctypes and keeps the
 * reference's class names, constructor signatures and state-dict keys.
 *
 * Conventions: plain device pointers and sizes, no torch types; every call is asynchronous on
 * `stream` (a hipStream_t passed as void*), never allocates, never synchronises, is re-entrant
 * and graph-capturable; returns 0 or a negative bevf_status, message via bevf_last_error().
 * Activations are fp32 NHWC ("pixel-major": [image][row][col][channel]) with an explicit
 * per-pixel channel stride so producers write straight into channel slices of a concat buffer.
 */
#ifndef BEVF_H
#define BEVF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  BEVF_OK = 0,
  BEVF_ERR_ARG = -1,      /* shape / alignment / null-pointer contract violated */
  BEVF_ERR_LAUNCH = -2,   /* hipLaunchKernel reported an error */
  BEVF_ERR_UNSUPPORTED = -3
} bevf_status;

int bevf_version(void);
const char* bevf_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM 2-D convolution on v_mfma_f32_32x32x2_f32 (exact fp32), fused epilogue
 *     y = act( conv(x, w) * scale + shift (+ res) )
 * Replaces every nn.Conv2d(+BatchNorm2d eval)(+ReLU)(+residual add) on the path:
 *   ResNet-18 layer1..3 + channel_proj   ref src/encoders.py:159-165 (torchvision BasicBlock)
 *   camera_proj / lidar_upsample / radar_refine / bev_fusion   ref src/fusion.py:126-207,239-295
 *   CenterNet 3x3 branches (5 fused into one Cout=320 conv)    ref src/fusion.py:822-854
 * and, with KH=KW=1, the shared per-point MLP layers (nn.Conv1d k=1 + BatchNorm1d + ReLU)
 *   PointNetLiDAREncoder conv2..conv5   ref src/encoders.py:290-295
 * With `colmax` set the output is not stored; instead the per-group column maximum
 * (torch.max(x, 2)[0], ref src/encoders.py:298) is accumulated with integer atomics on the
 * non-negative post-ReLU bit patterns (order-independent, hence deterministic).
 * Requires Cin % 32 == 0, x_cs % 4 == 0, 16-byte aligned x / w, and x / w below 2 GiB each (32-bit buffer
 * offsets; callers chunk larger batches over images).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const float* x;      /* [N][H][W][x_cs], first Cin channels of each pixel are read        */
  const float* w;      /* [Cout][KH][KW][Cin]  (OHWI)                                        */
  const float* scale;  /* [Cout] or NULL (1)   folded BN: gamma / sqrt(var + eps)            */
  const float* shift;  /* [Cout] or NULL (0)   beta - mean*scale + conv_bias*scale           */
  const float* res;    /* optional residual [N*Ho*Wo][res_cs], added before the activation   */
  float* y;            /* [N*Ho*Wo][y_cs] (may be NULL when colmax is set)                   */
  uint32_t* colmax;    /* optional [ceil(M/rows_per_group)][Cout], caller zeroes it          */
  int32_t N, H, W, Cin, x_cs;
  int32_t Ho, Wo, Cout, y_cs, res_cs;
  int32_t KH, KW, stride, pad;
  int32_t relu;            /* 0: none, 1: ReLU                                               */
  int32_t rows_per_group;  /* colmax grouping (points per batch element)                     */
  int32_t tile;            /* 0: auto (cost model); 1 128x128, 2 256x64, 3 128x64, 4 64x64, 5-7 hybrid
                              big + 64x64 tail of 1 / 2 / 3 -- forced variants for bench / tests     */
  /* bevf_conv3x3_wino_f32 only (training forward, ref train-mode nn.BatchNorm2d after the conv): when `stats` is set the
   * epilogue also leaves per-tile-block partial sums of the stored outputs, {sum(y - pivot), sum((y - pivot)^2)}
   * per channel, as rows [bevf_wino_stat_rows(N,H,W)][Cout][2] for bevf_bn_stats_from_partials_f32 -- the BatchNorm
   * batch statistics without re-reading the activation.  Needs relu = 0 and res = NULL. */
  float* stats;
  const float* stats_pivot; /* [Cout] any value near the channel mean (e.g. the running mean); NULL = 0 */
  /* bevf_conv3x3_wino_f32 only (training backward): when `bnb_x` is set, this convolution's output (+ res) is the gradient
   * dY reaching a train-mode BatchNorm(+ReLU) layer, and the epilogue does that layer's first backward pass on the way out
   * (what bevf_bn_backward_f32 would start with): dY is stored with the layer's ReLU mask applied -- mask from its output
   * `bnb_y`, or, if NULL (the layer had no residual input), recomputed from its raw input `bnb_x` exactly as the forward did --
   * and `stats` receives {sum dY, sum dY * xhat} per (tile block, channel) for bevf_bn_backward_from_partials_f32.
   * bnb_x / bnb_y: [N*H*W][Cout]; bnb_mean, bnb_invstd (required), bnb_gamma, bnb_beta (NULL = 1 / 0): [Cout].
   * Needs relu = 0, stats_pivot = NULL, y_cs == Cout. */
  const float* bnb_x;
  const float* bnb_y;
  const float* bnb_mean;
  const float* bnb_invstd;
  const float* bnb_gamma;
  const float* bnb_beta;
} bevf_conv_desc;
int bevf_conv2d_nhwc_f32(const bevf_conv_desc* d, void* stream);

/* The same convolution for the 3x3 / stride 1 / pad 1 layers (ResNet blocks, camera_proj, lidar_upsample, radar_refine,
 * bev_fusion, the fused CenterNet 3x3: ~90 % of the path's FLOPs) as fp32 Winograd F(2x2,3x3), fully fused, on
 * v_mfma_f32_16x16x4_f32: 16 multiplies per 2x2 outputs instead of 36.  fp32 products and fp32 accumulation as in the
 * direct kernel, a different summation structure: agrees with it to a few 1e-7 relative, not bit for bit.
 * `d->w` is the transformed-filter image made by bevf_wino_filter_transform_f32 from the OHWI filter
 * (bevf_wino_filter_floats(Cout, Cin) floats); everything else in the descriptor means what it means above
 * (colmax unsupported, `tile` ignored).  Requires KH = KW = 3, stride 1, pad 1, Cin % 32 == 0. */
size_t bevf_wino_filter_floats(int Cout, int Cin);
int bevf_wino_filter_transform_f32(const float* w_ohwi, float* u, int Cout, int Cin, void* stream);
int bevf_conv3x3_wino_f32(const bevf_conv_desc* d, void* stream);
int bevf_wino_stat_rows(int N, int H, int W);   /* rows of the `stats` partial buffer */

/* ResNet stem: 7x7 stride-2 pad-3 conv on a planar 3-channel image + BN (+ ReLU when relu != 0),
 * ref src/encoders.py:154-156 (torchvision conv1/bn1/relu).  x: [N][3][H][W] (NCHW, as the
 * reference's callers hand it over), w: the filter bank packed k-major [148][64] with
 * k = c*49 + kh*7 + kw and a zero row k = 147 (packed once per weight update by the host),
 * y: [N][Ho][Wo][64] NHWC.                                                                  */
int bevf_stem_conv7x7_f32(const float* x, const float* w, const float* scale, const float* shift,
                          float* y, int N, int H, int W, int relu, void* stream);
/* conv1 + bn1 + relu + maxpool(3, stride 2, pad 1) of ref src/encoders.py:154-157 in ONE kernel (inference): y is the pooled
 * NHWC map [N][Hp][Wp][64]; the stem map itself (the largest activation of the path) never reaches HBM.  Bit-identical to
 * bevf_stem_conv7x7_f32(relu = 1) followed by bevf_maxpool3x3s2_nhwc_f32. */
int bevf_stem_pool_f32(const float* x, const float* w_packed, const float* scale, const float* shift, float* y, int N,
                       int H, int W, void* stream);

/* 3x3 stride-2 pad-1 max-pool, NHWC, ref src/encoders.py:157.                               */
int bevf_maxpool3x3s2_nhwc_f32(const float* x, float* y, int N, int H, int W, int C, void* stream);

/* Small-K pointwise layer y[m][n] = act((sum_k x[m][k] w[n][k]) * scale[n] + shift[n]), K <= 16:
 * PointNet conv1 (K = 4|5), ref src/encoders.py:289.                                         */
int bevf_pointwise_smallk_f32(const float* x, const float* w, const float* scale, const float* shift,
                              float* y, int M, int K, int Cout, int relu, void* stream);

/* PointNet conv1 -> conv2 -> conv3 (K -> 64 -> 128 -> 256, each with folded BatchNorm + ReLU; ref src/encoders.py:289-291:
 * `x = F.relu(self.bn1(self.conv1(x)))` ... `bn3(conv3(x))`) as ONE launch whose 64- and 128-wide activations stay in
 * registers.  x: [M][K] points (K <= 8), w1: [64][K], w2f / w3f: the [128][64] and [256][128] filters in MFMA fragment
 * order (bevf_pointnet_front_pack_f32), s* / b*: per-channel scale / shift, y: [M][256].                              */
int bevf_pointnet_front_f32(const float* x, int M, int K, const float* w1, const float* s1, const float* b1,
                            const float* w2f, const float* s2, const float* b2, const float* w3f, const float* s3,
                            const float* b3, float* y, void* stream);
/* w: [Cout][Cin] row-major (Conv1d k=1 weight) -> wf: [Cout/32][Cin/32][4][64][4] fragments for the kernel above.     */
int bevf_pointnet_front_pack_f32(const float* w, float* wf, int Cout, int Cin, void* stream);

/* y[g][c] = max over the P rows of group g: the per-voxel max of VFELayer ("pillar reduction"),
 * ref src/encoders.py:451-452.  x: [G][P][C] post-ReLU point features -> y: [G][C].           */
int bevf_group_max_f32(const float* x, float* y, int G, int P, int C, void* stream);
/* The whole VFELayer for K <= 16 input channels in one pass (ref src/encoders.py:431-455): y[g][n] = max over the P rows of
 * group g of relu((x[g][p][:] . w[n][:]) * scale[n] + shift[n]); bit-identical to bevf_pointwise_smallk_f32 followed by
 * bevf_group_max_f32, without the [G][P][Cout] intermediate.  x: [G][P][K], w: [Cout][K], y: [G][Cout].               */
int bevf_vfe_smallk_max_f32(const float* x, const float* w, const float* scale, const float* shift, float* y, int G, int P,
                            int K, int Cout, void* stream);

/* One radar sweep -> 256-d feature: 4 x (Conv1d k=1 + BN + ReLU) + max over points, all in
 * LDS, one workgroup per (radar, batch element, 32-point chunk); `out` must be ZERO-FILLED (the chunk maxima
 * are merged with integer atomicMax).  ref src/encoders.py:549-555, loop :642-644.
 * x: [R][B][P][Cin]; w_i packed k-major [c_{i-1}][c_i]; out: [B][R][c4] (== torch.stack(dim=1)). */
typedef struct {
  const float* x;
  const float* w[4];
  const float* scale[4];
  const float* shift[4];
  float* out;
  int32_t R, B, P, Cin;
  int32_t c[4];
} bevf_radar_desc;
int bevf_radar_mlp_max_f32(const bevf_radar_desc* d, void* stream);

/* Dense layer for small batch (weight-streaming GEMV): y[b][perm(o)] = act(W[o].x[b] + bias[o]).
 * lidar_init (ref src/fusion.py:144-148,258), radar_proj (:183-186,274), fusion_fc
 * (ref src/encoders.py:624,653).  perm(o) = (o % perm_inner) * perm_outer + o / perm_inner when
 * perm_inner > 0 (writes the (B,128,25,25) view of ref :259 directly as NHWC), else o.        */
int bevf_linear_f32(const float* x, const float* w, const float* bias, float* y, int B, int K, int O,
                    int relu, int perm_inner, int perm_outer, void* stream);

/* Camera "BEV pooling" part 1: mean over the cameras, ref src/fusion.py:233-234.
 * x: [B][ncam][P][C] -> y: [B][P][C]; sum in camera order, then true division by ncam.       */
int bevf_cam_mean_f32(const float* x, float* y, int B, int ncam, int P, int C, void* stream);

/* Camera "BEV pooling" part 2 / nn.Upsample: bilinear resample, align_corners=False,
 * ref src/fusion.py:242-247 and :156.  NHWC in (stride x_cs) -> NHWC out slice (stride y_cs). */
int bevf_bilinear_nhwc_f32(const float* x, float* y, int B, int Hi, int Wi, int C, int x_cs,
                           int Ho, int Wo, int y_cs, void* stream);

/* radar broadcast, ref src/fusion.py:277-278: y[b][p][0:C] = v[b][0:C] for p < P.            */
int bevf_broadcast_nhwc_f32(const float* v, float* y, int B, int P, int C, int y_cs, void* stream);

/* Radar branch shortcut (exact): a 3x3/pad-1 conv stack on a spatially constant image yields at most 5x5
 * distinct pixels after two layers (a pixel's value depends only on its border class per axis), so
 * radar_refine (ref src/fusion.py:189-196,281) runs on a 5x5 image and this expands the classes:
 * y[b][i][j][0:C] = small[b][cls(i)][cls(j)][0:C], cls(i) = i<2 ? i : (i>=S-2 ? 4-(S-1-i) : 2).        */
int bevf_expand_border_classes_f32(const float* small, float* y, int B, int Sh, int Sw, int C, int y_cs,
                                   void* stream);

/* CenterNet head tail: block-diagonal 1x1 convs of the five branches + sigmoid on the heatmap,
 * ref src/fusion.py:825,832,839,846,853,869-884.  hid: [B*P][5*hc] post-ReLU hidden maps;
 * w: concatenated [sum(c_k)][hc]; bias: [sum(c_k)]; outputs NCHW (B,c_k,H,W) as the reference
 * returns them.  n_sigmoid = number of leading output channels passed through sigmoid.       */
typedef struct {
  const float* hid;
  const float* w;
  const float* bias;
  float* out[5];
  int32_t B, P, hc;
  int32_t c[5];
  int32_t n_sigmoid;
} bevf_head_desc;
int bevf_head_tail_f32(const bevf_head_desc* d, void* stream);

/* Layout changes at the module-API boundary (the reference's tensors are NCHW).              */
int bevf_nchw_to_nhwc_f32(const float* x, float* y, int N, int C, int P, int y_cs, void* stream);
int bevf_nhwc_to_nchw_f32(const float* x, float* y, int N, int C, int P, int x_cs, void* stream);

/* (N,P,Cin) point rows -> NHWC rows of a padded width (zero fill), for Cin not a multiple of 4 */
int bevf_fill_f32(float* y, float v, size_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Post-processing, ref src/centernet_target.py:326-452 / src/fusion_detection.py:695-820:
 * 3x3 keep-mask (_nms), per-class top-K then top-K of the C*K pool (_topk), gather + box
 * assembly.  One workgroup per batch element; outputs are fixed-size [B][K] records plus a
 * per-frame count of entries with score > thresh (they are a prefix: scores are sorted).
 * Ties are broken by the lower flattened index, like a stable descending sort.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const float* heat;   /* (B,C,H,W) post-sigmoid */
  const float* offset; /* (B,2,H,W) */
  const float* size;   /* (B,3,H,W) */
  const float* rot;    /* (B,2,H,W) */
  const float* vel;    /* (B,2,H,W) */
  float* boxes;        /* [B][K][7]  x,y,z,w,l,h,yaw */
  float* scores;       /* [B][K]     descending */
  int64_t* labels;     /* [B][K]     */
  float* velocities;   /* [B][K][2]  */
  int32_t* count;      /* [B]        entries with score > thresh (a prefix) */
  void* work;          /* scratch of bevf_centernet_decode_work_bytes(B,C,H,W,K) bytes */
  int64_t* pool_ind;   /* [B][K] or NULL: position of each winner in the flattened (C,K) pool of per-class
                          winners -- the second return value of the reference's _topk (ref centernet_target.py:441) */
  int32_t B, C, H, W, K;
  int32_t true_labels; /* 0: reference behaviour (label is always 0, ref centernet_target.py:434);
                          1: opt-in fix, label = class of the winning heatmap plane */
  int32_t raw_scores;  /* 0: apply the 3x3 keep mask first (decode_centernet_predictions calls _nms, ref :352);
                          1: rank `heat` as given (the reference's bare _topk, ref :424-452) */
  float thresh, voxel, x_min, y_min;
} bevf_decode_desc;
size_t bevf_centernet_decode_work_bytes(int B, int C, int H, int W, int K);
int bevf_centernet_decode_f32(const bevf_decode_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training targets, ref src/centernet_target.py:170-324 (+ gaussian_radius :128-150, gaussian_2d :118-125,
 * draw_gaussian :152-168): one workgroup per frame.  boxes are float32 [B][nmax][9] (x,y,z,w,l,h,yaw,vx,vy;
 * the last two ignored unless has_vel[b]), labels int32 (-1 or >= C skips the object, like the padding of
 * ref src/train_detect.py).  Grid index / radius arithmetic is fp32 in the reference's own operation order
 * (bit-exact ind / mask / reg_mask); the gaussian is float64 rounded to fp32; dense maps keep the LAST object
 * of a cell, as the reference's sequential writes do.  All outputs must be zero-filled by the caller.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const float* boxes;
  const int32_t* labels;
  const int32_t* has_vel;        /* [B] */
  float* heatmap;                /* (B,C,H,W) */
  float* offset; float* size; float* rot; float* vel;      /* (B,2|3|2|2,H,W) */
  uint8_t* mask; int64_t* ind; uint8_t* reg_mask;          /* [B][max_objects] */
  float* target_offset; float* target_size; float* target_rot; float* target_vel;   /* [B][max_objects][2|3|2|2] */
  int32_t* owner_scratch;        /* [B][H*W] int32, zero-filled */
  int32_t B, nmax, H, W, C, max_objects, min_radius;
  float pc_range[6];
  float gaussian_overlap;
} bevf_targets_desc;
int bevf_centernet_targets_f32(const bevf_targets_desc* d, void* stream);

/* _nms of ref src/centernet_target.py:416-421: out = heat * (maxpool3x3(heat) == heat), planes = B*C.  */
int bevf_nms_keep_f32(const float* heat, float* out, int planes, int H, int W, void* stream);

/* CenterNetLoss.forward, ref src/centernet_target.py:476-622: penalty-reduced focal loss on
 * clamp(sigmoid(pred)) -- the reference applies sigmoid to the already-sigmoided head output and so does
 * this -- and mask-weighted gather-L1 for offset/size/rot/vel.  out[6] = total, heatmap, offset, size, rot,
 * vel.  Two launches (fixed-grid partial sums, then a fixed-order final sum): deterministic.            */
typedef struct {
  const float* pred_heatmap;     /* (B,C,H,W) */
  const float* tgt_heatmap;
  const float* pred_reg[4];      /* offset,size,rot,vel (B,c,H,W) */
  const float* tgt_reg[4];       /* target_offset,... [B][K][c] */
  const int64_t* ind;            /* [B][K] */
  const uint8_t* reg_mask;       /* [B][K] */
  float* work;                   /* bevf_centernet_loss_work_floats() floats */
  float* out;                    /* [6] */
  int32_t B, C, H, W, K;
  float weights[5];              /* heatmap, offset, size, rot, vel */
} bevf_loss_desc;
size_t bevf_centernet_loss_work_floats(void);
int bevf_centernet_loss_f32(const bevf_loss_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * Hard voxelisation: points -> the (voxel_features, voxel_coords) layout of
 * VFELayer / VoxelNetLiDAREncoder (ref src/encoders.py:313-321, :385-387).  The reference contains no voxel
 * assignment, so the semantics are the usual deterministic ones: cell = floor((p - range_min) / voxel_size)
 * in fp32, out-of-grid points dropped, voxels numbered by first point, first max_points points per voxel in
 * input order, zero padding.  Outputs must be zero-filled by the caller.  voxel_coords are (z, y, x) = the
 * (D, H, W) index order of ref :399-410.  Grid dims = round((max - min) / voxel_size).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const float* points;       /* [B][N][C], x y z first */
  float* voxel_features;     /* [B][max_voxels][max_points][C] */
  int64_t* voxel_coords;     /* [B][max_voxels][3] */
  int32_t* num_points;       /* [B][max_voxels] */
  int32_t* num_voxels;       /* [B] */
  void* work;                /* bevf_voxelize_work_bytes(B, N) bytes */
  int32_t B, N, C, max_points, max_voxels;
  float pc_range[6];
  float voxel_size[3];
} bevf_voxelize_desc;
size_t bevf_voxelize_work_bytes(int B, int N);
int bevf_voxelize_f32(const bevf_voxelize_desc* d, void* stream);

/* Dense scatter of per-voxel features [B][Nv][C] into out [B][C][D][H][W] at voxel_coords (z,y,x) -- the pillar -> BEV
 * canvas step, ref src/encoders.py:399-410 (`feature_grid[b, :, c0, c1, c2] = features.T`, rows in order: when several
 * rows name one cell the LAST one wins; cells no row names are zero).  num_voxels [B] or NULL: with it only rows
 * v < num_voxels[b] take part (the padding rows of bevf_voxelize_f32 stay out); NULL = every row, exactly like the
 * reference.  owner: scratch of B*D*H*W int32.  Rows with coordinates outside the grid are ignored. */
int bevf_scatter_voxels_f32(const float* features, const int64_t* coords, const int32_t* num_voxels, int32_t* owner,
                            float* out, int B, int Nv, int C, int D, int H, int W, void* stream);

/* ==========================================================================================
 * bf16 storage, fp32 accumulate (BASELINE configs 3 and 5).  Same layouts and geometry as the fp32 entry
 * points, element type bfloat16 wherever a pointer is typed void*: the convolution runs on
 * v_mfma_f32_32x32x16_bf16 (a K step is 64 bf16 = the same 128-byte rows; needs Cin % 64 == 0), BN
 * scale/shift, biases, the small per-frame vectors (PointNet / radar features) and the head outputs stay
 * fp32.  bevf_conv2d_nhwc_bf16 takes the same descriptor with x / w / res / y pointing at bf16 data.
 * ========================================================================================== */
int bevf_conv2d_nhwc_bf16(const bevf_conv_desc* d, void* stream);

/* The 3x3 / stride 1 / pad 1 layers of the bf16 path (ResNet blocks ref src/encoders.py:158-160 via torchvision's BasicBlock,
 * camera_proj / lidar_upsample / bev_fusion ref src/fusion.py:141-207, the CenterNet 3x3 convs ref src/fusion.py:822-854) as a
 * direct convolution on v_mfma_f32_16x16x32_bf16 that stages each 18x18-pixel input patch ONCE per 32-channel chunk (LDS-DMA)
 * and reads all nine taps from it; same descriptor as bevf_conv2d_nhwc_bf16 (colmax / stats unsupported, `tile` ignored).
 * `d->w` = the filter image made by bevf_conv3x3_pack_bf16 from the OHWI bf16 filter (bevf_conv3x3_pack_elems(Cout, Cin)
 * bf16 elements).  Requires KH = KW = 3, stride 1, pad 1, Cin % 32 == 0, Cout % 64 == 0, x_cs % 8 == 0, y_cs % 4 == 0. */
size_t bevf_conv3x3_pack_elems(int Cout, int Cin);
int bevf_conv3x3_bf16_ct(int Cout);   /* output-channel tile (64 | 128) the kernel uses for this layer */
int bevf_conv3x3_pack_bf16(const void* w_ohwi, void* packed, int Cout, int Cin, void* stream);
int bevf_conv3x3_bf16(const bevf_conv_desc* d, void* stream);
/* Diagnostic only (in-kernel s_memtime stamps of the kernel above: start / operands landed / K loop done / stores acknowledged per
 * workgroup, 4 x uint64 each, into `buf` for every following launch; NULL switches it off).  Never set by the product path. */
int bevf_debug_conv3x3_stamps(void* buf);
/* The same for bevf_conv3x3_wino_f32 (plain ReLU launches): start / first operands transformed / K loop done / epilogue issued / stores
 * acknowledged, 5 x uint64 per workgroup. */
int bevf_debug_wino_stamps(void* buf);

/* Opt-in "f32x3" convolution: same contract as bevf_conv2d_nhwc_f32 (fp32 activations in and out, no colmax), but
 * the products run on the bf16 MFMA over an exact three-way bf16 split of both operands (six partial products,
 * fp32 accumulate): fp32-level error, not bit-identical to the fp32 FMA chain.  `w` = planes written by
 * bevf_split_weights_f32x3: [3][Cout][KH][KW][Cin] bf16 (hi, mid, lo).  tile: 0 auto, 1 / 3 / 4 as above.      */
int bevf_split_weights_f32x3(const float* w, void* planes, size_t n, void* stream);
int bevf_conv2d_nhwc_f32x3(const bevf_conv_desc* d, void* stream);
int bevf_stem_conv7x7_bf16out(const float* x, const float* w, const float* scale, const float* shift, void* y, int N,
                              int H, int W, int relu, void* stream);
/* bf16 stem on the bf16 MFMA (bf16-storage models): x fp32 NCHW (rounded to bf16 while it is staged), filter bank
 * bf16 [64][176] from bevf_stem_pack_bf16 (k = (c*7+kh)*8 + kw, zero columns for kw = 7 and k >= 168), fp32
 * accumulate, folded BN + ReLU, bf16 NHWC out.                                                                   */
int bevf_stem_pack_bf16(const float* w_oihw, void* packed, void* stream);
int bevf_stem_conv7x7_bf16mma(const float* x, const void* w_packed, const float* scale, const float* shift, void* y,
                              int N, int H, int W, int relu, void* stream); /* The bf16 stem and the 3x3/s2 max-pool in one kernel (bf16-storage inference): fp32 image in, pooled bf16 NHWC
 * [N][Hp][Wp][64] out, bit-identical to bevf_stem_conv7x7_bf16mma (relu = 1) followed by bevf_maxpool3x3s2_nhwc_bf16;
 * the 64-channel stem map never reaches HBM (ref src/encoders.py:154-157).                                          */
int bevf_stem_pool_bf16mma(const float* x, const void* w_packed, const float* scale, const float* shift, void* y, int N,
                           int H, int W, void* stream);
    /* fp32 image + fp32 MFMA, bf16 NHWC out */
int bevf_maxpool3x3s2_nhwc_bf16(const void* x, void* y, int N, int H, int W, int C, void* stream);
int bevf_pointwise_smallk_bf16out(const float* x, const float* w, const float* scale, const float* shift, void* y,
                                  int M, int K, int Cout, int relu, void* stream);
int bevf_linear_bf16w(const float* x, const void* w, const float* bias, void* y, int y_bf16, int B, int K, int O,
                      int relu, int perm_inner, int perm_outer, void* stream);
int bevf_cam_mean_bf16(const void* x, void* y, int B, int ncam, int P, int C, void* stream);
int bevf_bilinear_nhwc_bf16(const void* x, void* y, int B, int Hi, int Wi, int C, int x_cs, int Ho, int Wo, int y_cs,
                            void* stream);
int bevf_broadcast_nhwc_bf16(const float* v, void* y, int B, int P, int C, int y_cs, void* stream);
int bevf_expand_border_classes_bf16(const void* small, void* y, int B, int Sh, int Sw, int C, int y_cs, void* stream);
int bevf_head_tail_bf16(const bevf_head_desc* d, void* stream);          /* hid bf16, outputs fp32 NCHW */

/* ==========================================================================================
 * Training step (ref src/train_detect.py:401-434): the backward of every layer on
 * the path, train-mode BatchNorm, gradient clipping and AdamW.  Gradients accumulated with fp32 atomics
 * (conv weight gradient, bilinear / gather-L1 scatter) are not bitwise reproducible run to run.
 * ========================================================================================== */

/* Tap table for the weight gradient: tab[m][t] = byte offset of x[n][ih][iw][0] under filter tap t of output pixel
 * m (0x80000000 for padding taps and for the rows that pad M up to a multiple of 32); depends on the shape and on
 * x_cs only, so callers cache it.  bevf_conv_pixtab_bytes = size of the table.                                   */
size_t bevf_conv_pixtab_bytes(int N, int H, int W, int KH, int KW, int stride, int pad);
int bevf_conv_pixtab(int32_t* tab, int N, int H, int W, int KH, int KW, int stride, int pad, int x_cs, void* stream);

/* dW[co][kh][kw][ci] += sum_m dy[m][co] * x[pixel(m)+tap][ci]   (dw zero-filled by the caller; OHWI like the
 * forward weights; the host permutes back to the parameter's OIHW).  MFMA fp32, pixel range split over WGs. */
typedef struct {
  const float* x;          /* [N][H][W][x_cs] forward input */
  const float* dy;         /* [N*Ho*Wo][dy_cs] gradient of the raw conv output */
  float* dw;               /* [Cout][KH][KW][Cin] */
  const int32_t* pixtab;   /* bevf_conv_pixtab for this shape and x_cs */
  int32_t N, H, W, Cin, x_cs, Cout, dy_cs, KH, KW, stride, pad;
} bevf_wgrad_desc;
int bevf_conv2d_wgrad_f32(const bevf_wgrad_desc* d, void* stream);
/* The same gradient for 3x3 / stride 1 / pad 1 layers with Cin, Cout multiples of 64, in the Winograd F(2x2,3x3) domain
 * (csrc/conv_wino_wgrad.hip): 2.25x fewer MFMA FLOPs, deterministic (fixed-order sum of per-workgroup partial blocks, no
 * atomics).  `workspace`: bevf_wino_wgrad_workspace_floats() floats (0 = shape not supported: use bevf_conv2d_wgrad_f32);
 * `pixtab` of the descriptor = the per-shape tile table written by bevf_wino_wgrad_table (bevf_wino_wgrad_table_bytes()
 * bytes; depends on N, H, W and the two channel strides only, reusable across layers and steps); accumulate != 0 adds to
 * dw instead of overwriting it.  dw: [Cout][3][3][Cin]. */
size_t bevf_wino_wgrad_workspace_floats(int N, int H, int W, int Cin, int Cout);
size_t bevf_wino_wgrad_table_bytes(int N, int H, int W);
int bevf_wino_wgrad_table(int32_t* tab, int N, int H, int W, int x_cs, int dy_cs, void* stream);
int bevf_conv3x3_wgrad_wino_f32(const bevf_wgrad_desc* d, float* workspace, int accumulate, void* stream);

/* Data gradient: the forward kernel (bevf_conv2d_nhwc_f32) run on dy with the spatially flipped, channel-
 * transposed filter; strided convs first spread dy onto the input grid with zeros in between:            */
int bevf_zero_stuff_nhwc_f32(const float* dy, float* out, int N, int Ho, int Wo, int C, int H, int W, int s, void* stream);
/* Stride-2 data gradients without the zeros: the four input-parity classes are each a small stride-1 conv over dy
 * (1x1, 1x2, 2x1, 2x2 taps for a 3x3 filter); this writes dx[n][ih][iw] = cls[(ih&1)*2+(iw&1)][n][ih>>1][iw>>1]
 * (class q stored hq[q] x wq[q]; a null class is zeros, e.g. three of the four for a 1x1 stride-2 conv).          */
int bevf_interleave2x2_nhwc_f32(const float* const* cls4, const int32_t* hq4, const int32_t* wq4, float* dx, int N, int H,
                                int W, int C, void* stream);

/* Train-mode BatchNorm{1,2}d over rows [M][C] (channel stride cs): batch mean / biased variance / invstd
 * (two-stage, fixed order, shifted sums), apply (+residual)(+ReLU), and backward:
 *   dy <- dy * (y > 0) if relu;  dbeta = sum dy;  dgamma = sum dy*xhat;
 *   dx = gamma*invstd*(dy - dbeta/M - xhat*dgamma/M)   (dx == NULL: only the sums -> conv bias gradients) */
size_t bevf_bn_work_floats(int C);
/* running_mean/var <- (1-momentum)*running + momentum*batch (variance unbiased by M/(M-1)), num_batches_tracked += 1
 * (may be NULL): torch.nn.BatchNorm's training-mode buffer update in one launch.                                  */
int bevf_bn_update_running_f32(const float* mean, const float* var, float* running_mean, float* running_var,
                               int64_t* num_batches_tracked, int C, int M, float momentum, void* stream);
int bevf_bn_stats_f32(const float* x, float* work, float* mean, float* var, float* invstd, int M, int C, int cs,
                      float eps, void* stream);
/* The same statistics from partial sums a producer left behind (bevf_conv3x3_wino_f32 with `stats`): part [G][C][2] =
 * {sum(x - pivot), sum((x - pivot)^2)} over disjoint row sets covering all M rows; fixed-order merge in double. */
int bevf_bn_stats_from_partials_f32(const float* part, int G, const float* pivot, float* mean, float* var, float* invstd,
                                    int M, int C, float eps, void* stream);
int bevf_bn_apply_f32(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                      const float* res, float* y, int M, int C, int cs, int relu, void* stream);
/* relu with y == NULL: the mask is recomputed from x exactly as bn_apply computed it (gamma, beta as in the forward;
 * only valid when the forward had no residual input) -- saves reading the forward output.  relu == 2: the same, and
 * dy is left untouched (relu == 1 writes the masked gradient back in place): both passes mask on the fly.
 * relu | 4: the layer normalised with FIXED statistics (an eval-mode BatchNorm inside a module that trains, mean / invstd =
 * its running buffers): dgamma / dbeta as always, dx = gamma * invstd * dy (the two mean terms vanish).               */
int bevf_bn_backward_f32(float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                         const float* gamma, const float* beta, float* work, float* dgamma, float* dbeta, float* dx,
                         int M, int C, int cs, int relu, void* stream);
/* The same backward when the producer of dy has already applied the ReLU mask and left the per-channel partial sums
 * (bevf_conv3x3_wino_f32 with bnb_x): part [G][C][2] = {sum dy, sum dy * xhat}.  Merges them (fixed order, double) into
 * dbeta / dgamma and writes dx = gamma * invstd * (dy - sum_dy / M - xhat * sum_dyx / M). */
int bevf_bn_backward_from_partials_f32(const float* dy, const float* x, const float* mean, const float* invstd,
                                       const float* gamma, const float* part, int G, float* dgamma, float* dbeta, float* dx,
                                       int M, int C, int cs, void* stream);
/* BatchNorm(+ReLU) backward whose dY is the backward of a 3x3/s2/p1 max-pool (the ResNet stem in training, ref
 * src/encoders.py:154-157): dpool [N][Ho][Wo][C], idx from bevf_maxpool3x3s2_idx_f32, x the raw conv output [N][H][W][C].
 * Bit-identical to bevf_maxpool3x3s2_bwd_f32 + bevf_bn_backward_f32(relu = 1, y = NULL); the dense dY never exists.       */
int bevf_pool_bn_backward_f32(const float* dpool, const uint8_t* idx, const float* x, const float* mean, const float* invstd,
                              const float* gamma, const float* beta, float* work, float* dgamma, float* dbeta, float* dx,
                              int N, int H, int W, int C, void* stream);
/* Forward twin: y = maxpool3x3s2(relu(batchnorm(x))) with the argmax codes of bevf_maxpool3x3s2_idx_f32, the normalised map never
 * written (bit-identical to bevf_bn_apply_f32(relu = 1) followed by bevf_maxpool3x3s2_idx_f32).                                */
int bevf_bn_relu_maxpool3x3s2_idx_f32(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                      float* y, uint8_t* idx, int N, int H, int W, int C, void* stream);

int bevf_add_inplace_f32(float* y, const float* x, size_t n, void* stream);            /* y += x            */
int bevf_relu_mask_f32(float* dy, const float* y, size_t n, void* stream);             /* dy *= (y > 0)     */
int bevf_maxpool3x3s2_idx_f32(const float* x, float* y, uint8_t* idx, int N, int H, int W, int C, void* stream);
int bevf_maxpool3x3s2_bwd_f32(const float* dy, const uint8_t* idx, float* dx, int N, int H, int W, int C, void* stream);
int bevf_bilinear_bwd_nhwc_f32(const float* dy, float* dx, int B, int Hi, int Wi, int C, int x_cs, int Ho, int Wo,
                               int y_cs, void* stream);                                 /* dx zero-filled    */
int bevf_cam_mean_bwd_f32(const float* dy, float* dx, int B, int ncam, int P, int C, void* stream);
size_t bevf_group_max_idx_work_bytes(int G, int P, int C);
int bevf_group_max_idx_f32(const float* x, float* y, int32_t* idx, void* work, int G, int P, int C, void* stream);
int bevf_group_max_bwd_f32(const float* dy, const int32_t* idx, float* dx, int G, int P, int C, void* stream); /* dx zero-filled */
/* The sparse rows of the low-rank backward of conv -> BatchNorm -> ReLU -> max over points (ref src/encoders.py:296-299): S [G][C] = one entry
 * per (frame, channel), living at row g*P + idx[g][c] of the [G*P][K] layer input A.  out[c][k] = sum_g S[g][c] A[row(g,c)][k];
 * dA[row(g,c)][k] += S[g][c] W[c][k].  Fixed summation order, no atomics: bit-identical run to run. */
int bevf_sparse_rows_wgrad_f32(const float* S, const int32_t* idx, const float* A, float* out, int G, int P, int C, int K, void* stream);
int bevf_sparse_rows_scatter_add_f32(const float* S, const int32_t* idx, const float* W, float* dA, int G, int P, int C, int K, void* stream);
/* Same max / argmax (work: bevf_group_max_idx_work_bytes) over relu(batchnorm(x)) evaluated on the fly from the raw
 * rows x with bn_apply's own fma: the training forward of PointNet's last layer writes no activation.            */
int bevf_bn_relu_group_max_idx_f32(const float* x, const float* mean, const float* invstd, const float* gamma,
                                   const float* beta, float* y, int32_t* idx, void* work, int G, int P, int C, void* stream);
/* Backward of max-over-rows( relu( batchnorm(x) ) ) without the dense intermediate gradient (PointNet's last layer,
 * ref src/encoders.py:296-299): dg / gmax / idx [B][C] = gradient, value and argmax row of the max; writes dgamma,
 * dbeta, dx [B*P][cs]; dgm [B][C] scratch.                                                                        */
int bevf_gmax_bn_backward_f32(const float* dg, const float* gmax, const int32_t* idx, const float* x, const float* mean,
                              const float* invstd, const float* gamma, float* dgm, float* dgamma, float* dbeta, float* dx,
                              int B, int P, int C, int cs, void* stream);
/* Its first stage alone (masked dg, dgamma, dbeta), for callers that never build the dense dx (the low-rank backward of
 * PointNet's last layer in training.py).                                                                             */
int bevf_gmax_bn_sums_f32(const float* dg, const float* gmax, const int32_t* idx, const float* x, const float* mean,
                          const float* invstd, float* dgm, float* dgamma, float* dbeta, int B, int P, int C, int cs, void* stream);
size_t bevf_linear_bwd_work_floats(int B, int K, int O);
int bevf_linear_bwd_f32(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, float* work,
                        int B, int K, int O, int perm_inner, int perm_outer, void* stream);
typedef struct {
  const float* hid; const float* w; const float* out0;   /* out0: post-sigmoid heatmap (B,c0,H,W) */
  const float* dout[5];
  float* dhid; float* dw; float* db;                     /* dw, db zero-filled by the caller */
  int32_t B, P, hc;
  int32_t c[5];
  int32_t n_sigmoid;
} bevf_head_bwd_desc;
int bevf_head_tail_bwd_f32(const bevf_head_bwd_desc* d, void* stream);
/* d total_loss / d predictions for CenterNetLoss (ref src/centernet_target.py:544-622); dpred[5] zero-filled;
 * scratch2: 2 floats.                                                                                        */
int bevf_centernet_loss_bwd_f32(const bevf_loss_desc* d, float* const dpred[5], float* scratch2, void* stream);
int bevf_stem_im2col_f32(const float* x, float* col, int N, int H, int W, void* stream);   /* [M][160], k=c*49+kh*7+kw */
/* Stem weight gradient without the column matrix: dw [64][160] (k = c*49+kh*7+kw, columns >= 147 unused, zero-filled
 * by the caller) += sum over pixels dy[pixel][co] * x under tap k; x planar NCHW fp32, dy [N][Ho][Wo][64].       */
int bevf_stem_wgrad_f32(const float* x, const float* dy, float* dw, int N, int H, int W, void* stream);
int bevf_smallk_wgrad_f32(const float* dy, const float* x, float* dw, int M, int K, int Cout, void* stream);
/* clip_grad_norm_: out2 = {total L2 norm, min(1, max_norm/(norm+1e-6))}; work512: 512 doubles.                */
int bevf_grad_norm_f32(const float* g, size_t n, double* work512, float max_norm, float* out2, void* stream);
/* torch.optim.AdamW single step on a flat parameter range; gradients are scaled by clip2[1] when given.        */
int bevf_adamw_step_f32(float* p, const float* g, float* m, float* v, const float* clip2, size_t n, float lr,
                        float beta1, float beta2, float eps, float weight_decay, int step, void* stream);

/* ==========================================================================================
 * Input pipeline (ref src/train_detect.py:123-189, the dataset's per-sample host work)
 * ========================================================================================== */

/* uint8 HWC frames -> Pillow-identical antialiased bilinear resize (T.Resize on a PIL image: 22-bit fixed point,
 * uint8 after the horizontal pass) -> /255 -> (t-mean)/std -> planar fp32 [n][3][Ho][Wo].  bounds_* [out][2] =
 * (first source index, count), coef_* [out][ksize] = Pillow's integer weights for that axis (host:
 * preprocess.resample_tables).  Replaces ref src/train_detect.py:127-143.                                        */
int bevf_resize_normalize_u8(const unsigned char* x, float* out, int n, int H, int W, int Ho, int Wo,
                             const int32_t* bounds_h, const int32_t* coef_h, int ksize_h, const int32_t* bounds_v,
                             const int32_t* coef_v, int ksize_v, const float* mean3, const float* std3, void* stream);

/* One LiDAR sweep [N][C]: keep points strictly inside pc_range6 = (x0,y0,z0,x1,y1,z1), in input order; out
 * [max_points][C] = the survivors then zero rows, or survivors[choice[i]] when `choice` (max_points int64 indices,
 * the reference's np.random.choice) is given and at least max_points survive.  count = number of survivors.
 * work: N*C + ceil(N / 1024) floats (the compacted rows, then one counter per tile of 1024 points).  Replaces ref
 * src/train_detect.py:150-159, 181-189.                                                                          */
int bevf_lidar_filter_pad_f32(const float* points, float* out, int32_t* count, float* work, const int64_t* choice,
                              int N, int C, int max_points, const float* pc_range6, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BEVF_H */
