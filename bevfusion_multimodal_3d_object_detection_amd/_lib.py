"""ctypes binding of libbevf_hip.so (include/bevf.h).  Fails loudly if the library is missing.

Every wrapper takes torch CUDA tensors, checks device / dtype / contiguity on the host,
passes raw `data_ptr()`s and launches on torch's current HIP stream, so the launches are
ordered with (and graph-capturable alongside) everything else torch does on that stream.
There is no CPU fallback: a CPU tensor raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libbevf_hip.so")


class BevfError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("res", C.c_void_p), ("y", C.c_void_p), ("colmax", C.c_void_p)] + \
               [(n, C.c_int32) for n in ("N", "H", "W", "Cin", "x_cs", "Ho", "Wo", "Cout", "y_cs", "res_cs",
                                         "KH", "KW", "stride", "pad", "relu", "rows_per_group", "tile")] + \
               [(n, C.c_void_p) for n in ("stats", "stats_pivot", "bnb_x", "bnb_y", "bnb_mean", "bnb_invstd", "bnb_gamma",
                                          "bnb_beta")]


class RadarDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p * 4), ("scale", C.c_void_p * 4), ("shift", C.c_void_p * 4),
                ("out", C.c_void_p), ("R", C.c_int32), ("B", C.c_int32), ("P", C.c_int32), ("Cin", C.c_int32),
                ("c", C.c_int32 * 4)]


class HeadDesc(C.Structure):
    _fields_ = [("hid", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p * 5),
                ("B", C.c_int32), ("P", C.c_int32), ("hc", C.c_int32), ("c", C.c_int32 * 5),
                ("n_sigmoid", C.c_int32)]


class DecodeDesc(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("heat", "offset", "size", "rot", "vel", "boxes", "scores", "labels",
                                          "velocities", "count", "work", "pool_ind")] + \
               [(n, C.c_int32) for n in ("B", "C", "H", "W", "K", "true_labels", "raw_scores")] + \
               [(n, C.c_float) for n in ("thresh", "voxel", "x_min", "y_min")]


class TargetsDesc(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("boxes", "labels", "has_vel", "heatmap", "offset", "size", "rot", "vel",
                                          "mask", "ind", "reg_mask", "target_offset", "target_size", "target_rot",
                                          "target_vel", "owner_scratch")] + \
               [(n, C.c_int32) for n in ("B", "nmax", "H", "W", "C", "max_objects", "min_radius")] + \
               [("pc_range", C.c_float * 6), ("gaussian_overlap", C.c_float)]


class WgradDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("dy", C.c_void_p), ("dw", C.c_void_p), ("pixtab", C.c_void_p)] + \
               [(n, C.c_int32) for n in ("N", "H", "W", "Cin", "x_cs", "Cout", "dy_cs", "KH", "KW", "stride", "pad")]


class HeadBwdDesc(C.Structure):
    _fields_ = [("hid", C.c_void_p), ("w", C.c_void_p), ("out0", C.c_void_p), ("dout", C.c_void_p * 5),
                ("dhid", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p), ("B", C.c_int32), ("P", C.c_int32),
                ("hc", C.c_int32), ("c", C.c_int32 * 5), ("n_sigmoid", C.c_int32)]


class VoxelizeDesc(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("points", "voxel_features", "voxel_coords", "num_points", "num_voxels", "work")] + \
               [(n, C.c_int32) for n in ("B", "N", "C", "max_points", "max_voxels")] + \
               [("pc_range", C.c_float * 6), ("voxel_size", C.c_float * 3)]


class LossDesc(C.Structure):
    _fields_ = [("pred_heatmap", C.c_void_p), ("tgt_heatmap", C.c_void_p), ("pred_reg", C.c_void_p * 4),
                ("tgt_reg", C.c_void_p * 4), ("ind", C.c_void_p), ("reg_mask", C.c_void_p), ("work", C.c_void_p),
                ("out", C.c_void_p)] + [(n, C.c_int32) for n in ("B", "C", "H", "W", "K")] + \
               [("weights", C.c_float * 5)]


_lib: Optional[C.CDLL] = None

# name -> (restype, argtypes); the not-gpu test checks every one is exported by the .so
SIGNATURES = {
    "bevf_version": (C.c_int, []),
    "bevf_last_error": (C.c_char_p, []),
    "bevf_conv2d_nhwc_f32": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "bevf_stem_pool_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 3 + [C.c_void_p]),
    "bevf_stem_conv7x7_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_void_p]),
    "bevf_maxpool3x3s2_nhwc_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 4 + [C.c_void_p]),
    "bevf_pointwise_smallk_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_void_p]),
    "bevf_pointnet_front_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 11),
    "bevf_pointnet_front_pack_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 2 + [C.c_void_p]),
    "bevf_group_max_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 3 + [C.c_void_p]),
    "bevf_vfe_smallk_max_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_void_p]),
    "bevf_radar_mlp_max_f32": (C.c_int, [C.POINTER(RadarDesc), C.c_void_p]),
    "bevf_linear_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p]),
    "bevf_cam_mean_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 4 + [C.c_void_p]),
    "bevf_bilinear_nhwc_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 8 + [C.c_void_p]),
    "bevf_broadcast_nhwc_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 4 + [C.c_void_p]),
    "bevf_expand_border_classes_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 5 + [C.c_void_p]),
    "bevf_head_tail_f32": (C.c_int, [C.POINTER(HeadDesc), C.c_void_p]),
    "bevf_nchw_to_nhwc_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 4 + [C.c_void_p]),
    "bevf_nhwc_to_nchw_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 4 + [C.c_void_p]),
    "bevf_fill_f32": (C.c_int, [C.c_void_p, C.c_float, C.c_size_t, C.c_void_p]),
    "bevf_wino_filter_floats": (C.c_size_t, [C.c_int] * 2),
    "bevf_wino_filter_transform_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 2 + [C.c_void_p]),
    "bevf_conv3x3_wino_f32": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "bevf_wino_stat_rows": (C.c_int, [C.c_int] * 3),
    "bevf_bn_backward_from_partials_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p]),
    "bevf_bn_stats_from_partials_f32": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 2 + [C.c_float, C.c_void_p]),
    "bevf_centernet_decode_work_bytes": (C.c_size_t, [C.c_int] * 5),
    "bevf_centernet_decode_f32": (C.c_int, [C.POINTER(DecodeDesc), C.c_void_p]),
    "bevf_centernet_targets_f32": (C.c_int, [C.POINTER(TargetsDesc), C.c_void_p]),
    "bevf_nms_keep_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 3 + [C.c_void_p]),
    "bevf_centernet_loss_work_floats": (C.c_size_t, []),
    "bevf_centernet_loss_f32": (C.c_int, [C.POINTER(LossDesc), C.c_void_p]),
    "bevf_voxelize_work_bytes": (C.c_size_t, [C.c_int] * 2),
    "bevf_voxelize_f32": (C.c_int, [C.POINTER(VoxelizeDesc), C.c_void_p]),
    "bevf_scatter_voxels_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 6 + [C.c_void_p]),
    # ---- input pipeline ----
    "bevf_resize_normalize_u8": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 5 + [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 2 +
                                 [C.c_int] + [C.POINTER(C.c_float)] * 2 + [C.c_void_p]),
    "bevf_lidar_filter_pad_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 3 + [C.POINTER(C.c_float), C.c_void_p]),
    # ---- bf16 storage path ----
    "bevf_split_weights_f32x3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "bevf_conv2d_nhwc_f32x3": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "bevf_conv2d_nhwc_bf16": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "bevf_conv3x3_pack_elems": (C.c_size_t, [C.c_int] * 2),
    "bevf_conv3x3_bf16_ct": (C.c_int, [C.c_int]),
    "bevf_conv3x3_pack_bf16": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 2 + [C.c_void_p]),
    "bevf_conv3x3_bf16": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "bevf_debug_conv3x3_stamps": (C.c_int, [C.c_void_p]),
    "bevf_debug_wino_stamps": (C.c_int, [C.c_void_p]),
    "bevf_stem_pack_bf16": (C.c_int, [C.c_void_p] * 3),
    "bevf_stem_conv7x7_bf16mma": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_void_p]),
    "bevf_stem_pool_bf16mma": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 3 + [C.c_void_p]),
    "bevf_stem_conv7x7_bf16out": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_void_p]),
    "bevf_maxpool3x3s2_nhwc_bf16": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 4 + [C.c_void_p]),
    "bevf_pointwise_smallk_bf16out": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_void_p]),
    "bevf_linear_bf16w": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 7 + [C.c_void_p]),
    "bevf_cam_mean_bf16": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 4 + [C.c_void_p]),
    "bevf_bilinear_nhwc_bf16": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 8 + [C.c_void_p]),
    "bevf_broadcast_nhwc_bf16": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 4 + [C.c_void_p]),
    "bevf_expand_border_classes_bf16": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 5 + [C.c_void_p]),
    "bevf_head_tail_bf16": (C.c_int, [C.POINTER(HeadDesc), C.c_void_p]),
    # ---- training step ----
    "bevf_conv_pixtab_bytes": (C.c_size_t, [C.c_int] * 7),
    "bevf_conv_pixtab": (C.c_int, [C.c_void_p] + [C.c_int] * 8 + [C.c_void_p]),
    "bevf_conv2d_wgrad_f32": (C.c_int, [C.POINTER(WgradDesc), C.c_void_p]),
    "bevf_wino_wgrad_workspace_floats": (C.c_size_t, [C.c_int] * 5),
    "bevf_wino_wgrad_table_bytes": (C.c_size_t, [C.c_int] * 3),
    "bevf_wino_wgrad_table": (C.c_int, [C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]),
    "bevf_conv3x3_wgrad_wino_f32": (C.c_int, [C.POINTER(WgradDesc), C.c_void_p, C.c_int, C.c_void_p]),
    "bevf_zero_stuff_nhwc_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 7 + [C.c_void_p]),
    "bevf_stem_wgrad_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p]),
    "bevf_interleave2x2_nhwc_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]),
    "bevf_bn_work_floats": (C.c_size_t, [C.c_int]),
    "bevf_bn_update_running_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 2 + [C.c_float, C.c_void_p]),
    "bevf_bn_stats_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 3 + [C.c_float, C.c_void_p]),
    "bevf_bn_apply_f32": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 4 + [C.c_void_p]),
    "bevf_bn_relu_group_max_idx_f32": (C.c_int, [C.c_void_p] * 8 + [C.c_int] * 3 + [C.c_void_p]),
    "bevf_gmax_bn_backward_f32": (C.c_int, [C.c_void_p] * 11 + [C.c_int] * 4 + [C.c_void_p]),
    "bevf_gmax_bn_sums_f32": (C.c_int, [C.c_void_p] * 9 + [C.c_int] * 4 + [C.c_void_p]),
    "bevf_bn_backward_f32": (C.c_int, [C.c_void_p] * 11 + [C.c_int] * 4 + [C.c_void_p]),
    "bevf_pool_bn_backward_f32": (C.c_int, [C.c_void_p] * 11 + [C.c_int] * 4 + [C.c_void_p]),
    "bevf_bn_relu_maxpool3x3s2_idx_f32": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 4 + [C.c_void_p]),
    "bevf_add_inplace_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "bevf_relu_mask_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "bevf_maxpool3x3s2_idx_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p]),
    "bevf_maxpool3x3s2_bwd_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p]),
    "bevf_bilinear_bwd_nhwc_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 8 + [C.c_void_p]),
    "bevf_cam_mean_bwd_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 4 + [C.c_void_p]),
    "bevf_group_max_idx_work_bytes": (C.c_size_t, [C.c_int] * 3),
    "bevf_group_max_idx_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p]),
    "bevf_group_max_bwd_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p]),
    "bevf_sparse_rows_wgrad_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]),
    "bevf_sparse_rows_scatter_add_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]),
    "bevf_linear_bwd_work_floats": (C.c_size_t, [C.c_int] * 3),
    "bevf_linear_bwd_f32": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 5 + [C.c_void_p]),
    "bevf_head_tail_bwd_f32": (C.c_int, [C.POINTER(HeadBwdDesc), C.c_void_p]),
    "bevf_centernet_loss_bwd_f32": (C.c_int, [C.POINTER(LossDesc), C.c_void_p * 5, C.c_void_p, C.c_void_p]),
    "bevf_stem_im2col_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 3 + [C.c_void_p]),
    "bevf_smallk_wgrad_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p]),
    "bevf_grad_norm_f32": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "bevf_adamw_step_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_size_t] + [C.c_float] * 5 + [C.c_int, C.c_void_p]),
}


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BevfError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            f"or `make -C {os.path.dirname(LIB_PATH)}` -- there is no CPU fallback")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise BevfError(f"{what} failed ({rc}): {lib().bevf_last_error().decode()}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor], dtype=torch.float32) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise BevfError("HIP path needs CUDA/HIP tensors; got a CPU tensor (no CPU fallback in this package)")
    if t.dtype != dtype:
        raise BevfError(f"expected {dtype}, got {t.dtype}")
    return t.data_ptr()


BF16 = torch.bfloat16


def _sfx(t: torch.Tensor) -> str:
    """Entry-point suffix for a storage dtype."""
    if t.dtype == torch.float32:
        return "f32"
    if t.dtype == BF16:
        return "bf16"
    raise BevfError(f"unsupported storage dtype {t.dtype} (fp32 or bf16)")


def _pc(t: Optional[torch.Tensor], dtype=torch.float32) -> Optional[int]:
    if t is not None and not t.is_contiguous():
        raise BevfError("tensor must be contiguous")
    return _p(t, dtype)


# ---- wrappers -------------------------------------------------------------------------------------

def conv2d_nhwc(x: torch.Tensor, w: torch.Tensor, scale, shift, y: Optional[torch.Tensor], *, N: int, H: int,
                W: int, Cin: int, x_cs: int, Cout: int, y_cs: int, KH: int, KW: int, stride: int, pad: int,
                relu: bool, res: Optional[torch.Tensor] = None, res_cs: int = 0,
                colmax: Optional[torch.Tensor] = None, rows_per_group: int = 0, tile: int = 0) -> None:
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    M = N * Ho * Wo
    if x.numel() < (N * H * W - 1) * x_cs + Cin:
        raise BevfError("conv: input buffer smaller than N*H*W*x_cs")
    split = x.dtype == torch.float32 and w.dtype == torch.bfloat16      # f32x3: planes from split_weights_f32x3
    if w.numel() != Cout * KH * KW * Cin * (3 if split else 1):
        raise BevfError(f"conv: packed weight has {w.numel()} elements, expected {Cout * KH * KW * Cin * (3 if split else 1)}")
    if y is not None and y.numel() < (M - 1) * y_cs + Cout:
        raise BevfError("conv: output buffer too small")
    if res is not None and res.numel() < (M - 1) * res_cs + Cout:
        raise BevfError("conv: residual buffer too small")
    for v in (scale, shift):
        if v is not None and v.numel() != Cout:
            raise BevfError("conv: scale/shift length != Cout")
    if colmax is not None and colmax.numel() < -(-M // rows_per_group) * Cout:
        raise BevfError("conv: colmax buffer too small")
    dt = x.dtype
    d = ConvDesc(_p(x, dt), _pc(w, w.dtype), _pc(scale), _pc(shift), _p(res, dt), _p(y, dt), _pc(colmax, torch.int32),
                 N, H, W, Cin, x_cs, Ho, Wo, Cout, y_cs, res_cs, KH, KW, stride, pad, int(relu), rows_per_group, tile)
    fn = "bevf_conv2d_nhwc_f32x3" if split else "bevf_conv2d_nhwc_" + _sfx(x)
    _check(getattr(lib(), fn)(C.byref(d), _stream()), fn)


def wino_filter_transform(w_ohwi: torch.Tensor, Cout: int, Cin: int) -> torch.Tensor:
    """fp32 OHWI 3x3 filter -> the transformed-filter image of bevf_conv3x3_wino_f32."""
    w_ohwi = w_ohwi.contiguous()
    if w_ohwi.numel() != Cout * 9 * Cin:
        raise BevfError(f"wino_filter_transform: filter has {w_ohwi.numel()} elements, expected {Cout * 9 * Cin}")
    u = torch.empty(lib().bevf_wino_filter_floats(Cout, Cin), dtype=torch.float32, device=w_ohwi.device)
    _check(lib().bevf_wino_filter_transform_f32(_pc(w_ohwi), _p(u), Cout, Cin, _stream()), "bevf_wino_filter_transform_f32")
    return u


def conv3x3_wino(x: torch.Tensor, u: torch.Tensor, scale, shift, y: torch.Tensor, *, N: int, H: int, W: int, Cin: int,
                 x_cs: int, Cout: int, y_cs: int, relu: bool, res: Optional[torch.Tensor] = None, res_cs: int = 0,
                 stats: Optional[torch.Tensor] = None, stats_pivot: Optional[torch.Tensor] = None, bnb: Optional[dict] = None,
                 tile: int = 0) -> None:
    """3x3 / stride 1 / pad 1 convolution as fused fp32 Winograd F(2x2,3x3); `u` from wino_filter_transform.
    tile: 0 = the tiling that covers the batch with the fewest blocks, 1 = 16x16-pixel blocks per image, 2 = 32x8-pixel blocks per image,
    3 / 4 = the same two block shapes over the images' rows stacked into one map (bit-identical results, fewer dead rows).
    `stats` [bevf_wino_stat_rows(N,H,W)][Cout][2]: also leave the BatchNorm partial sums of the output (training)."""
    if stats is not None and stats.numel() < lib().bevf_wino_stat_rows(N, H, W) * Cout * 2:
        raise BevfError("conv_wino: stats buffer too small")
    if stats_pivot is not None and stats_pivot.numel() < Cout:
        raise BevfError("conv_wino: stats_pivot shorter than Cout")
    M = N * H * W
    if x.numel() < (M - 1) * x_cs + Cin:
        raise BevfError("conv_wino: input buffer smaller than N*H*W*x_cs")
    if u.numel() != lib().bevf_wino_filter_floats(Cout, Cin):
        raise BevfError("conv_wino: transformed filter has the wrong size")
    if y.numel() < (M - 1) * y_cs + Cout:
        raise BevfError("conv_wino: output buffer too small")
    if res is not None and res.numel() < (M - 1) * res_cs + Cout:
        raise BevfError("conv_wino: residual buffer too small")
    for v in (scale, shift):
        if v is not None and v.numel() != Cout:
            raise BevfError("conv_wino: scale/shift length != Cout")
    d = ConvDesc(_p(x), _pc(u), _pc(scale), _pc(shift), _p(res), _p(y), None, N, H, W, Cin, x_cs, H, W, Cout, y_cs, res_cs,
                 3, 3, 1, 1, int(relu), 0, int(tile), _p(stats), _pc(stats_pivot))
    if bnb is not None:       # this conv's output is dY of a train-mode BatchNorm(+ReLU) layer: mask + backward sums in the epilogue
        for k in ("x", "mean", "invstd"):
            if bnb.get(k) is None:
                raise BevfError(f"conv_wino: bnb needs '{k}'")
        if bnb["x"].numel() < M * Cout or (bnb.get("y") is not None and bnb["y"].numel() < M * Cout):
            raise BevfError("conv_wino: bnb x / y smaller than the output")
        d.bnb_x, d.bnb_y = _p(bnb["x"]), _p(bnb.get("y"))
        d.bnb_mean, d.bnb_invstd = _p(bnb["mean"]), _p(bnb["invstd"])
        d.bnb_gamma, d.bnb_beta = _p(bnb.get("gamma")), _p(bnb.get("beta"))
    _check(lib().bevf_conv3x3_wino_f32(C.byref(d), _stream()), "bevf_conv3x3_wino_f32")


def conv3x3_pack_bf16(w_ohwi: torch.Tensor, Cout: int, Cin: int) -> torch.Tensor:
    """bf16 OHWI 3x3 filter -> the MFMA-fragment-ordered filter image of bevf_conv3x3_bf16."""
    w_ohwi = w_ohwi.contiguous()
    if w_ohwi.dtype != torch.bfloat16 or w_ohwi.numel() != Cout * 9 * Cin:
        raise BevfError(f"conv3x3_pack: need a bf16 filter of {Cout * 9 * Cin} elements, got {w_ohwi.dtype} x {w_ohwi.numel()}")
    out = torch.empty(lib().bevf_conv3x3_pack_elems(Cout, Cin), dtype=torch.bfloat16, device=w_ohwi.device)
    _check(lib().bevf_conv3x3_pack_bf16(_pc(w_ohwi, torch.bfloat16), _p(out, torch.bfloat16), Cout, Cin, _stream()),
           "bevf_conv3x3_pack_bf16")
    return out


def conv3x3_bf16(x: torch.Tensor, wp: torch.Tensor, scale, shift, y: torch.Tensor, *, N: int, H: int, W: int, Cin: int,
                 x_cs: int, Cout: int, y_cs: int, relu: bool, res: Optional[torch.Tensor] = None, res_cs: int = 0,
                 tile: int = 0) -> None:
    """3x3 / stride 1 / pad 1 convolution of bf16 activations (fp32 accumulate, folded BN, residual, ReLU); `wp` from
    conv3x3_pack_bf16.  tile (64-channel tiles only): 0 auto, 1 = two patch buffers / two workgroups per CU, 2 = one / four,
    3 = one buffer and 32-row blocks."""
    M = N * H * W
    dt = torch.bfloat16
    if x.numel() < (M - 1) * x_cs + Cin:
        raise BevfError("conv3x3_bf16: input buffer smaller than N*H*W*x_cs")
    if wp.dtype != dt or wp.numel() != lib().bevf_conv3x3_pack_elems(Cout, Cin):
        raise BevfError("conv3x3_bf16: packed filter has the wrong size / dtype")
    if y.numel() < (M - 1) * y_cs + Cout:
        raise BevfError("conv3x3_bf16: output buffer too small")
    if res is not None and res.numel() < (M - 1) * res_cs + Cout:
        raise BevfError("conv3x3_bf16: residual buffer too small")
    for v in (scale, shift):
        if v is not None and v.numel() != Cout:
            raise BevfError("conv3x3_bf16: scale/shift length != Cout")
    d = ConvDesc(_p(x, dt), _pc(wp, dt), _pc(scale), _pc(shift), _p(res, dt), _p(y, dt), None, N, H, W, Cin, x_cs, H, W, Cout,
                 y_cs, res_cs, 3, 3, 1, 1, int(relu), 0, int(tile))
    _check(lib().bevf_conv3x3_bf16(C.byref(d), _stream()), "bevf_conv3x3_bf16")


def split_weights_f32x3(w: torch.Tensor) -> torch.Tensor:
    """fp32 packed filter -> [3][n] bf16 planes (hi, mid, lo) for the f32x3 convolution."""
    w = w.contiguous()
    out = torch.empty(3 * w.numel(), dtype=torch.bfloat16, device=w.device)
    _check(lib().bevf_split_weights_f32x3(_pc(w), _p(out, torch.bfloat16), w.numel(), _stream()), "bevf_split_weights_f32x3")
    return out


def stem_pack_bf16(w_oihw: torch.Tensor) -> torch.Tensor:
    out = torch.empty(64 * 176, dtype=torch.bfloat16, device=w_oihw.device)
    _check(lib().bevf_stem_pack_bf16(_pc(w_oihw.float().contiguous()), _p(out, torch.bfloat16), _stream()), "bevf_stem_pack_bf16")
    return out


def stem_conv7x7_bf16mma(x: torch.Tensor, w_packed: torch.Tensor, scale, shift, y: torch.Tensor, N: int, H: int, W: int,
                         relu: bool = True):
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if x.numel() != N * 3 * H * W or w_packed.numel() != 64 * 176 or y.numel() < N * Ho * Wo * 64:
        raise BevfError("stem bf16: buffer sizes do not match N,H,W")
    _check(lib().bevf_stem_conv7x7_bf16mma(_pc(x), _pc(w_packed, torch.bfloat16), _pc(scale), _pc(shift), _p(y, torch.bfloat16),
                                           N, H, W, int(relu), _stream()), "bevf_stem_conv7x7_bf16mma")


def stem_conv7x7(x: torch.Tensor, w_packed: torch.Tensor, scale, shift, y: torch.Tensor, N: int, H: int, W: int,
                 relu: bool = True):
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if x.numel() != N * 3 * H * W or w_packed.numel() != 148 * 64 or y.numel() < N * Ho * Wo * 64:
        raise BevfError("stem: buffer sizes do not match N,H,W")
    fn = "bevf_stem_conv7x7_f32" if y.dtype == torch.float32 else "bevf_stem_conv7x7_bf16out"
    _check(getattr(lib(), fn)(_pc(x), _pc(w_packed), _pc(scale), _pc(shift), _p(y, y.dtype), N, H, W, int(relu), _stream()), fn)


def stem_pool(x: torch.Tensor, w_packed: torch.Tensor, scale, shift, y: torch.Tensor, N: int, H: int, W: int):
    """stem 7x7/s2 + BN + ReLU + 3x3/s2 max-pool in one kernel: (N,3,H,W) fp32 -> pooled NHWC [N][Hp][Wp][64]."""
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
    if x.numel() != N * 3 * H * W or w_packed.numel() != 148 * 64 or y.numel() < N * Hp * Wp * 64:
        raise BevfError("stem_pool: buffer sizes do not match N,H,W")
    _check(lib().bevf_stem_pool_f32(_pc(x), _pc(w_packed), _pc(scale), _pc(shift), _p(y), N, H, W, _stream()), "bevf_stem_pool_f32")


def stem_pool_bf16mma(x: torch.Tensor, w_packed: torch.Tensor, scale, shift, y: torch.Tensor, N: int, H: int, W: int):
    """bf16 stem + BN + ReLU + 3x3/s2 max-pool in one kernel: (N,3,H,W) fp32 -> pooled bf16 NHWC [N][Hp][Wp][64]."""
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
    if x.numel() != N * 3 * H * W or w_packed.numel() != 64 * 176 or y.numel() < N * Hp * Wp * 64:
        raise BevfError("stem_pool bf16: buffer sizes do not match N,H,W")
    _check(lib().bevf_stem_pool_bf16mma(_pc(x), _pc(w_packed, torch.bfloat16), _pc(scale), _pc(shift), _p(y, torch.bfloat16),
                                        N, H, W, _stream()), "bevf_stem_pool_bf16mma")


def maxpool3x3s2(x: torch.Tensor, y: torch.Tensor, N: int, H: int, W: int, Cc: int):
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if x.numel() < N * H * W * Cc or y.numel() < N * Ho * Wo * Cc:
        raise BevfError("maxpool: buffer too small")
    fn = "bevf_maxpool3x3s2_nhwc_" + _sfx(x)
    _check(getattr(lib(), fn)(_p(x, x.dtype), _p(y, x.dtype), N, H, W, Cc, _stream()), fn)


def pointwise_smallk(x, w, scale, shift, y, M: int, K: int, Cout: int, relu: bool):
    if x.numel() < M * K or w.numel() != Cout * K or y.numel() < M * Cout:
        raise BevfError("pointwise: buffer sizes do not match M,K,Cout")
    fn = "bevf_pointwise_smallk_f32" if y.dtype == torch.float32 else "bevf_pointwise_smallk_bf16out"
    _check(getattr(lib(), fn)(_pc(x), _pc(w), _pc(scale), _pc(shift), _p(y, y.dtype), M, K, Cout, int(relu), _stream()), fn)


def pointnet_front_pack(w: torch.Tensor) -> torch.Tensor:
    """(Cout, Cin) fp32 pointwise filter -> MFMA fragment order for pointnet_front."""
    cout, cin = w.shape
    w = w.detach().float().contiguous()
    wf = torch.empty(cout * cin, device=w.device)
    _check(lib().bevf_pointnet_front_pack_f32(_p(w), _p(wf), cout, cin, _stream()), "bevf_pointnet_front_pack_f32")
    return wf


def pointnet_front(x, w1, s1, b1, w2f, s2, b2, w3f, s3, b3, y, M: int, K: int):
    if x.numel() < M * K or w1.numel() != 64 * K or w2f.numel() != 128 * 64 or w3f.numel() != 256 * 128 or y.numel() < M * 256 \
            or min(s1.numel(), b1.numel()) < 64 or min(s2.numel(), b2.numel()) < 128 or min(s3.numel(), b3.numel()) < 256:
        raise BevfError("pointnet_front: buffer sizes do not match M, K and the 64/128/256 widths")
    _check(lib().bevf_pointnet_front_f32(_pc(x), M, K, _pc(w1), _pc(s1), _pc(b1), _pc(w2f), _pc(s2), _pc(b2), _pc(w3f), _pc(s3),
                                         _pc(b3), _p(y), _stream()), "bevf_pointnet_front_f32")


def vfe_smallk_max(x, w, scale, shift, y, G: int, P: int, K: int, Cout: int) -> None:
    """VFELayer for K <= 16 in one pass: pointwise linear + folded BN + ReLU + max over the P rows of each group."""
    if x.numel() < G * P * K or w.numel() != Cout * K or y.numel() < G * Cout:
        raise BevfError("vfe_smallk_max: buffer sizes do not match G, P, K, Cout")
    _check(lib().bevf_vfe_smallk_max_f32(_pc(x), _pc(w), _pc(scale), _pc(shift), _p(y), G, P, K, Cout, _stream()),
           "bevf_vfe_smallk_max_f32")


def group_max(x, y, G: int, P: int, Cc: int):
    if x.numel() < G * P * Cc or y.numel() < G * Cc:
        raise BevfError("group_max: buffer too small")
    _check(lib().bevf_group_max_f32(_p(x), _p(y), G, P, Cc, _stream()), "bevf_group_max_f32")


def radar_mlp_max(x, ws: Sequence[torch.Tensor], scales, shifts, out, R: int, B: int, P: int, Cin: int,
                  widths: Sequence[int]):
    if x.numel() != R * B * P * Cin or out.numel() < B * R * widths[3]:
        raise BevfError("radar: buffer sizes do not match R,B,P,Cin")
    cin = Cin
    for i in range(4):
        if ws[i].numel() != cin * widths[i] or scales[i].numel() != widths[i] or shifts[i].numel() != widths[i]:
            raise BevfError(f"radar: layer {i} parameter sizes wrong")
        cin = widths[i]
    d = RadarDesc()
    d.x, d.out, d.R, d.B, d.P, d.Cin = _pc(x), _p(out), R, B, P, Cin
    for i in range(4):
        d.w[i], d.scale[i], d.shift[i], d.c[i] = _pc(ws[i]), _pc(scales[i]), _pc(shifts[i]), widths[i]
    _check(lib().bevf_radar_mlp_max_f32(C.byref(d), _stream()), "bevf_radar_mlp_max_f32")


def linear(x, w, bias, y, B: int, K: int, O: int, relu: bool, perm_inner: int = 0, perm_outer: int = 0):
    if x.numel() < B * K or w.numel() != O * K or y.numel() < B * O or (bias is not None and bias.numel() != O):
        raise BevfError("linear: buffer sizes do not match B,K,O")
    if w.dtype == torch.float32:
        _check(lib().bevf_linear_f32(_p(x), _pc(w), _pc(bias), _p(y), B, K, O, int(relu), perm_inner, perm_outer,
                                     _stream()), "bevf_linear_f32")
    else:
        _check(lib().bevf_linear_bf16w(_p(x), _pc(w, BF16), _pc(bias), _p(y, y.dtype), int(y.dtype == BF16), B, K, O,
                                       int(relu), perm_inner, perm_outer, _stream()), "bevf_linear_bf16w")


def cam_mean(x, y, B: int, ncam: int, P: int, Cc: int):
    if x.numel() < B * ncam * P * Cc or y.numel() < B * P * Cc:
        raise BevfError("cam_mean: buffer too small")
    fn = "bevf_cam_mean_" + _sfx(x)
    _check(getattr(lib(), fn)(_p(x, x.dtype), _p(y, x.dtype), B, ncam, P, Cc, _stream()), fn)


def bilinear_nhwc(x, y, B: int, Hi: int, Wi: int, Cc: int, x_cs: int, Ho: int, Wo: int, y_cs: int):
    if x.numel() < (B * Hi * Wi - 1) * x_cs + Cc or y.numel() < (B * Ho * Wo - 1) * y_cs + Cc:
        raise BevfError("bilinear: buffer too small")
    fn = "bevf_bilinear_nhwc_" + _sfx(x)
    _check(getattr(lib(), fn)(_p(x, x.dtype), _p(y, x.dtype), B, Hi, Wi, Cc, x_cs, Ho, Wo, y_cs, _stream()), fn)


def broadcast_nhwc(v, y, B: int, P: int, Cc: int, y_cs: int):
    if v.numel() < B * Cc or y.numel() < (B * P - 1) * y_cs + Cc:
        raise BevfError("broadcast: buffer too small")
    fn = "bevf_broadcast_nhwc_" + _sfx(y)
    _check(getattr(lib(), fn)(_p(v), _p(y, y.dtype), B, P, Cc, y_cs, _stream()), fn)


def expand_border_classes(small, y, B: int, Sh: int, Sw: int, Cc: int, y_cs: int):
    if small.numel() < B * 25 * Cc or y.numel() < (B * Sh * Sw - 1) * y_cs + Cc:
        raise BevfError("expand: buffer too small")
    fn = "bevf_expand_border_classes_" + _sfx(small)
    _check(getattr(lib(), fn)(_p(small, small.dtype), _p(y, small.dtype), B, Sh, Sw, Cc, y_cs, _stream()), fn)


def head_tail(hid, w, bias, outs: Sequence[torch.Tensor], B: int, P: int, hc: int, cs: Sequence[int], n_sigmoid: int):
    if hid.numel() < B * P * 5 * hc or w.numel() != sum(cs) * hc or bias.numel() != sum(cs):
        raise BevfError("head_tail: buffer sizes wrong")
    d = HeadDesc()
    d.hid, d.w, d.bias, d.B, d.P, d.hc, d.n_sigmoid = _p(hid, hid.dtype), _pc(w), _pc(bias), B, P, hc, n_sigmoid
    for k in range(5):
        if outs[k].numel() != B * cs[k] * P:
            raise BevfError("head_tail: output size wrong")
        d.out[k], d.c[k] = _pc(outs[k]), cs[k]
    fn = "bevf_head_tail_" + _sfx(hid)
    _check(getattr(lib(), fn)(C.byref(d), _stream()), fn)


def nchw_to_nhwc(x, y, N: int, Cc: int, P: int, y_cs: int):
    if x.numel() < N * Cc * P or y.numel() < (N * P - 1) * y_cs + Cc:
        raise BevfError("nchw_to_nhwc: buffer too small")
    _check(lib().bevf_nchw_to_nhwc_f32(_pc(x), _p(y), N, Cc, P, y_cs, _stream()), "bevf_nchw_to_nhwc_f32")


def nhwc_to_nchw(x, y, N: int, Cc: int, P: int, x_cs: int):
    if x.numel() < (N * P - 1) * x_cs + Cc or y.numel() < N * Cc * P:
        raise BevfError("nhwc_to_nchw: buffer too small")
    _check(lib().bevf_nhwc_to_nchw_f32(_p(x), _pc(y), N, Cc, P, x_cs, _stream()), "bevf_nhwc_to_nchw_f32")


def fill(y: torch.Tensor, v: float):
    _check(lib().bevf_fill_f32(_p(y), float(v), y.numel(), _stream()), "bevf_fill_f32")


def centernet_decode(pred: dict, K: int, thresh: float, voxel: float, x_min: float, y_min: float,
                     true_labels: bool = False, raw_scores: bool = False, pool_ind: Optional[torch.Tensor] = None):
    heat = pred["heatmap"]
    B, Cn, H, W = heat.shape
    dev = heat.device
    for k, c in (("offset", 2), ("size", 3), ("rot", 2), ("vel", 2)):
        if tuple(pred[k].shape) != (B, c, H, W):
            raise BevfError(f"decode: {k} has shape {tuple(pred[k].shape)}, expected {(B, c, H, W)}")
    boxes = torch.empty(B, K, 7, device=dev)
    scores = torch.empty(B, K, device=dev)
    labels = torch.empty(B, K, dtype=torch.int64, device=dev)
    vels = torch.empty(B, K, 2, device=dev)
    count = torch.empty(B, dtype=torch.int32, device=dev)
    work = torch.empty(lib().bevf_centernet_decode_work_bytes(B, Cn, H, W, K), dtype=torch.uint8, device=dev)
    d = DecodeDesc(_pc(heat), _pc(pred["offset"]), _pc(pred["size"]), _pc(pred["rot"]), _pc(pred["vel"]),
                   _p(boxes), _p(scores), _p(labels, torch.int64), _p(vels), _p(count, torch.int32),
                   _p(work, torch.uint8), _p(pool_ind, torch.int64), B, Cn, H, W, K, int(true_labels),
                   int(raw_scores), thresh, voxel, x_min, y_min)
    _check(lib().bevf_centernet_decode_f32(C.byref(d), _stream()), "bevf_centernet_decode_f32")
    return boxes, scores, labels, vels, count


def nms_keep(heat: torch.Tensor, out: torch.Tensor, planes: int, H: int, W: int):
    if heat.numel() != planes * H * W or out.numel() != planes * H * W:
        raise BevfError("nms_keep: buffer sizes do not match planes,H,W")
    _check(lib().bevf_nms_keep_f32(_pc(heat), _pc(out), planes, H, W, _stream()), "bevf_nms_keep_f32")


def centernet_targets(boxes, labels, has_vel, out: dict, B: int, nmax: int, H: int, W: int, Cn: int,
                      max_objects: int, pc_range, overlap: float, min_radius: int):
    dev = boxes.device
    if tuple(boxes.shape) != (B, nmax, 9) or tuple(labels.shape) != (B, nmax) or has_vel.numel() != B:
        raise BevfError("targets: boxes must be (B,nmax,9), labels (B,nmax), has_vel (B,)")
    want = dict(heatmap=(B, Cn, H, W), offset=(B, 2, H, W), size=(B, 3, H, W), rot=(B, 2, H, W), vel=(B, 2, H, W),
                mask=(B, max_objects), ind=(B, max_objects), reg_mask=(B, max_objects),
                target_offset=(B, max_objects, 2), target_size=(B, max_objects, 3), target_rot=(B, max_objects, 2),
                target_vel=(B, max_objects, 2))
    for k, shp in want.items():
        if tuple(out[k].shape) != shp:
            raise BevfError(f"targets: output {k} has shape {tuple(out[k].shape)}, expected {shp}")
    owner = torch.zeros(B, H * W, dtype=torch.int32, device=dev)
    d = TargetsDesc()
    d.boxes, d.labels, d.has_vel = _pc(boxes), _pc(labels, torch.int32), _pc(has_vel, torch.int32)
    for k in ("heatmap", "offset", "size", "rot", "vel", "target_offset", "target_size", "target_rot", "target_vel"):
        setattr(d, k, _pc(out[k]))
    d.mask, d.reg_mask, d.ind = _pc(out["mask"], torch.uint8), _pc(out["reg_mask"], torch.uint8), _pc(out["ind"], torch.int64)
    d.owner_scratch = _pc(owner, torch.int32)
    d.B, d.nmax, d.H, d.W, d.C, d.max_objects, d.min_radius = B, nmax, H, W, Cn, max_objects, min_radius
    for i in range(6):
        d.pc_range[i] = float(pc_range[i])
    d.gaussian_overlap = overlap
    _check(lib().bevf_centernet_targets_f32(C.byref(d), _stream()), "bevf_centernet_targets_f32")


def centernet_loss(pred: dict, tgt: dict, weights) -> torch.Tensor:
    heat = pred["heatmap"]
    B, Cn, H, W = heat.shape
    K = tgt["ind"].shape[1]
    dev = heat.device
    if tuple(tgt["heatmap"].shape) != (B, Cn, H, W):
        raise BevfError("loss: target heatmap shape differs from the prediction")
    d = LossDesc()
    keep = []

    def f32(t):
        t = t.float().contiguous()
        keep.append(t)
        return _pc(t)
    d.pred_heatmap, d.tgt_heatmap = f32(heat), f32(tgt["heatmap"])
    for q, (name, c) in enumerate((("offset", 2), ("size", 3), ("rot", 2), ("vel", 2))):
        if tuple(pred[name].shape) != (B, c, H, W) or tuple(tgt["target_" + name].shape) != (B, K, c):
            raise BevfError(f"loss: {name} shapes wrong")
        d.pred_reg[q], d.tgt_reg[q] = f32(pred[name]), f32(tgt["target_" + name])
    ind = tgt["ind"].to(torch.int64).contiguous()
    rm = tgt["reg_mask"].to(torch.uint8).contiguous()
    if int(ind.numel()) != B * K or rm.numel() != B * K:
        raise BevfError("loss: ind / reg_mask must be (B,K)")
    work = torch.empty(lib().bevf_centernet_loss_work_floats(), device=dev)
    out = torch.empty(6, device=dev)
    d.ind, d.reg_mask, d.work, d.out = _pc(ind, torch.int64), _pc(rm, torch.uint8), _pc(work), _pc(out)
    d.B, d.C, d.H, d.W, d.K = B, Cn, H, W, K
    for i in range(5):
        d.weights[i] = float(weights[i])
    _check(lib().bevf_centernet_loss_f32(C.byref(d), _stream()), "bevf_centernet_loss_f32")
    return out


def centernet_decode_raw(pred: dict, K: int):
    """Top-K bookkeeping only (no keep mask): voxel 1, origin 0, zero offsets -> boxes[...,0:2] are the integer
    (x, y) cells; also returns each winner's position in the (C,K) pool."""
    heat = pred["heatmap"]
    pool_ind = torch.empty(heat.shape[0], K, dtype=torch.int64, device=heat.device)
    boxes, scores, labels, _, _ = centernet_decode(pred, K, -1.0, 1.0, 0.0, 0.0, False, True, pool_ind)
    return boxes, scores, labels, pool_ind


def scatter_voxels(features: torch.Tensor, coords: torch.Tensor, grid, num_voxels: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(B,Nv,C) features + (B,Nv,3) int64 (z,y,x) -> dense (B,C,D,H,W); last row wins on duplicates (ref encoders.py:407-410)."""
    if features.dim() != 3 or coords.shape != (features.shape[0], features.shape[1], 3):
        raise BevfError("scatter_voxels: features must be (B,Nv,C) and coords (B,Nv,3)")
    B, Nv, Cc = features.shape
    D, H, W = (int(g) for g in grid)
    out = torch.empty(B, Cc, D, H, W, device=features.device)
    owner = torch.empty(B * D * H * W, dtype=torch.int32, device=features.device)
    _check(lib().bevf_scatter_voxels_f32(_pc(features), _pc(coords, torch.int64), _pc(num_voxels, torch.int32),
                                         _p(owner, torch.int32), _p(out), B, Nv, Cc, D, H, W, _stream()), "bevf_scatter_voxels_f32")
    return out


def voxelize(points: torch.Tensor, pc_range, voxel_size, max_points: int, max_voxels: int):
    if points.dim() != 3 or points.shape[2] < 3:
        raise BevfError("voxelize: points must be (B, N, C>=3)")
    B, N, Cc = points.shape
    dev = points.device
    feats = torch.zeros(B, max_voxels, max_points, Cc, device=dev)
    coords = torch.zeros(B, max_voxels, 3, dtype=torch.int64, device=dev)
    npts = torch.zeros(B, max_voxels, dtype=torch.int32, device=dev)
    nvox = torch.zeros(B, dtype=torch.int32, device=dev)
    work = torch.empty(lib().bevf_voxelize_work_bytes(B, N), dtype=torch.uint8, device=dev)
    d = VoxelizeDesc(_pc(points), _p(feats), _p(coords, torch.int64), _p(npts, torch.int32), _p(nvox, torch.int32),
                     _p(work, torch.uint8), B, N, Cc, max_points, max_voxels)
    for i in range(6):
        d.pc_range[i] = float(pc_range[i])
    for i in range(3):
        d.voxel_size[i] = float(voxel_size[i])
    _check(lib().bevf_voxelize_f32(C.byref(d), _stream()), "bevf_voxelize_f32")
    return feats, coords, npts, nvox
