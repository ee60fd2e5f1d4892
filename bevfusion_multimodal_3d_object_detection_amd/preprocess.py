"""Input pipeline on the device: what `NuScenesDetectionDataset._load_camera_images` and
`_load_lidar_points` / `_pad_or_subsample` do on the host in the reference (ref src/train_detect.py:123-189).

* camera: uint8 HWC frames -> `T.Resize((448, 800))` -> `T.ToTensor()` -> `T.Normalize(mean, std)` -> (n,3,448,800) fp32.
  torchvision resizes a PIL image with Pillow's antialiased bilinear `Image.resize`: a separable triangle filter
  whose support grows with the down-scale factor, evaluated in 22-bit fixed point with a uint8 intermediate after
  the horizontal pass.  `resample_tables` restates Pillow's coefficient computation (published algorithm of
  Pillow's libImaging/Resample.c; pinned in the tests against the installed Pillow itself), the kernel evaluates the
  two passes in the same integer arithmetic: the resized uint8 image is bit-identical to Pillow's, and the
  normalisation uses the same fp32 operations (x/255, -mean, /std).
* LiDAR: range filter with strict inequalities, order-preserving compaction, zero padding to `max_points`; when more
  points survive than fit, the reference draws `np.random.choice(N, max_points, replace=False)` -- pass those indices
  as `choice` (parity), or get the first `max_points` survivors (deterministic, documented deviation).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L

PRECISION_BITS = 32 - 8 - 2
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
_TABLES = {}


def resample_tables(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """Pillow's bilinear resampling coefficients for one axis: bounds (out,2) int32 = (first source index, count)
    and the 22-bit fixed-point weights (out, ksize) int32.  Same double-precision operations, in the same order, as
    Pillow's precompute_coeffs / normalize_coeffs_8bpc (box = the whole axis)."""
    scale = float(in_size) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale                                   # bilinear: support 1
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = []
        ww = 0.0
        for x in range(xmax):
            t = (x + xmin - center + 0.5) * ss
            if t < 0.0:
                t = -t
            v = 1.0 - t if t < 1.0 else 0.0
            w.append(v)
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _tables(in_size: int, out_size: int, dev):
    key = (in_size, out_size, str(dev))
    t = _TABLES.get(key)
    if t is None:
        b, k, ks = resample_tables(in_size, out_size)
        t = (torch.from_numpy(b).to(dev), torch.from_numpy(k).to(dev), ks)
        _TABLES[key] = t
    return t


def preprocess_camera_images(imgs: torch.Tensor, size: Tuple[int, int] = (448, 800),
                             mean: Sequence[float] = IMAGENET_MEAN, std: Sequence[float] = IMAGENET_STD) -> torch.Tensor:
    """uint8 (..., H, W, 3) cuda frames -> fp32 (..., 3, size[0], size[1]), resized like PIL and normalised
    (ref src/train_detect.py:127-143)."""
    if imgs.dtype != torch.uint8 or imgs.shape[-1] != 3 or imgs.dim() < 3:
        raise L.BevfError("preprocess_camera_images: expected uint8 (..., H, W, 3)")
    if not imgs.is_cuda:
        raise L.BevfError("HIP path needs CUDA/HIP tensors; got a CPU tensor (no CPU fallback in this package)")
    lead = tuple(imgs.shape[:-3])
    H, W = int(imgs.shape[-3]), int(imgs.shape[-2])
    x = imgs.reshape(-1, H, W, 3).contiguous()
    n = x.shape[0]
    Ho, Wo = size
    out = torch.empty(n, 3, Ho, Wo, device=x.device)
    bh, kh, ksh = _tables(W, Wo, x.device)
    bv, kv, ksv = _tables(H, Ho, x.device)
    m = (C.c_float * 3)(*mean)
    s = (C.c_float * 3)(*std)
    L._check(L.lib().bevf_resize_normalize_u8(x.data_ptr(), out.data_ptr(), n, H, W, Ho, Wo, bh.data_ptr(), kh.data_ptr(),
                                              ksh, bv.data_ptr(), kv.data_ptr(), ksv, m, s, L._stream()),
             "bevf_resize_normalize_u8")
    return out.reshape(*lead, 3, Ho, Wo)


def filter_pad_lidar(points: torch.Tensor, max_points: int = 35000,
                     pc_range: Sequence[float] = (-51.2, -51.2, -5.0, 51.2, 51.2, 3.0),
                     choice: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(N, C>=3) fp32 cuda sweep -> ((max_points, C) fp32, number of points inside the range (int32 scalar tensor)).
    ref src/train_detect.py:145-161, 181-189."""
    if points.dim() != 2 or points.shape[1] < 3 or points.dtype != torch.float32:
        raise L.BevfError("filter_pad_lidar: expected fp32 (N, C>=3)")
    if not points.is_cuda:
        raise L.BevfError("HIP path needs CUDA/HIP tensors; got a CPU tensor (no CPU fallback in this package)")
    pts = points.contiguous()
    N, Cc = pts.shape
    out = torch.empty(max_points, Cc, device=pts.device)
    count = torch.zeros(1, dtype=torch.int32, device=pts.device)
    work = torch.empty(max(N, 1) * Cc + (N + 1023) // 1024 + 64, device=pts.device)
    r = (C.c_float * 6)(*pc_range)
    ch = None
    if choice is not None:
        if choice.numel() != max_points:
            raise L.BevfError("filter_pad_lidar: `choice` must hold max_points indices")
        ch = choice.to(device=pts.device, dtype=torch.int64).contiguous()
    L._check(L.lib().bevf_lidar_filter_pad_f32(pts.data_ptr(), out.data_ptr(), count.data_ptr(), work.data_ptr(),
                                               None if ch is None else ch.data_ptr(), N, Cc, max_points, r, L._stream()),
             "bevf_lidar_filter_pad_f32")
    return out, count[0]
