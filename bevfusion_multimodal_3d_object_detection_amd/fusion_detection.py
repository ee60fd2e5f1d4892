"""Drop-in for the reference's `fusion_detection` module (ref src/fusion_detection.py).

The reference keeps this file as a legacy duplicate of fusion.py; what its drivers still import from it
is the post-processing (`eval.py` / `inference.py` call this decode, with 0.512 m cells -- wrong for a
50x50 grid, reproduced as is) and `CenterNetHead`.  Those run on device here.  The older fusion classes
(`BEVFusion`, `AttentionFusion`, `LateFusion`, `AnchorBasedHead`, `MultiModal3DDetector` -- the last of
which imports a module that does not exist in the reference, ref :593) and the unused `DetectionLoss` are
importable names that raise on construction (out of scope as compute).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn as nn

from . import _lib as L
from . import centernet_target as _ct
from .fusion import CenterNetHead, _OutOfScope  # noqa: F401  (same head; ref :376-473 duplicates fusion.py)


def decode_centernet_predictions(predictions: Dict[str, torch.Tensor], score_thresh: float = 0.3,
                                 max_detections: int = 100, true_labels: bool = False) -> List[Dict[str, torch.Tensor]]:
    """ref src/fusion_detection.py:695-780 (voxel_size 0.512)."""
    return _ct._decode(predictions, score_thresh, max_detections, 0.512, true_labels)


_nms = _ct._nms


def _topk(scores: torch.Tensor, K: int = 100) -> Tuple[torch.Tensor, ...]:
    """ref :792-820 (= src/centernet_target.py:424-452): per-class top-K of `scores` as given (no keep mask -- the
    caller applies `_nms` first), then top-K of the flattened (C,K) pool.  Returns (score, ind, classes, ys, xs)
    like the reference: `ind` indexes the C*K pool, `classes` is identically 0 there (index // (H*W) of an index
    that is already < H*W) and here.  Ties go to the lower index (torch.topk leaves their order unspecified)."""
    B, Cn, H, W = scores.shape
    z = torch.zeros(B, 2, H, W, device=scores.device)
    z3 = torch.zeros(B, 3, H, W, device=scores.device)
    pred = {"heatmap": scores.float().contiguous(), "offset": z, "size": z3, "rot": z, "vel": z}
    boxes, sc, labels, pool_ind = L.centernet_decode_raw(pred, K)
    return sc, pool_ind, labels, boxes[..., 1].long(), boxes[..., 0].long()


class BEVFusion(_OutOfScope):
    _what = "the legacy BEVFusion (ref src/fusion_detection.py:18-120; superseded by fusion.FlexibleBEVFusion)"


class CrossModalAttention(_OutOfScope):
    _what = "legacy cross-modal attention (ref src/fusion_detection.py:123-200)"


class AttentionFusion(_OutOfScope):
    _what = "legacy attention fusion (ref src/fusion_detection.py:203-290)"


class LateFusion(_OutOfScope):
    _what = "legacy late fusion (ref src/fusion_detection.py:293-369)"


class AnchorBasedHead(_OutOfScope):
    _what = "the anchor-based head (ref src/fusion_detection.py:476-560)"


class MultiModal3DDetector(_OutOfScope):
    _what = "the legacy detector (ref src/fusion_detection.py:563-688; imports a non-existent module at :593)"


class DetectionLoss(_OutOfScope):
    _what = "the MLP-path DetectionLoss (ref src/fusion_detection.py:827-940; unused by the bev path)"
