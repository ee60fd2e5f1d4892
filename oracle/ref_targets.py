"""TEST INFRASTRUCTURE -- CPU restatement of CenterNet targets / loss / decode.

Follows ref src/centernet_target.py (targets :118-324, decode :326-452, loss :455-622)
and the second decode variant in ref src/fusion_detection.py:695-820 (0.512 m cells).
The reference's quirks are reproduced on purpose: the loss applies
sigmoid to an already-sigmoided heatmap; decode labels are identically 0.
Pinned by tests/golden/targets_*.npz, loss_*.npz, decode_*.npz (imported reference).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

PC_RANGE = (-51.2, -51.2, -5.0, 51.2, 51.2, 3.0)


def gaussian_radius(height, width, min_overlap: float = 0.7):
    """ref src/centernet_target.py:128-150 -- smallest of the three quadratic roots.
    Operation order and operand types are kept exactly (inputs are numpy float32 scalars when
    the boxes are float32, so under numpy>=2 the arithmetic runs in float32; the integer
    radius derived from it must match bit for bit)."""
    a1 = 1
    b1 = (height + width)
    c1 = width * height * (1 - min_overlap) / (1 + min_overlap)
    r1 = (b1 + np.sqrt(b1 ** 2 - 4 * a1 * c1)) / 2
    a2 = 4
    b2 = 2 * (height + width)
    c2 = (1 - min_overlap) * width * height
    r2 = (b2 + np.sqrt(b2 ** 2 - 4 * a2 * c2)) / 2
    a3 = 4 * min_overlap
    b3 = -2 * min_overlap * (height + width)
    c3 = (min_overlap - 1) * width * height
    r3 = (b3 + np.sqrt(b3 ** 2 - 4 * a3 * c3)) / 2
    return min(r1, r2, r3)


def splat_gaussian(hm: np.ndarray, cx: int, cy: int, radius: int) -> None:
    """ref :118-125,152-168 -- exp(-(dx^2+dy^2)/(2 sigma^2)), sigma=(2r+1)/6, tiny values zeroed,
    merged into `hm` with element-wise max, clipped at the borders."""
    d = 2 * radius + 1
    sigma = d / 6
    ax = np.arange(-radius, radius + 1, dtype=np.float64)
    g = np.exp(-(ax[None, :] ** 2 + ax[:, None] ** 2) / (2 * sigma * sigma))
    g[g < np.finfo(g.dtype).eps * g.max()] = 0
    H, W = hm.shape
    l, r = min(cx, radius), min(W - cx, radius + 1)
    t, b = min(cy, radius), min(H - cy, radius + 1)
    dst = hm[cy - t:cy + b, cx - l:cx + r]
    src = g[radius - t:radius + b, radius - l:radius + r]
    if min(src.shape) > 0 and min(dst.shape) > 0:
        np.maximum(dst, src, out=dst)


def make_targets(gt_boxes: Sequence, gt_labels: Sequence, bev_size: Tuple[int, int] = (50, 50),
                 num_classes: int = 10, max_objects: int = 500, overlap: float = 0.7,
                 min_radius: int = 2, pc_range=PC_RANGE) -> Dict[str, torch.Tensor]:
    """ref src/centernet_target.py:170-324.  All float arithmetic in float64 (numpy on python
    floats / float32 inputs promoted by numpy exactly as the reference does), results stored fp32."""
    H, W = bev_size
    B = len(gt_boxes)
    f32 = lambda *s: torch.zeros(*s, dtype=torch.float32)
    out = dict(heatmap=f32(B, num_classes, H, W), offset=f32(B, 2, H, W), size=f32(B, 3, H, W),
               rot=f32(B, 2, H, W), vel=f32(B, 2, H, W),
               mask=torch.zeros(B, max_objects, dtype=torch.uint8),
               ind=torch.zeros(B, max_objects, dtype=torch.long),
               reg_mask=torch.zeros(B, max_objects, dtype=torch.uint8),
               target_offset=f32(B, max_objects, 2), target_size=f32(B, max_objects, 3),
               target_rot=f32(B, max_objects, 2), target_vel=f32(B, max_objects, 2))
    x_min, y_min, _, x_max, y_max, _ = pc_range
    vx, vy = (x_max - x_min) / W, (y_max - y_min) / H
    for b in range(B):
        boxes = gt_boxes[b].cpu().numpy() if isinstance(gt_boxes[b], torch.Tensor) else np.asarray(gt_boxes[b])
        labels = gt_labels[b].cpu().numpy() if isinstance(gt_labels[b], torch.Tensor) else np.asarray(gt_labels[b])
        hm = out["heatmap"][b].numpy()
        for k in range(min(len(boxes), max_objects)):
            cls = int(labels[k])
            if cls < 0 or cls >= num_classes:
                continue
            x, y, z, w, l, h, yaw = boxes[k][:7]
            px, py = (x - x_min) / vx, (y - y_min) / vy
            if px < 0 or px >= W or py < 0 or py >= H:
                continue
            cx, cy = int(px), int(py)
            if cx < 0 or cx >= W or cy < 0 or cy >= H:
                continue
            radius = max(min_radius, int(gaussian_radius(l / vy, w / vx, overlap)))
            splat_gaussian(hm[cls], cx, cy, radius)     # float64 gaussian max-merged into fp32 storage
            out["ind"][b, k] = cy * W + cx
            out["mask"][b, k] = 1
            out["reg_mask"][b, k] = 1
            off = (px - cx, py - cy)
            sc = (np.sin(yaw), np.cos(yaw))
            for key, dense, vals in (("target_offset", "offset", off), ("target_size", "size", (w, l, h)),
                                     ("target_rot", "rot", sc)):
                t = torch.tensor(np.array(vals), dtype=torch.float32) if key != "target_offset" \
                    else torch.tensor(np.array(vals)).float()
                out[key][b, k] = t
                out[dense][b, :, cy, cx] = t
            if boxes.shape[1] > 7:
                t = torch.tensor(np.array(boxes[k][7:9]), dtype=torch.float32)
                out["target_vel"][b, k] = t
                out["vel"][b, :, cy, cx] = t
    return out


def focal_loss(pred: torch.Tensor, target: torch.Tensor, alpha: float = 2.0, beta: float = 4.0) -> torch.Tensor:
    """ref :544-582 -- NOTE sigmoid applied again to the (already sigmoided) head output."""
    p = torch.clamp(torch.sigmoid(pred), min=1e-4, max=1 - 1e-4)
    pos = target.eq(1).float()
    neg = target.lt(1).float()
    pos_loss = (torch.log(p) * torch.pow(1 - p, alpha) * pos).sum()
    neg_loss = (torch.log(1 - p) * torch.pow(p, alpha) * torch.pow(1 - target, beta) * neg).sum()
    n_pos = pos.sum()
    return -neg_loss if n_pos == 0 else -(pos_loss + neg_loss) / n_pos


def gather_l1(pred: torch.Tensor, target: torch.Tensor, ind: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """ref :584-622 -- gather (B,C,H,W) at ind, |.-target| * mask, / (mask.sum() + 1e-4)."""
    B, C = pred.shape[:2]
    g = pred.view(B, C, -1).permute(0, 2, 1).gather(1, ind.unsqueeze(2).expand(B, ind.shape[1], C))
    m = mask.unsqueeze(2).expand_as(target).float()
    return (torch.abs(g - target) * m).sum() / (m.sum() + 1e-4)


LOSS_WEIGHTS = dict(heatmap=1.0, offset=1.0, size=1.0, rot=1.0, vel=0.1)   # ref :460-467


def centernet_loss(pred: Dict[str, torch.Tensor], tgt: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """ref :476-542."""
    out = {"heatmap_loss": focal_loss(pred["heatmap"], tgt["heatmap"])}
    for k in ("offset", "size", "rot", "vel"):
        out[f"{k}_loss"] = gather_l1(pred[k], tgt[f"target_{k}"], tgt["ind"], tgt["reg_mask"])
    out["total_loss"] = sum(LOSS_WEIGHTS[k] * out[f"{k}_loss"] for k in LOSS_WEIGHTS)
    return out


def decode(pred: Dict[str, torch.Tensor], score_thresh: float = 0.3, max_detections: int = 100,
           voxel_size: float = 2.048) -> List[Dict[str, torch.Tensor]]:
    """ref src/centernet_target.py:326-452 (voxel_size 2.048) and src/fusion_detection.py:695-820
    (voxel_size 0.512).  3x3 max-pool keep mask, per-class top-K, then top-K of the C*K pool.
    `labels` reproduces the reference bug: class = per-class index // (H*W) == 0 always."""
    heat = pred["heatmap"]
    B, C, H, W = heat.shape
    K = max_detections
    keep = (F.max_pool2d(heat, 3, stride=1, padding=1) == heat).float()
    heat = heat * keep
    s1, i1 = torch.topk(heat.view(B, C, -1), K, dim=2)
    cls = i1 // (H * W)
    i1 = i1 % (H * W)
    ys, xs = i1 // W, i1 % W
    s2, i2 = torch.topk(s1.view(B, -1), K, dim=1)
    cls = torch.gather(cls.view(B, -1), 1, i2)
    ys = torch.gather(ys.view(B, -1), 1, i2)
    xs = torch.gather(xs.view(B, -1), 1, i2)
    dets = []
    for b in range(B):
        m = s2[b] > score_thresh
        if m.sum() == 0:
            dets.append(dict(boxes=torch.zeros(0, 7), scores=torch.zeros(0),
                             labels=torch.zeros(0, dtype=torch.long), velocities=torch.zeros(0, 2)))
            continue
        by, bx = ys[b][m], xs[b][m]
        g = lambda t: t[b][:, by, bx].T
        off, size, rot, vel = g(pred["offset"]), g(pred["size"]), g(pred["rot"]), g(pred["vel"])
        wx = (bx.float() + off[:, 0]) * voxel_size + PC_RANGE[0]
        wy = (by.float() + off[:, 1]) * voxel_size + PC_RANGE[1]
        wz = torch.zeros_like(wx) - 1.0
        yaw = torch.atan2(rot[:, 0], rot[:, 1])
        dets.append(dict(boxes=torch.stack([wx, wy, wz, size[:, 0], size[:, 1], size[:, 2], yaw], dim=1),
                         scores=s2[b][m], labels=cls[b][m], velocities=vel))
    return dets
