"""TEST INFRASTRUCTURE -- PyTorch-CPU fp32 restatement of the detector forward.

State-dict keys and arithmetic follow the reference module for module so that a
state dict moves freely between the reference, this oracle and the HIP product.
Pinned by tests/golden/*.npz (outputs of the imported reference; see oracle/__init__).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .resnet18 import resnet18


class CameraEncoder(nn.Module):
    """ref src/encoders.py:53-172 -- ResNet-18 conv1..layer3 (stride 16) + 1x1 256->512 proj."""

    def __init__(self, out_channels: int = 512):
        super().__init__()
        trunk = resnet18()
        self.conv1, self.bn1, self.relu, self.maxpool = trunk.conv1, trunk.bn1, trunk.relu, trunk.maxpool
        self.layer1, self.layer2, self.layer3 = trunk.layer1, trunk.layer2, trunk.layer3
        self.channel_proj = nn.Sequential(nn.Conv2d(256, 512, 1, bias=False), nn.BatchNorm2d(512),
                                          nn.ReLU(inplace=True))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        five_d = x.dim() == 5                                   # ref :145-150
        if five_d:
            b, n = x.shape[:2]
            x = x.reshape(b * n, *x.shape[2:])
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))   # ref :154-157
        x = self.layer3(self.layer2(self.layer1(x)))           # ref :159-161
        x = self.channel_proj(x)                               # ref :165
        if five_d:
            x = x.view(b, n, *x.shape[1:])
        return x


class PointMLPMax(nn.Module):
    """Shared per-point MLP (Conv1d k=1 + BN1d + ReLU per layer) followed by max over points.

    widths [64,128,256,512,1024] = PointNetLiDAREncoder (ref src/encoders.py:252-298);
    widths [32,64,128,256]       = RadarEncoder         (ref src/encoders.py:515-555).
    Zero-padded points are NOT masked (ref keeps them in the max).
    """

    def __init__(self, cin: int, widths: List[int], use_bn: bool = True):
        super().__init__()
        self.input_channels = cin
        c = cin
        for i, w in enumerate(widths, 1):
            setattr(self, f"conv{i}", nn.Conv1d(c, w, 1))
            setattr(self, f"bn{i}", nn.BatchNorm1d(w) if use_bn else nn.Identity())
            c = w
        self.depth = len(widths)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() == 3 and x.shape[2] == self.input_channels:   # layout sniff, ref :282-284
            x = x.transpose(1, 2)
        for i in range(1, self.depth + 1):
            x = F.relu(getattr(self, f"bn{i}")(getattr(self, f"conv{i}")(x)))
        return torch.max(x, 2)[0]


class MultiRadar(nn.Module):
    """ref src/encoders.py:575-661 -- one shared RadarEncoder, stack, concat->Linear | max | mean."""

    def __init__(self, cin: int = 7, feat: int = 256, num_radars: int = 5, fusion_method: str = "concat"):
        super().__init__()
        self.radar_encoder = PointMLPMax(cin, [32, 64, 128, feat])
        self.fusion_method = fusion_method
        if fusion_method == "concat":
            self.fusion_fc = nn.Linear(feat * num_radars, feat)

    def forward(self, radar_list: List[torch.Tensor]) -> torch.Tensor:
        f = torch.stack([self.radar_encoder(r) for r in radar_list], dim=1)
        if self.fusion_method == "concat":
            return self.fusion_fc(f.view(f.shape[0], -1))
        if self.fusion_method == "max":
            return f.max(dim=1)[0]
        if self.fusion_method == "mean":
            return f.mean(dim=1)
        raise ValueError(f"Unknown fusion method: {self.fusion_method}")


def _cbr(cin, cout, k):
    return [nn.Conv2d(cin, cout, k, padding=k // 2), nn.BatchNorm2d(cout), nn.ReLU(inplace=True)]


class BEVFusion(nn.Module):
    """ref src/fusion.py:62-297 (FlexibleBEVFusion).

    One documented extension: the reference's LiDAR branch always
    emits 50x50 and crashes at the concat for any other BEV size; here the 50x50 map
    is bilinearly resized (align_corners=False, as the camera branch at ref :242-247)
    to (bev_h, bev_w) when they differ -- the identity at 50x50.
    """

    def __init__(self, use_camera=True, use_lidar=True, use_radar=True, camera_channels=512,
                 lidar_channels=1024, radar_channels=256, bev_h=50, bev_w=50, bev_channels=256):
        super().__init__()
        self.use_camera, self.use_lidar, self.use_radar = use_camera, use_lidar, use_radar
        self.bev_h, self.bev_w, self.bev_channels = bev_h, bev_w, bev_channels
        n_mod = int(use_camera) + int(use_lidar) + int(use_radar)
        assert n_mod > 0, "At least one modality must be enabled"
        if use_camera:
            self.camera_proj = nn.Sequential(*_cbr(camera_channels, 512, 3), *_cbr(512, bev_channels, 1))
        if use_lidar:
            self.lidar_start_size = 25
            self.lidar_init = nn.Sequential(nn.Linear(lidar_channels, 512), nn.ReLU(inplace=True),
                                            nn.Linear(512, 128 * 25 * 25))
            self.lidar_upsample = nn.Sequential(
                *_cbr(128, 128, 3), nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False),
                *_cbr(128, bev_channels, 3))
        if use_radar:
            self.radar_proj = nn.Sequential(nn.Linear(radar_channels, bev_channels), nn.ReLU(inplace=True))
            self.radar_refine = nn.Sequential(*_cbr(bev_channels, bev_channels, 3),
                                              *_cbr(bev_channels, bev_channels, 3))
        self.bev_fusion = nn.Sequential(*_cbr(bev_channels * n_mod, bev_channels * 2, 3),
                                        *_cbr(bev_channels * 2, bev_channels, 3))

    def forward(self, camera_features=None, lidar_features=None, radar_features=None) -> torch.Tensor:
        maps = []
        size = (self.bev_h, self.bev_w)
        if self.use_camera and camera_features is not None:
            cam = camera_features.mean(dim=1) if camera_features.dim() == 5 else camera_features
            cam = self.camera_proj(cam)
            maps.append(F.interpolate(cam, size=size, mode="bilinear", align_corners=False))
        if self.use_lidar and lidar_features is not None:
            b = lidar_features.shape[0]
            lid = self.lidar_init(lidar_features).view(b, 128, 25, 25)
            lid = self.lidar_upsample(lid)
            if tuple(lid.shape[-2:]) != size:                      # extension, see class doc
                lid = F.interpolate(lid, size=size, mode="bilinear", align_corners=False)
            maps.append(lid)
        if self.use_radar and radar_features is not None:
            b = radar_features.shape[0]
            r = self.radar_proj(radar_features).view(b, self.bev_channels, 1, 1)
            maps.append(self.radar_refine(r.expand(b, self.bev_channels, *size)))
        if not maps:
            raise ValueError("No modality features provided")
        return self.bev_fusion(torch.cat(maps, dim=1))


class CenterHead(nn.Module):
    """ref src/fusion.py:794-884 (CenterNetHead): 5 x [3x3 conv, ReLU, 1x1 conv]; sigmoid on heatmap."""

    BRANCHES = (("heatmap", None), ("offset", 2), ("size", 3), ("rot", 2), ("vel", 2))

    def __init__(self, in_channels=256, num_classes=10, head_conv=64):
        super().__init__()
        for name, c in self.BRANCHES:
            c = num_classes if c is None else c
            setattr(self, f"{name}_head", nn.Sequential(
                nn.Conv2d(in_channels, head_conv, 3, padding=1), nn.ReLU(inplace=True),
                nn.Conv2d(head_conv, c, 1)))
        for m in self.modules():                                  # ref :858-867
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, std=0.001)
                nn.init.constant_(m.bias, 0)
        nn.init.constant_(self.heatmap_head[-1].bias, -math.log((1 - 0.01) / 0.01))

    def forward(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        out = {name: getattr(self, f"{name}_head")(x) for name, _ in self.BRANCHES}
        out["heatmap"] = torch.sigmoid(out["heatmap"])
        return out


class Detector(nn.Module):
    """ref src/fusion.py:964-1137 (FlexibleMultiModal3DDetector, bev + centernet path)."""

    def __init__(self, use_camera=True, use_lidar=True, use_radar=True, num_classes=10,
                 bev_h=50, bev_w=50, lidar_channels_in=4, radar_channels_in=7, num_radars=5,
                 radar_fusion="concat"):
        super().__init__()
        self.use_camera, self.use_lidar, self.use_radar = use_camera, use_lidar, use_radar
        if use_camera:
            self.camera_encoder = CameraEncoder()
        if use_lidar:
            self.lidar_encoder = PointMLPMax(lidar_channels_in, [64, 128, 256, 512, 1024])
        if use_radar:
            self.radar_encoder = MultiRadar(radar_channels_in, 256, num_radars, radar_fusion)
        self.fusion = BEVFusion(use_camera, use_lidar, use_radar, bev_h=bev_h, bev_w=bev_w)
        self.det_head = CenterHead(256, num_classes)

    def forward(self, camera_imgs=None, lidar_points=None, radar_points=None) -> Dict[str, torch.Tensor]:
        cam = self.camera_encoder(camera_imgs) if self.use_camera and camera_imgs is not None else None
        lid = self.lidar_encoder(lidar_points) if self.use_lidar and lidar_points is not None else None
        rad = self.radar_encoder(radar_points) if self.use_radar and radar_points is not None else None
        return self.det_head(self.fusion(cam, lid, rad))


def make_detector(modality: str, bev_h: int = 50, bev_w: int = 50, **kw) -> Detector:
    m = modality.lower().replace(" ", "")                       # ref src/fusion.py:1197-1202
    return Detector("camera" in m or m == "all", "lidar" in m or m == "all",
                    "radar" in m or m == "all", bev_h=bev_h, bev_w=bev_w, **kw)


class VFE(nn.Module):
    """ref src/encoders.py:423-455 (VFELayer): Linear -> BN1d over all B*Nv*P rows -> ReLU -> max over P."""

    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.linear = nn.Linear(cin, cout)
        self.bn = nn.BatchNorm1d(cout)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        b, nv, p, c = x.shape
        y = self.bn(self.linear(x.reshape(b * nv * p, c)))
        return F.relu(y).view(b * nv, p, -1).max(dim=1)[0].view(b, nv, -1)
