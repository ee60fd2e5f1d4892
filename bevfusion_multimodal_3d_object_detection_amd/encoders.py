"""Drop-in for the reference's `encoders` module (ref src/encoders.py), MI355X-native.

Same class names, constructor signatures, attributes and state-dict keys; `forward` runs the
hand-written gfx950 kernels of libbevf_hip.so through engine.py.  Modules are parameter
containers (torch.nn layers are used only to hold tensors under the reference's key names).
"""
from __future__ import annotations

import warnings
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import yaml

from . import _lib as L
from . import engine as E


def load_config(config_path: str = "config.yaml") -> Dict:
    """ref src/encoders.py:16-33 -- yaml.safe_load; FileNotFoundError when the file is missing."""
    f = Path(config_path)
    if not f.exists():
        raise FileNotFoundError(f"Config file not found: {config_path}")
    with open(f, "r") as fh:
        return yaml.safe_load(fh)


def _cfg(config, config_path) -> Optional[Dict]:
    if config is None and config_path is not None:
        config = load_config(config_path)
    return config


# ---- ResNet-18 trunk parameter containers (torchvision layout: conv1,bn1,conv2,bn2[,downsample.0/.1]) ----

class _BasicBlockParams(nn.Module):
    def __init__(self, cin: int, cout: int, stride: int):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))


def _stage(cin: int, cout: int, stride: int) -> nn.Sequential:
    blocks = [_BasicBlockParams(cin, cout, stride), _BasicBlockParams(cout, cout, 1)]
    for b in blocks:                                   # torchvision's default init for the trunk
        nn.init.kaiming_normal_(b.conv1.weight, mode="fan_out", nonlinearity="relu")
        nn.init.kaiming_normal_(b.conv2.weight, mode="fan_out", nonlinearity="relu")
        if b.downsample is not None:
            nn.init.kaiming_normal_(b.downsample[0].weight, mode="fan_out", nonlinearity="relu")
    return nn.Sequential(*blocks)


class ResNetCameraEncoder(nn.Module):
    """ref src/encoders.py:36-189.  (B,6,3,H,W) or (B*6,3,H,W) -> (B,6,512,H/16,W/16) / (B*6,512,...)."""

    def __init__(self, backbone: Optional[str] = None, pretrained: Optional[bool] = None,
                 out_channels: Optional[int] = None, freeze_bn: Optional[bool] = None,
                 config: Optional[Dict] = None, config_path: Optional[str] = None):
        super().__init__()
        config = _cfg(config, config_path)
        if config is not None:
            c = config.get("model", {}).get("camera_encoder", {})
            self.backbone_name = c.get("backbone", "resnet18")
            pretrained = c.get("pretrained", True)
            freeze_bn = c.get("freeze_bn", False)
            self.out_channels = c.get("output_channels", 512)
            self.total_stride = c.get("total_stride", 16)
        else:
            self.backbone_name = backbone if backbone is not None else "resnet18"
            pretrained = True if pretrained is None else pretrained
            freeze_bn = False if freeze_bn is None else freeze_bn
            self.out_channels = out_channels if out_channels is not None else 512
            self.total_stride = 16
        if self.backbone_name != "resnet18":
            # the reference leaves `resnet` undefined for anything else (ref :97-99 -> UnboundLocalError)
            raise NotImplementedError(f"backbone '{self.backbone_name}': only 'resnet18' exists in the reference")
        if pretrained:
            warnings.warn("pretrained=True: ImageNet weights are a network fetch in the reference; this build never "
                          "downloads -- load them with load_state_dict() (keys are identical).", stacklevel=2)
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        nn.init.kaiming_normal_(self.conv1.weight, mode="fan_out", nonlinearity="relu")
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = _stage(64, 64, 1)
        self.layer2 = _stage(64, 128, 2)
        self.layer3 = _stage(128, 256, 2)
        self.channel_proj = nn.Sequential(nn.Conv2d(256, 512, 1, bias=False), nn.BatchNorm2d(512),
                                          nn.ReLU(inplace=True))
        if freeze_bn:
            self._freeze_bn()
        self._engine = None

    def _freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()
                for p in m.parameters():
                    p.requires_grad = False

    def _eng(self) -> E.CameraEncoderEngine:
        if self._engine is None:
            object.__setattr__(self, "_engine", E.CameraEncoderEngine(self))
        return self._engine

    def forward_nhwc(self, x: torch.Tensor):
        """Internal fast path: returns (NHWC feature buffer, (B, ncam, Hc, Wc))."""
        E.require_cuda(x)
        if x.dim() == 5:
            B, n = x.shape[:2]
            x = x.reshape(B * n, *x.shape[2:])
        else:
            B, n = x.shape[0], 1
        feat, hc, wc = self._eng().run(x.contiguous().float())
        return feat, (B, n, hc, wc)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training and _training().wants_train_path(self):
            # a freshly built encoder is in train mode (ref src/encoders.py:805, 829): batch statistics + a gradient path
            E.require_cuda(x)
            return _training().camera_encoder_train_forward(self, x)
        with torch.no_grad():
            return self._forward_eval(x)

    def _forward_eval(self, x: torch.Tensor) -> torch.Tensor:
        five_d = x.dim() == 5
        feat, (B, n, hc, wc) = self.forward_nhwc(x)
        out = E.to_nchw(feat, B * n, 512, hc, wc)
        return out.view(B, n, 512, hc, wc) if five_d else out

    def get_output_shape(self, input_height: int, input_width: int) -> Tuple[int, int, int]:
        return (self.out_channels, input_height // self.total_stride, input_width // self.total_stride)


def _training():
    from . import training            # train-mode BatchNorm + tape + hand-written backward; imported on first use
    return training


class _PointMLP(nn.Module):
    """conv{i}/bn{i} containers shared by the LiDAR and radar point encoders."""

    def _build(self, cin: int, widths: List[int], use_bn: bool):
        c = cin
        for i, w in enumerate(widths, 1):
            setattr(self, f"conv{i}", nn.Conv1d(c, w, 1))
            setattr(self, f"bn{i}", nn.BatchNorm1d(w) if use_bn else nn.Identity())
            c = w

    def _rows(self, x: torch.Tensor) -> torch.Tensor:
        """(B,N,C) or (B,C,N) -> contiguous (B,N,C), with the reference's layout sniff (ref :282-284)."""
        E.require_cuda(x)
        x = x.float()
        if x.dim() == 3 and x.shape[2] == self.input_channels:
            return x.contiguous()
        B, Cc, N = x.shape
        y = torch.empty(B, N, Cc, device=x.device)
        L.nchw_to_nhwc(x.contiguous(), y, B, Cc, N, Cc)
        return y


class PointNetLiDAREncoder(_PointMLP):
    """ref src/encoders.py:191-306.  (B,N,C)|(B,C,N) -> (B,feat_dim); padded points are not masked."""

    def __init__(self, input_channels: Optional[int] = None, feat_dim: Optional[int] = None,
                 use_bn: Optional[bool] = None, return_point_features: Optional[bool] = None,
                 config: Optional[Dict] = None, config_path: Optional[str] = None):
        super().__init__()
        config = _cfg(config, config_path)
        if config is not None:
            c = config.get("model", {}).get("lidar_encoder", {})
            self.input_channels = c.get("input_channels", 5)
            self.feat_dim = c.get("feature_dim", 1024)
            use_bn = c.get("use_batch_norm", True)
            self.return_point_features = False
            widths = c.get("mlp_layers", [64, 128, 256, 512, 1024])
        else:
            self.input_channels = input_channels if input_channels is not None else 5
            self.feat_dim = feat_dim if feat_dim is not None else 1024
            use_bn = True if use_bn is None else use_bn
            self.return_point_features = bool(return_point_features) if return_point_features is not None else False
            widths = [64, 128, 256, 512, 1024]
        self._build(self.input_channels, widths, use_bn)
        self._engine = None

    def _eng(self) -> E.PointNetEngine:
        if self._engine is None:
            object.__setattr__(self, "_engine", E.PointNetEngine(self))
        return self._engine

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training and _training().wants_train_path(self):
            E.require_cuda(x)
            return _training().pointnet_train_forward(self, x)
        with torch.no_grad():
            return self._forward_eval(x)

    def _forward_eval(self, x: torch.Tensor) -> torch.Tensor:
        rows = self._rows(x)
        B, N, _ = rows.shape
        g, last = self._eng().run(rows, keep_last=self.return_point_features)
        if self.return_point_features:                          # ref :300-304, (B,N,2*feat)
            f = g.shape[1]
            return torch.cat([last[:B * N * f].view(B, N, f), g.unsqueeze(1).expand(B, N, f)], dim=2)
        return g


class RadarEncoder(_PointMLP):
    """ref src/encoders.py:458-557."""

    def __init__(self, input_channels: Optional[int] = None, feat_dim: Optional[int] = None,
                 use_bn: Optional[bool] = None, config: Optional[Dict] = None, config_path: Optional[str] = None):
        super().__init__()
        config = _cfg(config, config_path)
        if config is not None:
            c = config.get("model", {}).get("radar_encoder", {})
            self.input_channels = c.get("input_channels", 7)
            self.feat_dim = c.get("feature_dim", 256)
            use_bn = c.get("use_batch_norm", True)
            widths = c.get("mlp_layers", [32, 64, 128, 256])
        else:
            self.input_channels = input_channels if input_channels is not None else 7
            self.feat_dim = feat_dim if feat_dim is not None else 256
            use_bn = True if use_bn is None else use_bn
            widths = [32, 64, 128, 256]
        self._build(self.input_channels, widths, use_bn)
        self._wrap = None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training and _training().wants_train_path(self):
            E.require_cuda(x)
            return _training().radar_train_forward(self, x)
        with torch.no_grad():
            return self._forward_eval(x)

    def _forward_eval(self, x: torch.Tensor) -> torch.Tensor:
        if self._wrap is None:
            object.__setattr__(self, "_wrap", _SingleRadar(self))
        return self._wrap(self._rows(x))


class _SingleRadar:
    """Runs one RadarEncoder through the fused radar kernel (R = 1, no fusion layer)."""

    class _Shim(nn.Module):
        def __init__(self, enc):
            super().__init__()
            self.radar_encoder = enc
            self.fusion_method = "max"

    def __init__(self, enc: RadarEncoder):
        self.engine = E.RadarEngine(self._Shim(enc))

    def __call__(self, rows: torch.Tensor) -> torch.Tensor:
        return self.engine.run([rows])


class MultiRadarEncoder(nn.Module):
    """ref src/encoders.py:560-661.  List of (B,N_i,C) sweeps -> (B,feat_dim)."""

    def __init__(self, input_channels: Optional[int] = None, feat_dim: Optional[int] = None,
                 num_radars: Optional[int] = None, fusion_method: Optional[str] = None,
                 config: Optional[Dict] = None, config_path: Optional[str] = None):
        super().__init__()
        config = _cfg(config, config_path)
        if config is not None:
            c = config.get("model", {}).get("radar_encoder", {})
            input_channels = c.get("input_channels", 7)
            self.feat_dim = c.get("feature_dim", 256)
            self.num_radars = c.get("num_radars", 5)
            self.fusion_method = c.get("fusion_method", "concat")
        else:
            input_channels = input_channels if input_channels is not None else 7
            self.feat_dim = feat_dim if feat_dim is not None else 256
            self.num_radars = num_radars if num_radars is not None else 5
            self.fusion_method = fusion_method if fusion_method is not None else "concat"
        self.radar_encoder = RadarEncoder(input_channels=input_channels, feat_dim=self.feat_dim, config=config)
        if self.fusion_method == "concat":
            self.fusion_fc = nn.Linear(self.feat_dim * self.num_radars, self.feat_dim)
        self.output_dim = self.feat_dim
        self._engine = None

    def _eng(self) -> E.RadarEngine:
        if self._engine is None:
            object.__setattr__(self, "_engine", E.RadarEngine(self))
        return self._engine

    def forward(self, radar_list: List[torch.Tensor]) -> torch.Tensor:
        if self.training and _training().wants_train_path(self):
            E.require_cuda(*radar_list)
            return _training().radar_train_forward(self, list(radar_list))
        with torch.no_grad():
            return self._forward_eval(radar_list)

    def _forward_eval(self, radar_list: List[torch.Tensor]) -> torch.Tensor:
        if self.fusion_method not in ("concat", "max", "mean"):
            raise ValueError(f"Unknown fusion method: {self.fusion_method}")
        rows = [self.radar_encoder._rows(r) for r in radar_list]
        return self._eng().run(rows)


def voxelize(points: torch.Tensor, point_cloud_range=(-51.2, -51.2, -5.0, 51.2, 51.2, 3.0),
             voxel_size=(2.048, 2.048, 8.0), max_points_per_voxel: int = 32, max_voxels: int = 12000):
    """Hard voxelisation feeding `VFELayer` / `VoxelNetLiDAREncoder` (their docstrings, ref src/encoders.py:313-321,
    describe exactly this input; the reference never builds it).  Defaults are the grid of configs/base.yaml:48-55
    (50 x 50 pillars) and the reference's "32 points, 12000 voxels".  points (B,N,C) ->
    voxel_features (B,Nv,P,C) zero padded, voxel_coords (B,Nv,3) int64 (z,y,x), num_points (B,Nv), num_voxels (B,)."""
    E.require_cuda(points)
    return L.voxelize(points.float().contiguous(), point_cloud_range, voxel_size, max_points_per_voxel, max_voxels)


def scatter_voxels(voxel_features: torch.Tensor, voxel_coords: torch.Tensor, voxel_grid_shape,
                   num_voxels: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The dense scatter of `VoxelNetLiDAREncoder.forward` (ref src/encoders.py:399-410) as a kernel: per-voxel features
    (B,Nv,C) at voxel_coords (B,Nv,3) = (z,y,x) -> (B,C,D,H,W), last row wins where rows collide, zeros elsewhere.  With
    `num_voxels` (from `voxelize`) the zero-padded rows stay out; without it every row is written, as the reference does.
    `voxelize` -> `VFELayer` -> `scatter_voxels(..., (1,50,50))[:, :, 0]` is the pillar canvas (B,C,50,50) of the config's grid."""
    E.require_cuda(voxel_features, voxel_coords)
    nv = None if num_voxels is None else num_voxels.to(torch.int32).contiguous()
    return L.scatter_voxels(voxel_features.float().contiguous(), voxel_coords.to(torch.int64).contiguous(), voxel_grid_shape, nv)


class VFELayer(nn.Module):
    """ref src/encoders.py:420-455 -- Linear -> BN1d over all B*Nv*P rows -> ReLU -> max over the P points
    of each voxel ("PointNet pillar reduction").  (B,Nv,P,C) -> (B,Nv,out_channels)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.linear = nn.Linear(in_channels, out_channels)
        self.bn = nn.BatchNorm1d(out_channels)
        self._engine = None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        E.require_cuda(x)
        B, Nv, P, Cc = x.shape                    # a 3-D input raises ValueError exactly like the reference
        if self.training and _training().wants_train_path(self):
            return _training().vfe_train_forward(self, x)          # batch statistics + gradients for linear / bn
        with torch.no_grad():
            if self._engine is None:
                object.__setattr__(self, "_engine", E.VFEEngine(self))
            return self._engine.run(x.contiguous().float()).view(B, Nv, self.out_channels)


class VoxelNetLiDAREncoder(nn.Module):
    """ref src/encoders.py:308-417.  Dead code in the reference (never imported by fusion.py) and its
    forward raises: vfe2 receives vfe1's 3-D output and cannot unpack 4 dims (ref :395-396 vs :438).
    The parameter layout is kept for checkpoint compatibility; forward reproduces that ValueError."""

    def __init__(self, input_channels: Optional[int] = None, voxel_feat_dim: Optional[int] = None,
                 output_feat_dim: Optional[int] = None, max_points_per_voxel: Optional[int] = None,
                 config: Optional[Dict] = None, config_path: Optional[str] = None):
        super().__init__()
        config = _cfg(config, config_path)
        if config is not None:
            c = config.get("model", {}).get("lidar_encoder", {})
            self.input_channels = c.get("input_channels", 5)
            self.voxel_feat_dim = 128
            self.output_feat_dim = c.get("feature_dim", 256)
        else:
            self.input_channels = input_channels if input_channels is not None else 5
            self.voxel_feat_dim = voxel_feat_dim if voxel_feat_dim is not None else 128
            self.output_feat_dim = output_feat_dim if output_feat_dim is not None else 256
        self.vfe1 = VFELayer(self.input_channels, self.voxel_feat_dim // 2)
        self.vfe2 = VFELayer(self.voxel_feat_dim // 2, self.voxel_feat_dim)
        self.conv3d_1 = nn.Conv3d(self.voxel_feat_dim, 128, 3, stride=2, padding=1)
        self.bn3d_1 = nn.BatchNorm3d(128)
        self.conv3d_2 = nn.Conv3d(128, 256, 3, stride=2, padding=1)
        self.bn3d_2 = nn.BatchNorm3d(256)
        self.conv3d_3 = nn.Conv3d(256, self.output_feat_dim, 3, stride=2, padding=1)
        self.bn3d_3 = nn.BatchNorm3d(self.output_feat_dim)

    def forward(self, voxel_features, voxel_coords, voxel_grid_shape):
        v = self.vfe1(voxel_features)
        B, N_voxels, N_points, Cc = v.shape       # raises ValueError: not enough values to unpack (as the reference)
        raise ValueError("unreachable")


def print_encoder_specs():
    """ref src/encoders.py:663-790 prints a static description of the encoders."""
    print("=" * 80)
    print("MULTI-MODAL ENCODER SPECIFICATIONS (MI355X-native build)")
    print("=" * 80)
    print("camera : ResNet-18 conv1..layer3 + 1x1 proj, (B,6,3,H,W) -> (B,6,512,H/16,W/16)")
    print("lidar  : PointNet shared MLP 64-128-256-512-1024 + max, (B,N,C) -> (B,1024)")
    print("radar  : 5 x shared MLP 32-64-128-256 + max, concat -> Linear(1280,256), -> (B,256)")
