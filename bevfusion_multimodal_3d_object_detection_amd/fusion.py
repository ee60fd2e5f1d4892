"""Drop-in for the reference's `fusion` module (ref src/fusion.py), MI355X-native.

`FlexibleBEVFusion`, `CenterNetHead`, `FlexibleMultiModal3DDetector` and `create_detector`
keep the reference's signatures, attributes, error behaviour and state-dict keys; their
forward passes run on hand-written gfx950 kernels (engine.py -> libbevf_hip.so).
Only the `bev` fusion + `centernet` head path is built: it is the hot path of BASELINE.json.
The attention / late fusion classes and the MLP head (<= 3 tokens, negligible compute,
SURVEY.md section 2 "OUT OF SCOPE") are importable names that raise on construction.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import engine as E
from .encoders import (MultiRadarEncoder, PointNetLiDAREncoder, ResNetCameraEncoder, _cfg, load_config)  # noqa: F401


def _cbr(cin: int, cout: int, k: int) -> List[nn.Module]:
    return [nn.Conv2d(cin, cout, k, padding=k // 2), nn.BatchNorm2d(cout), nn.ReLU(inplace=True)]


class FlexibleBEVFusion(nn.Module):
    """ref src/fusion.py:46-327.

    camera: mean over cameras -> conv3x3+BN+ReLU -> conv1x1+BN+ReLU -> bilinear to (bev_h,bev_w)
    lidar : Linear-ReLU-Linear to a 128x25x25 canvas -> conv -> x2 bilinear -> conv (50x50)
    radar : Linear-ReLU, broadcast to every cell, 2 x conv3x3+BN+ReLU
    concat [camera, lidar, radar] -> 2 x conv3x3+BN+ReLU.
    Extension: for bev sizes other than 50x50 -- where the reference raises at the
    concat -- the LiDAR map is bilinearly resized like the camera map; the identity at 50x50.
    """

    def __init__(self, use_camera: Optional[bool] = None, use_lidar: Optional[bool] = None,
                 use_radar: Optional[bool] = None, camera_channels: Optional[int] = None,
                 lidar_channels: Optional[int] = None, radar_channels: Optional[int] = None,
                 bev_h: Optional[int] = None, bev_w: Optional[int] = None, bev_channels: Optional[int] = None,
                 pc_range: Optional[List[float]] = None, config: Optional[Dict] = None,
                 config_path: Optional[str] = None):
        super().__init__()
        config = _cfg(config, config_path)
        default_range = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
        if config is not None:
            mc = config.get("model", {})
            bc, dc = mc.get("bev_fusion", {}), config.get("dataset", {})
            self.use_camera = mc.get("use_camera", True) if use_camera is None else use_camera
            self.use_lidar = mc.get("use_lidar", True) if use_lidar is None else use_lidar
            self.use_radar = mc.get("use_radar", True) if use_radar is None else use_radar
            if camera_channels is None:
                camera_channels = mc.get("camera_encoder", {}).get("output_channels", 512)
            if lidar_channels is None:
                lidar_channels = mc.get("lidar_encoder", {}).get("feature_dim", 1024)
            if radar_channels is None:
                radar_channels = mc.get("radar_encoder", {}).get("feature_dim", 256)
            self.bev_h = bc.get("bev_h", dc.get("bev_h", 200)) if bev_h is None else bev_h
            self.bev_w = bc.get("bev_w", dc.get("bev_w", 200)) if bev_w is None else bev_w
            self.bev_channels = bc.get("bev_channels", 256) if bev_channels is None else bev_channels
            self.pc_range = dc.get("point_cloud_range", default_range) if pc_range is None else pc_range
        else:
            self.use_camera = True if use_camera is None else use_camera
            self.use_lidar = True if use_lidar is None else use_lidar
            self.use_radar = True if use_radar is None else use_radar
            camera_channels = 512 if camera_channels is None else camera_channels
            lidar_channels = 1024 if lidar_channels is None else lidar_channels
            radar_channels = 256 if radar_channels is None else radar_channels
            self.bev_h = 200 if bev_h is None else bev_h
            self.bev_w = 200 if bev_w is None else bev_w
            self.bev_channels = 256 if bev_channels is None else bev_channels
            self.pc_range = default_range if pc_range is None else pc_range
        self.num_modalities = sum([self.use_camera, self.use_lidar, self.use_radar])
        assert self.num_modalities > 0, "At least one modality must be enabled"
        bevc = self.bev_channels
        if self.use_camera:
            self.camera_proj = nn.Sequential(*_cbr(camera_channels, 512, 3), *_cbr(512, bevc, 1))
        if self.use_lidar:
            hidden, start = 128, 25
            self.lidar_init = nn.Sequential(nn.Linear(lidar_channels, 512), nn.ReLU(inplace=True),
                                            nn.Linear(512, hidden * start * start))
            self.lidar_upsample = nn.Sequential(
                *_cbr(hidden, hidden, 3), nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False),
                *_cbr(hidden, bevc, 3))
            self.lidar_start_size = start
        if self.use_radar:
            self.radar_proj = nn.Sequential(nn.Linear(radar_channels, bevc), nn.ReLU(inplace=True))
            self.radar_refine = nn.Sequential(*_cbr(bevc, bevc, 3), *_cbr(bevc, bevc, 3))
        self.bev_fusion = nn.Sequential(*_cbr(bevc * self.num_modalities, bevc * 2, 3), *_cbr(bevc * 2, bevc, 3))
        self._engine = None

    def _eng(self) -> E.FusionEngine:
        if self._engine is None:
            self._engine = E.FusionEngine(self)
        return self._engine

    def forward_nhwc(self, cam_nhwc, cam_geom, lidar_features, radar_features):
        """Internal fast path on NHWC camera features (no layout change)."""
        return self._eng().run(cam_nhwc, cam_geom, lidar_features, radar_features)

    def forward(self, camera_features: Optional[torch.Tensor] = None, lidar_features: Optional[torch.Tensor] = None,
                radar_features: Optional[torch.Tensor] = None) -> torch.Tensor:
        E.require_cuda(camera_features, lidar_features, radar_features)
        if self.training:
            from . import training
            if training.wants_train_path(self):         # used outside the detector in train mode: batch statistics + gradients
                return training.fusion_train_forward(self, camera_features, lidar_features, radar_features)
        with torch.no_grad():
            return self._forward_eval(camera_features, lidar_features, radar_features)

    def _forward_eval(self, camera_features, lidar_features, radar_features) -> torch.Tensor:
        cam_nhwc = cam_geom = None
        if self.use_camera and camera_features is not None:
            x = camera_features.float()
            if x.dim() == 5:
                B, n, Cc, H, W = x.shape
                cam_nhwc, cam_geom = E.to_nhwc(x.reshape(B * n, Cc, H, W)).to(self._eng().dtype), (B, n, H, W)
            else:
                B, Cc, H, W = x.shape
                cam_nhwc, cam_geom = E.to_nhwc(x).to(self._eng().dtype), (B, 1, H, W)
        out, B = self.forward_nhwc(cam_nhwc, cam_geom,
                                   lidar_features.float() if lidar_features is not None else None,
                                   radar_features.float() if radar_features is not None else None)
        return E.to_nchw(out, B, self.bev_channels, self.bev_h, self.bev_w)

    def get_config_str(self) -> str:
        return "+".join(n for n, u in (("camera", self.use_camera), ("lidar", self.use_lidar),
                                       ("radar", self.use_radar)) if u)

    def count_parameters(self) -> Dict[str, int]:
        cnt = lambda m: sum(p.numel() for p in m.parameters())
        out = {}
        if self.use_camera:
            out["camera_proj"] = cnt(self.camera_proj)
        if self.use_lidar:
            out["lidar_init"], out["lidar_upsample"] = cnt(self.lidar_init), cnt(self.lidar_upsample)
            out["lidar_total"] = out["lidar_init"] + out["lidar_upsample"]
        if self.use_radar:
            out["radar_proj"], out["radar_refine"] = cnt(self.radar_proj), cnt(self.radar_refine)
            out["radar_total"] = out["radar_proj"] + out["radar_refine"]
        out["bev_fusion"] = cnt(self.bev_fusion)
        out["total"] = cnt(self)
        return out


class _OutOfScope(nn.Module):
    _what = ""

    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError(
            f"{type(self).__name__}: {self._what} is outside the accelerated hot path of this build "
            "(SURVEY.md section 2: <= 3 tokens, negligible compute); use fusion_type='bev' with detection_head='centernet'.")


class SpatialReshaper(_OutOfScope):
    _what = "the broadcast reshaper of the attention path (ref src/fusion.py:333-372)"


class CrossModalAttention(_OutOfScope):
    _what = "cross-modal attention (ref src/fusion.py:375-470)"


class FlexibleAttentionFusion(_OutOfScope):
    _what = "attention fusion (ref src/fusion.py:473-650)"


class FlexibleLateFusion(_OutOfScope):
    _what = "late fusion (ref src/fusion.py:653-781)"


class MLPDetectionHead(_OutOfScope):
    _what = "the MLP head of the non-spatial fusions (ref src/fusion.py:886-939)"


class CenterNetHead(nn.Module):
    """ref src/fusion.py:788-884.  Five branches conv3x3(+bias)+ReLU+conv1x1; sigmoid on the heatmap inside
    the head.  Init: weights N(0, 0.001), biases 0, heatmap bias -ln 99 (ref :858-867)."""

    def __init__(self, in_channels: Optional[int] = None, num_classes: Optional[int] = None,
                 head_conv: Optional[int] = None, config: Optional[Dict] = None, config_path: Optional[str] = None):
        super().__init__()
        config = _cfg(config, config_path)
        if config is not None:
            hc = config.get("model", {}).get("centernet_head", {})
            in_channels = hc.get("in_channels", 256) if in_channels is None else in_channels
            self.num_classes = config.get("dataset", {}).get("num_classes", 10) if num_classes is None else num_classes
            head_conv = hc.get("head_conv", 64) if head_conv is None else head_conv
        else:
            in_channels = 256 if in_channels is None else in_channels
            self.num_classes = 10 if num_classes is None else num_classes
            head_conv = 64 if head_conv is None else head_conv
        for name, c in zip(E.HEAD_BRANCHES, (self.num_classes, 2, 3, 2, 2)):
            setattr(self, f"{name}_head", nn.Sequential(nn.Conv2d(in_channels, head_conv, 3, padding=1, bias=True),
                                                        nn.ReLU(inplace=True), nn.Conv2d(head_conv, c, 1, bias=True)))
        self._init_weights()
        self._engine = None

    def _init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, std=0.001)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        nn.init.constant_(self.heatmap_head[-1].bias, -math.log((1 - 0.01) / 0.01))

    def _eng(self) -> E.HeadEngine:
        if self._engine is None:
            self._engine = E.HeadEngine(self)
        return self._engine

    def forward_nhwc(self, bev_nhwc: torch.Tensor, B: int, H: int, W: int) -> Dict[str, torch.Tensor]:
        return self._eng().run(bev_nhwc, B, H, W)

    def forward(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        E.require_cuda(x)
        if self.training and torch.is_grad_enabled() and self._eng().dtype == torch.float32 \
                and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            from . import training                      # no BatchNorm in the head: same values as eval mode, plus a gradient path
            return training.head_train_forward(self, x)
        with torch.no_grad():
            B, _, H, W = x.shape
            return self.forward_nhwc(E.to_nhwc(x.float()).to(self._eng().dtype), B, H, W)


def _any_bn_training(module: nn.Module) -> bool:
    return any(isinstance(m, nn.modules.batchnorm._BatchNorm) and m.training for m in module.modules())


class FlexibleMultiModal3DDetector(nn.Module):
    """ref src/fusion.py:946-1141.  `model(camera_imgs, lidar_points, radar_points)` -> dict of
    heatmap (B,C,H,W post-sigmoid), offset, size, rot, vel.  `None` inputs are skipped."""

    def __init__(self, use_camera: Optional[bool] = None, use_lidar: Optional[bool] = None,
                 use_radar: Optional[bool] = None, num_classes: Optional[int] = None,
                 fusion_type: Optional[str] = None, detection_head: Optional[str] = None,
                 bev_h: Optional[int] = None, bev_w: Optional[int] = None, config: Optional[Dict] = None,
                 config_path: Optional[str] = None):
        super().__init__()
        config = _cfg(config, config_path)
        if config is not None:
            mc, dc = config.get("model", {}), config.get("dataset", {})
            self.use_camera = mc.get("use_camera", True) if use_camera is None else use_camera
            self.use_lidar = mc.get("use_lidar", True) if use_lidar is None else use_lidar
            self.use_radar = mc.get("use_radar", True) if use_radar is None else use_radar
            num_classes = dc.get("num_classes", 10) if num_classes is None else num_classes
            self.fusion_type = mc.get("fusion_type", "bev") if fusion_type is None else fusion_type
            self.detection_head_type = mc.get("detection_head", "centernet") if detection_head is None else detection_head
            bev_h = dc.get("bev_h", 50) if bev_h is None else bev_h
            bev_w = dc.get("bev_w", 50) if bev_w is None else bev_w
        else:
            self.use_camera = True if use_camera is None else use_camera
            self.use_lidar = True if use_lidar is None else use_lidar
            self.use_radar = True if use_radar is None else use_radar
            num_classes = 10 if num_classes is None else num_classes
            self.fusion_type = "bev" if fusion_type is None else fusion_type
            self.detection_head_type = "centernet" if detection_head is None else detection_head
            bev_h = 50 if bev_h is None else bev_h
            bev_w = 50 if bev_w is None else bev_w
        assert sum([self.use_camera, self.use_lidar, self.use_radar]) > 0, "At least one modality must be enabled"
        if self.use_camera:
            self.camera_encoder = (ResNetCameraEncoder(config=config) if config is not None
                                   else ResNetCameraEncoder(backbone="resnet18", pretrained=False))
        if self.use_lidar:
            self.lidar_encoder = (PointNetLiDAREncoder(config=config) if config is not None
                                  else PointNetLiDAREncoder(input_channels=4, feat_dim=1024))
        if self.use_radar:
            self.radar_encoder = (MultiRadarEncoder(config=config) if config is not None
                                  else MultiRadarEncoder(input_channels=7, feat_dim=256, num_radars=5))
        if self.fusion_type == "bev":
            self.fusion = FlexibleBEVFusion(use_camera=self.use_camera, use_lidar=self.use_lidar,
                                            use_radar=self.use_radar, bev_h=bev_h, bev_w=bev_w, config=config)
        elif self.fusion_type == "attention":
            self.fusion = FlexibleAttentionFusion()
        elif self.fusion_type == "late":
            self.fusion = FlexibleLateFusion()
        else:
            raise ValueError(f"Unknown fusion type: {self.fusion_type}")
        if self.detection_head_type == "centernet":
            self.det_head = CenterNetHead(in_channels=self.fusion.bev_channels, num_classes=num_classes, config=config)
        else:
            self.det_head = MLPDetectionHead()

    def forward(self, camera_imgs: Optional[torch.Tensor] = None, lidar_points: Optional[torch.Tensor] = None,
                radar_points: Optional[List[torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        if self.training and (torch.is_grad_enabled() or _any_bn_training(self)):
            # under no_grad a train-mode model still normalises with batch statistics and updates the running buffers, as torch does
            from . import training                      # train-mode BN + tape + hand-written backward (training.py)
            E.require_cuda(camera_imgs, lidar_points)
            return training.detector_train_forward(self, camera_imgs, lidar_points, radar_points)
        with torch.no_grad():
            return self._forward_inference(camera_imgs, lidar_points, radar_points)

    def _forward_inference(self, camera_imgs, lidar_points, radar_points) -> Dict[str, torch.Tensor]:
        cam = geom = lid = rad = None
        if self.use_camera and camera_imgs is not None:
            cam, geom = self.camera_encoder.forward_nhwc(camera_imgs)       # stays NHWC: no layout change
        if self.use_lidar and lidar_points is not None:
            lid = self.lidar_encoder._forward_eval(lidar_points)
        if self.use_radar and radar_points is not None:
            rad = self.radar_encoder._forward_eval(radar_points)
        fused, B = self.fusion.forward_nhwc(cam, geom, lid, rad)
        return self.det_head.forward_nhwc(fused, B, self.fusion.bev_h, self.fusion.bev_w)

    def get_config_str(self) -> str:
        return f"{self.fusion.get_config_str()}_{self.fusion_type}_{self.detection_head_type}"

    def make_graphed(self, camera_imgs=None, lidar_points=None, radar_points=None) -> "GraphedDetector":
        """Capture the inference forward for these input shapes into one hipGraph (launch-bound small batches)."""
        return GraphedDetector(self, camera_imgs, lidar_points, radar_points)


class GraphedDetector:
    """The eval-mode detector forward captured as a hipGraph: ~40 kernel launches replayed with one call.

    Every launch of the HIP path goes to torch's current stream, so `torch.cuda.graph` records them; inputs are
    copied into static buffers, outputs are static tensors that the next replay overwrites (clone to keep).
    Weights are baked in as of capture time: re-capture after a parameter update."""

    def __init__(self, model: "FlexibleMultiModal3DDetector", camera_imgs, lidar_points, radar_points):
        assert not model.training, "capture the inference forward: call model.eval() first"
        E.require_cuda(camera_imgs, lidar_points)
        self.model = model
        c = lambda t: t.clone() if t is not None else None
        self.static_in = (c(camera_imgs), c(lidar_points), [r.clone() for r in radar_points] if radar_points else None)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):                                   # packs weights, sizes workspaces, sets kernel attributes
                model._forward_inference(*self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.static_out = model._forward_inference(*self.static_in)

    @torch.no_grad()
    def __call__(self, camera_imgs=None, lidar_points=None, radar_points=None) -> Dict[str, torch.Tensor]:
        si, sp, sr = self.static_in
        for dst, src in ((si, camera_imgs), (sp, lidar_points)):
            if dst is not None and src is not None and src.data_ptr() != dst.data_ptr():
                dst.copy_(src)
        if sr is not None and radar_points is not None:
            for d, s_ in zip(sr, radar_points):
                if d.data_ptr() != s_.data_ptr():
                    d.copy_(s_)
        self.graph.replay()
        return self.static_out


def create_detector(modality_config: Optional[str] = None, fusion_type: Optional[str] = None,
                    detection_head: Optional[str] = None, num_classes: Optional[int] = None,
                    config: Optional[Dict] = None, config_path: Optional[str] = None,
                    **kwargs) -> FlexibleMultiModal3DDetector:
    """ref src/fusion.py:1148-1221.  modality_config: 'camera_only' | 'camera+lidar' | ... | 'all'; the
    flags are substring tests on the lower-cased, space-stripped string (ref :1197-1202)."""
    config = _cfg(config, config_path)
    if config is not None and modality_config is None:
        modality_config = config.get("model", {}).get("modality_config", "all")
    use_camera = use_lidar = use_radar = None
    if modality_config is not None:
        m = modality_config.lower().replace(" ", "")
        use_camera = "camera" in m or m == "all"
        use_lidar = "lidar" in m or m == "all"
        use_radar = "radar" in m or m == "all"
    return FlexibleMultiModal3DDetector(use_camera=use_camera, use_lidar=use_lidar, use_radar=use_radar,
                                        num_classes=num_classes, fusion_type=fusion_type,
                                        detection_head=detection_head, config=config, **kwargs)


def test_all_configurations():
    """ref src/fusion.py:1228-1330 -- the reference's PASS/FAIL sweep, restricted to the built (bev) path."""
    dev = torch.device("cuda:0")
    results = {}
    for mod in ("camera+lidar", "camera+lidar+radar"):
        try:
            model = create_detector(mod, "bev", "centernet").to(dev).eval()
            imgs = torch.randn(2, 3, 3, 448, 800, device=dev)
            pts = torch.randn(2, 34720, 4, device=dev)
            radars = [torch.randn(2, 125, 7, device=dev) for _ in range(5)] if "radar" in mod else None
            out = model(imgs, pts, radars)
            n = sum(p.numel() for p in model.parameters())
            print(f"PASS {model.get_config_str()}: {n:,} params, heatmap {tuple(out['heatmap'].shape)}")
            results[mod] = True
        except Exception as e:  # noqa: BLE001 - mirrors the reference's try/except report
            print(f"FAIL {mod}: {e}")
            results[mod] = False
    return results
