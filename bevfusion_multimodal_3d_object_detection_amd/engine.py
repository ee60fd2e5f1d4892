"""Host-side execution engines: weight packing + kernel sequencing for the HIP path.

One engine per reference module (camera encoder, point MLPs, BEV fusion, CenterNet head).
An engine packs its module's parameters once per weight version (BatchNorm folded in fp64,
conv weights re-laid OIHW -> OHWI, the lidar_init rows left in place and permuted on store),
keeps grow-only device workspaces, and issues the C-ABI calls on torch's current stream.
Internally every activation is fp32 NHWC; NCHW exists only at the reference's API surface.
These engines are the eval-mode path (BatchNorm folded from its running statistics).  Training runs through
training.DetectorTape (train-mode BatchNorm + hand-written backward): the whole detector, and each camera / PointNet /
radar encoder, the BEV fusion and the head used on their own, dispatch there from their `forward` while in train mode.
An engine asked to fold a BatchNorm that is still in train mode (e.g. VFELayer, which has no tape) raises (_check_eval).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib as L


# ---- packing ---------------------------------------------------------------------------------------

def _bn_fold(bias: Optional[torch.Tensor], bn, cout: int, dev) -> Tuple[torch.Tensor, torch.Tensor]:
    """scale/shift with  y = conv_nobias(x) * scale + shift  ==  bn(conv(x) + bias)   (eval mode)."""
    b = torch.zeros(cout, dtype=torch.float64, device=dev) if bias is None else bias.detach().double()
    if bn is None or isinstance(bn, nn.Identity):
        return torch.ones(cout, device=dev), b.float().contiguous()
    g = bn.weight.detach().double() if bn.weight is not None else torch.ones(cout, dtype=torch.float64, device=dev)
    be = bn.bias.detach().double() if bn.bias is not None else torch.zeros(cout, dtype=torch.float64, device=dev)
    scale = g / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    shift = be + (b - bn.running_mean.detach().double()) * scale
    return scale.float().contiguous(), shift.float().contiguous()


@dataclass
class PackedConv:
    w: torch.Tensor          # OHWI, flat
    scale: Optional[torch.Tensor]
    shift: Optional[torch.Tensor]
    cin: int
    cout: int
    k: int
    stride: int
    pad: int
    relu: bool
    wino: bool = False       # w is the transformed-filter image of the fused Winograd kernel (csrc/conv_wino.hip)
    c3: bool = False         # w is the fragment-ordered filter image of the bf16 3x3 kernel (csrc/conv3x3_bf16.hip)


_CONV_MODE = "wino"
BF16_CONV3X3 = True          # bf16 models: 3x3 / stride 1 / pad 1 layers on csrc/conv3x3_bf16.hip (False: implicit GEMM everywhere, the round-2 path)


def set_conv_mode(mode: str) -> None:
    """How fp32 models multiply in their convolutions.  "wino" (default): the 3x3 / stride 1 / pad 1 layers
    (Cin % 32 == 0; ~85 % of the path's FLOPs) run as fused fp32 Winograd F(2x2,3x3) on v_mfma_f32_16x16x4_f32
    (csrc/conv_wino.hip: fp32 products and accumulation, 2.25x fewer of them; a few 1e-7 relative from the direct
    kernel per layer, the whole detector within 1e-4 of the oracle at full size -- tests/test_gpu_wino.py), every other
    layer as "f32".  "f32": v_mfma_f32_32x32x2_f32 everywhere, an exact fp32 FMA chain whose bits do not depend on
    tile shapes.  "f32x3": opt-in, fp32 operands split exactly into three bf16 planes and multiplied on the bf16 MFMA
    (six partial products, fp32 accumulate; csrc/conv_split.hip) -- fp32-level error, ~1.3-1.4x faster, not
    bit-identical to the default.  "wino_x3": opt-in, both reductions of MFMA work together -- Winograd where "wino" uses it,
    the three-plane kernel for every other layer (stride-2 3x3, 1x1 projections, PointNet incl. its fused point max).
    Engines repack on the next forward."""
    global _CONV_MODE
    if mode not in ("f32", "wino", "f32x3", "wino_x3"):
        raise ValueError(f"conv mode must be 'f32', 'wino', 'f32x3' or 'wino_x3', got {mode!r}")
    _CONV_MODE = mode


def conv_mode() -> str:
    return _CONV_MODE


def pack_conv(conv, bn=None, relu: bool = True, split_ok: bool = True, wino_ok: bool = True) -> PackedConv:
    """Weights keep the module's storage dtype (fp32 or bf16); the folded BN scale/shift are always fp32.
    In "f32x3" / "wino_x3" mode fp32 filters with Cin % 32 == 0 are stored as three bf16 planes (split_ok=False keeps a layer
    on the exact kernel)."""
    w = conv.weight.detach()
    if w.dim() == 3:                                   # Conv1d k=1 == pointwise
        w = w.unsqueeze(-1)
    cout, cin, kh, kw = w.shape
    assert kh == kw, "square kernels only"
    stride = conv.stride[0] if isinstance(conv.stride, tuple) else conv.stride
    pad = conv.padding[0] if isinstance(conv.padding, tuple) else conv.padding
    scale, shift = _bn_fold(conv.bias, bn, cout, w.device)
    packed = w.permute(0, 2, 3, 1).contiguous().view(-1)
    return _finish_pack(packed, scale, shift, cin, cout, kh, stride, pad, relu, split_ok, wino_ok)


def _finish_pack(packed, scale, shift, cin, cout, k, stride, pad, relu, split_ok=True, wino_ok=True) -> PackedConv:
    """OHWI filter -> what the active conv mode's kernel reads."""
    bf16_model = packed.dtype == torch.bfloat16          # (the f32x3 planes below are bf16 too, but belong to an fp32 model)
    if packed.dtype == torch.float32 and cin % 32 == 0:
        if _CONV_MODE in ("wino", "wino_x3") and wino_ok and (k, stride, pad) == (3, 1, 1):
            return PackedConv(L.wino_filter_transform(packed, cout, cin), scale, shift, cin, cout, k, stride, pad, relu, wino=True)
        if _CONV_MODE in ("f32x3", "wino_x3") and split_ok:
            packed = L.split_weights_f32x3(packed)
    if (BF16_CONV3X3 and bf16_model and (k, stride, pad) == (3, 1, 1) and cin % 32 == 0 and cout % 64 == 0
            and wino_ok):
        return PackedConv(L.conv3x3_pack_bf16(packed, cout, cin), scale, shift, cin, cout, k, stride, pad, relu, c3=True)
    return PackedConv(packed, scale, shift, cin, cout, k, stride, pad, relu)


def _check_eval(module: nn.Module) -> None:
    for m in module.modules():
        if isinstance(m, nn.modules.batchnorm._BatchNorm) and m.training:
            raise L.BevfError(
                f"{type(module).__name__} has a BatchNorm in train mode but was asked for the eval-mode kernels (BatchNorm "
                "folded from running statistics) -- call .eval() first.  Train-mode BatchNorm, backward and optimiser run "
                "through training.DetectorTape: the detector, ResNetCameraEncoder, PointNetLiDAREncoder, RadarEncoder / "
                "MultiRadarEncoder, VFELayer, FlexibleBEVFusion and CenterNetHead reach it from forward() in train mode.")


class _Engine:
    """Weight-version tracking + grow-only workspaces."""

    def __init__(self, module: nn.Module):
        self.module = module
        self._sig = None
        self._bufs: Dict[str, torch.Tensor] = {}

    def _signature(self):
        return (_CONV_MODE, BF16_CONV3X3) + tuple((t.data_ptr(), t._version) for t in self.module.state_dict(keep_vars=True).values())

    def ensure_packed(self) -> None:
        sig = self._signature()
        if sig != self._sig:
            _check_eval(self.module)
            with torch.no_grad():
                self.pack()
            self._sig = sig

    def pack(self) -> None:  # pragma: no cover - abstract
        raise NotImplementedError

    @property
    def device(self):
        return next(self.module.parameters()).device

    @property
    def dtype(self):
        """Storage dtype of activations = dtype of the module's parameters (fp32, or bf16 after model.bfloat16())."""
        dt = next(self.module.parameters()).dtype
        if dt not in (torch.float32, torch.bfloat16):
            raise L.BevfError(f"unsupported parameter dtype {dt}: the HIP path stores fp32 or bf16")
        return dt

    def buf(self, name: str, numel: int, dtype=None) -> torch.Tensor:
        dtype = self.dtype if dtype is None else dtype
        t = self._bufs.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype or t.device != self.device:
            t = torch.empty(max(numel, 4), dtype=dtype, device=self.device)
            self._bufs[name] = t
        return t


class KernelTimer:
    """HIP-event brackets around individual launches on the current stream (bench.py's live roofline
    measurement).  Off unless installed with set_timer(); adds two event records per bracket."""

    def __init__(self):
        self.spans = []          # (name, start, end, flops, bytes)

    def bracket(self, name: str, flops: float = 0.0, nbytes: float = 0.0):
        timer = self

        class _Span:
            def __enter__(self_):
                self_.s = torch.cuda.Event(enable_timing=True)
                self_.e = torch.cuda.Event(enable_timing=True)
                self_.s.record()

            def __exit__(self_, *exc):
                self_.e.record()
                timer.spans.append((name, self_.s, self_.e, flops, nbytes))
        return _Span()

    def totals(self) -> Dict[str, Dict[str, float]]:
        """name -> {launches, ms, flops, bytes}; call after a device synchronize."""
        out: Dict[str, Dict[str, float]] = {}
        for name, s, e, fl, by in self.spans:
            d = out.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms"] += s.elapsed_time(e)
            d["flops"] += fl
            d["bytes"] += by
        return out


_TIMER: Optional[KernelTimer] = None


def set_timer(t: Optional[KernelTimer]) -> None:
    global _TIMER
    _TIMER = t


class _NoSpan:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


def _span(name: str, flops: float = 0.0, nbytes: float = 0.0):
    return _TIMER.bracket(name, flops, nbytes) if _TIMER is not None else _NoSpan()


def _run_conv(pc: PackedConv, x, y, N, H, W, x_cs=None, y_cs=None, res=None, colmax=None, rows_per_group=0, tile=0):
    ho, wo = (H + 2 * pc.pad - pc.k) // pc.stride + 1, (W + 2 * pc.pad - pc.k) // pc.stride + 1
    flops = 2.0 * N * ho * wo * pc.cout * pc.k * pc.k * pc.cin          # algorithmic (direct convolution)
    with _span("conv_wino_f32" if pc.wino else ("conv3x3_bf16" if pc.c3 else "conv_igemm_f32"), flops=flops):
        _conv_call(pc, x, y, N, H, W, x_cs, y_cs, res, colmax, rows_per_group, tile)
    return ho, wo


def _conv_call(pc: PackedConv, x, y, N, H, W, x_cs=None, y_cs=None, res=None, colmax=None, rows_per_group=0, tile=0):
    if pc.wino:
        L.conv3x3_wino(x, pc.w, pc.scale, pc.shift, y, N=N, H=H, W=W, Cin=pc.cin, x_cs=x_cs or pc.cin, Cout=pc.cout,
                       y_cs=y_cs or pc.cout, relu=pc.relu, res=res, res_cs=pc.cout if res is not None else 0)
        return
    if pc.c3:
        L.conv3x3_bf16(x, pc.w, pc.scale, pc.shift, y, N=N, H=H, W=W, Cin=pc.cin, x_cs=x_cs or pc.cin, Cout=pc.cout,
                       y_cs=y_cs or pc.cout, relu=pc.relu, res=res, res_cs=pc.cout if res is not None else 0)
        return
    L.conv2d_nhwc(x, pc.w, pc.scale, pc.shift, y, N=N, H=H, W=W, Cin=pc.cin, x_cs=x_cs or pc.cin, Cout=pc.cout,
                  y_cs=y_cs or pc.cout, KH=pc.k, KW=pc.k, stride=pc.stride, pad=pc.pad, relu=pc.relu, res=res,
                  res_cs=pc.cout if res is not None else 0, colmax=colmax, rows_per_group=rows_per_group, tile=tile)


# ---- camera encoder (ref src/encoders.py:133-172) -----------------------------------------------------

FUSE_STEM_POOL = True        # inference: stem + max-pool as one kernel (csrc/stem.hip: stem_pool7x7, stem_pool7x7_bf16mma)


class CameraEncoderEngine(_Engine):
    def pack(self) -> None:
        m = self.module
        w = m.conv1.weight.detach().float()                             # (64,3,7,7); the stem computes in fp32
        assert tuple(w.shape) == (64, 3, 7, 7), "stem kernel supports the ResNet 7x7x3->64 stem only"
        packed = torch.zeros(148, 64, device=w.device)
        packed[:147] = w.reshape(64, 147).t()
        self.stem_w = packed.contiguous().view(-1)
        self.stem_scale, self.stem_shift = _bn_fold(None, m.bn1, 64, w.device)
        self.stem_w_bf16 = L.stem_pack_bf16(w) if self.dtype == torch.bfloat16 else None   # bf16 models: bf16-MFMA stem
        self.blocks = []
        for layer in (m.layer1, m.layer2, m.layer3):
            for blk in layer:
                down = None
                if blk.downsample is not None:
                    down = pack_conv(blk.downsample[0], blk.downsample[1], relu=False)
                self.blocks.append((pack_conv(blk.conv1, blk.bn1, True), pack_conv(blk.conv2, blk.bn2, True), down))
        self.proj = pack_conv(m.channel_proj[0], m.channel_proj[1], True)

    def run(self, x: torch.Tensor) -> Tuple[torch.Tensor, int, int]:
        """x: (N,3,H,W) contiguous NCHW -> (NHWC buffer [N*Hc*Wc*Cout], Hc, Wc).
        The conv kernels address their operands with 32-bit byte offsets (buffer instructions), so a trunk activation
        must stay below 2 GiB: larger batches run the trunk in image chunks that write into one feature buffer."""
        self.ensure_packed()
        N, _, H, W = x.shape
        H1, W1 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        H2, W2 = (H1 - 1) // 2 + 1, (W1 - 1) // 2 + 1
        es = 2 if self.dtype == torch.bfloat16 else 4
        n_max = max(1, ((1 << 31) - 1) // (H2 * W2 * 64 * es))
        h, w = H2, W2
        for c1, _, _ in self.blocks:
            h, w = (h + 2 - 3) // c1.stride + 1, (w + 2 - 3) // c1.stride + 1
        feat = self.buf("feat", N * h * w * self.proj.cout)
        if N <= n_max:
            self._run_chunk(x, N, H, W, feat)
        else:
            per = -(-N // -(-N // n_max))                    # balanced chunks
            for i0 in range(0, N, per):
                n = min(per, N - i0)
                self._run_chunk(x[i0:i0 + n], n, H, W, feat[i0 * h * w * self.proj.cout:])
        return feat, h, w

    def _run_chunk(self, x: torch.Tensor, N: int, H: int, W: int, feat: torch.Tensor) -> None:
        H1, W1 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        H2, W2 = (H1 - 1) // 2 + 1, (W1 - 1) // 2 + 1
        cur = self.buf("act0", N * H2 * W2 * 64)
        if FUSE_STEM_POOL:
            # conv1 + bn1 + relu + maxpool in ONE kernel -- the stem map (4.4 GB fp32 / 2.2 GB bf16 at 48 images of 900x1600, the
            # largest activation of the path) never reaches HBM; bit-identical to the two kernels below
            with _span("stem_conv7x7_f32", flops=2.0 * N * H1 * W1 * 64 * 147):
                if self.stem_w_bf16 is not None:
                    L.stem_pool_bf16mma(x.float(), self.stem_w_bf16, self.stem_scale, self.stem_shift, cur, N, H, W)
                else:
                    L.stem_pool(x, self.stem_w, self.stem_scale, self.stem_shift, cur, N, H, W)
        else:
            stem = self.buf("stem", N * H1 * W1 * 64)
            with _span("stem_conv7x7_f32", flops=2.0 * N * H1 * W1 * 64 * 147):
                if self.stem_w_bf16 is not None:
                    L.stem_conv7x7_bf16mma(x.float(), self.stem_w_bf16, self.stem_scale, self.stem_shift, stem, N, H, W)
                else:
                    L.stem_conv7x7(x, self.stem_w, self.stem_scale, self.stem_shift, stem, N, H, W)
            L.maxpool3x3s2(stem, cur, N, H1, W1, 64)
        h, w = H2, W2
        ping = 0                                       # activations ping-pong between act0 / act1
        for c1, c2, down in self.blocks:
            ho, wo = (h + 2 - 3) // c1.stride + 1, (w + 2 - 3) // c1.stride + 1
            t = self.buf("tmp", N * ho * wo * c1.cout)
            _run_conv(c1, cur, t, N, h, w)
            idt = cur
            if down is not None:
                idt = self.buf("down", N * ho * wo * down.cout)
                _run_conv(down, cur, idt, N, h, w)
            ping ^= 1
            out = self.buf(f"act{ping}", N * ho * wo * c2.cout)
            _run_conv(c2, t, out, N, ho, wo, res=idt)
            cur, h, w = out, ho, wo
        _run_conv(self.proj, cur, feat, N, h, w)


# ---- shared per-point MLP + max (ref src/encoders.py:271-306) -------------------------------------------

FUSE_POINTNET_FRONT = True   # inference, fp32: conv1 -> conv2 -> conv3 as one kernel, activations in registers (csrc/pointnet_front.hip)


class PointNetEngine(_Engine):
    def pack(self) -> None:
        m = self.module
        convs = [getattr(m, f"conv{i}") for i in range(1, 6)]
        bns = [getattr(m, f"bn{i}") for i in range(1, 6)]
        w0 = convs[0].weight.detach()
        self.cin = w0.shape[1]
        self.w0 = w0.reshape(w0.shape[0], self.cin).float().contiguous()
        self.s0, self.b0 = _bn_fold(convs[0].bias, bns[0], w0.shape[0], w0.device)
        self.c0 = w0.shape[0]
        self.layers = [pack_conv(c, b, True) for c, b in zip(convs[1:], bns[1:])]   # the last layer fuses the max over points (colmax)
        # fused front (exact fp32 modes, the reference's 64 / 128 / 256 widths): conv2 / conv3 filters in MFMA fragment order
        self.front = None
        if (FUSE_POINTNET_FRONT and self.dtype == torch.float32 and _CONV_MODE in ("f32", "wino") and self.cin <= 8
                and len(convs) >= 4 and [c.weight.shape[0] for c in convs[:3]] == [64, 128, 256]):
            self.front = [L.pointnet_front_pack(c.weight.detach().reshape(c.weight.shape[0], -1)) for c in convs[1:3]]

    def run(self, pts: torch.Tensor, keep_last: bool = False):
        """pts: (B,N,C) contiguous -> (B, feat) global max feature [and the (B*N, feat) last activations].
        Frames are processed in chunks when an activation of the whole batch would pass the kernels' 2 GiB limit."""
        self.ensure_packed()
        B, N, Cc = pts.shape
        es = 2 if self.dtype == torch.bfloat16 else 4
        widest = max(pc.cout for pc in self.layers)
        f_max = max(1, ((1 << 31) - 1) // (max(N, 1) * widest * es))
        last = self.layers[-1]
        gmax = torch.zeros(B, last.cout, dtype=torch.int32, device=pts.device)
        if B <= f_max:
            y = self._run_frames(pts, gmax, keep_last)
            return gmax.view(torch.float32), y
        if keep_last:
            raise L.BevfError(f"PointNet: per-point features of {B}x{N} points exceed the 2 GiB activation limit")
        for b0 in range(0, B, f_max):
            self._run_frames(pts[b0:b0 + f_max], gmax[b0:b0 + f_max], False)
        return gmax.view(torch.float32), None

    def _run_frames(self, pts: torch.Tensor, gmax: torch.Tensor, keep_last: bool):
        B, N, Cc = pts.shape
        M = B * N
        if self.front is not None:
            l2, l3 = self.layers[0], self.layers[1]
            a = self.buf("l2", M * l3.cout)
            with _span("pointnet_front_f32", flops=2.0 * M * (l2.cin * l2.cout + l3.cin * l3.cout)):     # the two MFMA layers
                L.pointnet_front(pts.float(), self.w0, self.s0, self.b0, self.front[0], l2.scale, l2.shift, self.front[1],
                                 l3.scale, l3.shift, a, M, Cc)
            rest = list(enumerate(self.layers[:-1]))[2:]
        else:
            a = self.buf("l0", M * self.c0)
            L.pointwise_smallk(pts.float(), self.w0, self.s0, self.b0, a, M, Cc, self.c0, True)
            rest = list(enumerate(self.layers[:-1]))
        for i, pc in rest:
            o = self.buf(f"l{i + 1}", M * pc.cout)
            _run_conv(pc, a, o, M, 1, 1)
            a = o
        last = self.layers[-1]
        y = self.buf("l_last", M * last.cout) if keep_last else None
        _run_conv(last, a, y, M, 1, 1, colmax=gmax, rows_per_group=N)
        return y


FUSE_VFE = True              # VFELayer with <= 16 input channels as one kernel (False: pointwise_smallk + group_max, bit-identical)


class VFEEngine(_Engine):
    """VFELayer (ref src/encoders.py:431-455): Linear + BN1d + ReLU per point, max over the points of a voxel."""

    def pack(self) -> None:
        m = self.module
        w = m.linear.weight.detach()
        self.cin, self.cout = w.shape[1], w.shape[0]
        self.scale, self.shift = _bn_fold(m.linear.bias, m.bn, self.cout, w.device)
        self.w = w.float().contiguous()
        if self.cin > 16:
            if self.cin % 32:
                raise L.BevfError(f"VFELayer: in_channels={self.cin} must be <= 16 or a multiple of 32")
            self.pc = PackedConv(self.w.view(-1), self.scale, self.shift, self.cin, self.cout, 1, 1, 0, True)

    def run(self, x: torch.Tensor) -> torch.Tensor:
        self.ensure_packed()
        B, Nv, P, Cc = x.shape
        G, M = B * Nv, B * Nv * P
        if Cc <= 16:
            out = torch.empty(G, self.cout, device=x.device)
            if FUSE_VFE:
                L.vfe_smallk_max(x, self.w, self.scale, self.shift, out, G, P, Cc, self.cout)
                return out
            t = self.buf("pts", M * self.cout)
            L.pointwise_smallk(x, self.w, self.scale, self.shift, t, M, Cc, self.cout, True)
            L.group_max(t, out, G, P, self.cout)
            return out
        gmax = torch.zeros(G, self.cout, dtype=torch.int32, device=x.device)
        _run_conv(self.pc, x, None, M, 1, 1, colmax=gmax, rows_per_group=P)
        return gmax.view(torch.float32)


class RadarEngine(_Engine):
    """MultiRadarEncoder: shared RadarEncoder per sweep + concat/max/mean (ref src/encoders.py:628-661)."""

    def pack(self) -> None:
        enc = self.module.radar_encoder
        self.cin = enc.conv1.weight.shape[1]
        self.ws, self.scales, self.shifts, self.widths = [], [], [], []
        for i in range(1, 5):
            conv, bn = getattr(enc, f"conv{i}"), getattr(enc, f"bn{i}")
            w = conv.weight.detach()
            self.ws.append(w.reshape(w.shape[0], w.shape[1]).t().float().contiguous())       # k-major, fp32 compute
            s, b = _bn_fold(conv.bias, bn, w.shape[0], w.device)
            self.scales.append(s); self.shifts.append(b); self.widths.append(w.shape[0])
        if self.module.fusion_method == "concat":
            fc = self.module.fusion_fc
            self.fc_w = fc.weight.detach().contiguous()                 # storage dtype (fp32 / bf16 weight stream)
            self.fc_b = fc.bias.detach().float().contiguous() if fc.bias is not None else None

    def run(self, radar_list: Sequence[torch.Tensor]) -> torch.Tensor:
        self.ensure_packed()
        R = len(radar_list)
        B = radar_list[0].shape[0]
        feat = self.widths[3]
        per = torch.zeros(B, R, feat, device=self.device)            # zero-filled: chunk maxima merge with atomicMax
        radar_list = [r.float() for r in radar_list]
        same = all(r.shape == radar_list[0].shape for r in radar_list)
        if same:
            x = torch.stack([r.contiguous() for r in radar_list], dim=0).contiguous()         # [R][B][P][Cin]
            L.radar_mlp_max(x, self.ws, self.scales, self.shifts, per, R, B, x.shape[2], self.cin, self.widths)
        else:
            for r, pts in enumerate(radar_list):
                one = torch.zeros(B, 1, feat, device=self.device)
                L.radar_mlp_max(pts.contiguous(), self.ws, self.scales, self.shifts, one, 1, B, pts.shape[1],
                                self.cin, self.widths)
                per[:, r] = one[:, 0]
        method = self.module.fusion_method
        if method == "concat":
            K = R * feat
            if K != self.fc_w.shape[1]:
                raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({B}x{K} and "
                                   f"{self.fc_w.shape[1]}x{self.fc_w.shape[0]})")
            out = torch.empty(B, self.fc_w.shape[0], device=self.device)
            L.linear(per, self.fc_w, self.fc_b, out, B, K, self.fc_w.shape[0], False)
            return out
        if method == "max":                                          # ref src/encoders.py:654-655
            out = torch.empty(B, feat, device=self.device)
            L.group_max(per, out, B, R, feat)
            return out
        if method == "mean":                                         # ref src/encoders.py:656-657
            out = torch.empty(B, feat, device=self.device)
            L.cam_mean(per, out, B, R, 1, feat)
            return out
        raise ValueError(f"Unknown fusion method: {method}")


# ---- BEV fusion (ref src/fusion.py:209-297) ----------------------------------------------------------------

class FusionEngine(_Engine):
    collapse_radar = True        # set False to run radar_refine on the full map (tests compare both, bit for bit)

    def pack(self) -> None:
        m = self.module
        if m.use_camera:
            self.cam1 = pack_conv(m.camera_proj[0], m.camera_proj[1], True)
            self.cam2 = pack_conv(m.camera_proj[3], m.camera_proj[4], True)
        if m.use_lidar:
            l0, l2 = m.lidar_init[0], m.lidar_init[2]
            self.li0 = (l0.weight.detach().contiguous(), l0.bias.detach().float().contiguous())
            self.li2 = (l2.weight.detach().contiguous(), l2.bias.detach().float().contiguous())
            self.lup1 = pack_conv(m.lidar_upsample[0], m.lidar_upsample[1], True)
            self.lup2 = pack_conv(m.lidar_upsample[4], m.lidar_upsample[5], True)
        if m.use_radar:
            r0 = m.radar_proj[0]
            self.rp = (r0.weight.detach().contiguous(), r0.bias.detach().float().contiguous())
            # exact kernel whatever the mode: the 5x5 border-class shortcut below must reproduce the full-map convolution
            # bit for bit, which a position-dependent Winograd tiling would not (and these two launches cost nothing)
            self.rr1 = pack_conv(m.radar_refine[0], m.radar_refine[1], True, wino_ok=False)
            self.rr2 = pack_conv(m.radar_refine[3], m.radar_refine[4], True, wino_ok=False)
        self.f1 = pack_conv(m.bev_fusion[0], m.bev_fusion[1], True)
        self.f2 = pack_conv(m.bev_fusion[3], m.bev_fusion[4], True)

    def run(self, cam: Optional[torch.Tensor], cam_geom: Optional[Tuple[int, int, int, int]],
            lidar: Optional[torch.Tensor], radar: Optional[torch.Tensor]) -> Tuple[torch.Tensor, int]:
        """cam: NHWC encoder features [B*ncam][Hc][Wc][C] with cam_geom = (B, ncam, Hc, Wc); lidar (B,1024);
        radar (B,256).  Returns the fused NHWC map [B][S_h*S_w][bev_channels] and B."""
        self.ensure_packed()
        m = self.module
        Sh, Sw, bc = m.bev_h, m.bev_w, m.bev_channels
        P = Sh * Sw
        present = []
        if m.use_camera and cam is not None:
            present.append("c")
        if m.use_lidar and lidar is not None:
            present.append("l")
        if m.use_radar and radar is not None:
            present.append("r")
        if not present:
            raise ValueError("No modality features provided")
        B = cam_geom[0] if "c" in present else (lidar.shape[0] if "l" in present else radar.shape[0])
        ccs = bc * len(present)
        if ccs != self.f1.cin:
            raise RuntimeError(f"Given groups=1, weight of size [{self.f1.cout}, {self.f1.cin}, 3, 3], expected input"
                               f"[{B}, {ccs}, {Sh}, {Sw}] to have {self.f1.cin} channels, but got {ccs} channels instead")
        concat = self.buf("concat", B * P * ccs)
        slot = 0
        if "c" in present:
            _, ncam, Hc, Wc = cam_geom
            Cc = self.cam1.cin
            pooled = cam
            if ncam > 1:
                pooled = self.buf("cam_mean", B * Hc * Wc * Cc)
                with _span("bev_pool", nbytes=float(cam.element_size()) * B * Hc * Wc * Cc * (ncam + 1)):
                    L.cam_mean(cam, pooled, B, ncam, Hc * Wc, Cc)
            t1 = self.buf("cam_t1", B * Hc * Wc * self.cam1.cout)
            _run_conv(self.cam1, pooled, t1, B, Hc, Wc)
            t2 = self.buf("cam_t2", B * Hc * Wc * self.cam2.cout)
            _run_conv(self.cam2, t1, t2, B, Hc, Wc)
            with _span("bev_pool", nbytes=float(t2.element_size()) * B * bc * (Hc * Wc + Sh * Sw)):
                L.bilinear_nhwc(t2, concat[slot * bc:], B, Hc, Wc, bc, bc, Sh, Sw, ccs)
            slot += 1
        if "l" in present:
            s0 = m.lidar_start_size
            hid = self.buf("lid_h", B * self.li0[0].shape[0], torch.float32)       # small per-frame vectors stay fp32
            L.linear(lidar.float().contiguous(), self.li0[0], self.li0[1], hid, B, self.li0[0].shape[1], self.li0[0].shape[0], True)
            O = self.li2[0].shape[0]
            ch = O // (s0 * s0)
            grid0 = self.buf("lid_g0", B * O)
            L.linear(hid, self.li2[0], self.li2[1], grid0, B, self.li2[0].shape[1], O, False, s0 * s0, ch)
            g1 = self.buf("lid_g1", B * s0 * s0 * self.lup1.cout)
            _run_conv(self.lup1, grid0, g1, B, s0, s0)
            s1 = 2 * s0
            g2 = self.buf("lid_g2", B * s1 * s1 * self.lup1.cout)
            L.bilinear_nhwc(g1, g2, B, s0, s0, self.lup1.cout, self.lup1.cout, s1, s1, self.lup1.cout)
            if (s1, s1) == (Sh, Sw):
                _run_conv(self.lup2, g2, concat[slot * bc:], B, s1, s1, y_cs=ccs)
            else:
                # extension beyond the reference (which crashes at the concat for BEV != 50x50):
                # bilinear resize of the 50x50 LiDAR map, exactly like the camera branch
                g3 = self.buf("lid_g3", B * s1 * s1 * bc)
                _run_conv(self.lup2, g2, g3, B, s1, s1)
                L.bilinear_nhwc(g3, concat[slot * bc:], B, s1, s1, bc, bc, Sh, Sw, ccs)
            slot += 1
        if "r" in present:
            rv = self.buf("rad_v", B * bc, torch.float32)
            L.linear(radar.float().contiguous(), self.rp[0], self.rp[1], rv, B, self.rp[0].shape[1], bc, True)
            if Sh >= 5 and Sw >= 5 and self.collapse_radar:
                # exact shortcut: two 3x3/pad-1 convs on a constant image have 5x5 distinct pixels (bevpool.hip)
                r0 = self.buf("rad_0", B * 25 * bc)
                L.broadcast_nhwc(rv, r0, B, 25, bc, bc)
                r1 = self.buf("rad_1", B * 25 * bc)
                _run_conv(self.rr1, r0, r1, B, 5, 5)
                r2 = self.buf("rad_2", B * 25 * bc)
                _run_conv(self.rr2, r1, r2, B, 5, 5)
                L.expand_border_classes(r2, concat[slot * bc:], B, Sh, Sw, bc, ccs)
            else:
                r0 = self.buf("rad_0", B * P * bc)
                L.broadcast_nhwc(rv, r0, B, P, bc, bc)
                r1 = self.buf("rad_1", B * P * bc)
                _run_conv(self.rr1, r0, r1, B, Sh, Sw)
                _run_conv(self.rr2, r1, concat[slot * bc:], B, Sh, Sw, y_cs=ccs)
            slot += 1
        f1 = self.buf("fus_1", B * P * self.f1.cout)
        _run_conv(self.f1, concat, f1, B, Sh, Sw)
        out = self.buf("fus_2", B * P * self.f2.cout)
        _run_conv(self.f2, f1, out, B, Sh, Sw)
        return out, B


# ---- CenterNet head (ref src/fusion.py:869-884) --------------------------------------------------------------

HEAD_BRANCHES = ("heatmap", "offset", "size", "rot", "vel")


class HeadEngine(_Engine):
    def pack(self) -> None:
        m = self.module
        convs3 = [getattr(m, f"{n}_head")[0] for n in HEAD_BRANCHES]
        convs1 = [getattr(m, f"{n}_head")[2] for n in HEAD_BRANCHES]
        w3 = torch.cat([c.weight.detach() for c in convs3], dim=0)                 # (5*hc, Cin, 3, 3)
        b3 = torch.cat([c.bias.detach() for c in convs3], dim=0)
        self.hc = convs3[0].weight.shape[0]
        self.conv = _finish_pack(w3.permute(0, 2, 3, 1).contiguous().view(-1), None, b3.float().contiguous(),
                                 w3.shape[1], w3.shape[0], 3, 1, 1, True)
        self.cs = [c.weight.shape[0] for c in convs1]
        self.w1 = torch.cat([c.weight.detach().reshape(c.weight.shape[0], self.hc) for c in convs1], 0).float().contiguous()
        self.b1 = torch.cat([c.bias.detach() for c in convs1], 0).float().contiguous()

    def run(self, bev_nhwc: torch.Tensor, B: int, H: int, W: int) -> Dict[str, torch.Tensor]:
        self.ensure_packed()
        P = H * W
        hid = self.buf("hid", B * P * self.conv.cout)
        _run_conv(self.conv, bev_nhwc, hid, B, H, W)
        outs = [torch.empty(B, c, H, W, device=bev_nhwc.device) for c in self.cs]        # fp32 also on the bf16 path
        L.head_tail(hid, self.w1, self.b1, outs, B, P, self.hc, self.cs, self.cs[0])
        return dict(zip(HEAD_BRANCHES, outs))


# ---- layout helpers at the API surface ----------------------------------------------------------------------

def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """(N,C,H,W) -> flat NHWC buffer."""
    N, Cc, H, W = x.shape
    dt = x.dtype
    y = torch.empty(N * H * W * Cc, device=x.device)
    L.nchw_to_nhwc(x.float().contiguous(), y, N, Cc, H * W, Cc)           # API-surface layout change runs in fp32
    return y if dt == torch.float32 else y.to(dt)


def to_nchw(buf: torch.Tensor, N: int, Cc: int, H: int, W: int) -> torch.Tensor:
    dt = buf.dtype
    y = torch.empty(N, Cc, H, W, device=buf.device)
    L.nhwc_to_nchw(buf.float(), y, N, Cc, H * W, Cc)
    return y if dt == torch.float32 else y.to(dt)


def require_cuda(*tensors) -> None:
    for t in tensors:
        if t is not None and isinstance(t, torch.Tensor) and not t.is_cuda:
            raise L.BevfError("this package runs the hot path on MI355X only: move the module and its inputs to "
                              "'cuda' (there is no CPU fallback; the CPU oracle lives under oracle/ for tests)")
