"""GPU parity: the HIP path (through the C-ABI of libbevf_hip.so) against the CPU oracle and the
golden fixtures minted from the imported reference.  Run with `-m gpu` on an MI355X.

Tolerances: BASELINE.json's north_star asks fp32 BEV features and CenterNet logits within
1e-4 rel; `rel_err` = max|a-b| / max|b| per tensor.  Kernel-level cases are held to 2e-5
(one conv's worth of summation-order noise), model-level cases to 1e-4.  Integer outputs exact.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from bevfusion_multimodal_3d_object_detection_amd import _lib as L
from bevfusion_multimodal_3d_object_detection_amd import encoders, engine, fusion, synth
from oracle import ref_model
from tests.conftest import load_golden, rel_err
from tests.golden import cases

pytestmark = pytest.mark.gpu
KTOL, MTOL = 2e-5, 1e-4


def nhwc(x):           # (N,C,H,W) cpu -> flat NHWC cuda
    return x.permute(0, 2, 3, 1).contiguous().view(-1).cuda()


def from_nhwc(buf, N, C, H, W):
    return buf[:N * H * W * C].view(N, H, W, C).permute(0, 3, 1, 2).cpu()


# ---- conv implicit GEMM ---------------------------------------------------------------------------------
CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, relu, res, tile
    (2, 9, 13, 32, 64, 3, 1, 1, True, False, 0),
    (1, 17, 11, 64, 128, 3, 2, 1, True, False, 1),      # stride 2, odd sizes (ResNet layer2.0.conv1)
    (2, 8, 8, 64, 128, 1, 2, 0, False, False, 4),       # 1x1 stride-2 downsample, no activation
    (1, 12, 20, 128, 128, 3, 1, 1, True, True, 1),      # residual add (BasicBlock conv2)
    (1, 30, 31, 64, 64, 3, 1, 1, True, True, 2),        # tall tile 256x64
    (1, 30, 31, 64, 64, 3, 1, 1, True, False, 3),       # 128x64
    (3, 7, 5, 256, 320, 3, 1, 1, True, False, 0),       # Cout not a multiple of the N tile (fused head conv)
    (1, 5, 5, 768, 512, 3, 1, 1, True, False, 4),       # 3-modality concat width
    (1, 1, 1, 32, 32, 1, 1, 0, True, False, 0),         # degenerate single pixel
    (1, 30, 31, 64, 128, 3, 1, 1, True, True, 5),       # hybrid: 3 big 128x128 M-tiles + 64x64 tail, residual
    (2, 33, 37, 64, 64, 3, 2, 1, True, True, 6),        # hybrid 256x64 + tail, stride 2
    (1, 64, 72, 32, 320, 1, 1, 0, False, False, 5),     # hybrid, ragged N in both tile shapes
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_igemm(gpu, case):
    N, H, W, Cin, Cout, k, stride, pad, relu, use_res, tile = case
    seed = hash(case) & 0xFFFF
    x = synth.normal((N, Cin, H, W), seed + 1)
    w = synth.normal((Cout, Cin, k, k), seed + 2, 0, (1.0 / (Cin * k * k)) ** 0.5)
    scale = synth.uniform((Cout,), seed + 3, 0.5, 1.5)
    shift = synth.normal((Cout,), seed + 4, 0, 0.3)
    ref = F.conv2d(x, w, None, stride, pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    Ho, Wo = ref.shape[-2:]
    res = synth.normal((N, Cout, Ho, Wo), seed + 5) if use_res else None
    if use_res:
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    y = torch.full((N * Ho * Wo * Cout,), float("nan"), device=gpu)
    L.conv2d_nhwc(nhwc(x), w.permute(0, 2, 3, 1).contiguous().view(-1).cuda(), scale.cuda(), shift.cuda(), y,
                  N=N, H=H, W=W, Cin=Cin, x_cs=Cin, Cout=Cout, y_cs=Cout, KH=k, KW=k, stride=stride, pad=pad,
                  relu=relu, res=nhwc(res) if use_res else None, res_cs=Cout if use_res else 0, tile=tile)
    assert rel_err(from_nhwc(y, N, Cout, Ho, Wo), ref) <= KTOL


def test_conv_channel_slices(gpu):
    """Reads the first Cin channels of a wider pixel and writes into a channel slice of a concat buffer."""
    N, H, W, Cin, Cout, x_cs, y_cs, off = 1, 6, 7, 32, 64, 48, 192, 64
    x = synth.normal((N, H, W, x_cs), 11)
    w = synth.normal((Cout, Cin, 3, 3), 12, 0, 0.06)
    ref = F.relu(F.conv2d(x[..., :Cin].permute(0, 3, 1, 2), w, None, 1, 1))
    y = torch.zeros(N * H * W * y_cs, device=gpu)
    L.conv2d_nhwc(x.view(-1).cuda(), w.permute(0, 2, 3, 1).contiguous().view(-1).cuda(), None, None, y[off:],
                  N=N, H=H, W=W, Cin=Cin, x_cs=x_cs, Cout=Cout, y_cs=y_cs, KH=3, KW=3, stride=1, pad=1, relu=True)
    got = y.view(N, H, W, y_cs).cpu()
    assert rel_err(got[..., off:off + Cout].permute(0, 3, 1, 2), ref) <= KTOL
    assert float(got[..., :off].abs().max()) == 0 and float(got[..., off + Cout:].abs().max()) == 0


@pytest.mark.parametrize("B,npts", [(1, 300), (3, 200), (2, 128), (2, 1)])
def test_conv_colmax(gpu, B, npts):
    """Pointwise layer + ReLU + per-batch-element column max fused in the epilogue (PointNet conv5 + max);
    group boundaries fall inside tiles (200, 300 are not multiples of the 64..256-row tiles)."""
    Cin, Cout = 64, 128
    x = synth.normal((B * npts, Cin), 21)
    w = synth.normal((Cout, Cin), 22, 0, 0.125)
    shift = synth.normal((Cout,), 23, 0, 0.2)
    ref = F.relu(x @ w.t() + shift).view(B, npts, Cout).max(dim=1)[0]
    for tile in (0, 1, 4, 5):
        if tile == 5 and B * npts < 256:
            continue
        cm = torch.zeros(B, Cout, dtype=torch.int32, device=gpu)
        L.conv2d_nhwc(x.view(-1).cuda(), w.contiguous().view(-1).cuda(), None, shift.cuda(), None, N=B * npts, H=1,
                      W=1, Cin=Cin, x_cs=Cin, Cout=Cout, y_cs=Cout, KH=1, KW=1, stride=1, pad=0, relu=True,
                      colmax=cm, rows_per_group=npts, tile=tile)
        assert rel_err(cm.view(torch.float32).cpu(), ref) <= KTOL


def test_conv_rejects_bad_shapes(gpu):
    x = torch.zeros(4 * 4 * 24, device=gpu)
    with pytest.raises(L.BevfError, match="multiple of 32"):
        L.conv2d_nhwc(x, torch.zeros(32 * 24 * 9, device=gpu), None, None, torch.zeros(4 * 4 * 32, device=gpu), N=1,
                      H=4, W=4, Cin=24, x_cs=24, Cout=32, y_cs=32, KH=3, KW=3, stride=1, pad=1, relu=True)
    with pytest.raises(L.BevfError, match="CPU tensor"):
        L.conv2d_nhwc(torch.zeros(4 * 4 * 32), torch.zeros(32 * 32 * 9, device=gpu), None, None,
                      torch.zeros(4 * 4 * 32, device=gpu), N=1, H=4, W=4, Cin=32, x_cs=32, Cout=32, y_cs=32, KH=3,
                      KW=3, stride=1, pad=1, relu=True)


# ---- stem / pool -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,H,W", [(2, 64, 96), (1, 37, 301), (1, 7, 5)])
def test_stem_and_maxpool(gpu, N, H, W):
    x = synth.normal((N, 3, H, W), 31)
    w = synth.normal((64, 3, 7, 7), 32, 0, 0.08)
    scale, shift = synth.uniform((64,), 33, 0.5, 1.5), synth.normal((64,), 34, 0, 0.2)
    ref = F.relu(F.conv2d(x, w, None, 2, 3) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    Ho, Wo = ref.shape[-2:]
    packed = torch.zeros(148, 64)
    packed[:147] = w.reshape(64, 147).t()
    y = torch.full((N * Ho * Wo * 64,), float("nan"), device=gpu)
    L.stem_conv7x7(x.cuda(), packed.view(-1).cuda(), scale.cuda(), shift.cuda(), y, N, H, W)
    assert rel_err(from_nhwc(y, N, 64, Ho, Wo), ref) <= KTOL
    pref = F.max_pool2d(ref, 3, 2, 1)
    Hp, Wp = pref.shape[-2:]
    p = torch.empty(N * Hp * Wp * 64, device=gpu)
    L.maxpool3x3s2(nhwc(ref), p, N, Ho, Wo, 64)
    assert torch.equal(from_nhwc(p, N, 64, Hp, Wp), pref)          # max is exact


# ---- BEV pooling glue ---------------------------------------------------------------------------------------
def test_cam_mean(gpu):
    B, n, P, C = 2, 6, 35, 64
    x = synth.normal((B, n, C, P), 41)
    ref = x.mean(dim=1)                                            # (B,C,P)
    y = torch.empty(B * P * C, device=gpu)
    L.cam_mean(x.permute(0, 1, 3, 2).contiguous().view(-1).cuda(), y, B, n, P, C)
    assert rel_err(y.view(B, P, C).permute(0, 2, 1).cpu(), ref) <= 1e-6


@pytest.mark.parametrize("Hi,Wi,Ho,Wo", [(28, 50, 128, 128), (25, 25, 50, 50), (57, 100, 16, 24), (5, 7, 5, 7), (1, 1, 4, 3)])
def test_bilinear(gpu, Hi, Wi, Ho, Wo):
    B, C = 2, 8
    x = synth.normal((B, C, Hi, Wi), 51)
    ref = F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=False)
    y = torch.zeros(B * Ho * Wo * 2 * C, device=gpu)
    L.bilinear_nhwc(nhwc(x), y[C:], B, Hi, Wi, C, C, Ho, Wo, 2 * C)        # into the upper channel slice
    got = y.view(B, Ho, Wo, 2 * C)[..., C:].permute(0, 3, 1, 2).cpu()
    assert rel_err(got, ref) <= 2e-6
    if (Hi, Wi) == (25, 25):                                       # nn.Upsample(scale_factor=2) is the same map
        assert rel_err(got, torch.nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False)(x)) <= 2e-6


@pytest.mark.parametrize("B", [1, 3, 8, 11])
def test_linear(gpu, B):
    K, O = 512, 1000
    x, w, b = synth.normal((B, K), 61), synth.normal((O, K), 62, 0, 0.05), synth.normal((O,), 63)
    y = torch.empty(B, O, device=gpu)
    L.linear(x.cuda(), w.cuda(), b.cuda(), y, B, K, O, True)
    assert rel_err(y.cpu(), F.relu(x @ w.t() + b)) <= KTOL
    L.linear(x.cuda(), w.cuda(), b.cuda(), y, B, K, O, False, 125, 8)      # permuted store: (B,8,125) -> [B][125][8]
    assert rel_err(y.view(B, 125, 8).permute(0, 2, 1).reshape(B, O).cpu(), x @ w.t() + b) <= KTOL
    # ragged row counts (from 16384 rows on a wave walks 4 or 8 rows at a time): nothing past the last row, bf16 weights, K = 1024
    for O2, K2 in ((1003, 512), (5, 1024), (1, 64), (16384 + 3, 128)):       # the last one takes the rows-in-flight variant
        x2, w2, b2 = synth.normal((B, K2), 64), synth.normal((O2, K2), 65, 0, 0.05), synth.normal((O2,), 66)
        y2 = torch.full((B * O2 + 16,), -3.0, device=gpu)
        L.linear(x2.cuda(), w2.cuda(), b2.cuda(), y2, B, K2, O2, False)
        assert rel_err(y2[:B * O2].view(B, O2).cpu(), x2 @ w2.t() + b2) <= KTOL
        assert torch.all(y2[B * O2:] == -3.0)
        wb = w2.cuda().bfloat16()
        L.linear(x2.cuda(), wb, b2.cuda(), y2, B, K2, O2, False)
        assert rel_err(y2[:B * O2].view(B, O2).cpu(), x2 @ wb.float().cpu().t() + b2) <= KTOL


def test_layout_roundtrip(gpu):
    x = synth.normal((3, 37, 5, 9), 71)
    buf = engine.to_nhwc(x.cuda())
    assert torch.equal(buf.view(3, 5, 9, 37).cpu(), x.permute(0, 2, 3, 1))
    assert torch.equal(engine.to_nchw(buf, 3, 37, 5, 9).cpu(), x)


# ---- modules against the golden fixtures (outputs of the imported reference) ---------------------------------
def _golden_check(out, gold, keys, tol=MTOL):
    for k in keys:
        o = out[k].cpu() if isinstance(out, dict) else out.cpu()
        assert tuple(o.shape) == gold[k].shape, (k, tuple(o.shape), gold[k].shape)
        e = rel_err(o, gold[k])
        assert e <= tol, (k, e)


def test_camera_encoder_golden(gpu):
    c = cases.CAMERA_ENCODER_CASE
    m = encoders.ResNetCameraEncoder(backbone="resnet18", pretrained=False)
    synth.fill_state_dict_(m, c["seed"])
    m = m.cuda().eval()
    _golden_check({"out": m(synth.normal(c["shape"], c["seed"] + 1).cuda())}, load_golden("camera_encoder"), ["out"])


def test_pointnet_golden(gpu):
    c = cases.POINTNET_CASE
    m = encoders.PointNetLiDAREncoder(input_channels=c["cin"], feat_dim=1024)
    synth.fill_state_dict_(m, c["seed"])
    m = m.cuda().eval()
    x = cases.pointnet_input(c)
    gold = load_golden("pointnet")
    _golden_check({"out": m(x.cuda())}, gold, ["out"])
    _golden_check({"out": m(x.transpose(1, 2).contiguous().cuda())}, gold, ["out"])     # (B,C,N) layout sniff


@pytest.mark.parametrize("method", ["concat", "max", "mean"])
def test_radar_golden(gpu, method):
    c = cases.RADAR_CASE
    m = encoders.MultiRadarEncoder(input_channels=7, feat_dim=256, num_radars=c["num_radars"], fusion_method=method)
    synth.fill_state_dict_(m, c["seed"])
    m = m.cuda().eval()
    _golden_check({"out": m([r.cuda() for r in cases.radar_input(c)])}, load_golden("radar_" + method), ["out"])


def test_radar_wrong_sweep_count_raises(gpu):
    """The reference's own sanity sweep fails 3 of 6 configs on this: 3 sweeps into num_radars=5 (demo.ipynb:417-444)."""
    m = encoders.MultiRadarEncoder(input_channels=7, feat_dim=256, num_radars=5).cuda().eval()
    with pytest.raises(RuntimeError, match="cannot be multiplied"):
        m([torch.zeros(2, 10, 7, device=gpu) for _ in range(3)])


def test_vfe_golden(gpu):
    c = cases.VFE_CASE
    m = encoders.VFELayer(c["cin"], c["cout"])
    synth.fill_state_dict_(m, c["seed"])
    m = m.cuda().eval()
    _golden_check({"out": m(synth.normal(c["shape"], c["seed"] + 1).cuda())}, load_golden("vfe"), ["out"])


@pytest.mark.parametrize("c", cases.FUSION_CASES, ids=lambda c: c["name"])
def test_fusion_golden(gpu, c):
    m = fusion.FlexibleBEVFusion(use_camera=c["cam"], use_lidar=c["lid"], use_radar=c["rad"], bev_h=c["bev_h"],
                                 bev_w=c["bev_w"])
    synth.fill_state_dict_(m, c["seed"])
    m = m.cuda().eval()
    cam, lid, rad = (t.cuda() if t is not None else None for t in cases.fusion_inputs(c))
    _golden_check({"out": m(cam, lid, rad)}, load_golden("fusion_" + c["name"]), ["out"])


def test_head_golden(gpu):
    c = cases.HEAD_CASE
    torch.manual_seed(0)
    m = fusion.CenterNetHead(in_channels=256, num_classes=10).cuda().eval()
    x = synth.normal(c["shape"], c["seed"] + 1).cuda()
    heat = m(x)["heatmap"]                                  # default init: sigmoid(-ln 99) = 0.01
    assert 0.0099 < float(heat.min()) and float(heat.max()) < 0.0101
    synth.fill_state_dict_(m, c["seed"])
    _golden_check(m(x), load_golden("head"), ["heatmap", "offset", "size", "rot", "vel"])


@pytest.mark.parametrize("c", cases.DETECTOR_CASES, ids=lambda c: c["name"])
def test_detector_golden(gpu, c):
    m = fusion.create_detector(c["modality"], "bev", "centernet", bev_h=c["bev_h"], bev_w=c["bev_w"])
    synth.fill_state_dict_(m, c["seed"])
    m = m.cuda().eval()
    imgs, pts, radars = cases.detector_inputs(c)
    out = m(imgs.cuda() if imgs is not None else None, pts.cuda() if pts is not None else None,
            [r.cuda() for r in radars] if radars else None)
    _golden_check(out, load_golden("detector_" + c["name"]), ["heatmap", "offset", "size", "rot", "vel"])


def test_detector_state_dict_roundtrip_with_oracle(gpu):
    """A checkpoint moves between the oracle (== reference layout) and the product; weight updates are picked up."""
    c = cases.DETECTOR_CASES[0]
    ora = ref_model.make_detector(c["modality"], 50, 50)
    synth.fill_state_dict_(ora, 7)
    ora.eval()
    m = fusion.create_detector(c["modality"], "bev", "centernet", bev_h=50, bev_w=50).cuda().eval()
    imgs, pts, _ = cases.detector_inputs(c)
    before = m(imgs.cuda(), pts.cuda(), None)
    m.load_state_dict(ora.state_dict(), strict=True)
    after = m(imgs.cuda(), pts.cuda(), None)
    with torch.no_grad():
        ref = ora(imgs, pts, None)
    for k in ref:
        assert rel_err(after[k].cpu(), ref[k]) <= MTOL, k
    assert rel_err(before["size"].cpu(), ref["size"]) > 1e-2        # the repack really happened


def test_lidar_resize_extension_128(gpu):
    """BEV 128x128 with LiDAR: beyond what the reference can run (it raises at the concat);
    checked against the oracle's documented bilinear-resize generalisation.  Parity unpinned by the reference."""
    ora = ref_model.make_detector("camera+lidar", 128, 96)
    synth.fill_state_dict_(ora, 9)
    ora.eval()
    m = fusion.create_detector("camera+lidar", "bev", "centernet", bev_h=128, bev_w=96).cuda().eval()
    m.load_state_dict(ora.state_dict())
    imgs, pts, _ = synth.frame_inputs(1, 2, 64, 96, 400, seed=77)
    out = m(imgs.cuda(), pts.cuda(), None)
    with torch.no_grad():
        ref = ora(imgs, pts, None)
    for k in ref:
        assert rel_err(out[k].cpu(), ref[k]) <= MTOL, k


def test_modules_without_a_train_mode_path_refuse_it(gpu):
    """Train-mode BatchNorm runs through training.DetectorTape for the detector and its module kinds
    (tests/test_gpu_standalone_train.py, mixed-mode BatchNorm included); a VFELayer wider than the small-K point kernel refuses loudly."""
    v = encoders.VFELayer(32, 32).cuda().train()
    with pytest.raises(RuntimeError, match="<= 16 channels"):
        v(torch.zeros(1, 3, 5, 32, device=gpu))


def test_no_modality_raises(gpu):
    m = fusion.FlexibleBEVFusion(use_camera=True, use_lidar=False, use_radar=False, bev_h=8, bev_w=8).cuda().eval()
    with pytest.raises(ValueError, match="No modality features provided"):
        m(None, None, None)


def test_radar_refine_collapse_is_bit_identical(gpu):
    """The 5x5 border-class shortcut for the radar branch must reproduce the full-map convs bit for bit."""
    for bev_h, bev_w in ((50, 50), (16, 24), (5, 7)):
        m = fusion.FlexibleBEVFusion(use_camera=False, use_lidar=False, use_radar=True, bev_h=bev_h, bev_w=bev_w)
        synth.fill_state_dict_(m, 31)
        m = m.cuda().eval()
        rad = synth.normal((3, 256), 32).cuda()
        fast = m(None, None, rad).clone()
        m._eng().collapse_radar = False
        full = m(None, None, rad)
        assert torch.equal(fast, full), (bev_h, bev_w)


def test_hipgraph_replay_matches_eager(gpu):
    """The forward captured into a hipGraph reproduces the eager launches bit for bit, also on new inputs."""
    m = fusion.create_detector("camera+lidar+radar", "bev", "centernet", bev_h=50, bev_w=50)
    synth.fill_state_dict_(m, 3)
    m = m.cuda().eval()
    a = [t.cuda() if not isinstance(t, list) else [r.cuda() for r in t] for t in synth.frame_inputs(1, 2, 64, 96, 300, 4, 5, 20, 7, seed=1)]
    b = [t.cuda() if not isinstance(t, list) else [r.cuda() for r in t] for t in synth.frame_inputs(1, 2, 64, 96, 300, 4, 5, 20, 7, seed=2)]
    eager_a = {k: v.clone() for k, v in m(*a).items()}
    eager_b = {k: v.clone() for k, v in m(*b).items()}
    g = m.make_graphed(*a)
    out = g(*a)
    assert all(torch.equal(out[k], eager_a[k]) for k in out)
    out = g(*b)
    assert all(torch.equal(out[k], eager_b[k]) for k in out)
    out = g(*a)
    assert all(torch.equal(out[k], eager_a[k]) for k in out)
