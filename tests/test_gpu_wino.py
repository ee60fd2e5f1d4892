"""Fused fp32 Winograd F(2x2,3x3) convolution (csrc/conv_wino.hip; conv mode "wino"): parity against fp64
torch and against the exact direct kernel over odd shapes / channel strides / residuals, the filter transform against a
float64 host restatement, and the whole detector in "wino" mode against the CPU oracle (north_star: 1e-4 rel) including
BASELINE config 2 at full size.  Winograd changes the summation structure, so tolerances are fp32-rounding-sized
(observed 3e-7 per layer), never bit-exact against the direct kernel; it IS bit-exact against itself across batch sizes."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from bevfusion_multimodal_3d_object_detection_amd import _lib as L
from bevfusion_multimodal_3d_object_detection_amd import engine, fusion, synth
from oracle import ref_model
from tests.conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture()
def wino_mode():
    before = engine.conv_mode()
    engine.set_conv_mode("wino")
    yield
    engine.set_conv_mode(before)


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def test_filter_transform_matches_float64_host(gpu):
    cout, cin = 80, 96
    w = synth.normal((cout, cin, 3, 3), 3, 0, 0.05)
    u = L.wino_filter_transform(_nhwc(w).view(-1).cuda(), cout, cin).cpu()
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
    U = torch.einsum("ia,ocab,jb->ocij", G, w.double(), G).float()                  # (cout, cin, 4, 4)
    nct, g8 = (cout + 63) // 64, cin // 8
    assert u.numel() == nct * 64 * 16 * cin
    img = u.view(nct, g8, 16, 2, 4, 16, 2, 2)                # [ct][g][f][np][kq = (cin & 7) >> 1][n][nb&1][cin&1]
    rng = np.random.RandomState(0)
    for _ in range(300):
        co, ci, f = int(rng.randint(cout)), int(rng.randint(cin)), int(rng.randint(16))
        nb, q = (co % 64) // 16, ci % 8
        got = img[co // 64, ci // 8, f, nb >> 1, q >> 1, co % 16, nb & 1, q & 1]
        assert float(got) == float(U[co, ci, f >> 2, f & 3]), (co, ci, f)
    # output channels 80..127 of the second 64-channel slab (nb = 2 np + nbl >= 1) do not exist: zero rows
    assert float(img[1, :, :, 1].abs().max()) == 0.0 and float(img[1, :, :, 0, :, :, 1].abs().max()) == 0.0


@pytest.mark.parametrize("N,H,W,cin,cout,res,relu,xcs,ycs", [
    (2, 13, 21, 64, 64, False, True, 64, 64), (1, 33, 18, 96, 80, True, True, 96, 80), (1, 16, 16, 32, 64, False, False, 32, 64),
    (3, 1, 1, 64, 16, False, True, 64, 16), (1, 2, 47, 128, 320, True, False, 128, 320), (2, 31, 17, 64, 100, False, True, 160, 256),
    (1, 50, 50, 256, 256, True, True, 768, 256), (1, 17, 64, 512, 72, False, True, 512, 72),
    # several images whose rows the stacked tilings (tile 3 / 4) cut across: odd H (one dead row between images), even H (two),
    # H just above one block height (only the 16-row blocks can stack)
    (5, 57, 20, 64, 64, True, True, 64, 64), (4, 40, 24, 32, 128, False, True, 32, 128), (7, 17, 9, 64, 64, True, False, 64, 64),
    (3, 34, 30, 32, 64, True, True, 96, 64)])
def test_conv_wino_against_fp64_and_direct(gpu, N, H, W, cin, cout, res, relu, xcs, ycs):
    s = N * 1000 + H * 10 + cin
    x = synth.normal((N, cin, H, W), s + 1).relu() * 2.0
    w = synth.normal((cout, cin, 3, 3), s + 2, 0, (2.0 / (9 * cin)) ** 0.5)
    scale, shift = synth.uniform((cout,), s + 3, 0.5, 1.5), synth.normal((cout,), s + 4, 0, 0.3)
    rs = synth.normal((N, cout, H, W), s + 5) if res else None
    ref = F.conv2d(x.double(), w.double(), None, 1, 1) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if res:
        ref = ref + rs.double()
    if relu:
        ref = ref.relu()
    M = N * H * W
    xb = torch.full((M, xcs), 7.0)
    xb[:, :cin] = _nhwc(x).view(M, cin)                                      # channel-strided input (concat slices)
    rb = _nhwc(rs).view(-1).cuda() if res else None
    w_ohwi = _nhwc(w).view(-1).cuda()
    u = L.wino_filter_transform(w_ohwi, cout, cin)
    outs = []
    for tile in (0, 1, 2, 3, 4):                                # auto; 16x16 / 32x8-pixel blocks per image; the same over stacked rows
        y = torch.full((M * ycs,), -5.0, device=gpu)
        L.conv3x3_wino(xb.view(-1).cuda(), u, scale.cuda(), shift.cuda(), y, N=N, H=H, W=W, Cin=cin, x_cs=xcs, Cout=cout, y_cs=ycs,
                       relu=relu, res=rb, res_cs=cout if res else 0, tile=tile)
        got = y.view(M, ycs).cpu()
        assert rel_err(got[:, :cout].view(N, H, W, cout).permute(0, 3, 1, 2), ref) <= 2e-6
        assert bool((got[:, cout:] == -5.0).all())                          # nothing written past Cout in a wider pixel
        outs.append(got)
    # the geometry changes which block a tile belongs to, not the tile's arithmetic: bit-identical results
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    yd = torch.zeros(M * ycs, device=gpu)
    L.conv2d_nhwc(xb.view(-1).cuda(), w_ohwi, scale.cuda(), shift.cuda(), yd, N=N, H=H, W=W, Cin=cin, x_cs=xcs, Cout=cout, y_cs=ycs,
                  KH=3, KW=3, stride=1, pad=1, relu=relu, res=rb, res_cs=cout if res else 0)
    assert rel_err(got[:, :cout], yd.view(M, ycs)[:, :cout].cpu()) <= 2e-6


def test_conv_wino_refuses_other_geometries(gpu):
    x, y = torch.zeros(64 * 64, device=gpu), torch.zeros(64 * 64, device=gpu)
    u = torch.zeros(L.lib().bevf_wino_filter_floats(64, 64), device=gpu)
    with pytest.raises(L.BevfError, match="multiple of 32"):
        L.conv3x3_wino(x, torch.zeros(L.lib().bevf_wino_filter_floats(64, 16), device=gpu), None, None, y, N=1, H=8, W=8, Cin=16,
                       x_cs=16, Cout=64, y_cs=64, relu=False)
    with pytest.raises(L.BevfError, match="wrong size"):
        L.conv3x3_wino(x, u[:-4], None, None, y, N=1, H=8, W=8, Cin=64, x_cs=64, Cout=64, y_cs=64, relu=False)


@pytest.mark.parametrize("modality,bev", [("camera+lidar+radar", (50, 50)), ("camera_only", (24, 40))])
def test_detector_wino_mode_against_oracle(gpu, wino_mode, modality, bev):
    ora = ref_model.make_detector(modality, *bev)
    synth.fill_state_dict_(ora, 6)
    ora.eval()
    m = fusion.create_detector(modality, "bev", "centernet", bev_h=bev[0], bev_w=bev[1])
    m.load_state_dict(ora.state_dict())
    m = m.cuda().eval()
    imgs, pts, radars = synth.frame_inputs(2, 3, 96, 160, 700 if "lidar" in modality else 0, 4, 5 if "radar" in modality else 0, 30, 7, seed=44)
    cu = lambda t: None if t is None else t.cuda()
    out = m(cu(imgs), cu(pts), [r.cuda() for r in radars] if radars else None)
    assert any(pc.wino for pc in [m.fusion._eng().f1, m.det_head._eng().conv])            # the mode really is in use
    with torch.no_grad():
        ref = ora(imgs, pts, radars or None)
    for k in ref:
        assert rel_err(out[k].cpu(), ref[k]) <= 1e-4, k
    engine.set_conv_mode("f32")
    exact = m(cu(imgs), cu(pts), [r.cuda() for r in radars] if radars else None)
    engine.set_conv_mode("wino")
    for k in ref:
        assert rel_err(out[k].cpu(), exact[k].cpu()) <= 2e-5, k


def test_config2_full_size_wino_against_oracle_and_batch_invariance(gpu, wino_mode):
    m = fusion.create_detector("camera+lidar", "bev", "centernet", bev_h=128, bev_w=128)
    synth.fill_state_dict_(m, 0)
    imgs, pts, _ = synth.frame_inputs(1, 6, 900, 1600, 35000, 4, seed=0x5EED + 2000)
    ora = ref_model.make_detector("camera+lidar", 128, 128)
    ora.load_state_dict(m.state_dict())
    ora.eval()
    with torch.no_grad():
        ref = ora(imgs, pts, None)
    m = m.cuda().eval()
    one = {k: v.clone() for k, v in m(imgs.cuda(), pts.cuda(), None).items()}
    for k in ref:
        assert rel_err(one[k].cpu(), ref[k]) <= 1e-4, k                                  # north_star tolerance at full size
    two = m(torch.cat([imgs, imgs]).cuda(), torch.cat([pts, pts]).cuda(), None)
    for k in one:                                                # the tiling is per image: batching changes no bit
        assert torch.equal(two[k][0:1], one[k]) and torch.equal(two[k][1:2], one[k]), k


def test_conv_wino_bn_partial_sums(gpu):
    """Training forward: the Winograd epilogue leaves {sum(y - pivot), sum((y - pivot)^2)} per (tile block, wave, channel);
    merged by bevf_bn_stats_from_partials_f32 they are the train-mode BatchNorm statistics of the conv output (edge tile
    blocks, a channel count that is not a multiple of 64, a pivot far from the mean)."""
    N, H, W, cin, cout = 2, 37, 29, 64, 80
    x = synth.normal((N, cin, H, W), 71)
    w = synth.normal((cout, cin, 3, 3), 72, 0, 0.06)
    bias = synth.normal((cout,), 73, 3.0, 1.0)                       # means well away from zero
    pivot = synth.normal((cout,), 74, 2.0, 2.0)
    ref = F.conv2d(x.double(), w.double(), bias.double(), 1, 1)
    M = N * H * W
    rows = L.lib().bevf_wino_stat_rows(N, H, W)
    part = torch.full((rows * cout * 2,), float("nan"), device=gpu)
    y = torch.empty(M * cout, device=gpu)
    u = L.wino_filter_transform(_nhwc(w).view(-1).cuda(), cout, cin)
    L.conv3x3_wino(_nhwc(x).view(-1).cuda(), u, None, bias.cuda(), y, N=N, H=H, W=W, Cin=cin, x_cs=cin, Cout=cout, y_cs=cout,
                   relu=False, stats=part, stats_pivot=pivot.cuda())
    assert bool(torch.isfinite(part).all())                          # every (row, channel) slot was written
    mean, var, invstd = (torch.empty(cout, device=gpu) for _ in range(3))
    rc = L.lib().bevf_bn_stats_from_partials_f32(part.data_ptr(), rows, pivot.cuda().data_ptr(), mean.data_ptr(), var.data_ptr(),
                                                 invstd.data_ptr(), M, cout, 1e-5, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    rm, rv = ref.mean(dim=(0, 2, 3)), ref.var(dim=(0, 2, 3), unbiased=False)
    assert rel_err(mean.cpu(), rm) <= 2e-6 and rel_err(var.cpu(), rv) <= 2e-5
    assert rel_err(invstd.cpu(), 1.0 / torch.sqrt(rv + 1e-5)) <= 2e-5
    with pytest.raises(L.BevfError, match="stats need relu = 0"):
        L.conv3x3_wino(_nhwc(x).view(-1).cuda(), u, None, None, y, N=N, H=H, W=W, Cin=cin, x_cs=cin, Cout=cout, y_cs=cout,
                       relu=True, stats=part)
