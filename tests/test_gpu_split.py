"""Opt-in f32x3 convolution (fp32 through three bf16 planes, csrc/conv_split.hip): error against an fp64 reference next
to the exact-fp32 kernel's.  Unpinned by the reference (it has no such mode); the bar is fp32-level error."""
import pytest
import torch

from bevfusion_multimodal_3d_object_detection_amd import _lib as L
from bevfusion_multimodal_3d_object_detection_amd import synth

pytestmark = pytest.mark.gpu

SHAPES = [  # N, H, W, Cin, Cout, k, stride, pad, residual
    (2, 20, 33, 64, 64, 3, 1, 1, True),
    (1, 17, 40, 128, 256, 3, 2, 1, False),
    (3, 9, 11, 256, 320, 1, 1, 0, False),
    (1, 30, 30, 32, 96, 3, 1, 1, True),
]


def _run(x, w_ohwi, sc, sh, res, geom, tile, split):
    N, H, W, Cin, Cout, k, s, p = geom
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    y = torch.empty(N * Ho * Wo * Cout, device=x.device)
    w = L.split_weights_f32x3(w_ohwi) if split else w_ohwi
    L.conv2d_nhwc(x, w, sc, sh, y, N=N, H=H, W=W, Cin=Cin, x_cs=Cin, Cout=Cout, y_cs=Cout, KH=k, KW=k, stride=s, pad=p,
                  relu=True, res=res, res_cs=Cout if res is not None else 0, tile=tile)
    return y.view(N, Ho, Wo, Cout)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("tile", [0, 1, 3, 4])
def test_f32x3_conv_has_fp32_level_error(gpu, shape, tile):
    N, H, W, Cin, Cout, k, s, p, with_res = shape
    x = synth.normal((N, H, W, Cin), 11).cuda()
    w = synth.normal((Cout, k, k, Cin), 12, 0.0, (2.0 / (k * k * Cin)) ** 0.5).cuda()
    sc, sh = synth.uniform((Cout,), 13, 0.5, 1.5).cuda(), synth.normal((Cout,), 14).cuda()
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    res = synth.normal((N * Ho * Wo * Cout,), 15).cuda() if with_res else None
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), stride=s, padding=p)
    ref = ref * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    if res is not None:
        ref = ref + res.double().view(N, Ho, Wo, Cout).permute(0, 3, 1, 2)
    ref = ref.relu().permute(0, 2, 3, 1)
    geom = (N, H, W, Cin, Cout, k, s, p)
    exact = _run(x.reshape(-1), w.reshape(-1), sc, sh, res, geom, 0, False).double()
    split = _run(x.reshape(-1), w.reshape(-1), sc, sh, res, geom, tile, True).double()
    scale = float(ref.abs().max())
    e_exact = float((exact - ref).abs().max()) / scale
    e_split = float((split - ref).abs().max()) / scale
    assert e_exact <= 2e-6
    assert e_split <= 4e-6 and e_split <= 4 * e_exact + 5e-7, (e_split, e_exact)


def test_split_planes_reconstruct_the_weights(gpu):
    w = synth.normal((4096,), 3).cuda() * 3.0
    pl = L.split_weights_f32x3(w).view(3, -1).float()
    back = pl[0].double() + pl[1].double() + pl[2].double()
    assert float((back - w.double()).abs().max()) <= 2.0 ** -24 * float(w.abs().max())


def test_detector_in_f32x3_mode_matches_oracle_like_fp32(gpu):
    """Whole camera+LiDAR detector with every eligible convolution on the f32x3 kernel: within the same 1e-4 of the
    CPU oracle as the exact path, and within 2e-5 of the exact path itself."""
    from bevfusion_multimodal_3d_object_detection_amd import engine, fusion
    from oracle import ref_model
    from tests.conftest import rel_err
    m = fusion.create_detector("camera+lidar", "bev", "centernet", bev_h=64, bev_w=64)
    synth.fill_state_dict_(m, 5)
    imgs, pts, _ = synth.frame_inputs(1, 2, 96, 160, 500, 4, seed=9)
    ora = ref_model.make_detector("camera+lidar", 64, 64)
    ora.load_state_dict(m.state_dict())
    ora.eval()
    with torch.no_grad():
        ref = ora(imgs, pts, None)
    m = m.cuda().eval()
    default = engine.conv_mode()
    engine.set_conv_mode("f32")
    try:
        exact = {k: v.clone() for k, v in m(imgs.cuda(), pts.cuda(), None).items()}
        engine.set_conv_mode("f32x3")
        split = {k: v.clone() for k, v in m(imgs.cuda(), pts.cuda(), None).items()}
        assert m.camera_encoder._engine.blocks[0][0].w.dtype == torch.bfloat16        # really repacked as planes
        engine.set_conv_mode("f32")
        again = m(imgs.cuda(), pts.cuda(), None)
    finally:
        engine.set_conv_mode(default)
    for k in ref:
        assert rel_err(split[k].cpu(), ref[k]) <= 1e-4, k
        assert rel_err(split[k].cpu(), exact[k].cpu()) <= 2e-5, k
        assert torch.equal(again[k], exact[k]), k                                      # back to the exact kernels


def test_f32x3_conv_with_fused_column_max(gpu):
    """The three-plane kernel with the fused max over each group of rows (PointNet's last layer: no activation written),
    against the exact kernel's column max and an fp64 reference."""
    M, P, cin, cout = 6 * 500, 500, 64, 160
    x = synth.normal((M, cin), 41).relu()
    w = synth.normal((cout, cin), 42, 0, (2.0 / cin) ** 0.5)
    scale, shift = synth.uniform((cout,), 43, 0.5, 1.5), synth.normal((cout,), 44, 0, 0.3)
    ref = ((x.double() @ w.double().t()) * scale.double() + shift.double()).relu().view(M // P, P, cout).amax(1)
    out = {}
    for split in (False, True):
        wg = w.contiguous().view(-1).cuda()
        if split:
            wg = L.split_weights_f32x3(wg)
        gmax = torch.zeros(M // P, cout, dtype=torch.int32, device=gpu)
        L.conv2d_nhwc(x.view(-1).cuda(), wg, scale.cuda(), shift.cuda(), None, N=M, H=1, W=1, Cin=cin, x_cs=cin, Cout=cout, y_cs=cout,
                      KH=1, KW=1, stride=1, pad=0, relu=True, colmax=gmax, rows_per_group=P)
        out[split] = gmax.view(torch.float32).cpu()
    from tests.conftest import rel_err
    assert rel_err(out[True], ref) <= 2e-6 and rel_err(out[False], ref) <= 2e-6
    assert rel_err(out[True], out[False]) <= 2e-6


def test_detector_in_wino_x3_mode_matches_oracle(gpu):
    """Opt-in mixed mode: Winograd for the 3x3 / stride 1 layers, the three-plane kernel for every other convolution (stride-2,
    1x1, PointNet incl. its fused point max): within the same 1e-4 of the CPU oracle, 2e-5 of the default mode."""
    from bevfusion_multimodal_3d_object_detection_amd import engine, fusion
    from oracle import ref_model
    from tests.conftest import rel_err
    m = fusion.create_detector("camera+lidar+radar", "bev", "centernet", bev_h=50, bev_w=50)
    synth.fill_state_dict_(m, 6)
    imgs, pts, radars = synth.frame_inputs(2, 2, 96, 160, 700, 4, n_radars=5, seed=11)
    ora = ref_model.make_detector("camera+lidar+radar", 50, 50)
    ora.load_state_dict(m.state_dict())
    ora.eval()
    with torch.no_grad():
        ref = ora(imgs, pts, radars)
    m = m.cuda().eval()
    rad = [r.cuda() for r in radars] if radars is not None else None
    default = engine.conv_mode()
    try:
        engine.set_conv_mode("wino")
        base = {k: v.clone() for k, v in m(imgs.cuda(), pts.cuda(), rad).items()}
        engine.set_conv_mode("wino_x3")
        mixed = {k: v.clone() for k, v in m(imgs.cuda(), pts.cuda(), rad).items()}
        eng = m.camera_encoder._engine
        assert eng.blocks[0][0].wino and eng.proj.w.dtype == torch.bfloat16               # Winograd kept, the 1x1 projection repacked as planes
        assert m.lidar_encoder._engine.layers[-1].w.dtype == torch.bfloat16                # PointNet's max layer too
    finally:
        engine.set_conv_mode(default)
    for k in ref:
        assert rel_err(mixed[k].cpu(), ref[k]) <= 1e-4, k
        assert rel_err(mixed[k].cpu(), base[k].cpu()) <= 2e-5, k
