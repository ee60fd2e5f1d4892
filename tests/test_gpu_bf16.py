"""bf16 storage / fp32 accumulate path (BASELINE configs 3 and 5).  Parity is against the fp32 CPU oracle run on
the SAME bf16-rounded weights: what differs is only the bf16 rounding of every stored activation, so tolerances
are bf16-sized (2^-9 per store) and stated per test.  Unpinned by the reference (it has no bf16 run)."""
import pytest
import torch
import torch.nn.functional as F

from bevfusion_multimodal_3d_object_detection_amd import _lib as L
from bevfusion_multimodal_3d_object_detection_amd import fusion, synth
from oracle import ref_model
from tests.conftest import rel_err

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def r16(t):
    return t.to(BF).float()


@pytest.mark.parametrize("N,H,W,cin,cout,k,stride,pad,res,tile", [(2, 9, 13, 64, 64, 3, 1, 1, False, 0), (1, 17, 11, 64, 128, 3, 2, 1, True, 1),
                                                                 (1, 30, 31, 128, 320, 3, 1, 1, False, 0), (2, 8, 8, 64, 128, 1, 2, 0, False, 4),
                                                                 (1, 40, 40, 256, 256, 3, 1, 1, True, 5), (1, 12, 12, 768, 512, 3, 1, 1, False, 3)])
def test_conv_bf16(gpu, N, H, W, cin, cout, k, stride, pad, res, tile):
    x = r16(synth.normal((N, cin, H, W), 1))
    w = r16(synth.normal((cout, cin, k, k), 2, 0, (1.0 / (cin * k * k)) ** 0.5))
    scale, shift = synth.uniform((cout,), 3, 0.5, 1.5), synth.normal((cout,), 4, 0, 0.3)
    ref = F.conv2d(x, w, None, stride, pad) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    Ho, Wo = ref.shape[-2:]
    rs = r16(synth.normal((N, cout, Ho, Wo), 5)) if res else None
    if res:
        ref = ref + rs
    ref = F.relu(ref)
    nh = lambda t: t.permute(0, 2, 3, 1).contiguous().view(-1).to(BF).cuda()
    y = torch.zeros(N * Ho * Wo * cout, dtype=BF, device=gpu)
    L.conv2d_nhwc(nh(x), w.permute(0, 2, 3, 1).contiguous().view(-1).to(BF).cuda(), scale.cuda(), shift.cuda(), y, N=N, H=H,
                  W=W, Cin=cin, x_cs=cin, Cout=cout, y_cs=cout, KH=k, KW=k, stride=stride, pad=pad, relu=True,
                  res=nh(rs) if res else None, res_cs=cout if res else 0, tile=tile)
    got = y.float().view(N, Ho, Wo, cout).permute(0, 3, 1, 2).cpu()
    assert rel_err(got, ref) <= 4e-3                        # exact products, fp32 accumulate, one bf16 rounding at the store


def test_conv_bf16_colmax(gpu):
    B, npts, cin, cout = 2, 300, 64, 128
    x, w = r16(synth.normal((B * npts, cin), 21)), r16(synth.normal((cout, cin), 22, 0, 0.125))
    ref = F.relu(x @ w.t()).view(B, npts, cout).max(dim=1)[0]
    cm = torch.zeros(B, cout, dtype=torch.int32, device=gpu)
    L.conv2d_nhwc(x.view(-1).to(BF).cuda(), w.view(-1).to(BF).cuda(), None, None, None, N=B * npts, H=1, W=1, Cin=cin, x_cs=cin,
                  Cout=cout, y_cs=cout, KH=1, KW=1, stride=1, pad=0, relu=True, colmax=cm, rows_per_group=npts)
    assert rel_err(cm.view(torch.float32).cpu(), ref) <= 2e-5   # the fused max keeps the fp32 accumulator


@pytest.mark.parametrize("modality,bev", [("camera+lidar", (50, 50)), ("camera+lidar+radar", (64, 48))])
def test_detector_bf16_vs_fp32_oracle_on_bf16_weights(gpu, modality, bev):
    ora = ref_model.make_detector(modality, *bev)
    synth.fill_state_dict_(ora, 5)
    with torch.no_grad():
        for p in ora.parameters():
            p.copy_(r16(p))
        for n, b in ora.named_buffers():
            if b.dtype.is_floating_point:
                b.copy_(r16(b))
    ora.eval()
    m = fusion.create_detector(modality, "bev", "centernet", bev_h=bev[0], bev_w=bev[1])
    m.load_state_dict(ora.state_dict())
    m = m.cuda().bfloat16().eval()
    imgs, pts, radars = synth.frame_inputs(1, 2, 64, 96, 300, 4, 5 if "radar" in modality else 0, 20, 7, seed=9)
    out = m(imgs.cuda(), pts.cuda(), [r.cuda() for r in radars] if radars else None)
    with torch.no_grad():
        ref = ora(imgs, pts, radars or None)
    for k in ref:
        assert out[k].dtype == torch.float32
        e = rel_err(out[k].cpu(), ref[k])
        assert e <= 3e-2, (k, e)                             # ~25 layers of bf16 activation rounding (2^-9 each)
    # and the fp32 model on the same weights agrees with the bf16 model to bf16 accuracy
    m32 = fusion.create_detector(modality, "bev", "centernet", bev_h=bev[0], bev_w=bev[1])
    m32.load_state_dict(ora.state_dict())
    o32 = m32.cuda().eval()(imgs.cuda(), pts.cuda(), [r.cuda() for r in radars] if radars else None)
    assert rel_err(out["size"].cpu(), o32["size"].cpu()) <= 3e-2


@pytest.mark.parametrize("N,H,W", [(2, 64, 96), (1, 37, 50), (1, 33, 46), (2, 450, 800)])
def test_stem_bf16_mfma(gpu, N, H, W):
    """bf16-MFMA stem (column tile expanded in LDS) against fp32 torch on the bf16-rounded image and filter."""
    x = synth.normal((N, 3, H, W), 1)
    w = synth.normal((64, 3, 7, 7), 2, 0, (1.0 / 147) ** 0.5)
    scale, shift = synth.uniform((64,), 3, 0.5, 1.5), synth.normal((64,), 4, 0, 0.3)
    ref = F.relu(F.conv2d(r16(x), r16(w), None, 2, 3) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    Ho, Wo = ref.shape[-2:]
    wp = L.stem_pack_bf16(w.cuda())
    y = torch.zeros(N * Ho * Wo * 64, dtype=BF, device=gpu)
    xg = x.cuda()
    L.stem_conv7x7_bf16mma(xg, wp, scale.cuda(), shift.cuda(), y, N, H, W, relu=True)
    got = y.float().view(N, Ho, Wo, 64).permute(0, 3, 1, 2).cpu()
    assert rel_err(got, ref) <= 4e-3
