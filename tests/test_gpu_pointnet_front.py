"""PointNet conv1 -> conv2 -> conv3 as one kernel with the activations in registers (csrc/pointnet_front.hip; VERDICT r1 item 8,
ref src/encoders.py:289-291): against fp64 torch, against the three separate launches it replaces, and through the encoder."""
import pytest
import torch

from bevfusion_multimodal_3d_object_detection_amd import _lib as L
from bevfusion_multimodal_3d_object_detection_amd import encoders, engine, synth
from tests.conftest import rel_err

pytestmark = pytest.mark.gpu


def _params(K, seed):
    ws = [synth.normal((64, K), seed, 0, 0.5), synth.normal((128, 64), seed + 1, 0, 0.15), synth.normal((256, 128), seed + 2, 0, 0.1)]
    ss = [synth.normal((c,), seed + 3 + i, 1.0, 0.2) for i, c in enumerate((64, 128, 256))]
    bs = [synth.normal((c,), seed + 6 + i, 0.0, 0.3) for i, c in enumerate((64, 128, 256))]
    return ws, ss, bs


def test_fragment_pack_is_the_documented_permutation(gpu):
    w = synth.normal((256, 128), 5).cuda()
    wf = L.pointnet_front_pack(w)
    # (mb, row, kb, j, h, r) -> (mb, kb, j, h, row, r); lane = 32 h + row
    ref = w.view(8, 32, 4, 4, 2, 4).permute(0, 2, 3, 4, 1, 5).contiguous().view(-1)
    assert torch.equal(wf, ref)
    with pytest.raises(L.BevfError):
        L.pointnet_front_pack(torch.zeros(48, 64, device=gpu))


@pytest.mark.parametrize("M,K", [(1, 4), (31, 4), (32, 5), (1000 + 17, 4), (4096 + 5, 5), (300, 7), (70001, 4)])
def test_fused_front_matches_fp64_and_the_separate_launches(gpu, M, K):
    ws, ss, bs = _params(K, 40 + K)
    x = synth.normal((M, K), 3, 0, 2.0)
    a = x.double()
    for w, s, b in zip(ws, ss, bs):
        a = torch.relu((a @ w.double().t()) * s.double() + b.double())
    g = [t.cuda() for t in ws], [t.cuda() for t in ss], [t.cuda() for t in bs]
    y = torch.full((M * 256 + 64,), -7.0, device=gpu)
    L.pointnet_front(x.cuda(), g[0][0], g[1][0], g[2][0], L.pointnet_front_pack(g[0][1]), g[1][1], g[2][1],
                     L.pointnet_front_pack(g[0][2]), g[1][2], g[2][2], y, M, K)
    assert torch.all(y[M * 256:] == -7.0)                               # nothing past the last row (tail tile masked)
    out = y[:M * 256].view(M, 256)
    assert rel_err(out.cpu(), a.float()) <= 2e-6
    # the launches it replaces: conv1 on the vector ALU (same FMA order: bit-identical), conv2 / conv3 on the implicit-GEMM kernel
    h1 = torch.empty(M * 64, device=gpu)
    L.pointwise_smallk(x.cuda(), g[0][0], g[1][0], g[2][0], h1, M, K, 64, True)
    h2, h3 = torch.empty(M * 128, device=gpu), torch.empty(M * 256, device=gpu)
    L.conv2d_nhwc(h1, g[0][1].contiguous().view(-1), g[1][1], g[2][1], h2, N=M, H=1, W=1, Cin=64, x_cs=64, Cout=128, y_cs=128,
                  KH=1, KW=1, stride=1, pad=0, relu=True)
    L.conv2d_nhwc(h2, g[0][2].contiguous().view(-1), g[1][2], g[2][2], h3, N=M, H=1, W=1, Cin=128, x_cs=128, Cout=256, y_cs=256,
                  KH=1, KW=1, stride=1, pad=0, relu=True)
    assert rel_err(out, h3.view(M, 256)) <= 2e-6
    # a point's features do not depend on where it sits in the batch (every output element is one fixed FMA sequence)
    if M > 64:
        perm = torch.randperm(M, generator=torch.Generator().manual_seed(1))
        y2 = torch.empty(M * 256, device=gpu)
        L.pointnet_front(x[perm].contiguous().cuda(), g[0][0], g[1][0], g[2][0], L.pointnet_front_pack(g[0][1]), g[1][1], g[2][1],
                         L.pointnet_front_pack(g[0][2]), g[1][2], g[2][2], y2, M, K)
        assert torch.equal(y2.view(M, 256), out[perm.cuda()])


def test_fused_front_refuses_bad_arguments(gpu):
    ws, ss, bs = _params(4, 9)
    g = [t.cuda() for t in ws + ss + bs]
    y = torch.empty(10 * 256, device=gpu)
    with pytest.raises(L.BevfError):                                    # fragments of the wrong size
        L.pointnet_front(torch.zeros(10, 4, device=gpu), g[0], g[3], g[6], g[1].view(-1)[:100], g[4], g[7],
                         L.pointnet_front_pack(g[2]), g[5], g[8], y, 10, 4)
    with pytest.raises(L.BevfError):                                    # K > 8
        L.pointnet_front(torch.zeros(10, 9, device=gpu), torch.zeros(64, 9, device=gpu), g[3], g[6], L.pointnet_front_pack(g[1]),
                         g[4], g[7], L.pointnet_front_pack(g[2]), g[5], g[8], y, 10, 9)


@pytest.mark.parametrize("cin", [4, 5])
def test_encoder_with_and_without_the_fused_front(gpu, monkeypatch, cin):
    from oracle import ref_model
    ora = ref_model.PointMLPMax(cin, [64, 128, 256, 512, 1024])
    synth.fill_state_dict_(ora, 61)
    ora.eval()
    pts = synth.frame_inputs(2, 1, 32, 32, 3000 + 7, cin, seed=62)[1]
    with torch.no_grad():
        ref = ora(pts)
    outs = []
    for fuse in (True, False):
        monkeypatch.setattr(engine, "FUSE_POINTNET_FRONT", fuse)
        enc = encoders.PointNetLiDAREncoder(input_channels=cin, feat_dim=1024)
        enc.load_state_dict(ora.state_dict())
        enc = enc.cuda().eval()
        outs.append(enc(pts.cuda()))
        assert (enc._eng().front is not None) == fuse
        assert rel_err(outs[-1].cpu(), ref) <= 1e-4
    assert rel_err(outs[0], outs[1]) <= 2e-6
