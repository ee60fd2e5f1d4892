"""Fused stem + max-pool kernel (csrc/stem.hip: stem_pool7x7; ref src/encoders.py:154-157) against the two separate
kernels it replaces: same patch staging, MFMA schedule, fma and max, so the pooled map must be BIT-identical -- over
odd / tiny / unaligned image sizes (edge strips, segments that end mid-chunk, W % 4 != 0 staging path) and the full
900x1600 frame."""
import pytest
import torch

from bevfusion_multimodal_3d_object_detection_amd import _lib as L
from bevfusion_multimodal_3d_object_detection_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,H,W", [(2, 64, 96), (1, 37, 50), (3, 33, 46), (1, 7, 9), (2, 1, 1), (1, 450, 800), (1, 129, 1027),
                                   (6, 900, 1600), (1, 255, 482)])
def test_stem_pool_is_bit_identical_to_stem_then_maxpool(gpu, N, H, W):
    x = synth.normal((N, 3, H, W), 1000 + H).cuda()
    w = synth.normal((64, 3, 7, 7), 2, 0, (1.0 / 147) ** 0.5)
    packed = torch.zeros(148, 64)
    packed[:147] = w.reshape(64, 147).t()
    packed = packed.contiguous().view(-1).cuda()
    scale, shift = synth.uniform((64,), 3, 0.5, 1.5).cuda(), synth.normal((64,), 4, 0, 0.3).cuda()
    H1, W1 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (H1 - 1) // 2 + 1, (W1 - 1) // 2 + 1
    stem = torch.empty(N * H1 * W1 * 64, device=gpu)
    L.stem_conv7x7(x, packed, scale, shift, stem, N, H, W)
    ref = torch.empty(N * Hp * Wp * 64, device=gpu)
    L.maxpool3x3s2(stem, ref, N, H1, W1, 64)
    got = torch.full((N * Hp * Wp * 64,), float("nan"), device=gpu)
    L.stem_pool(x, packed, scale, shift, got, N, H, W)
    assert bool(torch.isfinite(got).all())                       # every pooled element was written
    assert torch.equal(got, ref)


def test_detector_with_and_without_the_fused_stem_pool(gpu):
    from bevfusion_multimodal_3d_object_detection_amd import engine, fusion
    m = fusion.create_detector("camera_only", "bev", "centernet", bev_h=16, bev_w=24)
    synth.fill_state_dict_(m, 4)
    m = m.cuda().eval()
    imgs = synth.normal((2, 2, 3, 97, 131), 6).cuda()
    assert engine.FUSE_STEM_POOL
    a = {k: v.clone() for k, v in m(imgs, None, None).items()}
    engine.FUSE_STEM_POOL = False
    try:
        b = m(imgs, None, None)
    finally:
        engine.FUSE_STEM_POOL = True
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("N,H,W", [(2, 64, 96), (1, 37, 50), (3, 33, 46), (1, 7, 9), (2, 1, 1), (1, 450, 800), (1, 129, 1027),
                                   (6, 900, 1600), (1, 255, 482)])
def test_bf16_stem_pool_is_bit_identical_to_stem_then_maxpool(gpu, N, H, W):
    """stem_pool7x7_bf16mma against bevf_stem_conv7x7_bf16mma -> bevf_maxpool3x3s2_nhwc_bf16 (max commutes with the
    monotonic bf16 rounding, a missing neighbour is 0 after the ReLU)."""
    x = synth.normal((N, 3, H, W), 2000 + H).cuda()
    w = synth.normal((64, 3, 7, 7), 5, 0, (1.0 / 147) ** 0.5).cuda()
    packed = L.stem_pack_bf16(w)
    scale, shift = synth.uniform((64,), 6, 0.5, 1.5).cuda(), synth.normal((64,), 7, 0, 0.3).cuda()
    H1, W1 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (H1 - 1) // 2 + 1, (W1 - 1) // 2 + 1
    stem = torch.empty(N * H1 * W1 * 64, device=gpu, dtype=torch.bfloat16)
    L.stem_conv7x7_bf16mma(x, packed, scale, shift, stem, N, H, W)
    ref = torch.empty(N * Hp * Wp * 64, device=gpu, dtype=torch.bfloat16)
    L.maxpool3x3s2(stem, ref, N, H1, W1, 64)
    got = torch.full((N * Hp * Wp * 64,), float("nan"), device=gpu, dtype=torch.bfloat16)
    L.stem_pool_bf16mma(x, packed, scale, shift, got, N, H, W)
    assert bool(torch.isfinite(got.float()).all())               # every pooled element was written
    assert torch.equal(got, ref)


def test_bf16_detector_with_and_without_the_fused_stem_pool(gpu):
    from bevfusion_multimodal_3d_object_detection_amd import engine, fusion
    m = fusion.create_detector("camera_only", "bev", "centernet", bev_h=16, bev_w=24)
    synth.fill_state_dict_(m, 4)
    m = m.cuda().eval().bfloat16()
    imgs = synth.normal((2, 2, 3, 97, 131), 6).cuda()
    a = {k: v.clone() for k, v in m(imgs, None, None).items()}
    engine.FUSE_STEM_POOL = False
    try:
        b = m(imgs, None, None)
    finally:
        engine.FUSE_STEM_POOL = True
    for k in a:
        assert torch.equal(a[k], b[k]), k
