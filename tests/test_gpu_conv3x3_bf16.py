"""The bf16 3x3 / stride 1 / pad 1 direct convolution (csrc/conv3x3_bf16.hip: LDS patch shared by the nine taps, filters in
MFMA fragment order) against fp64 torch, against the implicit-GEMM bf16 kernel it replaces, and -- on small-integer data,
where every partial sum is exact whatever the summation order -- bit for bit.  Unpinned by the reference (no bf16 run there)."""
import pytest
import torch
import torch.nn.functional as F

from bevfusion_multimodal_3d_object_detection_amd import _lib as L
from bevfusion_multimodal_3d_object_detection_amd import synth
from tests.conftest import rel_err

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def r16(t):
    return t.to(BF).float()


def nhwc(t, cs=None):
    """NCHW float -> flat bf16 NHWC with channel stride cs (the gap holds a poison value the kernel must not read into y)."""
    n, c, h, w = t.shape
    v = t.permute(0, 2, 3, 1).contiguous()
    if cs and cs > c:
        v = torch.cat([v, torch.full((n, h, w, cs - c), 77.0)], dim=3)
    return v.reshape(-1).to(BF).cuda()


def run(x, w, scale, shift, res, relu, x_cs=None, y_cs=None, res_cs=None, tile=0):
    N, cin, H, W = x.shape
    cout = w.shape[0]
    x_cs, y_cs = x_cs or cin, y_cs or cout
    res_cs = (res_cs or cout) if res is not None else 0
    wp = L.conv3x3_pack_bf16(w.permute(0, 2, 3, 1).contiguous().view(-1).to(BF).cuda(), cout, cin)
    y = torch.full((N * H * W * y_cs,), 55.0, dtype=BF, device="cuda")
    L.conv3x3_bf16(nhwc(x, x_cs), wp, None if scale is None else scale.cuda(), None if shift is None else shift.cuda(), y,
                   N=N, H=H, W=W, Cin=cin, x_cs=x_cs, Cout=cout, y_cs=y_cs, relu=relu,
                   res=None if res is None else nhwc(res, res_cs), res_cs=res_cs, tile=tile)
    torch.cuda.synchronize()
    full = y.float().view(N, H, W, y_cs).cpu()
    if y_cs > cout:
        assert torch.all(full[..., cout:] == 55.0), "wrote outside its channel slice"
    return full[..., :cout].permute(0, 3, 1, 2)


SHAPES = [  # N, H, W, cin, cout, res, relu, x_cs, y_cs
    (2, 9, 13, 64, 64, False, True, None, None),
    (1, 16, 16, 32, 64, False, False, None, None),
    (1, 1, 1, 64, 128, False, True, None, None),
    (1, 17, 33, 64, 128, True, True, None, None),
    (3, 5, 40, 96, 192, True, False, 128, 256),            # channel slices on both sides, CT = 64 (192 % 128 != 0)
    (1, 30, 31, 128, 320, False, True, None, None),        # the CenterNet 3x3 (5 x 64)
    (1, 40, 40, 256, 256, True, True, None, 512),
    (1, 12, 12, 768, 512, False, True, None, None),        # bev_fusion of the three-modality model
    (2, 57, 100, 64, 64, True, True, None, None),          # a layer-3-sized map: edge blocks in both directions
    (1, 20, 24, 64, 192, True, True, 128, 256),            # Cin = 64 (the one-image kernel) with three channel tiles and channel slices
    (1, 33, 18, 64, 128, False, True, None, None),         # Cin = 64 with a 128-channel tile (stays on the two-buffer kernel)
]


@pytest.mark.parametrize("N,H,W,cin,cout,res,relu,x_cs,y_cs", SHAPES)
def test_conv3x3_bf16_against_fp64(gpu, N, H, W, cin, cout, res, relu, x_cs, y_cs):
    x = r16(synth.normal((N, cin, H, W), 1))
    w = r16(synth.normal((cout, cin, 3, 3), 2, 0, (1.0 / (cin * 9)) ** 0.5))
    scale, shift = synth.uniform((cout,), 3, 0.5, 1.5), synth.normal((cout,), 4, 0, 0.3)
    rs = r16(synth.normal((N, cout, H, W), 5)) if res else None
    ref = F.conv2d(x.double(), w.double(), None, 1, 1) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if res:
        ref = ref + rs.double()
    if relu:
        ref = F.relu(ref)
    got = run(x, w, scale, shift, rs, relu, x_cs, y_cs)
    assert rel_err(got, ref.float()) <= 4e-3                  # exact products, fp32 accumulate, one bf16 rounding at the store
    for tile in (1, 2, 3, 4, 5):                              # the buffering / block-height / persistent variants run the same arithmetic in the same order
        assert torch.equal(run(x, w, scale, shift, rs, relu, x_cs, y_cs, tile=tile), got), tile
    # the kernel it replaces on the same operands: both round the same fp32-level value to bf16
    y2 = torch.zeros(N * H * W * cout, dtype=BF, device=gpu)
    L.conv2d_nhwc(nhwc(x), w.permute(0, 2, 3, 1).contiguous().view(-1).to(BF).cuda(), scale.cuda(), shift.cuda(), y2, N=N, H=H, W=W,
                  Cin=cin, x_cs=cin, Cout=cout, y_cs=cout, KH=3, KW=3, stride=1, pad=1, relu=relu,
                  res=nhwc(rs) if res else None, res_cs=cout if res else 0) if cin % 64 == 0 else None
    if cin % 64 == 0:
        old = y2.float().view(N, H, W, cout).permute(0, 3, 1, 2).cpu()
        assert rel_err(got, old) <= 4e-3
        assert float((got != old).float().mean()) <= 0.08     # they differ only where a value sits on a bf16 rounding boundary


@pytest.mark.parametrize("N,H,W,cin,cout", [(1, 19, 23, 64, 64), (2, 16, 35, 128, 128), (1, 33, 17, 768, 512), (1, 7, 5, 32, 320)])
def test_conv3x3_bf16_exact_on_integer_data(gpu, N, H, W, cin, cout):
    """x in {0, 1} (sparse), w in {-1, 0, 1}, asymmetric in every index: all partial sums are small integers, exact in fp32 in
    any order and exact in bf16 -> the result must EQUAL the reference (a wrong tap, channel, swizzle or border shows up)."""
    g = torch.Generator().manual_seed(1234 + cin + cout)
    x = (torch.rand((N, cin, H, W), generator=g) < 1.0 / 16).float()
    w = torch.randint(-1, 2, (cout, cin, 3, 3), generator=g).float()
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    assert float(ref.abs().max()) <= 256                      # integers up to 256 are bf16 numbers
    for tile in (1, 2, 3, 4, 5):                              # two patch buffers / one (4 workgroups per CU) / one with 32-row blocks / persistent
        got = run(x, w, None, None, None, False, tile=tile)
        assert torch.equal(got.double(), ref), tile


def test_conv3x3_bf16_refuses_what_it_cannot_do(gpu):
    x = torch.zeros(16 * 16 * 48, dtype=BF, device=gpu)
    y = torch.zeros(16 * 16 * 64, dtype=BF, device=gpu)
    wp = torch.zeros(L.lib().bevf_conv3x3_pack_elems(64, 64), dtype=BF, device=gpu)
    with pytest.raises(L.BevfError):                          # Cin not a multiple of 32
        L.conv3x3_bf16(x, wp, None, None, y, N=1, H=16, W=16, Cin=48, x_cs=48, Cout=64, y_cs=64, relu=False)
    with pytest.raises(L.BevfError):                          # packed filter of another layer
        L.conv3x3_bf16(torch.zeros(16 * 16 * 64, dtype=BF, device=gpu), wp[:100], None, None, y, N=1, H=16, W=16, Cin=64, x_cs=64,
                       Cout=64, y_cs=64, relu=False)
    with pytest.raises(L.BevfError):                          # fp32 filter
        L.conv3x3_pack_bf16(torch.zeros(64 * 9 * 64, device=gpu), 64, 64)


@pytest.mark.parametrize("N,H,W,cin,cout", [(8, 225, 400, 64, 64), (4, 128, 128, 256, 320), (2, 113, 200, 128, 128)])
def test_conv3x3_bf16_variants_agree_bit_for_bit_at_full_occupancy(gpu, N, H, W, cin, cout):
    """Synchronisation errors in the hand-written LDS-DMA waits only show when the chip is full and the DMA queues are long (round 3:
    a deeper filter ring let patch pieces fly across a chunk boundary -- every small-shape test passed, BASELINE config 5 did not).
    The buffering variants share the arithmetic and its order but not the wait structure, so at a size that fills every CU several
    times over they must agree bit for bit, run after run, and with the implicit-GEMM kernel to bf16 rounding."""
    g = torch.Generator().manual_seed(7)
    x = torch.randn((N * H * W * cin,), generator=g).clamp_(min=0).to(BF).cuda()
    w = (torch.randn((cout * 9 * cin,), generator=g) * (1.0 / (9 * cin)) ** 0.5).to(BF).cuda()
    wp = L.conv3x3_pack_bf16(w, cout, cin)
    res = torch.randn((N * H * W * cout,), generator=g).to(BF).cuda()
    outs = []
    for tile in (1, 2, 3, 4, 5, 1, 4, 5):
        y = torch.empty(N * H * W * cout, dtype=BF, device=gpu)
        L.conv3x3_bf16(x, wp, None, None, y, N=N, H=H, W=W, Cin=cin, x_cs=cin, Cout=cout, y_cs=cout, relu=True, res=res, res_cs=cout, tile=tile)
        outs.append(y)
    torch.cuda.synchronize()
    for y in outs[1:]:
        assert torch.equal(y, outs[0])
    if cin % 64 == 0:
        ref = torch.empty_like(outs[0])
        L.conv2d_nhwc(x, w, None, None, ref, N=N, H=H, W=W, Cin=cin, x_cs=cin, Cout=cout, y_cs=cout, KH=3, KW=3, stride=1, pad=1, relu=True,
                      res=res, res_cs=cout)
        assert rel_err(outs[0].float().cpu(), ref.float().cpu()) <= 8e-3
