"""Seeded random sweep of the convolution entry points (exact fp32, f32x3, bf16) over shapes the fixed cases do not
name: non-square filters, odd sizes, padding 0..k-1, strides, channel strides on input / output / residual, every tile
variant.  Reference: torch conv2d in fp64 on the CPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from bevfusion_multimodal_3d_object_detection_amd import _lib as L
from bevfusion_multimodal_3d_object_detection_amd import synth

pytestmark = pytest.mark.gpu


def _cases(n, seed):
    rs = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        kh, kw = int(rs.choice([1, 2, 3, 5])), int(rs.choice([1, 2, 3, 5]))
        stride = int(rs.choice([1, 1, 2, 3]))
        pad = int(rs.randint(0, max(kh, kw)))
        N, H, W = int(rs.randint(1, 4)), int(rs.randint(1, 34)), int(rs.randint(1, 41))
        if (H + 2 * pad - kh) // stride + 1 <= 0 or (W + 2 * pad - kw) // stride + 1 <= 0 or H + 2 * pad < kh or W + 2 * pad < kw:
            continue
        cin = int(rs.choice([32, 64, 96, 128]))
        cout = int(rs.choice([4, 20, 32, 64, 72, 128, 132, 200]))
        out.append(dict(N=N, H=H, W=W, cin=cin, cout=cout, kh=kh, kw=kw, stride=stride, pad=pad,
                        x_cs=cin + int(rs.choice([0, 0, 32, 64])), y_cs=cout + int(rs.choice([0, 0, 4, 60])),
                        res=bool(rs.randint(0, 2)), relu=bool(rs.randint(0, 2)), affine=bool(rs.randint(0, 2)),
                        tile=int(rs.choice([0, 0, 1, 2, 3, 4, 5, 6, 7])), seed=int(rs.randint(1, 1 << 20))))
    return out


def _run(c, mode):
    N, H, W, cin, cout, kh, kw, s, p = (c[k] for k in ("N", "H", "W", "cin", "cout", "kh", "kw", "stride", "pad"))
    if kh != kw and p > min(kh, kw) - 1 + 10:
        pass
    x = synth.normal((N, H, W, c["x_cs"]), c["seed"])
    w = synth.normal((cout, kh, kw, cin), c["seed"] + 1, 0.0, (2.0 / (kh * kw * cin)) ** 0.5)
    sc = synth.uniform((cout,), c["seed"] + 2, 0.5, 1.5) if c["affine"] else None
    sh = synth.normal((cout,), c["seed"] + 3) if c["affine"] else None
    Ho, Wo = (H + 2 * p - kh) // s + 1, (W + 2 * p - kw) // s + 1
    res_cs = cout + 8
    res = synth.normal((N, Ho, Wo, res_cs), c["seed"] + 4) if c["res"] else None
    if mode == "bf16":
        x, w = x.bfloat16().float(), w.bfloat16().float()
        res = res.bfloat16().float() if res is not None else None
    ref = F.conv2d(x[..., :cin].double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), stride=s, padding=p)
    if sc is not None:
        ref = ref * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    if res is not None:
        ref = ref + res[..., :cout].double().permute(0, 3, 1, 2)
    if c["relu"]:
        ref = ref.relu()
    dt = torch.bfloat16 if mode == "bf16" else torch.float32
    y = torch.full((N * Ho * Wo * c["y_cs"],), 7.0, dtype=dt, device="cuda")
    wg = w.reshape(-1).cuda()
    tile = c["tile"]
    if mode == "f32x3":
        wg = L.split_weights_f32x3(wg)
        tile = tile if tile in (0, 1, 3, 4) else 0
    elif mode == "bf16":
        wg = wg.bfloat16()
    L.conv2d_nhwc(x.reshape(-1).to(dt).cuda(), wg, None if sc is None else sc.cuda(), None if sh is None else sh.cuda(), y,
                  N=N, H=H, W=W, Cin=cin, x_cs=c["x_cs"], Cout=cout, y_cs=c["y_cs"], KH=kh, KW=kw, stride=s, pad=p,
                  relu=c["relu"], res=None if res is None else res.reshape(-1).to(dt).cuda(), res_cs=res_cs if res is not None else 0,
                  tile=tile)
    got = y.float().cpu().view(N, Ho, Wo, c["y_cs"])
    assert bool((got[..., cout:] == 7.0).all()), "wrote outside its channel slice"
    err = float((got[..., :cout].double().permute(0, 3, 1, 2) - ref).abs().max()) / max(float(ref.abs().max()), 1e-6)
    return err


@pytest.mark.parametrize("c", _cases(40, 1234), ids=lambda c: f"{c['N']}x{c['H']}x{c['W']}_{c['cin']}to{c['cout']}_k{c['kh']}x{c['kw']}s{c['stride']}p{c['pad']}t{c['tile']}")
def test_conv_fuzz_fp32(gpu, c):
    assert _run(c, "f32") <= 2e-5


@pytest.mark.parametrize("c", _cases(16, 99), ids=lambda c: f"{c['N']}x{c['H']}x{c['W']}_{c['cin']}to{c['cout']}_k{c['kh']}x{c['kw']}s{c['stride']}p{c['pad']}t{c['tile']}")
def test_conv_fuzz_f32x3(gpu, c):
    assert _run(c, "f32x3") <= 2e-5


@pytest.mark.parametrize("c", _cases(16, 7), ids=lambda c: f"{c['N']}x{c['H']}x{c['W']}_{c['cin']}to{c['cout']}_k{c['kh']}x{c['kw']}s{c['stride']}p{c['pad']}t{c['tile']}")
def test_conv_fuzz_bf16(gpu, c):
    c = dict(c, cin=64 if c["cin"] % 64 else c["cin"])
    c["x_cs"] = max(c["x_cs"], c["cin"]) // 8 * 8
    c["y_cs"] = (c["y_cs"] + 7) // 8 * 8
    assert _run(c, "bf16") <= 6e-3


def _grad_cases(n, seed):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        k = int(rs.choice([1, 3]))
        stride = int(rs.choice([1, 1, 2]))
        out.append(dict(N=int(rs.randint(1, 4)), H=int(rs.randint(2, 30)), W=int(rs.randint(2, 30)),
                        cin=int(rs.choice([32, 64, 128])), cout=int(rs.choice([32, 64, 96, 160])), k=k, stride=stride,
                        pad=k // 2, seed=int(rs.randint(1, 1 << 20))))
    return out


@pytest.mark.parametrize("c", _grad_cases(24, 4321), ids=lambda c: f"{c['N']}x{c['H']}x{c['W']}_{c['cin']}to{c['cout']}_k{c['k']}s{c['stride']}")
def test_conv_gradients_fuzz(gpu, c):
    """Weight and data gradients (MFMA wgrad with the tap table; dgrad incl. the stride-2 parity classes) vs autograd."""
    from bevfusion_multimodal_3d_object_detection_amd import training
    N, H, W, cin, cout, k, s, p = (c[q] for q in ("N", "H", "W", "cin", "cout", "k", "stride", "pad"))
    x = synth.normal((N, cin, H, W), c["seed"]).double().requires_grad_(True)
    w = synth.normal((cout, cin, k, k), c["seed"] + 1, 0.0, 0.05).double().requires_grad_(True)
    yr = F.conv2d(x, w, None, s, p)
    dy = synth.normal(tuple(yr.shape), c["seed"] + 2)
    yr.backward(dy.double())
    nh = lambda t: t.permute(0, 2, 3, 1).contiguous().view(-1).float().cuda()
    xg, dyg, wg = nh(x.detach()), nh(dy), w.detach().float().cuda()
    dw = training.conv_wgrad(xg, dyg, N, H, W, cin, cout, k, s, p)
    dx = training.conv_dgrad(dyg, wg, N, H, W, cin, cout, k, s, p)
    ew = float((dw.permute(0, 3, 1, 2).double().cpu() - w.grad).abs().max()) / max(float(w.grad.abs().max()), 1e-9)
    ex = float((dx[:N * H * W * cin].view(N, H, W, cin).permute(0, 3, 1, 2).double().cpu() - x.grad).abs().max()) / max(float(x.grad.abs().max()), 1e-9)
    assert ew <= 2e-5 and ex <= 2e-5, (ew, ex)
