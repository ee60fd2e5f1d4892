#!/usr/bin/env python3
"""Weight-gradient kernels A/B: the Winograd-domain kernel (csrc/conv_wino_wgrad.hip) against the pixel-GEMM with atomics
(csrc/conv_wgrad.hip).  First an fp64 check of both on small odd shapes (torch autograd of F.conv2d in float64 on the
GPU), then TF direct-equivalent on the 3x3 / stride 1 layer shapes of the training step (config 4, B=8: 48 images of
448x800).  Usage: wgrad_bench.py [check]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from bevfusion_multimodal_3d_object_detection_amd import _lib as _L
if os.environ.get("BEVF_AB_LIB"): _L.LIB_PATH = os.environ["BEVF_AB_LIB"]          # A/B against another build of the library
from bevfusion_multimodal_3d_object_detection_amd import training as T

dev = torch.device("cuda")


def run(kind, x, dy, N, H, W, cin, cout):
    T.WINO_WGRAD = kind == "wino"
    return T.conv_wgrad(x, dy, N, H, W, cin, cout, 3, 1, 1)


def check(N, H, W, cin, cout):
    g = torch.Generator(device="cuda").manual_seed(N * 1000 + H)
    x = torch.randn(N, H, W, cin, device=dev, generator=g)
    dy = torch.randn(N, H, W, cout, device=dev, generator=g)
    w = torch.zeros(cout, cin, 3, 3, device=dev, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.double().permute(0, 3, 1, 2), w, padding=1)
    (ref,) = torch.autograd.grad(y, w, dy.double().permute(0, 3, 1, 2))
    ref = ref.permute(0, 2, 3, 1)                                        # OHWI
    out = {}
    for kind in ("wino", "gemm"):
        dw = run(kind, x.reshape(-1), dy.reshape(-1), N, H, W, cin, cout).double()
        out[kind] = float((dw - ref).abs().max() / ref.abs().max())
    a = run("wino", x.reshape(-1), dy.reshape(-1), N, H, W, cin, cout).clone()
    b = run("wino", x.reshape(-1), dy.reshape(-1), N, H, W, cin, cout)
    print(f"check N={N} {H}x{W} {cin}->{cout}: max err / max |dW|  wino {out['wino']:.2e}  gemm {out['gemm']:.2e}  "
          f"wino run-to-run identical: {bool(torch.equal(a, b))}", flush=True)
    assert out["wino"] < 2e-5, out


if not os.environ.get("BEVF_WW_DEAD"):
  for a in ((2, 13, 21, 64, 64), (1, 33, 18, 128, 64), (3, 8, 7, 64, 192), (5, 30, 50, 64, 64)):
    check(*a)
if len(sys.argv) > 1 and sys.argv[1] == "check":
    sys.exit(0)

SHAPES = [("layer1", 48, 112, 200, 64, 64), ("layer2", 48, 56, 100, 128, 128), ("layer3", 48, 28, 50, 256, 256),
          ("layer4", 48, 14, 25, 512, 512), ("fusion1", 8, 50, 50, 512, 512), ("fusion2", 8, 50, 50, 512, 256),
          ("layer1_900", 48, 225, 400, 64, 64)]
for name, N, H, W, cin, cout in SHAPES:
    x = torch.randn(N * H * W * cin, device=dev)
    dy = torch.randn(N * H * W * cout, device=dev)
    flops = 2.0 * N * H * W * cout * 9 * cin
    line = f"{name:11s} N={N} {H}x{W} {cin}->{cout}:"
    for kind in ("gemm", "wino"):
        for _ in range(3):
            run(kind, x, dy, N, H, W, cin, cout)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run(kind, x, dy, N, H, W, cin, cout)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        line += f"  {kind} {ms * 1e3:8.1f} us {flops / ms / 1e9:6.1f} TF"
    print(line, flush=True)
