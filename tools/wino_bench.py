#!/usr/bin/env python3
"""Fused fp32 Winograd convolution on the 3x3 / stride 1 layer shapes of config 2 at B=8 (TF direct-equivalent), with both
block geometries (16x16-pixel and 32x8-pixel blocks), each per image and over the images' rows stacked into one map; the launcher
(auto) picks the tiling with the fewest blocks."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bevfusion_multimodal_3d_object_detection_amd import _lib as L

SHAPES = {  # name: (N, H, W, Cin, Cout, residual)
    "layer1": (48, 225, 400, 64, 64, False), "layer1r": (48, 225, 400, 64, 64, True),
    "layer2": (48, 113, 200, 128, 128, True), "layer3": (48, 57, 100, 256, 256, True),
    "fusion1": (8, 128, 128, 512, 512, False), "fusion2": (8, 128, 128, 512, 256, False), "head": (8, 128, 128, 256, 320, False),
    "lift": (8, 128, 128, 64, 128, False), "odd": (3, 37, 45, 96, 80, True),
    "t_layer2": (48, 56, 100, 128, 128, True), "t_layer3": (48, 28, 50, 256, 256, True), "t_layer4": (48, 14, 25, 512, 512, True),
}
dev = torch.device("cuda")
names = sys.argv[1].split(",") if len(sys.argv) > 1 else list(SHAPES)
for name in names:
    N, H, W, Cin, Cout, res = SHAPES[name]
    x = torch.randn(N * H * W * Cin, device=dev)
    w = torch.randn(Cout * 9 * Cin, device=dev) * 0.05
    u = L.wino_filter_transform(w, Cout, Cin)
    sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
    r = torch.randn(N * H * W * Cout, device=dev) if res else None
    flops = 2.0 * N * H * W * Cout * 9 * Cin
    line = f"{name:8s} N={N} {H}x{W} {Cin}->{Cout}{' +res' if res else ''}:"
    for tile, label in ((1, "16x16"), (2, "32x8"), (3, "16x16 stacked"), (4, "32x8 stacked"), (0, "auto")):
        y = torch.empty(N * H * W * Cout, device=dev)
        run = lambda: L.conv3x3_wino(x, u, sc, sh, y, N=N, H=H, W=W, Cin=Cin, x_cs=Cin, Cout=Cout, y_cs=Cout, relu=True,
                                     res=r, res_cs=Cout if res else 0, tile=tile)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        line += f"  {label} {ms * 1e3:7.1f} us {flops / ms / 1e9:5.1f} TF"
    print(line, flush=True)
