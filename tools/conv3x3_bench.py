#!/usr/bin/env python3
"""A/B of the bf16 3x3 / stride 1 layers: bevf_conv3x3_bf16 (csrc/conv3x3_bf16.hip) against the implicit-GEMM bf16 kernel it
replaces (bevf_conv2d_nhwc_bf16), interleaved rounds in one process, random post-ReLU-like data, on the layer shapes of
BASELINE configs 3 (B = 8) and 5 (B = 2).  usage: conv3x3_bench.py [names] [rounds]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bevfusion_multimodal_3d_object_detection_amd import _lib as L

SHAPES = {  # name: (N, H, W, Cin, Cout, residual)
    "layer1": (48, 225, 400, 64, 64, True),
    "layer2": (48, 113, 200, 128, 128, True),
    "layer3": (48, 57, 100, 256, 256, True),
    "cam_proj": (8, 57, 100, 512, 512, False),
    "fusion1_c3": (8, 128, 128, 768, 512, False),
    "fusion1_c2": (8, 128, 128, 512, 512, False),
    "fusion2": (8, 128, 128, 512, 256, False),
    "head": (8, 128, 128, 256, 320, False),
    "fusion1_c5": (2, 256, 256, 512, 512, False),
    "head_c5": (2, 256, 256, 256, 320, False),
}
dev = torch.device("cuda")
names = sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] != "all" else list(SHAPES)
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
BF = torch.bfloat16
for name in names:
    N, H, W, Cin, Cout, has_res = SHAPES[name]
    x = torch.randn(N * H * W * Cin, device=dev).clamp_(min=0).to(BF)          # post-ReLU activations: half zeros
    w = (torch.randn(Cout * 9 * Cin, device=dev) * (1.0 / (9 * Cin)) ** 0.5).to(BF)
    wp = L.conv3x3_pack_bf16(w, Cout, Cin)
    sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
    res = torch.randn(N * H * W * Cout, device=dev).to(BF) if has_res else None
    y0, y1 = torch.empty(N * H * W * Cout, device=dev, dtype=BF), torch.empty(N * H * W * Cout, device=dev, dtype=BF)
    kw = dict(N=N, H=H, W=W, Cin=Cin, x_cs=Cin, Cout=Cout, y_cs=Cout, relu=True, res=res, res_cs=Cout if has_res else 0)

    def old():
        L.conv2d_nhwc(x, w, sc, sh, y0, KH=3, KW=3, stride=1, pad=1, **kw)

    variants = {"old": old}
    for tile, label in ((0, "auto"), (1, "pb2"), (2, "pb1"), (3, "pb1/32rows"), (4, "persistent"), (5, "wide64")):
        variants[label] = (lambda tl: (lambda: L.conv3x3_bf16(x, wp, sc, sh, y1, tile=tl, **kw)))(tile)
    t = {k: [] for k in variants}
    for _ in range(2):
        for fn in variants.values():
            fn()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for key, fn in variants.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record(); torch.cuda.synchronize()
            t[key].append(e0.elapsed_time(e1) / 5)
    flops = 2.0 * N * H * W * Cout * 9 * Cin
    diff = float((y0.float() - y1.float()).abs().max() / y0.float().abs().max())
    med = lambda v: sorted(v)[len(v) // 2]
    print(f"{name:11s} {flops / 1e9:7.1f} GF  " + "  ".join(f"{k} {med(v) * 1e3:6.1f}us {flops / med(v) / 1e9:6.0f}TF" for k, v in t.items())
          + f"  max rel diff {diff:.1e}", flush=True)
