"""PointNet front (conv1..conv3) at config 2's size: the fused kernel against the three launches it replaces.
    python tools/pointnet_front_bench.py [points-per-frame] [frames]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevfusion_multimodal_3d_object_detection_amd import _lib as L  # noqa: E402
from bevfusion_multimodal_3d_object_detection_amd import synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 35000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
M, K = N * B, 4
dev = torch.device("cuda")
x = synth.normal((M, K), 1).to(dev)
w = [synth.normal(s, 2 + i, 0, 0.1).to(dev) for i, s in enumerate(((64, K), (128, 64), (256, 128)))]
s = [torch.ones(c, device=dev) for c in (64, 128, 256)]
b = [torch.zeros(c, device=dev) for c in (64, 128, 256)]
f2, f3 = L.pointnet_front_pack(w[1]), L.pointnet_front_pack(w[2])
h1, h2, y = (torch.empty(M * c, device=dev) for c in (64, 128, 256))


def fused():
    L.pointnet_front(x, w[0], s[0], b[0], f2, s[1], b[1], f3, s[2], b[2], y, M, K)


def separate():
    L.pointwise_smallk(x, w[0], s[0], b[0], h1, M, K, 64, True)
    L.conv2d_nhwc(h1, w[1].view(-1), s[1], b[1], h2, N=M, H=1, W=1, Cin=64, x_cs=64, Cout=128, y_cs=128, KH=1, KW=1, stride=1, pad=0, relu=True)
    L.conv2d_nhwc(h2, w[2].view(-1), s[2], b[2], y, N=M, H=1, W=1, Cin=128, x_cs=128, Cout=256, y_cs=256, KH=1, KW=1, stride=1, pad=0, relu=True)


flops = 2.0 * M * (64 * 128 + 128 * 256)
for name, fn in (("fused", fused), ("separate", separate)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name:9s} M={M}: {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.1f} TF (MFMA layers)")
