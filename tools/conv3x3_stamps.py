#!/usr/bin/env python3
"""Where a conv3x3_bf16 workgroup spends its life: in-kernel s_memtime stamps (diagnostic launches only) at start / operands landed /
K loop done / stores acknowledged, per workgroup; prints medians in cycles and the share of each phase.  usage: conv3x3_stamps.py [layer] [tile]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bevfusion_multimodal_3d_object_detection_amd import _lib as L
SHAPES = {"layer1": (48, 225, 400, 64, 64, True), "layer2": (48, 113, 200, 128, 128, True), "layer3": (48, 57, 100, 256, 256, True),
          "fusion1_c3": (8, 128, 128, 768, 512, False), "head": (8, 128, 128, 256, 320, False)}
name = sys.argv[1] if len(sys.argv) > 1 else "layer1"
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 0
N, H, W, Cin, Cout, has_res = SHAPES[name]
dev, BF = torch.device("cuda"), torch.bfloat16
x = torch.randn(N * H * W * Cin, device=dev).clamp_(min=0).to(BF)
w = (torch.randn(Cout * 9 * Cin, device=dev) * (1.0 / (9 * Cin)) ** 0.5).to(BF)
wp = L.conv3x3_pack_bf16(w, Cout, Cin)
sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
res = torch.randn(N * H * W * Cout, device=dev).to(BF) if has_res else None
y = torch.empty(N * H * W * Cout, device=dev, dtype=BF)
kw = dict(N=N, H=H, W=W, Cin=Cin, x_cs=Cin, Cout=Cout, y_cs=Cout, relu=True, res=res, res_cs=Cout if has_res else 0, tile=tile)
for _ in range(20):                                   # warm: clocks settle
    L.conv3x3_bf16(x, wp, sc, sh, y, **kw)
nwg = 4 * N * ((H + 15) // 16) * ((W + 15) // 16) * max(1, Cout // 64)
buf = torch.zeros(nwg * 4, dtype=torch.int64, device=dev)
L.lib().bevf_debug_conv3x3_stamps(buf.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
L.conv3x3_bf16(x, wp, sc, sh, y, **kw)
e1.record(); torch.cuda.synchronize()
L.lib().bevf_debug_conv3x3_stamps(None)
t = buf.view(-1, 4).cpu()
t = t[t[:, 0] != 0]
if t.shape[0] == 0:
    sys.exit(f"{name} tile={tile}: this launch ran a kernel variant without stamps (the Cin = 64 one-image and persistent kernels); pass tile 1, 2 or 3")
d = (t[:, 1:] - t[:, :-1]).double()
life = (t[:, 3] - t[:, 0]).double()
q = lambda v: [float(v.quantile(p)) for p in (0.1, 0.5, 0.9)]
print(f"{name} tile={tile}: {t.shape[0]} workgroups, launch {e0.elapsed_time(e1) * 1e3:.1f} us (with stamps)")
for i, lab in enumerate(("prologue (DMA issue + first operands)", "K loop", "epilogue (+ store acknowledgement)")):
    print(f"  {lab:40s} p10/p50/p90 cycles {q(d[:, i])}  share of life {float(d[:, i].sum() / life.sum()):.3f}")
print(f"  workgroup life p10/p50/p90 {q(life)}; span first start -> last end {int(t[:, 3].max() - t[:, 0].min())} cycles")
