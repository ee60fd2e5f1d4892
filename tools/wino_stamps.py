#!/usr/bin/env python3
"""Where a wino_f32 workgroup spends its life: in-kernel s_memtime stamps of a diagnostic launch (start / first operands transformed /
K loop done / epilogue issued / stores acknowledged), per workgroup; p10 / p50 / p90 in s_memtime ticks and shares.  The last line gives the
tick rate: one workgroup per CU is resident, so sum of lives / (256 CUs x launch time) = ticks per second (it comes out at ~2.0-2.4 GHz:
the ticks are shader cycles).
usage: wino_stamps.py [layer] [tile]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bevfusion_multimodal_3d_object_detection_amd import _lib as L
SHAPES = {"layer1": (48, 225, 400, 64, 64, False), "layer1r": (48, 225, 400, 64, 64, True), "layer2": (48, 113, 200, 128, 128, True),
          "layer3": (48, 57, 100, 256, 256, True), "fusion1": (8, 128, 128, 512, 512, False)}
name = sys.argv[1] if len(sys.argv) > 1 else "layer1"
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 0
N, H, W, Cin, Cout, has_res = SHAPES[name]
dev = torch.device("cuda")
x = torch.randn(N * H * W * Cin, device=dev).clamp_(min=0)
w = torch.randn(Cout * 9 * Cin, device=dev) * (1.0 / (9 * Cin)) ** 0.5
u = L.wino_filter_transform(w, Cout, Cin)
sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev)
res = torch.randn(N * H * W * Cout, device=dev) if has_res else None
y = torch.empty(N * H * W * Cout, device=dev)
kw = dict(N=N, H=H, W=W, Cin=Cin, x_cs=Cin, Cout=Cout, y_cs=Cout, relu=True, res=res, res_cs=Cout if has_res else 0, tile=tile)
for _ in range(10):
    L.conv3x3_wino(x, u, sc, sh, y, **kw)
nwg = N * ((H + 15) // 16 + 1) * ((W + 7) // 8 + 1) * max(1, (Cout + 63) // 64)      # an upper bound for every tiling
buf = torch.zeros(nwg * 5, dtype=torch.int64, device=dev)
L.lib().bevf_debug_wino_stamps(buf.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
L.conv3x3_wino(x, u, sc, sh, y, **kw)
e1.record(); torch.cuda.synchronize()
L.lib().bevf_debug_wino_stamps(None)
t = buf.view(-1, 5).cpu()
t = t[t[:, 0] != 0]
d = (t[:, 1:] - t[:, :-1]).double()
life = (t[:, 4] - t[:, 0]).double()
q = lambda v: [round(float(v.quantile(p))) for p in (0.1, 0.5, 0.9)]
print(f"{name} tile={tile}: {t.shape[0]} workgroups, launch {e0.elapsed_time(e1) * 1e3:.1f} us (with stamps)")
for i, lab in enumerate(("prologue (first DMA, wait, first input transform)", "K loop", "epilogue issue (transform, residual, stores)",
                         "store acknowledgement")):
    print(f"  {lab:50s} p10/p50/p90 ticks {q(d[:, i])}  share of life {float(d[:, i].sum() / life.sum()):.3f}")
print(f"  workgroup life p10/p50/p90 {q(life)} ticks; sum of lives / (256 CUs x launch) = "
      f"{float(life.sum()) / (256 * e0.elapsed_time(e1) * 1e-3) / 1e9:.2f} G ticks/s")
