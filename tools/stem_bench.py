#!/usr/bin/env python3
"""Micro-benchmark of the stem conv + max-pool kernels at BASELINE config 2's image size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bevfusion_multimodal_3d_object_detection_amd import _lib as L
if os.environ.get("BEVF_AB_LIB"): L.LIB_PATH = os.environ["BEVF_AB_LIB"]
dev = torch.device("cuda")
N, H, W = (int(sys.argv[1]) if len(sys.argv) > 1 else 24), 900, 1600
x = torch.rand(N, 3, H, W, device=dev)
w = torch.zeros(148, 64, device=dev); w[:147] = torch.randn(147, 64, device=dev) * 0.05
sc, sh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev)
Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
y = torch.empty(N * Ho * Wo * 64, device=dev)
Hp, Wp = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
p = torch.empty(N * Hp * Wp * 64, device=dev)
def t(fn, n=10):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = t(lambda: L.stem_conv7x7(x, w.view(-1), sc, sh, y, N, H, W, relu=True))
fl = 2.0 * N * Ho * Wo * 64 * 147
print(f"stem   {ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} TF   (write {y.numel()*4/ms/1e6:.0f} GB/s)")
ms = t(lambda: L.maxpool3x3s2(y, p, N, Ho, Wo, 64))
print(f"maxpool{ms*1e3:8.1f} us  algorithmic {(y.numel()+p.numel())*4/ms/1e6:.0f} GB/s")
ms = t(lambda: L.stem_pool(x, w.view(-1), sc, sh, p, N, H, W))
print(f"stem+pool fp32 {ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} TF")
wp = L.stem_pack_bf16(w[:147].t().reshape(64, 3, 7, 7).contiguous())
yb = torch.empty(N * Ho * Wo * 64, dtype=torch.bfloat16, device=dev)
ms = t(lambda: L.stem_conv7x7_bf16mma(x, wp, sc, sh, yb, N, H, W, relu=True))
print(f"stem bf16-MFMA {ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} TF   (read {x.numel()*4/ms/1e6:.0f} + write {yb.numel()*2/ms/1e6:.0f} GB/s)")
pb = torch.empty(N * Hp * Wp * 64, dtype=torch.bfloat16, device=dev)
ms = t(lambda: L.stem_pool_bf16mma(x, wp, sc, sh, pb, N, H, W))
print(f"stem+pool bf16 {ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} TF   (read {x.numel()*4/ms/1e6:.0f} + write {pb.numel()*2/ms/1e6:.0f} GB/s)")
