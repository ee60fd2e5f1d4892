#!/usr/bin/env python3
"""Stability soak on one MI355X: N training steps (loss finite and decreasing on a fixed batch, allocator steady) and
repeated inference forwards (bit-identical outputs run to run)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bevfusion_multimodal_3d_object_detection_amd import fusion, synth, training, centernet_target as ct
dev = torch.device("cuda")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
model = fusion.create_detector("camera+lidar", "bev", "centernet", bev_h=50, bev_w=50)
synth.fill_state_dict_(model, 0)
model = model.to(dev).train()
imgs, pts, _ = synth.frame_inputs(4, 6, 224, 400, 20000, 4, seed=5)
imgs, pts = imgs.to(dev), pts.to(dev)
boxes, labels = synth.gt_boxes(4, 20, seed=3)
tgt = ct.prepare_centernet_targets({"gt_boxes": boxes.to(dev), "gt_labels": labels.to(dev)}, dev)
crit = ct.CenterNetLoss()
opt = training.FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.01, max_grad_norm=10.0)
first = last = None
for i in range(steps):
    losses = crit(model(imgs, pts, None), tgt)
    opt.zero_grad()
    losses["total_loss"].backward()
    opt.step()
    v = float(losses["total_loss"])
    assert v == v and abs(v) < 1e6, f"step {i}: loss {v}"
    first = v if first is None else first
    last = v
    if i in (5, steps - 1):
        st = torch.cuda.memory_stats()
        print(f"step {i}: loss {v:.4f} reserved {st['reserved_bytes.all.current']/2**30:.2f} GiB mallocs {st['num_device_alloc']}", flush=True)
print(f"training: loss {first:.4f} -> {last:.4f} over {steps} steps on a fixed batch")
assert last < first
model.eval()
ref = {k: v.clone() for k, v in model(imgs, pts, None).items()}
for _ in range(50):
    out = model(imgs, pts, None)
    assert all(torch.equal(out[k], ref[k]) for k in ref)
print("inference: 50 repeated forwards bit-identical")
# full-size config 2 (B = 2): races in the Winograd kernel's hand-written wait / barrier scheme would show as run-to-run differences
del model, opt
torch.cuda.empty_cache()
model = fusion.create_detector("camera+lidar", "bev", "centernet", bev_h=128, bev_w=128)
synth.fill_state_dict_(model, 0)
model = model.to(dev).eval()
imgs, pts, _ = synth.frame_inputs(2, 6, 900, 1600, 35000, 4, seed=5)
imgs, pts = imgs.to(dev), pts.to(dev)
ref = {k: v.clone() for k, v in model(imgs, pts, None).items()}
for _ in range(300):
    out = model(imgs, pts, None)
    assert all(torch.equal(out[k], ref[k]) for k in ref)
print("inference, full size (6 x 900x1600 + 35k points, BEV 128): 300 repeated forwards bit-identical")
