#!/usr/bin/env python3
"""The non-convolution kernels the north star names, each timed on its own against its algorithmic bytes (HBM roofline, 8 TB/s spec):
K20 front end (voxelize -> VFELayer -> scatter_voxels) at B = 8 x 35 k and B = 2 x 120 k points, the input pipeline
(resize_normalize_u8 on 6 x 900 x 1600 uint8 frames per sample, lidar_filter_pad), camera BEV pooling and CenterNet decode at
128^2 / 256^2.  usage: frontend_bench.py [rounds]   (prints one JSON object per kernel; bench.py runs it as an extra leg)"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bevfusion_multimodal_3d_object_detection_amd import _lib as _L
if os.environ.get("BEVF_AB_LIB"):                      # A/B against another build of the library (e.g. an older voxeliser)
    _L.LIB_PATH = os.environ["BEVF_AB_LIB"]
from bevfusion_multimodal_3d_object_detection_amd import centernet_target, encoders, preprocess, synth

PEAK = 8000.0


def timed(fn, rounds=5, inner=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record(); torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1) / inner)
    return sorted(t)[len(t) // 2] * 1e3          # us


def run(rounds=5):
    dev = torch.device("cuda")
    out = []

    def rec(name, us, nbytes, note):
        gbs = nbytes / us / 1e3
        out.append({"kernel": name, "us": round(us, 1), "algorithmic_mb": round(nbytes / 1e6, 2), "gb_per_s": round(gbs, 1),
                    "frac_of_hbm_peak": round(gbs / PEAK, 4), "note": note})
    RANGE, VS = (-51.2, -51.2, -5.0, 51.2, 51.2, 3.0), (2.048, 2.048, 8.0)
    for B, N in ((8, 35000), (2, 120000)):
        _, pts, _ = synth.frame_inputs(B, 0, 0, 0, N, 4, seed=77)
        pts = pts.to(dev)
        P, NV = 32, 2500
        us = timed(lambda: encoders.voxelize(pts, RANGE, VS, P, NV), rounds)
        # points read once (16 B), 8-byte keys written + read by 2 sort passes (x2 each) + heads, the P x C pillar rows written
        nb = B * N * (16 + 8 * 6 + 8) + B * NV * P * 16
        rec(f"voxelize B={B} N={N} (50x50 pillars, P=32)", us, nb, "keys + 2 radix passes (3 launches each) + heads + scan (2) + fill: 11 launches + 5 torch allocations, launch-bound")
        f, c, n, v = encoders.voxelize(pts, RANGE, VS, P, NV)
        vfe = encoders.VFELayer(4, 64).to(dev).eval()
        us = timed(lambda: vfe(f), rounds)
        rec(f"VFELayer 4->64 + max over P, B={B}", us, B * NV * (P * 16 + 64 * 4), "one fused kernel (linear + BN + ReLU + max in registers)")
        pf = vfe(f)
        us = timed(lambda: encoders.scatter_voxels(pf, c, (1, 50, 50), v), rounds)
        rec(f"scatter_voxels -> (B,64,50,50), B={B}", us, B * (NV * 64 * 4 + 2500 * 64 * 4 + NV * 24), "owner atomicMax + coalesced write")
    g = torch.Generator().manual_seed(5)
    imgs = torch.randint(0, 256, (8, 6, 900, 1600, 3), dtype=torch.uint8, generator=g).to(dev)
    us = timed(lambda: preprocess.preprocess_camera_images(imgs), rounds)
    rec("resize_normalize_u8 8x6x900x1600 -> 448x800 fp32", us, imgs.numel() + 8 * 6 * 3 * 448 * 800 * 4, "Pillow-exact antialiased bilinear + normalise")
    sweep = torch.rand(120000, 5, generator=g).mul_(120).sub_(60).to(dev)
    us = timed(lambda: preprocess.filter_pad_lidar(sweep, 35000), rounds)
    rec("lidar_filter_pad 120k x 5 -> 35k", us, 120000 * 20 + 35000 * 20, "range filter + ordered compaction + pad (one sweep: count, compact, output = 3 launches over 118 tiles)")
    from bevfusion_multimodal_3d_object_detection_amd import _lib as L
    for B, S in ((8, 128), (2, 256)):
        cam = torch.randn(B, 6, 57, 100, 512, device=dev)                # NHWC per camera
        mean = torch.empty(B * 57 * 100 * 512, device=dev)
        us = timed(lambda: L.cam_mean(cam, mean, B, 6, 57 * 100, 512), rounds)
        rec(f"cam_mean B={B} (6 x 57x100x512 fp32)", us, B * 7 * 57 * 100 * 512 * 4, "BEV pooling, part 1 (mean over cameras)")
        pred = {"heatmap": torch.rand(B, 10, S, S, device=dev), "offset": torch.rand(B, 2, S, S, device=dev),
                "size": torch.rand(B, 3, S, S, device=dev), "rot": torch.randn(B, 2, S, S, device=dev),
                "vel": torch.randn(B, 2, S, S, device=dev)}
        us = timed(lambda: centernet_target.decode_centernet_predictions(pred, 0.3, 100), rounds)
        rec(f"decode_centernet B={B} {S}x{S}", us, B * 19 * S * S * 4, "nms + per-class top-K + frame top-K + boxes (latency-bound; includes the host-side result lists)")
    return out


if __name__ == "__main__":
    for r in run(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
        print(json.dumps(r), flush=True)
