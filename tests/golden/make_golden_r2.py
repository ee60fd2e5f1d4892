#!/usr/bin/env python3
"""Round-2 golden fixtures, minted by running the REFERENCE itself (build container only).

    python tests/golden/make_golden_r2.py         # needs /root/reference; writes tests/golden/*

Same rules as make_golden.py: the reference's src/*.py are imported from /root/reference/src (never copied),
torchvision is satisfied by oracle/resnet18.py, inputs and weights come from synth.py so the fixtures hold
OUTPUTS only.  What this adds:

* `metrics_report_<case>.txt`  -- the file `utils_v2.save_and_print_metrics` writes (ref src/utils_v2.py:208-233)
* `topk_raw.npz`               -- `centernet_target._topk` on an UN-masked heatmap (ref :424-452)
* `gaussian.npz`               -- `gaussian_2d` / `draw_gaussian` (ref :118-125, :152-168)
* `config1_full.npz`           -- BASELINE config 1 at full size on the reference: camera_only, 6x448x800, BEV 128^2,
                                  B=1: every 4th row/column of the five head tensors + float64 checksums
"""
import io
import json
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from bevfusion_multimodal_3d_object_detection_amd import synth            # noqa: E402
from tests.golden import cases                                             # noqa: E402
from tests.golden.make_golden import import_reference, save               # noqa: E402


@torch.no_grad()
def main():
    torch.set_num_threads(1)
    enc, fus, ct, fd = import_reference()
    import utils_v2 as ref_utils                                            # the reference's module

    # ---- metrics report text ---------------------------------------------------------------
    gold = json.load(open(os.path.join(HERE, "metrics.json")))
    for name, metrics in gold.items():
        path = os.path.join(HERE, f"metrics_report_{name}.txt")
        with redirect_stdout(io.StringIO()) as out:
            ref_utils.save_and_print_metrics(metrics, path)
        with open(os.path.join(HERE, f"metrics_report_{name}.stdout.txt"), "w") as f:
            f.write(out.getvalue().replace(path, "<save_path>"))
        print("wrote", path)

    # ---- bare _topk on raw scores ----------------------------------------------------------------
    for c in cases.TOPK_CASES:
        heat = cases.topk_scores(c)
        for tag, mod in (("ct", ct), ("fd", fd)):
            score, ind, classes, ys, xs = mod._topk(heat, K=c["K"])
            save(f"topk_{tag}_{c['name']}", score=score, ind=ind, classes=classes, ys=ys, xs=xs)

    # ---- gaussian helpers --------------------------------------------------------------------------
    arrs = {}
    for i, (shape, sigma) in enumerate(cases.GAUSSIAN_2D_CASES):
        arrs[f"g2d_{i}"] = ct.gaussian_2d(shape, sigma)
    for i, c in enumerate(cases.DRAW_GAUSSIAN_CASES):
        hm = cases.draw_gaussian_canvas(c)
        for center, radius, k in c["splats"]:
            ct.draw_gaussian(hm, center, radius, k)
        arrs[f"draw_{i}"] = hm
    save("gaussian", **arrs)

    # ---- BASELINE config 1 at full size, on the reference itself ------------------------------------
    c = cases.CONFIG1_CASE
    torch.set_num_threads(8)
    torch.manual_seed(0)
    model = fus.create_detector("camera_only", "bev", "centernet", bev_h=c["bev"], bev_w=c["bev"])
    synth.fill_state_dict_(model, c["seed"])
    model.eval()
    imgs = synth.normal((1, 6, 3, c["h"], c["w"]), c["seed"] * 7919)
    out = model(imgs, None, None)
    arrs = {}
    for k, v in out.items():
        v64 = v.double()
        arrs["slice__" + k] = v[:, :, ::c["stride"], ::c["stride"]].contiguous()
        arrs["sum__" + k] = v64.sum()
        arrs["abssum__" + k] = v64.abs().sum()
        arrs["sqsum__" + k] = (v64 * v64).sum()
        arrs["max__" + k] = v64.abs().max()
        arrs["rowsum__" + k] = v64.sum(dim=3)              # (1,C,128): a checksum per BEV row and channel
    save("config1_full", **arrs)


if __name__ == "__main__":
    main()
