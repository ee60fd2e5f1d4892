#!/usr/bin/env python3
"""Mint tests/golden/metrics.json by running the REFERENCE's src/utils_v2.py (torch/numpy only) in the build container.
Inputs come from tests/golden/cases.metrics_inputs (counter-based generator); the fixture holds outputs only."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")
from tests.golden import cases                       # noqa: E402
import utils_v2 as ref                               # noqa: E402  (the reference's module)

out = {}
for c in cases.METRICS_CASES:
    preds, gts = cases.metrics_inputs(c)
    out[c["name"]] = ref.compute_metrics(preds, gts)
json.dump(out, open(os.path.join(HERE, "metrics.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: (v["mAP"], v["NDS"]) for k, v in out.items()}))
