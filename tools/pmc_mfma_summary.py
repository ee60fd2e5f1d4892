#!/usr/bin/env python3
"""Per-kernel MFMA-pipe utilisation and wait breakdown from one rocprofv3 --pmc pass (tools/pmc_mfma.sh).
MFMA util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); the SQ wait counters are quad-cycles and are
shown as shares of SQ_WAVE_CYCLES (MI355X_MICROARCH.md, "rocprofv3 PMC slots"); LDS busy = SQ_LDS_IDX_ACTIVE / (active cycles x 256 CUs)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
per = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    per[name][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"], name)
    if key not in seen:
        seen.add(key)
        cnt[name] += 1
print(f"{'kernel':44s} {'n':>4s} {'MFMA util':>9s} {'wait_any':>8s} {'wait_inst':>9s} {'active':>7s} {'LDS confl':>9s} {'LDS busy':>8s}")
for name, c in sorted(per.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    gui = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    if gui <= 0:
        continue
    wave = max(c.get("SQ_WAVE_CYCLES", 0), 1.0)
    util = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui * 1024.0)
    lds = c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0), 1.0)
    print(f"{name:44s} {cnt[name]:4d} {util:9.3f} {c.get('SQ_WAIT_ANY', 0) / wave:8.3f} {c.get('SQ_WAIT_INST_ANY', 0) / wave:9.3f} "
          f"{c.get('SQ_ACTIVE_INST_ANY', 0) / wave:7.3f} {lds:9.3f} {c.get('SQ_LDS_IDX_ACTIVE', 0) / (gui * 256.0):8.3f}")
